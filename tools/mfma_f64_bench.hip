// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (cycles per instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void mfma_loop(double* out, unsigned long long* cyc, int iters, int nacc) {
    v4d a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        if (nacc > 1) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        if (nacc > 2) { a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void fma_loop(double* out, unsigned long long* cyc, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = 1.0000001, y = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
        a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    unsigned long long h[256];
    const int iters = 20000;
    for (int threads : {64, 256, 512, 1024}) for (int nacc : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        mfma_loop<<<256, threads>>>(out, cyc, iters, nacc);
        hipEventRecord(e0); mfma_loop<<<256, threads>>>(out, cyc, iters, nacc); hipEventRecord(e1);
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        int per = nacc == 4 ? 4 : nacc;
        double nm = (double)iters * per;                      // MFMAs per wave
        double waves_per_simd = threads / 64.0 / 4.0; if (waves_per_simd < 1) waves_per_simd = 1;
        double flop = 256.0 * (threads / 64) * nm * 2048;
        printf("mfma threads=%4d nacc=%d: %.1f ticks/MFMA/wave, %.1f ticks per MFMA per SIMD, %.2f TFLOP/s, kernel %.3f ms, tick-GHz %.3f\n",
               threads, nacc, h[0] / nm, h[0] / nm / waves_per_simd, flop / ms / 1e9, ms, h[0] / (ms * 1e6));
    }
    for (int threads : {64, 256, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        fma_loop<<<256, threads>>>(out, cyc, iters);
        hipEventRecord(e0); fma_loop<<<256, threads>>>(out, cyc, iters); hipEventRecord(e1);
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double flop = 256.0 * threads * iters * 8.0 * 2;
        printf("fma  threads=%4d: %.2f ticks per v_fma_f64 per wave, %.2f TFLOP/s, %.3f ms\n", threads, h[0] / (iters * 8.0), flop / ms / 1e9, ms);
    }
    return 0;
}

// Per-CU read throughput ceiling: ONE workgroup (or 256) streams a private 16 MB buffer with 16-byte loads,
// (a) into VGPRs with UNR loads in flight per lane, (b) with LDS-DMA (global_load_lds, 16 B/lane).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNR>
__global__ __launch_bounds__(1024) void stream_vgpr(const double2* __restrict__ src, double* out, size_t n2) {
    const double2* p = src + (size_t)blockIdx.x * n2;
    double acc = 0.0;
    for (size_t i = threadIdx.x; i + (UNR - 1) * 1024 < n2; i += UNR * 1024) {
        double2 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = p[i + u * 1024];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void stream_lds(const double2* __restrict__ src, double* out, size_t n2) {
    extern __shared__ double2 buf[];      // [4][1024] ring
    const double2* p = src + (size_t)blockIdx.x * n2;
    double acc = 0.0;
    const int wv = threadIdx.x >> 6;
    for (size_t i = 0; i + 4 * 1024 <= n2; i += 4 * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // wave-uniform LDS base + lane*16
            __builtin_amdgcn_global_load_lds((const void*)(p + i + u * 1024 + threadIdx.x), (__attribute__((address_space(3))) void*)(buf + u * 1024 + wv * 64), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
        __syncthreads();
        acc += buf[threadIdx.x].x + buf[1024 + threadIdx.x].y + buf[2048 + threadIdx.x].x + buf[3072 + threadIdx.x].y;
        __syncthreads();
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = 16u << 20, n2 = bytes / 16;
    double2* src; double* out;
    hipMalloc(&src, 256 * bytes); hipMemset(src, 0, 256 * bytes); hipMalloc(&out, 256 * 1024 * 8);
    for (int nwg : {1, 256}) {
        auto run = [&](auto kern, size_t lds, const char* name) {
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(1024), lds, 0, src, out, n2);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nwg), dim3(1024), lds, 0, src, out, n2); hipEventRecord(e1);
            hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-22s nwg=%3d: %.3f ms, %.1f GB/s per CU (%.1f B/clk @2.3GHz), %.2f TB/s total\n", name, nwg, ms,
                   bytes / ms / 1e6, bytes / (ms * 2.3e6), nwg * (double)bytes / ms / 1e9);
        };
        run(stream_vgpr<1>, 0, "vgpr 1 load/lane");
        run(stream_vgpr<4>, 0, "vgpr 4 loads/lane");
        run(stream_vgpr<8>, 0, "vgpr 8 loads/lane");
        run(stream_lds, 4 * 1024 * 16, "lds-dma 4KBx16 ring");
    }
    return 0;
}

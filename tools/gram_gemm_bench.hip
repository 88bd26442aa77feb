// Feasibility bench for the Gram as ONE shared-operand GEMM: P[n][M] = sum_k K[M][k] * W2[k][n] with
// M = 561 * 256 entries of the factor's tile order, k = 512 data rows, n = up to 1024 spectra, FP64 MFMA 16x16x4, 128 x 128 tile
// per 256-thread workgroup (64 x 64 per wavefront = 16 accumulators), K slabs of 16 through LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/gram_gemm_bench.hip -o tools/gram_gemm_bench.bin && ./tools/gram_gemm_bench.bin [nspec]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int TM = 128, TN = 128, SK = 16, LD = 144;     // LD: k rows land 32 banks apart
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Kp: [mblk][k][TM] (a slab of 16 k rows of one M block is 16 kB contiguous), W2: [k][N], P: [n][M]
template <bool XCD, bool STORE, bool PIPE = false, int MODE = 0>      // MODE 1: no global fetch / LDS refill inside the loop; 2: no LDS reads either
__global__ __launch_bounds__(256, 2) void gemm_kernel(const double* __restrict__ Kp, const double* __restrict__ W2,
                                                      double* __restrict__ P, int M, int N, int KD, int nblk_n,
                                                      unsigned long long* __restrict__ clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __shared__ double sA[2][SK * LD];
    __shared__ double sB[2][SK * LD];
    int mb, nb;
    if (XCD) {      // the N blocks of one M block on one XCD (blocks b, b + 8, ... share an L2)
        const int b = blockIdx.x, grp = b / (8 * nblk_n), r = b % (8 * nblk_n);
        mb = grp * 8 + (r & 7); nb = r >> 3;
    } else { mb = blockIdx.x / nblk_n; nb = blockIdx.x % nblk_n; }
    if ((size_t)mb * TM >= (size_t)M) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    const double* Ka = Kp + (size_t)mb * KD * TM;
    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0, 0, 0, 0};
    // staging: thread -> (k = tid / 16, 8 consecutive columns)
    const int sk = tid >> 4, sc = (tid & 15) * 8;
    double2 ra[4], rb[4];
    auto fetch = [&](int k0) {
        const double* pa = Ka + (size_t)(k0 + sk) * TM + sc;
        const double* pb = W2 + (size_t)(k0 + sk) * N + (size_t)nb * TN + sc;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ra[e] = *reinterpret_cast<const double2*>(pa + 2 * e); rb[e] = *reinterpret_cast<const double2*>(pb + 2 * e); }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<double2*>(&sA[buf][sk * LD + sc + 2 * e]) = ra[e];
            *reinterpret_cast<double2*>(&sB[buf][sk * LD + sc + 2 * e]) = rb[e];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < KD; k0 += SK) {
        if (MODE == 0 && k0 + SK < KD) fetch(k0 + SK);
        if (PIPE) {
            // operands of k step kk + 4 requested in front of the sixteen matrix instructions of step kk
            double a[2][4], b[2][4];
            auto ld = [&](int slot, int kk) {
                const int kr = kk + (lane >> 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[slot][i] = sA[buf][kr * LD + wm + 16 * i + (lane & 15)]; b[slot][i] = sB[buf][kr * LD + wn + 16 * i + (lane & 15)]; }
            };
            ld(0, 0);
#pragma unroll
            for (int q = 0; q < SK / 4; ++q) {
                if (q + 1 < SK / 4) ld((q + 1) & 1, 4 * (q + 1));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[q & 1][j], a[q & 1][i], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < SK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 2) { a[i] = 1.0 + kr + i + k0; b[i] = 2.0 + kr - i + k0; }
                else { a[i] = sA[buf][kr * LD + wm + 16 * i + (lane & 15)]; b[i] = sB[buf][kr * LD + wn + 16 * i + (lane & 15)]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        }
        if (MODE == 0 || MODE == 4) {             // MODE 4: LDS refill + barrier, but from registers fetched once (no global loads in the loop)
            if (k0 + SK < KD) { stash(buf ^ 1); }
            __syncthreads();
            buf ^= 1;
        } else if (MODE == 3) {                   // MODE 3: the barrier alone
            __syncthreads();
            buf ^= 1;
        }
    }
    if (threadIdx.x == 0 && clk && blockIdx.x == gridDim.x / 2) {     // shader clock over this workgroup's life: ticks per 100 MHz tick
        clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // acc[i][j], lane l, register r = element (m = wm + 16 i + (l & 15), n = wn + 16 j + (l >> 4) + 4 r): 16 consecutive m per n
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t n = (size_t)nb * TN + wn + 16 * j + (lane >> 4) + 4 * r;
                const size_t m = (size_t)mb * TM + wm + 16 * i + (lane & 15);
                if (STORE || acc[i][j][r] == 1.2345e300) __builtin_nontemporal_store(acc[i][j][r], &P[n * (size_t)M + m]);
            }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1024, KD = 512, M = 561 * 256;
    const int nbm = (M + TM - 1) / TM, nbn = N / TN;
    std::vector<double> hK((size_t)nbm * KD * TM), hW((size_t)KD * N);
    for (size_t i = 0; i < hK.size(); ++i) hK[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = ((i * 40503u) % 977) / 977.0;
    double *dK, *dW, *dP;
    CK(hipMalloc(&dK, hK.size() * 8)); CK(hipMalloc(&dW, hW.size() * 8)); CK(hipMalloc(&dP, (size_t)N * M * 8));
    CK(hipMemcpy(dK, hK.data(), hK.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice));
    unsigned long long* dC; CK(hipMalloc(&dC, 16)); CK(hipMemset(dC, 0, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int xcd = 0; xcd < 9; ++xcd) {
        const int grid = ((nbm + 7) / 8) * 8 * nbn;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (xcd == 1) hipLaunchKernelGGL((gemm_kernel<true, true>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 2) hipLaunchKernelGGL((gemm_kernel<true, false>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 3) hipLaunchKernelGGL((gemm_kernel<true, false, true>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 4) hipLaunchKernelGGL((gemm_kernel<true, true, true>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 5) hipLaunchKernelGGL((gemm_kernel<true, false, false, 1>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 6) hipLaunchKernelGGL((gemm_kernel<true, false, false, 2>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 7) hipLaunchKernelGGL((gemm_kernel<true, false, false, 3>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else if (xcd == 8) hipLaunchKernelGGL((gemm_kernel<true, false, false, 4>), dim3(grid), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            else hipLaunchKernelGGL((gemm_kernel<false, true>), dim3(nbm * nbn), dim3(256), 0, 0, dK, dW, dP, M, N, KD, nbn, dC);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = 2.0 * M * (double)N * KD;
            unsigned long long hc[2]; CK(hipMemcpy(hc, dC, 16, hipMemcpyDeviceToHost));
            if (rep == 2 && hc[1]) printf("   shader clock while a mid-grid workgroup ran: %.2f GHz (%llu shader ticks over %llu ticks of 100 MHz)\n", hc[0] / (double)hc[1] * 0.1, hc[0], hc[1]);
            if (rep == 2) printf("N = %d spectra, %s block order: %.3f ms, %.1f TFLOP/s = %.3f of 78.6; P written %.2f GB -> %.2f TB/s of stores\n", N,
                                 xcd == 8 ? "NO STORES, LDS refill + barrier per slab, no global loads in the loop" : xcd == 7 ? "NO STORES, barrier per slab only" : xcd == 6 ? "NO STORES, no global fetch, no LDS reads (registers only)" : xcd == 5 ? "NO STORES, no global fetch / LDS refill / barrier in the loop (LDS reads only)" : xcd == 4 ? "XCD-grouped, operands one k step ahead" : xcd == 3 ? "XCD-grouped, operands one k step ahead, NO STORES" : xcd == 2 ? "XCD-grouped, NO STORES (K loop only)" : xcd ? "XCD-grouped" : "row-major", ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6, (double)N * M * 8 / 1e9, (double)N * M * 8 / ms / 1e9);
        }
    }
    // spot check
    std::vector<double> hp(M);
    CK(hipMemcpy(hp.data(), dP + (size_t)5 * M, (size_t)M * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m : {0, 1, 127, 128, 4097, M - 1}) {
        double ref = 0;
        const int mb = m / TM, mi = m % TM;
        for (int k = 0; k < KD; ++k) ref += hK[((size_t)mb * KD + k) * TM + mi] * hW[(size_t)k * N + 5];
        worst = fmax(worst, fabs(ref - hp[m]) / (fabs(ref) + 1e-30));
    }
    printf("spot check, spectrum 5: max relative deviation %.2e\n", worst);
    return 0;
}

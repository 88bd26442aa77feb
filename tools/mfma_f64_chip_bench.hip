// micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate of the WHOLE chip (all SIMDs busy, wall time by HIP events) against the
// 78.6 TFLOP/s that 256 CUs x 4 pipes x 2048 flop / 64 cycles x 2.4 GHz give on paper; also the shader clock under that load
// (mfma_cycles below: s_memtime counts shader cycles, s_memrealtime 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_chip_bench.hip -o tools/mfma_f64_chip_bench.bin && tools/mfma_f64_chip_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
// s_memtime counts shader clock cycles, s_memrealtime a constant 100 MHz: their ratio is the shader clock under this very load.
// EVERY wavefront reports: cycles and 100 MHz ticks of its timed window (1000 x 4 MFMAs behind a warm-up under the same load),
// the absolute start / end of its whole run, and where it ran (HW_ID: SIMD, CU, SE; XCC_ID).
struct WaveRec { unsigned cyc, ticks, hw, xcc; unsigned long long run0, run1; };
__global__ __launch_bounds__(1024) void mfma_cycles(double* out, WaveRec* rec, int warm) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < warm; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 1000; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < warm; ++i) {      // keep the load up while the others are being timed
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if ((threadIdx.x & 63) == 0) {
        WaveRec& r = rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6];
        r.cyc = (unsigned)(c1 - c0); r.ticks = (unsigned)(t1 - t0); r.run0 = r0; r.run1 = r1;
        r.hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11));            // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;     // HW_REG_XCC_ID[3:0]
    }
}
// The same timed window with the accumulators PINNED in AccVGPRs by inline asm ("+a"): v_mfma_f64_16x16x4_f64 a[..], v, v, a[..].
// The compiler-chosen AccVGPR form of mfma_loop below copies all 32 accumulator registers VGPR -> AccVGPR and back in EVERY
// iteration (32 v_accvgpr_write + 32 v_accvgpr_read + s_nop 13 around 4 MFMAs: profiles/r04_mfma_agpr_disasm.txt), which is
// what made that leg 2.4x slower -- not the register class.  Same launch geometry cases as mfma_cycles.
template <int BOUND>
__global__ __launch_bounds__(BOUND) void mfma_cycles_agpr(double* out, WaveRec* rec, int warm) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
#define AMFMA(A) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(A) : "v"(x), "v"(y))
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < warm; ++i) { AMFMA(a0); AMFMA(a1); AMFMA(a2); AMFMA(a3); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 1000; ++i) { AMFMA(a0); AMFMA(a1); AMFMA(a2); AMFMA(a3); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < warm; ++i) { AMFMA(a0); AMFMA(a1); AMFMA(a2); AMFMA(a3); }
#undef AMFMA
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if ((threadIdx.x & 63) == 0) {
        WaveRec& r = rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6];
        r.cyc = (unsigned)(c1 - c0); r.ticks = (unsigned)(t1 - t0); r.run0 = r0; r.run1 = r1;
        r.hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11));
        r.xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;
    }
}
// ... and with A / B operands in AccVGPRs too (gfx90a+: any MFMA source may be an AccVGPR), one dependent accumulator chain of
// 2, 3 and 4 accumulators: how many independent tiles a wavefront needs in rotation to issue every 64 cycles.
template <int NACC, bool ABAGPR>
__global__ __launch_bounds__(256) void mfma_chain(double* out, unsigned* cyc) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
#define VMFMA(A) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(A) : "v"(x), "v"(y))
#define GMFMA(A) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(A) : "a"(x), "a"(y))
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4000; ++i) {
        if (ABAGPR) { GMFMA(a0); if (NACC > 1) GMFMA(a1); if (NACC > 2) GMFMA(a2); if (NACC > 3) GMFMA(a3); }
        else { VMFMA(a0); if (NACC > 1) VMFMA(a1); if (NACC > 2) VMFMA(a2); if (NACC > 3) VMFMA(a3); }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#undef VMFMA
#undef GMFMA
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (unsigned)(c1 - c0);
}
// Does other work on the same SIMD take time from the FP64 matrix pipe?  512-thread workgroups, wavefronts 0..3 (one per SIMD) run
// the MFMA loop, wavefronts 4..7 (their SIMD partners) one of: nothing, 32-bit integer VALU, FP32 FMA, FP64 FMA, LDS reads.
__global__ __launch_bounds__(512) void mfma_partner(double* out, unsigned* cyc, int kind, int iters) {
    __shared__ double lds[2048];
    const int wv = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.0; lds[threadIdx.x + 1024] = 2.0; lds[threadIdx.x + 1536] = 3.0;
    __syncthreads();
    if (wv < 4) {
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = (unsigned)(c1 - c0);
    } else {
        // about as long as the MFMA loop (iters x 256 cycles): 16 instructions of >= 4 cycles per round, 4 x iters rounds
        unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
        float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3;
        double d0 = threadIdx.x, d1 = 1, d2 = 2, d3 = 3;
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
        unsigned long long n = 0;
        for (int i = 0; i < 4 * iters; ++i) {
            if (kind == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { u0 = u0 * 3u + u1; u1 = u1 * 5u + u2; u2 = u2 * 7u + u3; u3 = u3 * 9u + u0; }
            } else if (kind == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { f0 = fmaf(f0, 1.0001f, f1); f1 = fmaf(f1, 1.0001f, f2); f2 = fmaf(f2, 1.0001f, f3); f3 = fmaf(f3, 1.0001f, f0); }
            } else if (kind == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { d0 = fma(d0, 1.0001, d1); d1 = fma(d1, 1.0001, d2); d2 = fma(d2, 1.0001, d3); d3 = fma(d3, 1.0001, d0); }
            } else if (kind == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0 += lds[(threadIdx.x + 64 * r) & 2047]; d1 += lds[(threadIdx.x + 64 * r + 512) & 2047];
                    d2 += lds[(threadIdx.x + 64 * r + 1024) & 2047]; d3 += lds[(threadIdx.x + 64 * r + 1536) & 2047];
                }
            } else {
                break;
            }
            ++n;
        }
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 512 + threadIdx.x] = u0 + u1 + u2 + u3 + f0 + f1 + f2 + f3 + d0 + d1 + d2 + d3;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = (unsigned)(c1 - c0);
    }
}
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double* out; hipMalloc(&out, sizeof(double) * 256 * cus * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, clock rate reported %.0f MHz\n", cus, p.clockRate / 1e3);
    printf("accumulators in AccVGPRs (what the compiler picks for this loop under __launch_bounds__(256)):\n");
    for (int wgs_per_cu : {1, 2, 4}) {
        for (int iters : {20000}) {
            const int grid = cus * wgs_per_cu;           // 256-thread workgroups: one wavefront per SIMD each
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 1000);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfmas_per_simd = 4.0 * iters * wgs_per_cu;
            const double flop = mfmas_per_simd * 2048.0 * 4 * cus;
            printf("%d wavefront(s) per SIMD, %7d x 4 MFMAs each: %8.3f ms, %6.1f TFLOP/s, implied clock at 64 cycles per MFMA %.0f MHz\n",
                   wgs_per_cu, iters, ms, flop / ms * 1e-9, mfmas_per_simd * 64 / (ms * 1e-3) * 1e-6);
        }
    }
    printf("accumulators in ordinary VGPRs (__launch_bounds__(1024): at most 128 registers, no AccVGPRs):\n");
    // Who pays: the clock (cycles per MFMA stay 64, the cycle gets longer) or the pipe (more cycles per MFMA)?  And is it the
    // chip's budget (few CUs run at full rate) or the CU's (one SIMD alone runs at full rate)?
    const int maxw = 256 * 16 * 2;
    WaveRec* rec; hipMalloc(&rec, sizeof(WaveRec) * maxw);
    static WaveRec h[maxw];
    struct Case { const char* what; int grid, block; } cases[] = {
        {"all CUs x 16 wavefronts (4 per SIMD)", cus, 1024},
        {"all CUs x 8 wavefronts (2 per SIMD)", cus, 512},
        {"all CUs x 4 wavefronts (1 per SIMD)", cus, 256},
        {"all CUs x 1 wavefront", cus, 64},
        {"128 workgroups x 16 wavefronts", 128, 1024},
        {"64 workgroups x 16 wavefronts", 64, 1024},
        {"32 workgroups x 16 wavefronts", 32, 1024},
        {"8 workgroups x 16 wavefronts", 8, 1024},
    };
    for (const Case& c : cases) {
        const int nw = c.grid * c.block / 64, wps = c.block >= 256 ? c.block / 256 : 1;
        const int warm = 40000 / wps;
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_cycles, dim3(c.grid), dim3(c.block), 0, 0, out, rec, warm);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, rec, sizeof(WaveRec) * nw, hipMemcpyDeviceToHost);
        double sc = 0, st = 0, cmin = 1e30, cmax = 0;
        unsigned long long r0 = ~0ull, r1 = 0;
        int cu_used[8][64] = {};
        for (int i = 0; i < nw; ++i) {
            sc += h[i].cyc; st += h[i].ticks;
            cmin = h[i].cyc < cmin ? h[i].cyc : cmin; cmax = h[i].cyc > cmax ? h[i].cyc : cmax;
            r0 = h[i].run0 < r0 ? h[i].run0 : r0; r1 = h[i].run1 > r1 ? h[i].run1 : r1;
            const unsigned hw = h[i].hw, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cu_used[h[i].xcc & 7][(se * 2 + sh) * 16 + cu & 63]++;
        }
        int ncu = 0, maxper = 0;
        for (int x = 0; x < 8; ++x) for (int k = 0; k < 64; ++k) { if (cu_used[x][k]) ++ncu; if (cu_used[x][k] > maxper) maxper = cu_used[x][k]; }
        const double total_mfma = (double)nw * (2.0 * warm + 1000) * 4;
        const double span_s = (double)(r1 - r0) * 10e-9;
        printf("%-40s: window: %6.1f cycles per MFMA per wavefront (min %.1f max %.1f), clock %4.0f MHz | whole run: %6.1f TFLOP/s by s_memrealtime span "
               "(%.2f ms; events %.2f ms) | %d CUs used, at most %d wavefronts on one\n",
               c.what, sc / nw / 4000.0, cmin / 4000.0, cmax / 4000.0, sc / (st * 10e-9) * 1e-6, total_mfma * 2048 / span_s * 1e-12,
               span_s * 1e3, ms, ncu, maxper);
    }
    printf("accumulators pinned in AccVGPRs by inline asm (same in-kernel window; __launch_bounds__(256) = up to 512 registers per lane):\n");
    {
        struct Case2 { const char* what; int grid, block; } cases2[] = {
            {"AGPR acc, all CUs x 4 wavefronts (1 per SIMD)", cus, 256},
            {"AGPR acc, 2 x CUs x 4 wavefronts (2 per SIMD)", 2 * cus, 256},
            {"AGPR acc, 8 workgroups x 4 wavefronts", 8, 256},
        };
        for (const Case2& c : cases2) {
            const int nw = c.grid * c.block / 64;
            const int warm = 40000 / (c.grid > cus ? 2 : 1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_cycles_agpr<256>, dim3(c.grid), dim3(c.block), 0, 0, out, rec, warm);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, rec, sizeof(WaveRec) * nw, hipMemcpyDeviceToHost);
            double sc = 0, st = 0, cmin = 1e30, cmax = 0;
            unsigned long long r0 = ~0ull, r1 = 0;
            for (int i = 0; i < nw; ++i) {
                sc += h[i].cyc; st += h[i].ticks;
                cmin = h[i].cyc < cmin ? h[i].cyc : cmin; cmax = h[i].cyc > cmax ? h[i].cyc : cmax;
                r0 = h[i].run0 < r0 ? h[i].run0 : r0; r1 = h[i].run1 > r1 ? h[i].run1 : r1;
            }
            const double total_mfma = (double)nw * (2.0 * warm + 1000) * 4;
            const double span_s = (double)(r1 - r0) * 10e-9;
            printf("%-48s: window: %6.1f cycles per MFMA per wavefront (min %.1f max %.1f), clock %4.0f MHz | whole run: %6.1f TFLOP/s (%.2f ms; events %.2f ms)\n",
                   c.what, sc / nw / 4000.0, cmin / 4000.0, cmax / 4000.0, sc / (st * 10e-9) * 1e-6, total_mfma * 2048 / span_s * 1e-12, span_s * 1e3, ms);
        }
        unsigned* cyc2; hipMalloc(&cyc2, sizeof(unsigned) * 4 * cus);
        static unsigned hc2[4 * 1024];
        auto report = [&](const char* what, int nacc) {
            hipDeviceSynchronize();
            hipMemcpy(hc2, cyc2, sizeof(unsigned) * 4 * cus, hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < 4 * cus; ++i) m += hc2[i];
            printf("dependent chains, %-44s %d accumulator(s) in rotation: %.1f cycles per MFMA\n", what, nacc, m / (4.0 * cus) / (4000.0 * nacc));
        };
        hipLaunchKernelGGL((mfma_chain<1, false>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, VGPR A/B:", 1);
        hipLaunchKernelGGL((mfma_chain<2, false>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, VGPR A/B:", 2);
        hipLaunchKernelGGL((mfma_chain<3, false>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, VGPR A/B:", 3);
        hipLaunchKernelGGL((mfma_chain<4, false>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, VGPR A/B:", 4);
        hipLaunchKernelGGL((mfma_chain<1, true>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, AGPR A/B:", 1);
        hipLaunchKernelGGL((mfma_chain<2, true>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, AGPR A/B:", 2);
        hipLaunchKernelGGL((mfma_chain<4, true>), dim3(cus), dim3(256), 0, 0, out, cyc2); report("AGPR acc, AGPR A/B:", 4);
    }
    {
        unsigned* cyc; hipMalloc(&cyc, sizeof(unsigned) * 8 * cus);
        static unsigned hc[8 * 1024];
        const char* kinds[] = {"idle", "32-bit integer multiply-add", "FP32 FMA", "FP64 FMA", "LDS reads (ds_read_b64) + FP64 add"};
        const int iters = 20000;
        for (int kind = 0; kind < 5; ++kind) {
            hipMemset(cyc, 0, sizeof(unsigned) * 8 * cus);
            hipLaunchKernelGGL(mfma_partner, dim3(cus), dim3(512), 0, 0, out, cyc, kind, iters);
            hipDeviceSynchronize();
            hipMemcpy(hc, cyc, sizeof(unsigned) * 8 * cus, hipMemcpyDeviceToHost);
            double m = 0, o = 0;
            for (int i = 0; i < cus; ++i) for (int w = 0; w < 4; ++w) { m += hc[i * 8 + w]; o += hc[i * 8 + 4 + w]; }
            m /= 4.0 * cus; o /= 4.0 * cus;
            printf("SIMD partner: %-36s: %.1f cycles per MFMA (partner: %.1f cycles per instruction while it ran)\n", kinds[kind],
                   m / (4.0 * iters), kind ? o / (4.0 * iters * 16) : 0.0);
        }
    }
    return 0;
}

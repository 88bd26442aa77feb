// Stand-alone model of the GEMM phase of the left-looking Cholesky in qp_resident.hpp (n = 514, NB = 32,
// tile-packed L), to compare operand-delivery variants in isolation.  Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_phase_bench.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int N = 514, NB = 32, TSZ = 256, NCH = 34, RT = 1024, RNW = 16, RMAXT = 2;

template <int VAR>
__global__ __launch_bounds__(RT) void gemm_phase(const double* __restrict__ Lall, double* out, int reps) {
    const double* L = Lall + (size_t)blockIdx.x * NCH * NCH * TSZ;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    double total = 0.0;
    for (int rep = 0; rep < reps; ++rep)
    for (int jb = 1; jb < 17; ++jb) {
        const int j0 = jb * NB, R = N - j0, ntile = (R + 15) >> 4, tb = j0 >> 4;
        v4d acc[RMAXT][2];
        for (int u = 0; u < RMAXT; ++u) { acc[u][0] = (v4d){0,0,0,0}; acc[u][1] = (v4d){0,0,0,0}; }
        if (wv < ntile) {
            const int fo = (VAR == 8) ? lane : li * 8 + 2 * kq;
            const int o2 = (VAR == 8) ? 64 : 1;     // second half of the fragment
            auto tile2 = [&](int t, int c) { return reinterpret_cast<const double2*>(L) + (size_t)((t * NCH + c) * (TSZ / 2)); };
            const double2* pb0 = tile2(tb, 0) + fo;
            const double2* pb1 = tile2(tb + 1 < NCH ? tb + 1 : tb, 0) + fo;
            const double2* pa[RMAXT];
            for (int u = 0; u < RMAXT; ++u) { int t = tb + wv + u * RNW; if (t > NCH - 1) t = NCH - 1; pa[u] = tile2(t, 0) + fo; }
            struct Slab { double2 b0a, b0b, b1a, b1b, aa[RMAXT], ab[RMAXT]; };
            auto load = [&](Slab& s_, int c) {
                const int o = c * (TSZ / 2);
                if (VAR == 3) {   // A from memory, B constant
                    s_.b0a = s_.b0b = s_.b1a = s_.b1b = make_double2(1.0 + c, 2.0 + lane);
                    for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = pa[u][o]; s_.ab[u] = pa[u][o + 1]; }
                } else if (VAR == 4) {   // B from memory, A constant
                    s_.b0a = pb0[o]; s_.b0b = pb0[o + 1]; s_.b1a = pb1[o]; s_.b1b = pb1[o + 1];
                    for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = make_double2(3.0, c); s_.ab[u] = make_double2(lane, 4.0); }
                } else if (VAR == 1) {   // no memory traffic
                    s_.b0a = s_.b0b = s_.b1a = s_.b1b = make_double2(1.0 + c, 2.0 + lane);
                    for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = make_double2(3.0, c); s_.ab[u] = make_double2(lane, 4.0); }
                } else {
                    s_.b0a = pb0[o]; s_.b0b = pb0[o + o2]; s_.b1a = pb1[o]; s_.b1b = pb1[o + o2];
                    for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = pa[u][o]; s_.ab[u] = pa[u][o + o2]; }
                }
            };
            auto mult = [&](const Slab& s_) {
                for (int u = 0; u < RMAXT; ++u) if (wv + u * RNW < ntile) {
                    if (VAR == 2) {  // loads only: consume with cheap VALU
                        acc[u][0][0] += s_.aa[u].x * s_.b0a.x + s_.aa[u].y * s_.b0a.y + s_.ab[u].x * s_.b0b.x + s_.ab[u].y * s_.b0b.y;
                        acc[u][1][0] += s_.aa[u].x * s_.b1a.x + s_.aa[u].y * s_.b1a.y + s_.ab[u].x * s_.b1b.x + s_.ab[u].y * s_.b1b.y;
                    } else {
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[u].x, s_.b0a.x, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[u].x, s_.b1a.x, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[u].y, s_.b0a.y, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[u].y, s_.b1a.y, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[u].x, s_.b0b.x, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[u].x, s_.b1b.x, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[u].y, s_.b0b.y, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[u].y, s_.b1b.y, acc[u][1], 0, 0, 0);
                    }
                }
            };
            Slab sa, sb;
            const int nc = 2 * jb;
            load(sa, 0);
            for (int c = 0; c < nc; c += 2) { load(sb, c + 1); mult(sa); if (c + 2 < nc) load(sa, c + 2); mult(sb); }
        }
        for (int u = 0; u < RMAXT; ++u) total += acc[u][0][0] + acc[u][1][1];
        __syncthreads();
    }
    out[blockIdx.x * RT + tid] = total;
}


// LDS-B: the 32 block rows (B operand) are staged once per workgroup into a double-buffered LDS slab of KS
// 16-column chunks; A fragments stream from global memory, register double buffered per slab.
template <int KS>
__global__ __launch_bounds__(RT) void gemm_phase_ldsb(const double* __restrict__ Lall, double* out, int reps) {
    __shared__ double bbuf[2][KS][2][TSZ];          // [buffer][chunk][tile][16x16 swizzled]
    const double* L = Lall + (size_t)blockIdx.x * NCH * NCH * TSZ;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    auto tile2 = [&](int t, int c) { return reinterpret_cast<const double2*>(L) + (size_t)((t * NCH + c) * (TSZ / 2)); };
    double total = 0.0;
    for (int rep = 0; rep < reps; ++rep)
    for (int jb = 1; jb < 17; ++jb) {
        const int j0 = jb * NB, R = N - j0, ntile = (R + 15) >> 4, tb = j0 >> 4;
        const int nc = 2 * jb, nslab = (nc + KS - 1) / KS;
        v4d acc[RMAXT][2];
        for (int u = 0; u < RMAXT; ++u) { acc[u][0] = (v4d){0,0,0,0}; acc[u][1] = (v4d){0,0,0,0}; }
        const bool mine = wv < ntile;
        const int fo = li * 8 + 2 * kq;
        const double2* pa[RMAXT];
        for (int u = 0; u < RMAXT; ++u) { int t = tb + wv + u * RNW; if (t > NCH - 1) t = NCH - 1; pa[u] = tile2(t, 0) + fo; }
        // staging: a slab is KS chunks x 2 tiles x 2 KB = KS*4 one-KB pieces; piece p -> (chunk p>>2, tile (p>>1)&1, half p&1)
        constexpr int NP_ = KS * 4;                  // pieces per slab
        constexpr int PPW = (NP_ + RNW - 1) / RNW;   // pieces per wavefront
        const int frd = li * 16 + 4 * (kq ^ ((li >> 1) & 3));
        double2 breg[PPW];
        auto stage_load = [&](int slab) {
            for (int q = 0; q < PPW; ++q) {
                const int p = wv + q * RNW;
                const int c = slab * KS + (p >> 2);
                if (p < NP_ && c < nc) {
                    const int bt = tb + ((p >> 1) & 1) < NCH ? tb + ((p >> 1) & 1) : NCH - 1;
                    breg[q] = tile2(bt, c)[(p & 1) * 64 + lane];
                }
            }
        };
        auto stage_store = [&](int slab) {
            for (int q = 0; q < PPW; ++q) {
                const int p = wv + q * RNW;
                const int c = slab * KS + (p >> 2);
                if (p < NP_ && c < nc) {
                    const int si = (p & 1) * 8 + (lane >> 3), skq = (lane & 7) >> 1;
                    double* dst = &bbuf[slab & 1][p >> 2][(p >> 1) & 1][si * 16 + 4 * (skq ^ ((si >> 1) & 3)) + 2 * (lane & 1)];
                    *reinterpret_cast<double2*>(dst) = breg[q];
                }
            }
        };
        struct SlabA { double2 aa[KS][RMAXT], ab[KS][RMAXT]; };
        auto loadA = [&](SlabA& s_, int slab) {
            for (int k = 0; k < KS; ++k) {
                const int c = slab * KS + k;
                if (c < nc) for (int u = 0; u < RMAXT; ++u) { s_.aa[k][u] = pa[u][c * (TSZ / 2)]; s_.ab[k][u] = pa[u][c * (TSZ / 2) + 1]; }
            }
        };
        auto mult = [&](const SlabA& s_, int slab) {
            for (int k = 0; k < KS; ++k) {
                const int c = slab * KS + k;
                if (c < nc) {
                    const double* bs = &bbuf[slab & 1][k][0][0];
                    const double2 b0a = *reinterpret_cast<const double2*>(bs + frd), b0b = *reinterpret_cast<const double2*>(bs + frd + 2);
                    const double2 b1a = *reinterpret_cast<const double2*>(bs + TSZ + frd), b1b = *reinterpret_cast<const double2*>(bs + TSZ + frd + 2);
                    for (int u = 0; u < RMAXT; ++u) if (wv + u * RNW < ntile) {
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[k][u].x, b0a.x, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[k][u].x, b1a.x, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[k][u].y, b0a.y, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.aa[k][u].y, b1a.y, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[k][u].x, b0b.x, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[k][u].x, b1b.x, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[k][u].y, b0b.y, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.ab[k][u].y, b1b.y, acc[u][1], 0, 0, 0);
                    }
                }
            }
        };
        SlabA sa, sb;
        stage_load(0); stage_store(0);
        if (nslab > 1) stage_load(1);
        if (mine) loadA(sa, 0);
        __syncthreads();
        for (int sl = 0; sl < nslab; sl += 2) {
            if (mine && sl + 1 < nslab) loadA(sb, sl + 1);
            if (mine) mult(sa, sl);
            if (sl + 1 < nslab) { stage_store(sl + 1); if (sl + 2 < nslab) stage_load(sl + 2); }
            __syncthreads();
            if (sl + 1 < nslab) {
                if (mine && sl + 2 < nslab) loadA(sa, sl + 2);
                if (mine) mult(sb, sl + 1);
                if (sl + 2 < nslab) { stage_store(sl + 2); if (sl + 3 < nslab) stage_load(sl + 3); }
                __syncthreads();
            }
        }
        for (int u = 0; u < RMAXT; ++u) total += acc[u][0][0] + acc[u][1][1];
    }
    out[blockIdx.x * RT + tid] = total;
}
template <int KS>
void run_ldsb(const double* L, double* out, int nwg, const char* name) {
    const int reps = 7;
    gemm_phase_ldsb<KS><<<nwg, RT>>>(L, out, reps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); gemm_phase_ldsb<KS><<<nwg, RT>>>(L, out, reps); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s nwg=%3d: %.3f ms for %d factorizations' GEMM phase = %.0f kcycles each (@2.3GHz)\n", name, nwg, ms, reps, ms * 2.3e3 / reps);
}

template <int VAR>
void run(const double* L, double* out, int nwg, const char* name) {
    const int reps = 7;
    gemm_phase<VAR><<<nwg, RT>>>(L, out, reps);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); gemm_phase<VAR><<<nwg, RT>>>(L, out, reps); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s nwg=%3d: %.3f ms for %d factorizations' GEMM phase = %.0f kcycles each (@2.3GHz)\n", name, nwg, ms, reps, ms * 2.3e3 / reps);
}
int main() {
    const size_t per = (size_t)NCH * NCH * TSZ;
    double *L, *out; hipMalloc(&L, 256 * per * 8); hipMalloc(&out, 256 * RT * 8);
    std::vector<double> h(per); for (size_t i = 0; i < per; ++i) h[i] = 1e-3 * ((i * 2654435761u) % 1000) ;
    for (int b = 0; b < 256; ++b) hipMemcpy(L + b * per, h.data(), per * 8, hipMemcpyHostToDevice);
    for (int nwg : {1, 256}) {
        run<0>(L, out, nwg, "V0 loads+mfma (current)");
        run<1>(L, out, nwg, "V1 mfma only (no loads)");
        run<2>(L, out, nwg, "V2 loads only (no mfma)");
        run<3>(L, out, nwg, "V3 A loads + mfma (B const)");
        run<4>(L, out, nwg, "V4 B loads + mfma (A const)");
        run<8>(L, out, nwg, "V8 contiguous 1KB loads");
        run_ldsb<1>(L, out, nwg, "V5 LDS-B, 16 k per barrier");
        run_ldsb<2>(L, out, nwg, "V6 LDS-B, 32 k per barrier");
        run_ldsb<4>(L, out, nwg, "V7 LDS-B, 64 k per barrier");
    }
    return 0;
}

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
            const int fo = li * 8 + 2 * kq;
            auto tile2 = [&](int t, int c) { return reinterpret_cast<const double2*>(L) + (size_t)((t * NCH + c) * (TSZ / 2)); };
            const double2* pb0 = tile2(tb, 0) + fo;
            const double2* pb1 = tile2(tb + 1 < NCH ? tb + 1 : tb, 0) + fo;
            const double2* pa[RMAXT];
            for (int u = 0; u < RMAXT; ++u) { int t = tb + wv + u * RNW; if (t > NCH - 1) t = NCH - 1; pa[u] = tile2(t, 0) + fo; }
            struct Slab { double2 b0a, b0b, b1a, b1b, aa[RMAXT], ab[RMAXT]; };
            auto load = [&](Slab& s_, int c) {
                const int o = c * (TSZ / 2);
                if (VAR == 1) {   // no memory traffic
                    s_.b0a = s_.b0b = s_.b1a = s_.b1b = make_double2(1.0 + c, 2.0 + lane);
                    for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = make_double2(3.0, c); s_.ab[u] = make_double2(lane, 4.0); }
                } else {
                    s_.b0a = pb0[o]; s_.b0b = pb0[o + 1]; s_.b1a = pb1[o]; s_.b1b = pb1[o + 1];
                    for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = pa[u][o]; s_.ab[u] = pa[u][o + 1]; }
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
    }
    return 0;
}

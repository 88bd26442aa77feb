// Diagnostic: the 16 x 16 "factor and invert" step of the chain wavefront (qp_resident.hpp: cholinv16_blocked) on one wavefront:
// cycles per call (s_memtime), alone and beside a SIMD partner that issues v_mfma_f64_16x16x4 back to back (what wavefront 4's
// history pass does to wavefront 0's chains), and the error against a host computation.  Round 5: variants of the block step
// (VARIANT 1: lean reciprocal square root without the special-value select, minimum pivot instead of four compares).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ihybrid-drt_amd/csrc tools/cholinv16_bench.hip -o tools/cholinv16_bench.bin
#include "qp_resident.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
using namespace hipdrt;

// the library's rsqrt(double) is v_rsq_f64 + one refinement + a select that keeps the raw result for 0 / inf / nan inputs
// (three more dependent instructions per pivot); the pivots here are checked for > 0 anyway
static __device__ __forceinline__ double rsq_lean(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(y0 * -x, y0, 1.0);
    return __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
}

template <int VARIANT>
static __device__ __forceinline__ bool cholinv16_var(const double* D, double* U, int r0, int c0) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int li = lane & 15, kq = lane >> 4;
    v4d aA, aW;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int i = kq + 4 * rg;
        aA[rg] = D[(i > li ? i : li) * DLD + (i > li ? li : i)];
        aW[rg] = (i == li) ? 1.0 : 0.0;
    }
    double pmin = 1.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double tA = aA[k], tW = aW[k];
        auto dg = [&](int a_, int b_) { return bcast_lane(tA, 4 * k + b_ + 16 * a_); };
        const double d00 = dg(0, 0), d10 = dg(1, 0), d20 = dg(2, 0), d30 = dg(3, 0);
        const double d11 = dg(1, 1), d21 = dg(2, 1), d31 = dg(3, 1), d22 = dg(2, 2), d32 = dg(3, 2), d33 = dg(3, 3);
        const double i0 = rsq_lean(d00);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double p1 = d11 - l10 * l10;
        const double i1 = rsq_lean(p1);
        const double l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
        const double p2 = d22 - l20 * l20 - l21 * l21;
        const double i2 = rsq_lean(p2);
        const double l32 = (d32 - l30 * l20 - l31 * l21) * i2;
        const double p3 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
        const double i3 = rsq_lean(p3);
        pmin = fmin(fmin(pmin, fmin(d00, p1)), fmin(p2, p3));       // NaN-safe enough: a NaN pivot poisons everything after it
        const double w10 = -(l10 * i0) * i1;
        const double w21 = -(l21 * i1) * i2, w20 = -(l20 * i0 + l21 * w10) * i2;
        const double w32 = -(l32 * i2) * i3, w31 = -(l31 * i1 + l32 * w21) * i3, w30 = -(l30 * i0 + l31 * w10 + l32 * w20) * i3;
        const double c0_ = li == 0 ? i0 : li == 1 ? w10 : li == 2 ? w20 : w30;
        const double c1_ = li == 1 ? i1 : li == 2 ? w21 : w31;
        const double c2_ = li == 2 ? i2 : w32;
        double wsel = kq == 0 ? c0_ : kq == 1 ? c1_ : kq == 2 ? c2_ : i3;
        wsel = (li < 4 && kq <= li) ? wsel : 0.0;
        const v4d z4 = (v4d){0, 0, 0, 0};
        const v4d X = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, tA, z4, 0, 0, 0);
        const v4d Y = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, tW, z4, 0, 0, 0);
        U[(size_t)(r0 + 4 * k + kq) * PLD + c0 + li] = (li <= 4 * k + kq) ? Y[0] : 0.0;
        if (k < 3) {
            const double pan = (li > 4 * k + 3) ? -X[0] : 0.0;
            aA = __builtin_amdgcn_mfma_f64_16x16x4f64(pan, X[0], aA, 0, 0, 0);
            aW = __builtin_amdgcn_mfma_f64_16x16x4f64(pan, Y[0], aW, 0, 0, 0);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return pmin > 0.0;
}

// wave 0: the chain; wave `partner` (4 = same SIMD as wave 0 under the w % 4 placement, 5 = another SIMD, -1 = none): MFMAs back to back
__global__ __launch_bounds__(512) void bench(const double* Din, double* Wout, unsigned long long* out, int reps, int mode, int partner) {
    __shared__ double dsc[16 * DLD];
    __shared__ double U[32 * PLD];
    __shared__ int stop;
    OpsResidentT<false, 512> ops;
    ops.sm.dsc = dsc; ops.sm.U = U;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {
        for (int i = lane; i < 256; i += 64) dsc[(i / 16) * DLD + i % 16] = Din[i];
        for (int i = lane; i < 32 * PLD; i += 64) U[i] = -77.0;
        if (lane == 0) stop = 0;
    }
    __syncthreads();
    if (wv == 0) {
        bool ok = true;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int r = 0; r < reps; ++r) {
            ok = (mode == 0 ? ops.cholinv16_blocked(0, 0) : cholinv16_var<1>(dsc, U, 0, 0)) && ok;
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[0] = (t1 - t0) / reps; out[1] = ok ? 1 : 0; __hip_atomic_store(&stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        for (int i = lane; i < 256; i += 64) Wout[i] = U[(i / 16) * PLD + i % 16];
    } else if (wv == partner) {
        v4d a0 = (v4d){0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = 1.0 + lane * 1e-9, y = 1.0 - lane * 1e-9;
        unsigned long long n_ = 0;
        while (!__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
            }
            n_ += 32;
        }
        if (lane == 0) { out[2] = n_; out[3] = (unsigned long long)(a0[0] + a1[1] + a2[2] + a3[3]); }
    }
}

int main() {
    std::mt19937_64 rng(3);
    std::normal_distribution<double> nd;
    std::vector<double> A(16 * 20), D(256), L(256, 0.0), W(256, 0.0);
    for (auto& v : A) v = nd(rng);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = (i == j) ? 0.5 : 0.0;
            for (int k = 0; k < 20; ++k) s += A[i * 20 + k] * A[j * 20 + k];
            D[i * 16 + j] = s;
        }
    for (int j = 0; j < 16; ++j) {                     // host reference: L, then W = L^-1
        double s = D[j * 16 + j];
        for (int k = 0; k < j; ++k) s -= L[j * 16 + k] * L[j * 16 + k];
        L[j * 16 + j] = std::sqrt(s);
        for (int i = j + 1; i < 16; ++i) {
            double t = D[i * 16 + j];
            for (int k = 0; k < j; ++k) t -= L[i * 16 + k] * L[j * 16 + k];
            L[i * 16 + j] = t / L[j * 16 + j];
        }
    }
    for (int c = 0; c < 16; ++c)
        for (int i = c; i < 16; ++i) {
            double t = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) t -= L[i * 16 + k] * W[k * 16 + c];
            W[i * 16 + c] = t / L[i * 16 + i];
        }
    double *dD, *dW; unsigned long long* dO;
    hipMalloc(&dD, 256 * 8); hipMalloc(&dW, 256 * 8); hipMalloc(&dO, 64);
    hipMemcpy(dD, D.data(), 256 * 8, hipMemcpyHostToDevice);
    std::vector<double> W0(256);
    for (int mode = 0; mode < 2; ++mode)
        for (int partner : {-1, 4, 5, 2}) {
            std::vector<double> Wd(256);
            unsigned long long o[4] = {0, 0, 0, 0};
            hipMemset(dO, 0, 64);
            hipLaunchKernelGGL(bench, dim3(1), dim3(512), 0, 0, dD, dW, dO, 2000, mode, partner);
            hipDeviceSynchronize();
            hipMemcpy(Wd.data(), dW, 256 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(o, dO, 32, hipMemcpyDeviceToHost);
            if (mode == 0 && partner == -1) W0 = Wd;
            double err = 0, mx = 0, up = 0, dv = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    mx = std::fmax(mx, std::fabs(W[i * 16 + j]));
                    if (j <= i) err = std::fmax(err, std::fabs(Wd[i * 16 + j] - W[i * 16 + j]));
                    else up = std::fmax(up, std::fabs(Wd[i * 16 + j]));
                    dv = std::fmax(dv, std::fabs(Wd[i * 16 + j] - W0[i * 16 + j]));
                }
            printf("%s, MFMA partner wave %2d: %5llu cycles per call (partner issued %llu MFMAs), ok %llu, max |W - W_host| / max |W| = %.2e, "
                   "upper triangle %.1e, max |W - W_library_form| = %.1e\n", mode ? "lean variant  " : "library form  ", partner, o[0], o[2], o[1],
                   err / mx, up, dv);
        }
    return 0;
}

// Kernel-matrix builders for gfx950 (compiled with -ffp-contract=off so that the interpolation
// arithmetic rounds exactly like numpy's un-fused slope*(x-xp[j])+fp[j]).
//
//  lookup_kernel          basis.generate_impedance_lookup        hybdrt/matrices/basis.py:648-669
//  impedance_*_kernel     mat1d.construct_impedance_matrix        hybdrt/matrices/mat1d.py:212-374
//  response_lookup_kernel basis.generate_response_lookup          hybdrt/matrices/basis.py:672-689, 616-618
//  response_*_kernel      mat1d.construct_response_matrix         hybdrt/matrices/mat1d.py:16-122
//  penalty_kernel         mat1d.construct_integrated_derivative_matrix  mat1d.py:125-209, basis.py:382-395
//  eis_vmm_kernel         mat1d.construct_eis_var_matrix          mat1d.py:493-515
//  phasor_z_kernel / phasor_v_kernel  phasance.construct_phasor_z_matrix / _v_matrix   hybdrt/matrices/phasance.py:108-144
//  chrono_vmm_kernel      mat1d.construct_chrono_var_matrix       mat1d.py:457-490 (transformed times: utils/chrono.py:5-44)
//
// Layout: every matrix row-major float64.  The interp build is HBM-write bound (2*nf*ntau*8 B per
// frequency grid); the lookup tables (3 arrays x 2 parts x ngrid x 8 B = 96 kB at ngrid=2000) are staged
// once per workgroup into LDS, each thread produces two adjacent tau columns of both parts and stores them
// as 16-byte vectors, so a wavefront writes 1 KiB contiguous per store instruction.
#include "common.hpp"

namespace hipdrt {


// ---------------------------------------------------------------------------------------------------------
// integrands (basis.py:93-95, 565-570) evaluated for one y; lw = log(w*t), wt = w*t
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void integrand(double phi, double ey, double y, double lw, double w, double t,
                                          double& fre, double& fim) {
    const double den = 1.0 + exp(2.0 * (y + lw));
    fre = phi / den;
    fim = (((-phi) * ey) * w) * t / den;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wavefront integrates one (w, t) pair with the ny-point trapezoid rule of np.trapezoid:
// sum_j d_j * (f_{j+1} + f_j) / 2 on y = linspace(-20, 20, ny).  ys/phis/eys are LDS tables [ny].
__device__ __forceinline__ void trapz_pair(const double* ys, const double* phis, const double* eys, int ny,
                                           double w, double t, int lane, double& zre, double& zim) {
    const double lw = log(w * t);
    double sre = 0.0, sim = 0.0;
    for (int j = lane; j < ny - 1; j += 64) {
        double f0r, f0i, f1r, f1i;
        integrand(phis[j], eys[j], ys[j], lw, w, t, f0r, f0i);
        integrand(phis[j + 1], eys[j + 1], ys[j + 1], lw, w, t, f1r, f1i);
        const double d = ys[j + 1] - ys[j];
        sre += d * (f1r + f0r) / 2.0;
        sim += d * (f1i + f0i) / 2.0;
    }
    zre = wave_sum(sre);
    zim = wave_sum(sim);
}

__device__ __forceinline__ void fill_y_tables(double* ys, double* phis, double* eys, int ny, double eps) {
    const double step = 40.0 / (double)(ny - 1);   // np.linspace(-20, 20, ny)
    for (int j = threadIdx.x; j < ny; j += blockDim.x) {
        double y = (double)j * step + (-20.0);
        if (j == ny - 1) y = 20.0;
        ys[j] = y;
        const double ey2 = eps * y;
        phis[j] = exp(-(ey2 * ey2));
        eys[j] = exp(y);
    }
}

// grid: ceil(2*ngrid / 4) blocks of 256 threads; wave g handles table entry g (re for g < ngrid, else im)
__global__ __launch_bounds__(256) void lookup_kernel(double eps, int ngrid, int ny, const double* __restrict__ wt_re,
                                                     const double* __restrict__ wt_im, double* __restrict__ z_re,
                                                     double* __restrict__ z_im) {
    extern __shared__ double sm[];
    double* ys = sm;
    double* phis = sm + ny;
    double* eys = sm + 2 * ny;
    fill_y_tables(ys, phis, eys, ny, eps);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= 2 * ngrid) return;
    const bool im = g >= ngrid;
    const int i = im ? g - ngrid : g;
    const double wt = im ? wt_im[i] : wt_re[i];
    double zr, zi;
    trapz_pair(ys, phis, eys, ny, wt, 1.0, lane, zr, zi);
    if (lane == 0) {
        if (im) z_im[i] = zi; else z_re[i] = zr;
    }
}

__global__ void slopes_kernel(int ngrid, const double* __restrict__ xp, const double* __restrict__ fp,
                              double* __restrict__ slopes) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ngrid - 1) slopes[j] = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    else if (j == ngrid - 1) slopes[j] = 0.0;
}

// np.interp(x, xp, fp) with numpy's semantics (end clamping, exact-knot shortcut, un-fused evaluation).
// xp is log(logspace(..)), i.e. uniform to ~1e-15, so the bin is found arithmetically and then verified
// against the true knots (result identical to numpy's binary search).
__device__ __forceinline__ double np_interp(double x, const double* xp, const double* fp, const double* sl,
                                            int ng, double x0, double inv_dx) {
    if (x > xp[ng - 1]) return fp[ng - 1];
    if (x < xp[0]) return fp[0];
    int j = (int)((x - x0) * inv_dx);
    j = j < 0 ? 0 : (j > ng - 1 ? ng - 1 : j);
    while (j > 0 && xp[j] > x) --j;
    while (j < ng - 1 && xp[j + 1] <= x) ++j;
    if (j == ng - 1) return fp[j];
    const double xj = xp[j];
    if (xj == x) return fp[j];
    return sl[j] * (x - xj) + fp[j];
}

// ln(2 pi f) per (grid, frequency) and ln(tau): the general build evaluates ln(omega tau) as their sum (one add
// per entry instead of one FP64 log; differs from log(omega*tau) by <= 2 ulp of the abscissa, i.e. ~1e-16
// relative in the interpolated value, and not at all in the clamped regions).
__global__ void log_grid_kernel(int count_f, const double* __restrict__ freq, int ntau, const double* __restrict__ tau,
                                double* __restrict__ lw, double* __restrict__ lt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count_f) lw[i] = log(freq[i] * 2.0 * 3.141592653589793);
    if (i < ntau) lt[i] = log(tau[i]);
}

// LDS image of one lookup for the general build: knots x[ng] and (value, slope) pairs fs[ng], so an interpolation is one
// 16-byte pair of knots (ds_read2_b64) plus one 16-byte (f, s) read instead of four to six dependent 8-byte reads
struct LutFS {
    const double* x;
    const double2* fs;
    double x_lo, x_hi, f_lo, f_hi, x0, inv_dx;
    int ng;
};

__device__ __forceinline__ LutFS stage_lut_fs(double* sm_x, double2* sm_fs, const double* __restrict__ xp,
                                              const double* __restrict__ fp, const double* __restrict__ sl, int ng) {
    for (int i = threadIdx.x; i < ng; i += blockDim.x) {
        sm_x[i] = xp[i];
        sm_fs[i] = make_double2(fp[i], sl[i]);
    }
    return LutFS{sm_x, sm_fs, xp[0], xp[ng - 1], fp[0], fp[ng - 1], xp[0], (double)(ng - 1) / (xp[ng - 1] - xp[0]), ng};
}

// Two np.interp evaluations on one table, written so that their LDS reads are independent and issue together.  bins2:
// arithmetic bin (the knots are log(logspace(..)), uniform to ~1e-15), checked against the true knots; the search only runs
// for a lane that sits on a rounding edge, so the result is numpy's for any increasing grid.  eval2: value from (f, s).
struct Bins2 { int ja, jb; double a0, b0; };

__device__ __forceinline__ Bins2 bins2(const LutFS& T, double xa, double xb) {
    int ja = (int)((xa - T.x0) * T.inv_dx), jb = (int)((xb - T.x0) * T.inv_dx);
    ja = min(max(ja, 0), T.ng - 2);
    jb = min(max(jb, 0), T.ng - 2);
    double a0 = T.x[ja], a1 = T.x[ja + 1], b0 = T.x[jb], b1 = T.x[jb + 1];
    if (__builtin_expect((a0 > xa) | (a1 <= xa) | (b0 > xb) | (b1 <= xb), 0)) {
        while (ja > 0 && T.x[ja] > xa) --ja;
        while (ja < T.ng - 2 && T.x[ja + 1] <= xa) ++ja;
        while (jb > 0 && T.x[jb] > xb) --jb;
        while (jb < T.ng - 2 && T.x[jb + 1] <= xb) ++jb;
        a0 = T.x[ja];
        b0 = T.x[jb];
    }
    return Bins2{ja, jb, a0, b0};
}

__device__ __forceinline__ void eval2(const LutFS& T, const Bins2& k, double xa, double xb, double& va, double& vb) {
    const double2 fa = T.fs[k.ja], fb = T.fs[k.jb];
    const double la = (k.a0 == xa) ? fa.x : fa.y * (xa - k.a0) + fa.x;
    const double lb = (k.b0 == xb) ? fb.x : fb.y * (xb - k.b0) + fb.x;
    va = xa >= T.x_hi ? T.f_hi : (xa < T.x_lo ? T.f_lo : la);
    vb = xb >= T.x_hi ? T.f_hi : (xb < T.x_lo ? T.f_lo : lb);
}

// INTERP, general (non-Toeplitz) build.  grid = (row chunks, B), 1024 threads: the 96 kB of tables are staged
// once per workgroup and amortised over `rows_per_block` rows; thread (tr, tc) walks rows tr, tr+RP, ... and owns
// two adjacent tau columns, so a wavefront stores 1 KiB contiguous per matrix per row.
__global__ __launch_bounds__(1024) void impedance_interp_kernel(
    int freq_batched, const double* __restrict__ lw, int nf, const double* __restrict__ lt, int ntau, int ng,
    const double* __restrict__ lut6, int rows_per_block, int tpr, double* __restrict__ a_re,
    double* __restrict__ a_im) {
    extern __shared__ double sm[];
    // LDS: x_re[ng] x_im[ng] | fs_re[ng] fs_im[ng] (16-byte aligned: the dynamic segment starts aligned and 2 ng doubles
    // precede the pairs only when ng is even; odd ng gets one pad double)
    const int npad = ng + (ng & 1);
    double* sx = sm;
    double2* sfs = reinterpret_cast<double2*>(sm + 2 * npad);
    const LutFS Tr = stage_lut_fs(sx, sfs, lut6, lut6 + ng, lut6 + 2 * ng, ng);
    const LutFS Ti = stage_lut_fs(sx + npad, sfs + ng, lut6 + 3 * ng, lut6 + 4 * ng, lut6 + 5 * ng, ng);
    const int b = blockIdx.y;
    const double* lwb = lw + (freq_batched ? (size_t)b * nf : 0);
    const int row0 = blockIdx.x * rows_per_block;
    const int rend = min(row0 + rows_per_block, nf);
    // ln(omega) of this workgroup's rows: an LDS read per row instead of a global load in front of every dependent chain
    double* slw = sm + 6 * npad;
    for (int i = threadIdx.x; i < rend - row0; i += blockDim.x) slw[i] = lwb[row0 + i];
    // Z' and Z'' tabulated on the same abscissae (generate_impedance_lookup gives both one grid): one bin search serves both
    int mine = 1;
    for (int i = threadIdx.x; i < ng; i += blockDim.x) mine &= (lut6[i] == lut6[3 * ng + i]);
    const bool same_x = __syncthreads_and(mine) != 0;
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr, rp = blockDim.x / tpr;
    const bool even = (ntau & 1) == 0;
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    for (int c = 2 * tc; c < ntau; c += 2 * tpr) {
        const bool two = (c + 1 < ntau);
        const double t0 = lt[c], t1 = two ? lt[c + 1] : lt[c];
        auto entry = [&](int r) {
            const double w = slw[r - row0];
            const double xa = w + t0, xb = w + t1;
            double re0, re1, im0, im1;
            Bins2 k = bins2(Tr, xa, xb);
            eval2(Tr, k, xa, xb, re0, re1);
            if (!same_x) k = bins2(Ti, xa, xb);
            eval2(Ti, k, xa, xb, im0, im1);
            const size_t o = ((size_t)b * nf + r) * ntau + c;
            if (two) {
                if (even) {
                    // written once, read later by another kernel: non-temporal 16-byte stores (plain stores measured
                    // 2-4 % slower; with the interpolation replaced by constants this mapping writes 5.0-5.9 TB/s)
                    const v2d_t vre = {re0, re1}, vim = {im0, im1};
                    __builtin_nontemporal_store(vre, reinterpret_cast<v2d_t*>(a_re + o));
                    __builtin_nontemporal_store(vim, reinterpret_cast<v2d_t*>(a_im + o));
                } else {
                    a_re[o] = re0; a_re[o + 1] = re1;
                    a_im[o] = im0; a_im[o + 1] = im1;
                }
            } else {
                a_re[o] = re0;
                a_im[o] = im0;
            }
        };
        int r = row0 + tr;
        for (; r + rp < rend; r += 2 * rp) { entry(r); entry(r + rp); }      // two rows in flight per pass
        if (r < rend) entry(r);
    }
}

// first column c[nf] (omega_n * tau_0) and first row r[ntau] (omega_0 * tau_m) of the Toeplitz shortcut
// (mat1d.py:353-360).  cr = {c_re[nf], r_re[ntau], c_im[nf], r_im[ntau]}.
// INTERP: one thread per entry; TRAPZ: one wavefront per entry.
__global__ void toeplitz_cr_interp_kernel(const double* __restrict__ freq, int nf, const double* __restrict__ tau,
                                          int ntau, int ng, const double* __restrict__ lut6, double* __restrict__ cr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf + ntau) return;
    const double* xr = lut6; const double* fr = lut6 + ng; const double* sr = lut6 + 2 * ng;
    const double* xi = lut6 + 3 * ng; const double* fi = lut6 + 4 * ng; const double* si = lut6 + 5 * ng;
    const double idr = (double)(ng - 1) / (xr[ng - 1] - xr[0]);
    const double idi = (double)(ng - 1) / (xi[ng - 1] - xi[0]);
    double w, t;
    if (i < nf) { w = freq[i] * 2.0 * 3.141592653589793; t = tau[0]; }
    else { w = freq[0] * 2.0 * 3.141592653589793; t = tau[i - nf]; }
    const double x = log(w * t);
    cr[i] = np_interp(x, xr, fr, sr, ng, xr[0], idr);
    cr[nf + ntau + i] = np_interp(x, xi, fi, si, ng, xi[0], idi);
}

__global__ __launch_bounds__(256) void toeplitz_cr_trapz_kernel(const double* __restrict__ freq, int nf,
                                                                const double* __restrict__ tau, int ntau, double eps,
                                                                int ny, double* __restrict__ cr) {
    extern __shared__ double sm[];
    double* ys = sm; double* phis = sm + ny; double* eys = sm + 2 * ny;
    fill_y_tables(ys, phis, eys, ny, eps);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nf + ntau) return;
    double w, t;
    if (i < nf) { w = freq[i] * 2.0 * 3.141592653589793; t = tau[0]; }
    else { w = freq[0] * 2.0 * 3.141592653589793; t = tau[i - nf]; }
    double zr, zi;
    trapz_pair(ys, phis, eys, ny, w, t, lane, zr, zi);
    if (lane == 0) { cr[i] = zr; cr[nf + ntau + i] = zi; }
}

// scipy.linalg.toeplitz(c, r): A[i][j] = c[i-j] (i >= j) else r[j-i]
__global__ void toeplitz_fill_kernel(int B, int nf, int ntau, const double* __restrict__ cr, double* __restrict__ a_re,
                                     double* __restrict__ a_im) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= ntau) return;
    const double* c_re = cr; const double* r_re = cr + nf;
    const double* c_im = cr + nf + ntau; const double* r_im = c_im + nf;
    const double vr = (i >= j) ? c_re[i - j] : r_re[j - i];
    const double vi = (i >= j) ? c_im[i - j] : r_im[j - i];
    for (int b = blockIdx.z; b < B; b += gridDim.z) {
        const size_t o = ((size_t)b * nf + i) * ntau + j;
        a_re[o] = vr;
        a_im[o] = vi;
    }
}

// TRAPZ, general: one wavefront per matrix entry, 4 entries per 256-thread block along tau.
__global__ __launch_bounds__(256) void impedance_trapz_kernel(int freq_batched, const double* __restrict__ freq, int nf,
                                                              const double* __restrict__ tau, int ntau, double eps, int ny,
                                                              double* __restrict__ a_re, double* __restrict__ a_im) {
    extern __shared__ double sm[];
    double* ys = sm; double* phis = sm + ny; double* eys = sm + 2 * ny;
    fill_y_tables(ys, phis, eys, ny, eps);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int r = blockIdx.y, b = blockIdx.z;
    const double* fr = freq + (freq_batched ? (size_t)b * nf : 0);
    const double w = fr[r] * 2.0 * 3.141592653589793;
    // each block covers 16 consecutive tau columns
    for (int k = 0; k < 4; ++k) {
        const int c = blockIdx.x * 16 + k * 4 + wv;
        if (c >= ntau) continue;
        double zr, zi;
        trapz_pair(ys, phis, eys, ny, w, tau[c], lane, zr, zi);
        if (lane == 0) {
            const size_t o = ((size_t)b * nf + r) * ntau + c;
            a_re[o] = zr;
            a_im[o] = zi;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// penalty matrices: orders 0..2 in one pass.  out matrices have leading dimension ld and the DRT block
// starts at (pad, pad) (pad = number of special parameters in the plan; 0 for the stand-alone API).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void penalty_funcs(double x_n, double x_m, double eps, double& f0, double& f1, double& f2) {
    const double a = eps * (x_m - x_n);
    const double a2 = a * a;
    const double e = exp(-(a2 / 2.0));
    const double rpi = sqrt(3.141592653589793 / 2.0);
    f0 = rpi * (1.0 / eps) * e;
    f1 = (-rpi) * eps * (-1.0 + a2) * e;
    f2 = rpi * (eps * eps * eps) * (3.0 - 6.0 * a2 + a2 * a2) * e;
}

__global__ void penalty_kernel(const double* __restrict__ ln_tau, int n, double eps, int toeplitz,
                               double* __restrict__ m0, double* __restrict__ m1, double* __restrict__ m2, int ld,
                               int pad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    double f0, f1, f2;
    if (toeplitz) {
        const int dd = i > j ? i - j : j - i;
        penalty_funcs(ln_tau[dd], ln_tau[0], eps, f0, f1, f2);   // c[d] = func(x_d, x_0)
    } else {
        penalty_funcs(ln_tau[i], ln_tau[j], eps, f0, f1, f2);
    }
    const size_t o = (size_t)(i + pad) * ld + (j + pad);
    m0[o] = f0;
    m1[o] = f1;
    m2[o] = f2;
}

// ---------------------------------------------------------------------------------------------------------
// Time response of the Gaussian basis to ideal galvanostatic steps (chrono / hybrid fits)
// ---------------------------------------------------------------------------------------------------------
// integrand (basis.py:616-618): phi(y) * (1 - exp(-t / (tau * e^y))); one wavefront integrates one (t, tau) pair
__device__ __forceinline__ double trapz_response(const double* ys, const double* phis, const double* eys, int ny,
                                                 double tau, double t, int lane) {
    double s = 0.0;
    for (int j = lane; j < ny - 1; j += 64) {
        const double f0 = phis[j] * (1.0 - exp(-t / (tau * eys[j])));
        const double f1 = phis[j + 1] * (1.0 - exp(-t / (tau * eys[j + 1])));
        const double d = ys[j + 1] - ys[j];
        s += d * (f1 + f0) / 2.0;
    }
    return wave_sum(s);
}

// v[i] = trapz over y of the integrand at t/tau = td[i]; grid: ceil(ngrid / 4) blocks of 256 threads
__global__ __launch_bounds__(256) void response_lookup_kernel(double eps, int ngrid, int ny,
                                                              const double* __restrict__ td, double* __restrict__ v) {
    extern __shared__ double sm[];
    double* ys = sm; double* phis = sm + ny; double* eys = sm + 2 * ny;
    fill_y_tables(ys, phis, eys, ny, eps);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ngrid) return;
    const double r = trapz_response(ys, phis, eys, ny, 1.0, td[g], lane);
    if (lane == 0) v[g] = r;
}

// interp mode: rows after a step hold np.interp(ln((t - t_k)/tau), log_td, v) * size_k, earlier rows 0; A = sum over
// the steps in order.  lut3 = {log_td, v, slope}[ngrid] staged in LDS; block = 256 columns x rows_per_block rows.
__global__ __launch_bounds__(256) void response_interp_kernel(const double* __restrict__ times, int nt,
                                                              const double* __restrict__ tau, int ntau,
                                                              const double* __restrict__ step_times,
                                                              const double* __restrict__ step_sizes, int nsteps,
                                                              int ngrid, const double* __restrict__ lut3, int rpb,
                                                              double* __restrict__ a, double* __restrict__ layered) {
    extern __shared__ double sm[];
    for (int i = threadIdx.x; i < 3 * ngrid; i += blockDim.x) sm[i] = lut3[i];
    __syncthreads();
    const double* xp = sm; const double* fp = sm + ngrid; const double* sl = sm + 2 * ngrid;
    const double x0 = xp[0], inv_dx = (double)(ngrid - 1) / (xp[ngrid - 1] - xp[0]);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ntau) return;
    const double tc = tau[c];
    const int r1 = min(nt, (int)(blockIdx.y + 1) * rpb);
    for (int r = blockIdx.y * rpb; r < r1; ++r) {
        const double t = times[r];
        double acc = 0.0;
        for (int k = 0; k < nsteps; ++k) {
            const double st = step_times[k];
            double val = 0.0;
            if (t > st) val = np_interp(log((t - st) / tc), xp, fp, sl, ngrid, x0, inv_dx) * step_sizes[k];
            if (layered) layered[((size_t)k * nt + r) * ntau + c] = val;
            acc += val;
        }
        a[(size_t)r * ntau + c] = acc;
    }
}

// trapz mode: one wavefront per entry, ny-point trapezoid per step; grid (ceil(ntau/16), nt)
__global__ __launch_bounds__(256) void response_trapz_kernel(const double* __restrict__ times, int nt,
                                                             const double* __restrict__ tau, int ntau,
                                                             const double* __restrict__ step_times,
                                                             const double* __restrict__ step_sizes, int nsteps,
                                                             double eps, int ny, double* __restrict__ a,
                                                             double* __restrict__ layered) {
    extern __shared__ double sm[];
    double* ys = sm; double* phis = sm + ny; double* eys = sm + 2 * ny;
    fill_y_tables(ys, phis, eys, ny, eps);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.y;
    const double t = times[r];
    for (int q = 0; q < 4; ++q) {
        const int c = blockIdx.x * 16 + q * 4 + wv;
        if (c >= ntau) continue;
        double acc = 0.0;
        for (int k = 0; k < nsteps; ++k) {
            const double st = step_times[k];
            double val = 0.0;
            if (t > st) val = trapz_response(ys, phis, eys, ny, tau[c], t - st, lane) * step_sizes[k];
            if (layered && lane == 0) layered[((size_t)k * nt + r) * ntau + c] = val;
            acc += val;
        }
        if (lane == 0) a[(size_t)r * ntau + c] = acc;
    }
}


// ---- the non-default forms of construct_response_matrix (mat1d.py:96-118) --------------------------------------------------
// expdecay step model (basis.py:619-637): the current rises as 1 - exp(-t / tau_rise); integrand
//   phi(y) * (1 - e^{-t/T} + tau_rise / (tau_rise - T) * (e^{-t/T} - e^{-t/tau_rise})),  T = e^y tau
__device__ __forceinline__ double trapz_response_expdecay(const double* ys, const double* phis, const double* eys, int ny,
                                                          double tau, double t, double tr, int lane) {
    const double etr = exp(-t / tr);
    auto f = [&](int j) {
        const double T = eys[j] * tau;
        const double eT = exp(-t / T);
        return phis[j] * ((1.0 - eT) + (tr / (tr - T)) * (eT - etr));
    };
    double s = 0.0;
    for (int j = lane; j < ny - 1; j += 64) {
        const double d = ys[j + 1] - ys[j];
        s += d * (f(j + 1) + f(j)) / 2.0;
    }
    return wave_sum(s);
}

// variant 1 (expdecay, trapz): one wavefront per entry like response_trapz_kernel, tau_rise[k] per step
__global__ __launch_bounds__(256) void response_expdecay_kernel(const double* __restrict__ times, int nt,
                                                                const double* __restrict__ tau, int ntau,
                                                                const double* __restrict__ step_times,
                                                                const double* __restrict__ step_sizes,
                                                                const double* __restrict__ tau_rise, int nsteps,
                                                                double eps, int ny, double* __restrict__ a,
                                                                double* __restrict__ layered) {
    extern __shared__ double sm[];
    double* ys = sm; double* phis = sm + ny; double* eys = sm + 2 * ny;
    fill_y_tables(ys, phis, eys, ny, eps);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.y;
    const double t = times[r];
    for (int q = 0; q < 4; ++q) {
        const int c = blockIdx.x * 16 + q * 4 + wv;
        if (c >= ntau) continue;
        double acc = 0.0;
        for (int k = 0; k < nsteps; ++k) {
            const double st = step_times[k];
            double val = 0.0;
            if (t > st) val = trapz_response_expdecay(ys, phis, eys, ny, tau[c], t - st, tau_rise[k], lane) * step_sizes[k];
            if (layered && lane == 0) layered[((size_t)k * nt + r) * ntau + c] = val;
            acc += val;
        }
        if (lane == 0) a[(size_t)r * ntau + c] = acc;
    }
}

// variant 0 (potentiostatic, mat1d.py:114-118): the basis is a delta function, the current after a voltage step decays as
// exp(-(t - t_k) / tau) from the step on (unit_step: t >= t_k); rows before the step are 0 (the reference's nan_to_num)
__global__ __launch_bounds__(256) void response_pot_kernel(const double* __restrict__ times, int nt,
                                                           const double* __restrict__ tau, int ntau,
                                                           const double* __restrict__ step_times,
                                                           const double* __restrict__ step_sizes, int nsteps,
                                                           double* __restrict__ a, double* __restrict__ layered) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= ntau) return;
    const double t = times[r], tc = tau[c];
    double acc = 0.0;
    for (int k = 0; k < nsteps; ++k) {
        const double st = step_times[k];
        double val = 0.0;
        if (t >= st) val = exp(-(t - st) / tc) * step_sizes[k];
        if (layered) layered[((size_t)k * nt + r) * ntau + c] = val;
        acc += val;
    }
    a[(size_t)r * ntau + c] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// EIS variance-estimation matrix: one 256-thread block per row of the (2nf x 2nf) matrix
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eis_vmm_kernel(const double* __restrict__ freq, int nf, double ve, double cor,
                                                      int uniform, double* __restrict__ vmm) {
    __shared__ double red[4];
    const int i = blockIdx.x;          // row in [0, 2nf)
    const int ih = i >= nf ? i - nf : i;
    const int m = 2 * nf;
    const double lfi = log(freq[ih]);
    double s = 0.0;
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const int jh = j >= nf ? j - nf : j;
        double v = 1.0;
        if (!uniform) {
            const double dd = ve * (lfi - log(freq[jh]));
            v = exp(-(dd * dd));
        }
        if ((i >= nf) != (j >= nf)) v = v * cor;
        vmm[(size_t)i * m + j] = v;
        s += v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int j = threadIdx.x; j < m; j += blockDim.x) vmm[(size_t)i * m + j] /= tot;
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
void launch_lookup(hipStream_t st, double eps, int ngrid, int ny, const double* wt_re, const double* wt_im,
                   double* z_re, double* z_im) {
    const int blocks = (2 * ngrid + 3) / 4;
    hipLaunchKernelGGL(lookup_kernel, dim3(blocks), dim3(256), 3 * ny * sizeof(double), st, eps, ngrid, ny, wt_re,
                       wt_im, z_re, z_im);
}

void launch_lookup_slopes(hipStream_t st, int ngrid, const double* xp, const double* fp, double* slopes) {
    hipLaunchKernelGGL(slopes_kernel, dim3((ngrid + 255) / 256), dim3(256), 0, st, ngrid, xp, fp, slopes);
}

void launch_impedance_matrix(hipStream_t st, int B, int freq_batched, const double* freq, int nf, const double* tau,
                             int ntau, int mode, int toeplitz, double eps, int ngrid, const double* lut6, int ny,
                             double* a_re, double* a_im, double* cr_scratch) {
    if (toeplitz) {
        const int tot = nf + ntau;
        if (mode == HIPDRT_MODE_INTERP)
            hipLaunchKernelGGL(toeplitz_cr_interp_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, freq, nf, tau, ntau,
                               ngrid, lut6, cr_scratch);
        else
            hipLaunchKernelGGL(toeplitz_cr_trapz_kernel, dim3((tot + 3) / 4), dim3(256), 3 * ny * sizeof(double), st,
                               freq, nf, tau, ntau, eps, ny, cr_scratch);
        const int zb = B < 64 ? B : 64;
        hipLaunchKernelGGL(toeplitz_fill_kernel, dim3((ntau + 255) / 256, nf, zb), dim3(256), 0, st, B, nf, ntau,
                           cr_scratch, a_re, a_im);
        return;
    }
    if (mode == HIPDRT_MODE_INTERP) {
        // cr_scratch doubles as storage for ln(omega) [B*nf] and ln(tau) [ntau] (sized by the caller)
        double* lw = cr_scratch;
        const int count_f = (freq_batched ? B : 1) * nf;
        double* lt = cr_scratch + count_f;
        const int cnt = count_f > ntau ? count_f : ntau;
        hipLaunchKernelGGL(log_grid_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, count_f, freq, ntau, tau, lw, lt);
        int tpr = ((ntau + 1) / 2 + 63) / 64 * 64;           // threads per row (two columns each)
        if (tpr > 1024) tpr = 1024;
        while (1024 % tpr) tpr += 64;                         // 64,128,256,512,1024
        int rpb = 256;
        while (rpb > 4 && (long long)B * ((nf + rpb - 1) / rpb) < 1024) rpb /= 2;   // enough workgroups for 256 CUs
        const int bx = (nf + rpb - 1) / rpb;
        hipLaunchKernelGGL(impedance_interp_kernel, dim3(bx, B), dim3(1024), (6 * (size_t)(ngrid + (ngrid & 1)) + rpb) * sizeof(double), st,
                           freq_batched, lw, nf, lt, ntau, ngrid, lut6, rpb, tpr, a_re, a_im);
    } else {
        hipLaunchKernelGGL(impedance_trapz_kernel, dim3((ntau + 15) / 16, nf, B), dim3(256), 3 * ny * sizeof(double),
                           st, freq_batched, freq, nf, tau, ntau, eps, ny, a_re, a_im);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Distribution of phasances (gaussian nu basis, normalize=False)
// ---------------------------------------------------------------------------------------------------------
// erf(x + i y) by Abramowitz & Stegun 7.1.29 (|y| <= ~1: here y = -pi / (4 eps)); absolute error ~1e-16
__device__ __forceinline__ void cerf(double x, double y, double& re, double& im) {
    const double PI = 3.141592653589793;
    const double ex2 = exp(-x * x);
    re = erf(x);
    const double s = sin(x * y), xy2 = 2.0 * x * y;
    const double c2 = cos(xy2), s2 = sin(xy2);
    if (x != 0.0) { re += ex2 * 2.0 * s * s / (2.0 * PI * x); im = ex2 * s2 / (2.0 * PI * x); }
    else im = ex2 * y / PI;
    double sre = 0.0, sim = 0.0;
    for (int n = 1; n <= 32; ++n) {
        const double dn = (double)n;
        const double en = exp(-0.25 * dn * dn) / (dn * dn + 4.0 * x * x);
        const double ch = cosh(dn * y), sh = sinh(dn * y);
        sre += en * (2.0 * x - 2.0 * x * ch * c2 + dn * sh * s2);
        sim += en * (2.0 * x * ch * s2 + dn * sh * c2);
    }
    re += 2.0 / PI * ex2 * sre;
    im += 2.0 / PI * ex2 * sim;
}

// zm[r][c] = F(b) - F(a),  F(nu) = sqrt(pi)/2 (j w)^nu_m / eps * (j w)^(ln(j w) / 4 eps^2) * erf(eps (nu - nu_m) - ln(j w) / 2 eps),
// (a, b) = (min(0, sgn nu_m), max(0, sgn nu_m))   (phasance.py:19-33, 62-82)
__global__ void phasor_z_kernel(const double* __restrict__ freq, int nf, const double* __restrict__ nu, int nnu, double eps,
                                double* __restrict__ zre, double* __restrict__ zim) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= nnu) return;
    const double PI = 3.141592653589793;
    const double lw = log(2.0 * PI * freq[r]), num = nu[c];
    const double sg = (num > 0.0) ? 1.0 : ((num < 0.0) ? -1.0 : 0.0);
    const double a = fmin(0.0, sg), b = fmax(0.0, sg);
    // prefactor exp(nu_m L + L^2 / (4 eps^2)), L = ln(j w) = lw + j pi/2
    const double q = 1.0 / (4.0 * eps * eps);
    const double e_re = num * lw + (lw * lw - 0.25 * PI * PI) * q;
    const double e_im = num * 0.5 * PI + PI * lw * q;
    const double mag = 0.5 * sqrt(PI) / eps * exp(e_re);
    const double p_re = mag * cos(e_im), p_im = mag * sin(e_im);
    const double y = -PI / (4.0 * eps), xoff = -lw / (2.0 * eps);
    double br, bi, ar, ai;
    cerf(eps * (b - num) + xoff, y, br, bi);
    cerf(eps * (a - num) + xoff, y, ar, ai);
    const double dr = br - ar, di = bi - ai;
    zre[(size_t)r * nnu + c] = p_re * dr - p_im * di;
    zim[(size_t)r * nnu + c] = p_re * di + p_im * dr;
}

// rm_layered[k][r][c] = size_k (G(b) - G(a)) for t_r > t_k, G(nu) = sqrt(pi)/2 dt^-nu_m / Gamma(1 - nu_m) / eps *
// dt^(ln dt / 4 eps^2) * erf(eps (nu - nu_m) + ln dt / 2 eps), dt = t_r - t_k; rm = sum over k  (phasance.py:38-52, 121-144)
__global__ void phasor_v_kernel(const double* __restrict__ times, int nt, const double* __restrict__ nu, int nnu, double eps,
                                const double* __restrict__ step_times, const double* __restrict__ step_sizes, int nsteps,
                                double* __restrict__ rm, double* __restrict__ layered) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= nnu) return;
    const double PI = 3.141592653589793;
    const double num = nu[c], t = times[r];
    const double sg = (num > 0.0) ? 1.0 : ((num < 0.0) ? -1.0 : 0.0);
    const double a = fmin(0.0, sg), b = fmax(0.0, sg);
    const double gin = 1.0 / tgamma(1.0 - num);
    double acc = 0.0;
    for (int k = 0; k < nsteps; ++k) {
        double val = 0.0;
        if (t > step_times[k]) {
            const double ldt = log(t - step_times[k]);
            const double pre = 0.5 * sqrt(PI) * (exp(-num * ldt) * gin) / eps * exp(ldt * ldt / (4.0 * eps * eps));
            const double xo = ldt / (2.0 * eps);
            val = step_sizes[k] * (pre * erf(eps * (b - num) + xo) - pre * erf(eps * (a - num) + xo));
        }
        if (layered) layered[((size_t)k * nt + r) * nnu + c] = val;
        acc += val;
    }
    rm[(size_t)r * nnu + c] = acc;
}

void launch_phasor_z(hipStream_t st, const double* freq, int nf, const double* nu, int nnu, double eps, double* zre,
                     double* zim) {
    hipLaunchKernelGGL(phasor_z_kernel, dim3((nnu + 63) / 64, nf), dim3(64), 0, st, freq, nf, nu, nnu, eps, zre, zim);
}

void launch_phasor_v(hipStream_t st, const double* times, int nt, const double* nu, int nnu, double eps,
                     const double* step_times, const double* step_sizes, int nsteps, double* rm, double* layered) {
    hipLaunchKernelGGL(phasor_v_kernel, dim3((nnu + 63) / 64, nt), dim3(64), 0, st, times, nt, nu, nnu, eps, step_times,
                       step_sizes, nsteps, rm, layered);
}

// Chrono variance-estimation matrix: one 256-thread block per row.  tt = transformed sample times, seg[nseg+1] =
// sample index bounds of the step segments (no correlation across segments), rows normalised to sum 1.
__global__ __launch_bounds__(256) void chrono_vmm_kernel(const double* __restrict__ tt, int nt,
                                                         const int* __restrict__ seg, int nseg, double eps, int uniform,
                                                         double* __restrict__ vmm) {
    __shared__ double red[4];
    const int i = blockIdx.x;
    int a = 0, b = nt;
    if (!uniform) {
        for (int k = 0; k < nseg; ++k)
            if (i >= seg[k] && i < seg[k + 1]) { a = seg[k]; b = seg[k + 1]; }
    }
    const double ti = tt[i];
    double s = 0.0;
    for (int j = threadIdx.x; j < nt; j += blockDim.x) {
        double v = 0.0;
        if (uniform) v = 1.0;
        else if (j >= a && j < b) { const double dd = eps * (ti - tt[j]); v = exp(-(dd * dd)); }
        vmm[(size_t)i * nt + j] = v;
        s += v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int j = threadIdx.x; j < nt; j += blockDim.x) vmm[(size_t)i * nt + j] /= tot;
}

void launch_chrono_vmm(hipStream_t st, const double* tt, int nt, const int* seg, int nseg, double eps, int uniform,
                       double* vmm) {
    hipLaunchKernelGGL(chrono_vmm_kernel, dim3(nt), dim3(256), 0, st, tt, nt, seg, nseg, eps, uniform, vmm);
}

void launch_response_lookup(hipStream_t st, double eps, int ngrid, int ny, const double* td, double* v) {
    hipLaunchKernelGGL(response_lookup_kernel, dim3((ngrid + 3) / 4), dim3(256), 3 * ny * sizeof(double), st, eps, ngrid, ny,
                       td, v);
}

void launch_response_matrix(hipStream_t st, const double* times, int nt, const double* tau, int ntau,
                            const double* step_times, const double* step_sizes, int nsteps, int mode, double eps,
                            int ngrid, const double* lut3, int ny, double* a, double* layered) {
    if (mode == HIPDRT_MODE_INTERP) {
        int rpb = 32;
        while (rpb > 1 && (long long)((ntau + 255) / 256) * ((nt + rpb - 1) / rpb) < 1024) rpb /= 2;
        hipLaunchKernelGGL(response_interp_kernel, dim3((ntau + 255) / 256, (nt + rpb - 1) / rpb), dim3(256),
                           3 * (size_t)ngrid * sizeof(double), st, times, nt, tau, ntau, step_times, step_sizes, nsteps,
                           ngrid, lut3, rpb, a, layered);
    } else {
        hipLaunchKernelGGL(response_trapz_kernel, dim3((ntau + 15) / 16, nt), dim3(256), 3 * ny * sizeof(double), st,
                           times, nt, tau, ntau, step_times, step_sizes, nsteps, eps, ny, a, layered);
    }
}

void launch_response_variant(hipStream_t st, const double* times, int nt, const double* tau, int ntau,
                             const double* step_times, const double* step_sizes, const double* tau_rise, int nsteps,
                             int variant, double eps, int ny, double* a, double* layered) {
    if (variant == HIPDRT_RESPONSE_POT)
        hipLaunchKernelGGL(response_pot_kernel, dim3((ntau + 255) / 256, nt), dim3(256), 0, st, times, nt, tau, ntau,
                           step_times, step_sizes, nsteps, a, layered);
    else
        hipLaunchKernelGGL(response_expdecay_kernel, dim3((ntau + 15) / 16, nt), dim3(256), 3 * ny * sizeof(double), st,
                           times, nt, tau, ntau, step_times, step_sizes, tau_rise, nsteps, eps, ny, a, layered);
}

void launch_penalty(hipStream_t st, const double* ln_tau, int n, double eps, int toeplitz, double* m0, double* m1,
                    double* m2, int ld, int pad) {
    hipLaunchKernelGGL(penalty_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, ln_tau, n, eps, toeplitz, m0, m1,
                       m2, ld, pad);
}

void launch_eis_vmm(hipStream_t st, const double* freq, int nf, double vmm_eps, double reim_cor, int uniform,
                    double* vmm) {
    hipLaunchKernelGGL(eis_vmm_kernel, dim3(2 * nf), dim3(256), 0, st, freq, nf, vmm_eps, reim_cor, uniform, vmm);
}

// ---------------------------------------------------------------------------------------------------------
// filters.nonuniform_gaussian_filter1d (hybdrt/filters/_filters.py:261-343), order 0, mode 'reflect', empty=False: the
// anti-aliasing filter of the chrono down-sampling (preprocessing.filter_chrono_signal, 507-572).  Every sample has its own
// sigma; the reference evaluates scipy's gaussian_filter1d at log-spaced sigma nodes and blends the node outputs with
// triangular weights in log sigma.  Here one thread per sample evaluates only the (at most two) nodes that carry weight,
// with scipy's symmetric correlation order  y[i] w0 + sum_j (y[i-j] + y[i+j]) w_j  and its 'reflect' boundary
// (d c b a | a b c d | d c b a).  weights[woff[k] + j], j = 0..radius[k], are the normalised kernel halves (host: numpy,
// exactly as scipy builds them); a node with radius < 0 returns the input (sigma below min_sigma).
// seg[s], seg[s+1] bound the step segments: the reference filters every segment separately.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nonuniform_gauss_kernel(const double* __restrict__ y, int n,
                                                               const double* __restrict__ sigma,
                                                               const int* __restrict__ seg_of, const int* __restrict__ seg,
                                                               const double* __restrict__ nodes, int K,
                                                               const double* __restrict__ node_delta,
                                                               const double* __restrict__ weights,
                                                               const int* __restrict__ woff, const int* __restrict__ radius,
                                                               double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = seg_of[i];
    if (s < 0) { out[i] = y[i]; return; }                 // segment without filtering (all sigma zero)
    const int a = seg[s], len = seg[s + 1] - a, li = i - a;
    const double sg = sigma[i];
    const int kb = s * K;                                  // nodes are per segment
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
        const double node = nodes[kb + k];
        if (!(node > 0.0)) continue;                       // unused slot
        double nw = fabs(log(sg / node)) / node_delta[s];
        if (nw >= 1.0) nw = 1.0;
        nw = 1.0 - nw;
        if (nw == 0.0) continue;                           // exact zero: adds nothing in the reference either
        double v;
        const int r = radius[kb + k];
        if (r < 0) {
            v = y[i];
        } else {
            const double* w = weights + woff[kb + k];
            v = y[i] * w[0];
            for (int j = 1; j <= r; ++j) {
                int lo = li - j, hi = li + j;
                // scipy 'reflect': period 2 len, mirrored about the half-sample edges
                if (lo < 0) { lo = -lo - 1; if (lo >= len) { lo %= 2 * len; if (lo >= len) lo = 2 * len - 1 - lo; } }
                if (hi >= len) { hi %= 2 * len; if (hi >= len) hi = 2 * len - 1 - hi; }
                v += (y[a + lo] + y[a + hi]) * w[j];
            }
        }
        acc += v * nw;
    }
    out[i] = acc;
}

void launch_nonuniform_gauss(hipStream_t st, const double* y, int n, const double* sigma, const int* seg_of, const int* seg,
                             const double* nodes, int K, const double* node_delta, const double* weights, const int* woff,
                             const int* radius, double* out) {
    hipLaunchKernelGGL(nonuniform_gauss_kernel, dim3((n + 255) / 256), dim3(256), 0, st, y, n, sigma, seg_of, seg, nodes, K,
                       node_delta, weights, woff, radius, out);
}

}  // namespace hipdrt

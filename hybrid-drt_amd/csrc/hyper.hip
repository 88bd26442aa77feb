// Per-spectrum pieces of the QPHB outer loop that are not the QP itself (one 512-thread workgroup per spectrum):
//
//   prep_kernel       DRTBase.scale_data / pp.estimate_rp (EIS branch)   hybdrt/models/drtbase.py:439-514,
//                     preprocessing.py:828-841; state initialisation     drt1d.py:542-566, 612
//   weights_kernel    qphb.estimate_weights                               hybdrt/models/qphb.py:1545-1594
//   hyper_kernel      the s_k / rho_k updates of qphb.iterate_qphb (qphb.py:718-816) = solve_s (320-356) +
//                     solve_rho (385-405), the xmx_norms freeze (drt1d.py:946-951), estimate_weights and the
//                     convergence rule (qphb.py:597-603, 969-970)
//
// The reference codes solve_s / calculate_qp_l2_matrix as dense n^3 diagonal products; algebraically they are
// row sums of elementwise products, done here as one wavefront per matrix row (coalesced row reads of the shared
// penalty / response / variance matrices, which stay L2/Infinity-Cache resident across the batch).
#include "common.hpp"

namespace hipdrt {

// PROFILE=1 builds: shader-clock ticks per phase of hyper_kernel's workgroup 0 (slots 48.. of hipdrt_qp_profile)
#ifdef HIPDRT_QP_PROFILE
__device__ unsigned long long g_hyper_prof[16];
struct HProf {
    unsigned long long t;
    __device__ __forceinline__ void start() { t = clock64(); }
    __device__ __forceinline__ void mark(int slot) {
        if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long n_ = clock64(); atomicAdd(&g_hyper_prof[slot], n_ - t); t = n_; }
    }
};
#define HPROF_START() HProf hp_; hp_.start()
#define HPROF(slot) hp_.mark(slot)
#define HPROF_COUNT(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_hyper_prof[slot], 1ull); } while (0)
#else
#define HPROF_COUNT(slot) do {} while (0)
#define HPROF_START() do {} while (0)
#define HPROF(slot) do {} while (0)
#endif
int hyper_profile_read(unsigned long long* out, int n, int reset) {
#ifdef HIPDRT_QP_PROFILE
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hyper_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 16; ++i) out[i] = h[i];
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hyper_prof), z, sizeof(z)); }
#else
    for (int i = 0; i < n && i < 16; ++i) out[i] = 0;
#endif
    return 0;
}

static constexpr int HT = 512;
static constexpr int HNW = HT / 64;

// DPP moves of a double (two dwords); ctrl: quad_perm 0x00-0xFF, row_ror:n = 0x120 + n
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// wavefront sum, every lane gets the total: quad permutes and row rotations (VALU speed) inside the 16-lane rows,
// v_readlane across the four rows -- no LDS crossbar traffic
__device__ __forceinline__ double hw_sum(double v) {
    v += dpp_mov<0xB1>(v);          // lane ^ 1
    v += dpp_mov<0x4E>(v);          // lane ^ 2
    v += dpp_mov<0x124>(v);         // row_ror:4
    v += dpp_mov<0x128>(v);         // row_ror:8
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ double hw_max(double v) {
    v = fmax(v, dpp_mov<0xB1>(v));
    v = fmax(v, dpp_mov<0x4E>(v));
    v = fmax(v, dpp_mov<0x124>(v));
    v = fmax(v, dpp_mov<0x128>(v));
    return fmax(fmax(lane_bcast(v, 0), lane_bcast(v, 16)), fmax(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ __forceinline__ double hw_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

// block-wide reductions over HT threads; red = LDS [HNW]; two syncs so `red` is immediately reusable
__device__ __forceinline__ double blk_sum(double v, double* red) {
    v = hw_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < HNW; ++w) t += red[w];
    return t;
}
__device__ __forceinline__ double blk_max(double v, double* red) {
    v = hw_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < HNW; ++w) t = fmax(t, red[w]);
    return t;
}
__device__ __forceinline__ double blk_min(double v, double* red) {
    v = hw_min(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < HNW; ++w) t = fmin(t, red[w]);
    return t;
}

// ---------------------------------------------------------------------------------------------------------
// prep: scaling + state init.  grid = B
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HT) void prep_kernel(FitState st) {
    __shared__ double red[HNW];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nf = st.nf, m = st.m, n = st.n;
    const double* zr = st.z_re + (size_t)b * nf;
    const double* zi = st.z_im + (size_t)b * nf;
    double mx = -INFINITY, mn = INFINITY;
    for (int i = tid; i < nf; i += HT) { mx = fmax(mx, zr[i]); mn = fmin(mn, zr[i]); }
    mx = blk_max(mx, red);
    mn = blk_min(mn, red);
    double cs = 1.0;
    if (st.opts.scale_data) cs = (mx - mn) / st.opts.rp_scale;     // rp_est / rp_scale
    double* rv = st.rv + (size_t)b * m;
    double s1 = 0.0;
    for (int i = tid; i < m; i += HT) {
        const double v = (i < nf ? zr[i] : zi[i - nf]) / cs;
        rv[i] = v;
        s1 += v;
    }
    const double mean = blk_sum(s1, red) / (double)m;
    double s2 = 0.0;
    for (int i = tid; i < m; i += HT) { const double dv = rv[i] - mean; s2 += dv * dv; }
    const double var = blk_sum(s2, red) / (double)m;                 // np.var
    for (int i = tid; i < m; i += HT) st.w[(size_t)b * m + i] = 1.0;  // unweighted initial-weights QP
    for (int k = 0; k < 3; ++k)
        for (int i = tid; i < n; i += HT) st.s[((size_t)b * 3 + k) * n + i] = st.opts.s_0[k];
    for (int i = tid; i < n; i += HT) { st.x[(size_t)b * n + i] = 1e-6; st.x_in[(size_t)b * n + i] = 1e-6; }
    if (tid == 0) {
        st.coef_scale[b] = cs;
        st.var_floor[b] = var * 1e-7;
        for (int k = 0; k < 3; ++k) { st.rho[(size_t)b * 3 + k] = st.opts.rho_0[k]; st.xmx[(size_t)b * 3 + k] = 1.0; }
        st.active[b] = 1;
        st.outer_iters[b] = 0;
        st.fit_status[b] = 1;
        st.qp_iters_total[b] = 0;
    }
}

// prepared plans: the data vector is already scaled; var floor of estimate_weights + state init.  grid = B
__global__ __launch_bounds__(HT) void prep_prepared_kernel(FitState st) {
    __shared__ double red[HNW];
    const int b = blockIdx.x, tid = threadIdx.x, m = st.m, n = st.n;
    const double* rv = st.rv + (size_t)b * m;
    double s1 = 0.0;
    for (int i = tid; i < m; i += HT) s1 += rv[i];
    const double mean = blk_sum(s1, red) / (double)m;
    double s2 = 0.0;
    for (int i = tid; i < m; i += HT) { const double dv = rv[i] - mean; s2 += dv * dv; }
    const double var = blk_sum(s2, red) / (double)m;
    for (int i = tid; i < m; i += HT) st.w[(size_t)b * m + i] = 1.0;
    for (int k = 0; k < 3; ++k)
        for (int i = tid; i < n; i += HT) st.s[((size_t)b * 3 + k) * n + i] = st.opts.s_0[k];
    for (int i = tid; i < n; i += HT) { st.x[(size_t)b * n + i] = 1e-6; st.x_in[(size_t)b * n + i] = 1e-6; }
    if (tid == 0) {
        st.coef_scale[b] = 1.0;
        st.var_floor[b] = var * 1e-7;
        for (int k = 0; k < 3; ++k) {
            st.rho[(size_t)b * 3 + k] = st.opts.rho_0[k]; st.xmx[(size_t)b * 3 + k] = 1.0;
            st.dop_rho[(size_t)b * 3 + k] = st.desc.dop_rho_0[k]; st.dop_xmx[(size_t)b * 3 + k] = 1.0;
        }
        st.active[b] = 1;
        st.outer_iters[b] = 0;
        st.fit_status[b] = 1;
        st.qp_iters_total[b] = 0;
    }
}

// y[i] = sum_j M[i][j] * v[j] for the rows owned by this wavefront; v in LDS; result to LDS out.  Four rows per
// pass with 16-byte loads, MV_UC column chunks of 128 requested before the first product: 4 * MV_UC independent loads
// in flight per lane (one L2 round trip per 512 columns instead of one per chunk; more would not fit 128 VGPRs).  Columns past the end are
// loaded from a clamped address and left out of the sum, so the per-lane order of additions is the plain loop's.
static constexpr int MV_UC = 4;
__device__ __forceinline__ void rows_matvec(const double* __restrict__ M, int ld, int nrow, int ncol,
                                            const double* __restrict__ v, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (((ld | ncol) & 1) == 0 && (reinterpret_cast<size_t>(M) & 15) == 0) {
        for (int i0 = 4 * wv; i0 < nrow; i0 += 4 * HNW) {
            const double* r0 = M + (size_t)i0 * ld;
            const double* r1 = M + (size_t)(i0 + 1 < nrow ? i0 + 1 : i0) * ld;
            const double* r2 = M + (size_t)(i0 + 2 < nrow ? i0 + 2 : i0) * ld;
            const double* r3 = M + (size_t)(i0 + 3 < nrow ? i0 + 3 : i0) * ld;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int base = 0; base < ncol; base += 128 * MV_UC) {
                const int jb = base + 2 * lane;
                double2 a0[MV_UC], a1[MV_UC], a2[MV_UC], a3[MV_UC];
#pragma unroll
                for (int c = 0; c < MV_UC; ++c) {
                    if (base + 128 * c >= ncol) break;               // the whole chunk lies past the end (uniform)
                    const int j = jb + 128 * c, jj = j < ncol ? j : ncol - 2;
                    a0[c] = *reinterpret_cast<const double2*>(r0 + jj);
                    a1[c] = *reinterpret_cast<const double2*>(r1 + jj);
                    a2[c] = *reinterpret_cast<const double2*>(r2 + jj);
                    a3[c] = *reinterpret_cast<const double2*>(r3 + jj);
                }
#pragma unroll
                for (int c = 0; c < MV_UC; ++c) {
                    const int j = jb + 128 * c;
                    if (j < ncol) {
                        const double vx = v[j], vy = v[j + 1];
                        s0 += a0[c].x * vx + a0[c].y * vy;
                        s1 += a1[c].x * vx + a1[c].y * vy;
                        s2 += a2[c].x * vx + a2[c].y * vy;
                        s3 += a3[c].x * vx + a3[c].y * vy;
                    }
                }
            }
            s0 = hw_sum(s0); s1 = hw_sum(s1); s2 = hw_sum(s2); s3 = hw_sum(s3);
            if (lane == 0) {
                out[i0] = s0;
                if (i0 + 1 < nrow) out[i0 + 1] = s1;
                if (i0 + 2 < nrow) out[i0 + 2] = s2;
                if (i0 + 3 < nrow) out[i0 + 3] = s3;
            }
        }
        return;
    }
    for (int i = wv; i < nrow; i += HNW) {
        const double* row = M + (size_t)i * ld;
        double s = 0.0;
        for (int j = lane; j < ncol; j += 64) s += row[j] * v[j];
        s = hw_sum(s);
        if (lane == 0) out[i] = s;
    }
}

// same for a symmetric Toeplitz matrix given by its mirrored first column (LDS, c[d] valid for -(nd-1) <= d <= nd-1):
// y[i] = sum_j c[i - j] v[j].  Two adjacent rows per thread: v[j] is a broadcast read, c[i - j] runs over consecutive
// addresses across the lanes, nothing is reduced across lanes, and row i+1's operand at column j+1 is row i's at column j
// (kept in a register) -- one LDS read per product instead of two.  Four partial sums per row (j mod 4), as before.
// reach >= 0: c[d] is exactly zero for |d| > reach (the Gaussian penalties underflow a few dozen grid points off the diagonal):
// the columns outside [ia - reach, ia + 1 + reach], widened to multiples of four so that every product keeps its partial sum,
// would add exact zeros and are left out -- the same bits from a fifth of the products at nd = 512.
__device__ __forceinline__ void toeplitz_matvec(const double* __restrict__ c, int nd, const double* __restrict__ v,
                                                double* __restrict__ out, int reach = -1) {
    for (int p = threadIdx.x; 2 * p < nd; p += HT) {
        const int ia = 2 * p;
        const bool hb = ia + 1 < nd;
        const double* ca = c + ia;                 // ca[-j] = c[ia - j]
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        int j = 0, jhi = nd;
        if (reach >= 0) {
            j = ia - reach > 0 ? (ia - reach) & ~3 : 0;
            const int e_ = (ia + 2 + reach + 3) & ~3;
            jhi = e_ < nd ? e_ : nd;
        }
        double prev = hb ? ca[1 - j] : ca[-j];     // c[ib - j]
        for (; j + 3 < jhi; j += 4) {
            const double c0 = ca[-j], c1 = ca[-j - 1], c2 = ca[-j - 2], c3 = ca[-j - 3];
            const double v0 = v[j], v1 = v[j + 1], v2 = v[j + 2], v3 = v[j + 3];
            a0 += c0 * v0; a1 += c1 * v1; a2 += c2 * v2; a3 += c3 * v3;
            b0 += prev * v0; b1 += c0 * v1; b2 += c1 * v2; b3 += c2 * v3;
            prev = c3;
        }
        for (; j < jhi; ++j) { const double c0 = ca[-j]; a0 += c0 * v[j]; b0 += prev * v[j]; prev = c0; }
        out[ia] = (a0 + a1) + (a2 + a3);
        if (hb) out[ia + 1] = (b0 + b1) + (b2 + b3);
    }
}

// a per-lane value the compiler may not hoist address arithmetic out of a loop for (hyper_kernel at 128 VGPRs: offsets formed where
// they are used instead of being carried in scratch from the kernel's start)
__device__ __forceinline__ int opaque_lane(int v) { asm volatile("" : "+v"(v)); return v; }

// estimate_weights for spectrum b: xs = LDS x[n]; tmp = LDS [m]; result written to w_out[m] (global)
// estimate_weights (qphb.py:1545-1594) for spectrum b with variance matrix V: xs = LDS x[n]; tmp, tmp2 = LDS [m]; with
// outlier_p > 0 also tmp3, tmp4 = LDS [m] (solve_outlier_t / outlier_tvt, qphb.py:1497-1539:
// s_hat = sqrt(t) o V (sqrt(t) o r^2) + (1 - t) o r^2).  Result written to w_out[m] (global).
__device__ void estimate_weights_dev(const FitState& st, int b, const double* V, const double* xs, double* tmp, double* tmp2,
                                     double* tmp3, double* tmp4, const double* est_w, double* w_out, int r0 = 0, int r1 = -1,
                                     double vf_range = -1.0, const double* pre = nullptr) {
    const int m = st.m, n = st.n, tid = opaque_lane(threadIdx.x);
    const double* rv = st.rv + (size_t)b * m;
    const double op = st.opts.outlier_p;
    HPROF_START();
    // (pre: both products of this call were computed by premv_kernel, same arithmetic per row)
    const double* pre1 = pre ? pre + (size_t)b * m : nullptr;
    const double* pre2 = pre ? pre + ((size_t)gridDim.x + b) * m : nullptr;
    if (pre1) { for (int i = tid; i < m; i += HT) tmp[i] = pre1[i]; }
    else rows_matvec(st.rm + (size_t)b * st.rm_stride, st.ldrm, m, n, xs, tmp);        // rm @ x
    __syncthreads();
    HPROF(8);
    for (int i = tid; i < m; i += HT) {
        const double r = tmp[i] - rv[i];
        if (op > 0.0) tmp3[i] = r;
        tmp[i] = r * r;
    }
    __syncthreads();
    // V @ resid**2.  A uniform chrono block (error_structure='uniform': every chrono row of V is the same vector, zero on the
    // impedance columns) needs one row only -- same arithmetic per row, so the same bits, without streaming nc x m entries
    const int ncu = (st.prepared && st.desc.chrono_vmm_uniform && V == st.vmm && op <= 0.0) ? st.desc.num_chrono : 0;
    if (pre2) {
        for (int i = tid; i < m; i += HT) if (i == 0 || i >= ncu) tmp2[i] = pre2[i];
        __syncthreads();
        if (ncu > 0) {
            const double s0 = tmp2[0];
            __syncthreads();
            for (int i = 1 + tid; i < ncu; i += HT) tmp2[i] = s0;
        }
    } else if (ncu > 0) {
        rows_matvec(V, m, 1, m, tmp, tmp2);
        if (ncu < m) rows_matvec(V + (size_t)ncu * m, m, m - ncu, m, tmp, tmp2 + ncu);
        __syncthreads();
        const double s0 = tmp2[0];
        __syncthreads();
        for (int i = 1 + tid; i < ncu; i += HT) tmp2[i] = s0;
    } else {
        rows_matvec(V, m, m, m, tmp, tmp2);
    }
    __syncthreads();
    HPROF(9);
    if (op > 0.0) {
        const double s2pi = sqrt(2.0 * 3.141592653589793);
        for (int i = tid; i < m; i += HT) {
            const double r = tmp3[i], ar = fabs(r), sb = sqrt(tmp2[i]);
            const double pdf_in = 1.0 / (sb * s2pi) * exp(-0.5 * (r * r) / (sb * sb));
            const double pdf_out = 1.0 / (ar * s2pi) * exp(-0.5 * (r * r) / (ar * ar));
            double t = 1.0 - op * pdf_out / ((1.0 - op) * pdf_in + op * pdf_out);
            if (sb > ar) t = 1.0;
            tmp3[i] = t;                                // outlier_t
            if (st.outlier_t) st.outlier_t[(size_t)b * m + i] = t;
            tmp4[i] = sqrt(t) * tmp[i];                 // sqrt(t) o r^2
        }
        __syncthreads();
        rows_matvec(V, m, m, m, tmp4, tmp2);
        __syncthreads();
        for (int i = tid; i < m; i += HT) {
            const double t = tmp3[i];
            tmp2[i] = sqrt(t) * tmp2[i] + (1.0 - t) * tmp[i];
        }
        __syncthreads();
    }
    const double vf = vf_range >= 0.0 ? vf_range : st.var_floor[b];
    if (r1 < 0) r1 = m;
    for (int i = r0 + tid; i < r1; i += HT) {
        double sh = tmp2[i];
        if (sh < vf) sh = vf;
        double wh = 1.0 / sqrt(sh);                      // s_hat ** -0.5
        if (est_w) {
            const double ew = est_w[i];
            const double fc = wh / (wh + ew);
            const double fe = 1.0 - fc;
            wh = fc * wh + fe * ew;
        }
        w_out[i] = fmax(wh, 1e-10);
    }
    __syncthreads();
}

// after an initial-weights QP: est_weights = estimate_weights(x_overfit, est_weights=None) on the initialize_weights
// variance matrix; stage 1 also sets weights = solve_init_weight_scale(est_weights) (qphb.py:1471-1479, 1679) and resets x
// stage 2 (init_weights_separately, drt1d.py:648-672): est_weights of the rows [r0, r1) only, from the QP that saw only
// those rows, with the variance floor of that data block; stage 3: just the final step (weights from est_weights, x reset)
__global__ __launch_bounds__(HT) void init_weights_kernel(FitState st, int stage, int r0, int r1) {
    extern __shared__ double sm[];
    __shared__ double red[HNW];
    const int b = blockIdx.x, tid = threadIdx.x, n = st.n, m = st.m;
    double* xs = sm;
    double* tmp = xs + n;
    double* tmp2 = tmp + m;
    double* tmp3 = tmp2 + m;      // [2][m] present only when outlier_p is set
    double* tmp4 = tmp3 + m;
    if (st.qp_status[b] < 0) {
        if (tid == 0) { st.active[b] = 0; st.fit_status[b] = -1; }
        return;
    }
    if (stage != 3) {
        for (int i = tid; i < n; i += HT) xs[i] = st.x[(size_t)b * n + i];
        __syncthreads();
        if (stage == 2) {
            const double* rv = st.rv + (size_t)b * m;
            double s1 = 0.0;
            for (int i = r0 + tid; i < r1; i += HT) s1 += rv[i];
            const double mean = blk_sum(s1, red) / (double)(r1 - r0);
            double s2 = 0.0;
            for (int i = r0 + tid; i < r1; i += HT) { const double dv = rv[i] - mean; s2 += dv * dv; }
            const double var = blk_sum(s2, red) / (double)(r1 - r0);          // np.var of this block's data
            estimate_weights_dev(st, b, st.vmm_iw, xs, tmp, tmp2, tmp3, tmp4, nullptr, st.est_w + (size_t)b * m, r0, r1,
                                 var * 1e-7);
            return;
        }
        estimate_weights_dev(st, b, st.vmm_iw, xs, tmp, tmp2, tmp3, tmp4, nullptr, st.est_w + (size_t)b * m);
    }
    if (stage == 0) return;
    const double al = st.opts.iw_alpha, be = st.opts.iw_beta;
    for (int i = tid; i < m; i += HT) {
        double w = st.est_w[(size_t)b * m + i];
        if (al > 0.0) {
            const double bq = 0.5 - al + 1.0;
            const double sh = (-bq + sqrt(bq * bq + 2.0 * be * (1.0 / (w * w)))) / (2.0 * be);
            w = 1.0 / sqrt(sh);
        }
        st.w[(size_t)b * m + i] = w;
    }
    // the outer loop starts from x = 1e-6 (drt1d.py:612), not from x_overfit
    for (int i = tid; i < n; i += HT) st.x[(size_t)b * n + i] = 1e-6;
}

// variance matrix without each point's own residual, rows renormalised (qphb.py:1644-1648)
__global__ void vmm_exclude_self_kernel(const double* __restrict__ vmm, int m, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= m) return;
    const double d = vmm[(size_t)i * m + i];
    out[(size_t)i * m + j] = (i == j) ? 0.0 : vmm[(size_t)i * m + j] / (1.0 - d);
}

void launch_vmm_exclude_self(hipStream_t s, const double* vmm, int m, double* out) {
    hipLaunchKernelGGL(vmm_exclude_self_kernel, dim3((m + 255) / 256, m), dim3(256), 0, s, vmm, m, out);
}

// ---------------------------------------------------------------------------------------------------------
// Few, large fits (one joint fit of BASELINE configs[4]: rm is 5120 x 1078, vmm 5120 x 5120): hyper_kernel is one workgroup
// per fit, and its three matrix-vector products would stream ~130 MB through ONE CU per outer iteration (2.6 of its
// 4.5 ms).  Here the same products, row slab by row slab on many workgroups, with rows_matvec itself -- the same
// arithmetic per row, hence the same bits -- into premv[3][B][m]: phase 0  rm @ x and (vz_offset fits) rm @ x with the
// baseline / offset entries zeroed; phase 1  vmm @ (rm x - rv)^2 (uniform chrono block: row 0 stands for the block).
// grid = (row slabs, B)
// ---------------------------------------------------------------------------------------------------------
static constexpr int PREMV_ROWS = 32;      // rows per workgroup (4 per wavefront and pass)
__global__ __launch_bounds__(HT) void premv_kernel(FitState st, int phase, int B) {
    extern __shared__ double sm[];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (!st.active[b] || st.qp_status[b] < 0) return;
    const int n = st.n, m = st.m;
    const int r0 = blockIdx.x * PREMV_ROWS, r1 = r0 + PREMV_ROWS < m ? r0 + PREMV_ROWS : m;
    double* y1 = st.premv + (size_t)b * m;
    double* y2 = st.premv + ((size_t)B + b) * m;
    double* y3 = st.premv + (2 * (size_t)B + b) * m;
    if (phase == 0) {
        double* xs = sm;
        const double* xg = st.x + (size_t)b * n;
        for (int i = tid; i < n; i += HT) xs[i] = xg[i];
        __syncthreads();
        rows_matvec(st.rm + (size_t)b * st.rm_stride + (size_t)r0 * st.ldrm, st.ldrm, r1 - r0, n, xs, y1 + r0);
        if (st.prepared && st.desc.vz_index >= 0 && st.continue_mode != 2) {
            __syncthreads();
            const int vz = st.desc.vz_index;
            for (int i = tid; i < n; i += HT)
                if (i == vz || (i >= st.desc.vb_start && i < st.desc.vb_start + st.desc.vb_size)) xs[i] = 0.0;
            __syncthreads();
            rows_matvec(st.rm_rw + (size_t)b * st.rm_stride + (size_t)r0 * st.ldrm, st.ldrm, r1 - r0, n, xs, y3 + r0);
        }
    } else {
        double* r2 = sm;
        const double* rv = st.rv + (size_t)b * m;
        for (int i = tid; i < m; i += HT) { const double r = y1[i] - rv[i]; r2[i] = r * r; }
        __syncthreads();
        const int ncu = (st.prepared && st.desc.chrono_vmm_uniform) ? st.desc.num_chrono : 0;
        if (ncu > 0 && r0 == 0) rows_matvec(st.vmm, m, 1, m, r2, y2);
        const int a0 = r0 > ncu ? r0 : ncu;
        if (a0 < r1) rows_matvec(st.vmm + (size_t)a0 * m, m, r1 - a0, m, r2, y2 + a0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Many fits that SHARE their matrices (an EIS plan: one response matrix rm and one variance matrix vmm for the whole batch):
// the two products of estimate_weights, rm @ x_b and vmm @ (rm x_b - rv_b)^2 for every spectrum b, and the linear term of the
// QP, rm' (w_b^2 rv_b), are matrix-matrix products with a shared operand -- [B][n] x [m][n]', [B][m] x [m][m]' and [B][m] x [m][n]
// -- and belong on the matrix pipe, where they read the shared matrix once per 32 spectra instead of once per spectrum
// (hyper_kernel streamed 4.2 MB from L2 per spectrum and outer iteration: 126 GB per 1024-spectrum step, more than half of its
// 22 ms; qvec_kernel 2.1 MB).  C[b][i] = sum_k f(A[b][k]) M[i][k] with
//   MODE 0: f = identity                      (A = x: the model impedance rm x)
//   MODE 1: f(v) = (v - rv[b][k])^2           (A = the MODE-0 result: vmm @ resid^2)
//   MODE 2: f(v) = v * (v * rv[b][k])         (A = w, rv = the data: C = -sum + l1, the QP's q; M is given TRANSposed, [K][ni])
// on v_mfma_f64_16x16x4 from slabs of 16 k staged through LDS, the next slab requested while this one is multiplied.  A row's
// sum runs over k in ascending order whatever the batch holds, so a spectrum's bits do not depend on its neighbours or on B.
// grid = (ceil(ni / 64), ceil(B / 32)), 256 threads (4 wavefronts as 2 x 2 of 16 spectra x 32 rows).
// ---------------------------------------------------------------------------------------------------------
static constexpr int BG_TB = 32, BG_TI = 64, BG_K = 16, BG_LD = BG_K + 1;     // 32 spectra x 64 rows per workgroup, K slabs of 16
template <int MODE, bool TRANS>
__global__ __launch_bounds__(256) void batch_products_kernel(int B, int ni, int K, const double* __restrict__ A, int lda,
                                                             const double* __restrict__ rv, const double* __restrict__ M, int ldm,
                                                             const int* __restrict__ active, double* __restrict__ C,
                                                             const double* __restrict__ l1, double l1_scalar) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    __shared__ double sA[BG_TB * BG_LD];
    __shared__ double sM[BG_TI * BG_LD];
    const int b0 = blockIdx.y * BG_TB, i0 = blockIdx.x * BG_TI;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // nothing to do for a slab of spectra that have all converged
    if (active && !__any(lane < BG_TB && b0 + lane < B && active[b0 + lane] != 0)) return;
    const int wb = (wv >> 1) * 16, wi = (wv & 1) * 32;       // wavefront: 16 spectra x 32 rows
    const int li = lane & 15, kq = lane >> 4;
    v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
    // staging.  A (and M when it is stored [ni][K]): row sr, four consecutive k.  M stored [K][ni]: slab row tk, four consecutive i
    const int sr = tid >> 2, sk = (tid & 3) * 4;
    const int tk = tid >> 4, ti = (tid & 15) * 4;
    const bool evenA = ((lda | K) & 1) == 0, evenM = TRANS ? (ldm & 1) == 0 : ((ldm | K) & 1) == 0;
    const bool ldA = sr < BG_TB && b0 + sr < B, ldM = i0 + sr < ni;
    const double* arow = A + (size_t)(ldA ? b0 + sr : 0) * lda;
    const double* rrow = MODE ? rv + (size_t)(ldA ? b0 + sr : 0) * lda : nullptr;
    const double* mrow = TRANS ? M + i0 + ti : M + (size_t)(ldM ? i0 + sr : 0) * ldm;
    double va[4], vr[4], vm[4];
    auto fetch = [&](int k0) {
        const int kk = k0 + sk;
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = 0.0; vr[e] = 0.0; vm[e] = 0.0; }
        if (ldA) {
            if (evenA && kk + 3 < K) {
                const double2 t0 = *reinterpret_cast<const double2*>(arow + kk), t1 = *reinterpret_cast<const double2*>(arow + kk + 2);
                va[0] = t0.x; va[1] = t0.y; va[2] = t1.x; va[3] = t1.y;
                if (MODE) {
                    const double2 r0 = *reinterpret_cast<const double2*>(rrow + kk), r1 = *reinterpret_cast<const double2*>(rrow + kk + 2);
                    vr[0] = r0.x; vr[1] = r0.y; vr[2] = r1.x; vr[3] = r1.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kk + e < K) { va[e] = arow[kk + e]; if (MODE) vr[e] = rrow[kk + e]; }
            }
        }
        if (TRANS) {
            if (k0 + tk < K) {
                const double* mp = mrow + (size_t)(k0 + tk) * ldm;
                if (evenM && i0 + ti + 3 < ni) {
                    const double2 t0 = *reinterpret_cast<const double2*>(mp), t1 = *reinterpret_cast<const double2*>(mp + 2);
                    vm[0] = t0.x; vm[1] = t0.y; vm[2] = t1.x; vm[3] = t1.y;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (i0 + ti + e < ni) vm[e] = mp[e];
                }
            }
        } else if (ldM) {
            if (evenM && kk + 3 < K) {
                const double2 t0 = *reinterpret_cast<const double2*>(mrow + kk), t1 = *reinterpret_cast<const double2*>(mrow + kk + 2);
                vm[0] = t0.x; vm[1] = t0.y; vm[2] = t1.x; vm[3] = t1.y;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (kk + e < K) vm[e] = mrow[kk + e];
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BG_K) {
        __syncthreads();       // previous slab consumed
        if (sr < BG_TB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double v = va[e];
                if (MODE == 1) { const double r = va[e] - vr[e]; v = r * r; }      // (entries past K: 0 - 0)
                if (MODE == 2) v = va[e] * (va[e] * vr[e]);
                sA[sr * BG_LD + sk + e] = v;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (TRANS) sM[(ti + e) * BG_LD + tk] = vm[e];
            else sM[sr * BG_LD + sk + e] = vm[e];
        }
        __syncthreads();
        if (k0 + BG_K < K) fetch(k0 + BG_K);
#pragma unroll
        for (int s_ = 0; s_ < BG_K; s_ += 4) {
            // operands of v_mfma_f64_16x16x4: lane (li, kq) supplies element [row li][k = kq] of either factor; with the shared
            // matrix as the first operand the accumulator of lane (li, kq), register r is C[spectrum li][row kq + 4 r]
            const double a0 = sA[(wb + li) * BG_LD + s_ + kq];
            const double m0 = sM[(wi + li) * BG_LD + s_ + kq], m1 = sM[(wi + 16 + li) * BG_LD + s_ + kq];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(m0, a0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(m1, a0, acc[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bb = b0 + wb + li, ii = i0 + wi + c * 16 + kq + 4 * r;
            if (bb < B && ii < ni && (!active || MODE != 2 || active[bb])) {
                double v = acc[c][r];
                if (MODE == 2) v = -v + (l1 ? l1[ii] : l1_scalar);
                C[(size_t)bb * ni + ii] = v;
            }
        }
}

// q_b = -rm' (w_b (w_b y_b)) + l1 for a batch that shares rm (row-major [m][n])
void launch_qvec_batched(hipStream_t s, int B, int m, int n, const double* rm, int ldrm, const double* w, const double* y,
                         const double* l1, double l1_scalar, double* q, const int* active) {
    const dim3 grid((n + BG_TI - 1) / BG_TI, (B + BG_TB - 1) / BG_TB);
    hipLaunchKernelGGL((batch_products_kernel<2, true>), grid, dim3(256), 0, s, B, n, m, w, m, y, rm, ldrm, active, q, l1, l1_scalar);
}

// ---------------------------------------------------------------------------------------------------------
// hyper-parameter update + weights + convergence for one outer iteration.  grid = B
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(HT, 4) void hyper_kernel(FitState st, int it) {     // <= 128 VGPRs: two workgroups per CU
    extern __shared__ double sm[];
    __shared__ double red[HNW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!st.active[b]) return;
    HPROF_START();
    const int n = st.n, m = st.m, ns = st.ns, nd = n - ns;
    double* xs = sm;              // [n]   new x
    double* bsum = xs + n;        // [nd]  off-diagonal row sums of gamma @ diag(sqrt s)
    double* gdia = bsum + nd;     // [nd]  diag(gamma)
    double* sq = gdia + nd;       // [nd]  sqrt(s_k)
    double* xh = sq + nd;         // [nd]  sign(x) sqrt|x|
    double* tmp = xh + nd;        // [max(m, nd)]
    const int tl = m > nd ? m : nd;
    double* tmp2 = tmp + tl;      // [max(m, nd)]
    // [3][2 nd - 1] first columns of the Toeplitz penalty blocks (uniform ln-tau grids), mirrored around index nd - 1
    // so that entry (i, j) is cx[i - j + nd - 1] without an absolute value
    // (toeplitz_m == 2, large joint fits: they start inside the second m-vector, behind the nd entries the Toeplitz phases use
    // of it -- the weights phase, which needs all m, comes after the last use of the columns)
    double* ctp = st.toeplitz_m == 2 ? tmp2 + nd : tmp2 + tl;
    const int cw = 2 * nd - 1;
    const bool tpl = st.toeplitz_m != 0;
    if (tpl) {
        for (int e = tid; e < 3 * cw; e += HT) {
            const int k = e / cw, o = e % cw;
            const int dd = o >= nd - 1 ? o - (nd - 1) : (nd - 1) - o;
            ctp[e] = st.mk[k][(size_t)ns * st.ldm + ns + dd];
        }
    }

    if (st.qp_status[b] < 0) {    // QP broke down at its start point: cvxopt raises, DRTMD flags the observation
        if (tid == 0) { st.active[b] = 0; st.fit_status[b] = -1; st.outer_iters[b] = it + 1; }
        return;
    }
    double* xg = st.x + (size_t)b * n;
    for (int i = tid; i < n; i += HT) xs[i] = xg[i];
    __syncthreads();
    const double* xd = xs + ns;
    for (int i = tid; i < nd; i += HT) {
        const double v = xd[i];
        const double sg = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
        xh[i] = sg * sqrt(fabs(v));
    }
    __syncthreads();
    HPROF(0);

    // solve_s + solve_rho (qphb.py:320-356, 385-405) of one derivative order on one block of x: the DRT coefficients
    // (with the G matrix for k = 0) or the x_dop block (qphb.py:822-933, no G matrix)
    const int tid0 = tid;
    auto update_block = [&](const int k, const double* xd, const int nd, const int off, const bool tpl, const bool use_g,
                            const double alpha, const double s0, const double sigma, const double ra, const double r0,
                            double* rho_out, const double* xmx_in, const double reff, const bool fresh_rows = false) {
        // (fresh_rows, the second use of this block: thread and row offsets are formed again instead of being carried -- in
        // scratch, at 128 VGPRs -- from the kernel's start)
        const int tid = fresh_rows ? opaque_lane(tid0) : tid0;
        double* sk = st.s + ((size_t)b * 3 + k) * n + off;
        const double* Mk = st.mk[k] + (size_t)off * st.ldm + off;
        const double* M1 = st.mk[1] + (size_t)off * st.ldm + off;
        const double beta = (alpha - 1.0) / s0;
        const double sig2 = 2.0 * sigma * sigma;
        for (int i = tid; i < nd; i += HT) sq[i] = sqrt(sk[i]);
        __syncthreads();
        // gamma = X M X + G/(2 sigma^2) + beta I ; gu = gamma @ diag(sqrt s), zero diagonal
        double gmax = 0.0;
        if (tpl) {
            // Toeplitz blocks: gu_ij = x_i m_|i-j| (x_j sqrt s_j) [+ (xh_i / sig2) m1_|i-j| (xh_j sqrt s_j) for k = 0],
            // so the row sums are two convolutions with vectors prepared once per k; per element: two LDS reads, a
            // multiply, an add and a max.  The maximum is kept per lane and reduced once after all rows.
            double* vs = tmp;            // x_j sqrt(s_j)
            double* vh = tmp2;           // xh_j sqrt(s_j)
            for (int i = tid; i < nd; i += HT) { vs[i] = xd[i] * sq[i]; vh[i] = xh[i] * sq[i]; }
            __syncthreads();
            const double* ck = ctp + k * cw + (nd - 1);      // ck[i - j]
            const double* c1 = ctp + cw + (nd - 1);
            // the row sums leave out j = i: for their duration the lag-0 entries of the Toeplitz columns are zero in LDS (a
            // product with zero adds nothing and cannot raise the maximum) instead of two selects per product
            const double ck00 = ck[0], c100 = c1[0];
            __syncthreads();
            if (tid == 0) { ctp[k * cw + (nd - 1)] = 0.0; if (use_g) ctp[cw + (nd - 1)] = 0.0; }
            __syncthreads();
            // two adjacent rows per thread (see toeplitz_matvec): the column index is uniform across the wavefront and row
            // i+1 reuses row i's Toeplitz operands one column later; per row the same sums in the same order as one row per
            // thread gave
            double lmax = 0.0;
            for (int p = tid; 2 * p < nd; p += HT) {
                const int ia = 2 * p, ib = ia + 1;
                const bool hb = ib < nd;
                const double xia = xd[ia], xra = reff * xia;                 // rho_k_eff * x_i (1 under eff_hp, qphb.py:747-750)
                const double xib = hb ? xd[ib] : 0.0, xrb = reff * xib;
                const double* cka = ck + ia;
                // (columns beyond the reach of the penalty matrices hold exact zeros in ck and c1: see toeplitz_matvec)
                int jlo = 0, jhi = nd;
                if (st.toep_reach >= 0) {
                    jlo = ia - st.toep_reach > 0 ? (ia - st.toep_reach) & ~3 : 0;
                    const int e_ = (ia + 2 + st.toep_reach + 3) & ~3;
                    jhi = e_ < nd ? e_ : nd;
                }
                if (use_g) {
                    const double xhsa = xh[ia] / sig2, xhsb = hb ? xh[ib] / sig2 : 0.0;
                    const double* c1a = c1 + ia;
                    double sa0 = 0.0, sa1 = 0.0, sb0 = 0.0, sb1 = 0.0, ma0 = 0.0, mb0 = 0.0;
                    double pk = hb ? cka[1 - jlo] : cka[-jlo], p1 = hb ? c1a[1 - jlo] : c1a[-jlo];
                    int j = jlo;
                    for (; j + 1 < jhi; j += 2) {
                        const double k0 = cka[-j], k1 = cka[-j - 1], q0 = c1a[-j], q1 = c1a[-j - 1];
                        const double v0 = vs[j], v1 = vs[j + 1], h0 = vh[j], h1 = vh[j + 1];
                        const double ga0 = xra * (k0 * v0) + xhsa * (q0 * h0);
                        const double ga1 = xra * (k1 * v1) + xhsa * (q1 * h1);
                        const double gb0 = xrb * (pk * v0) + xhsb * (p1 * h0);
                        const double gb1 = xrb * (k0 * v1) + xhsb * (q0 * h1);
                        pk = k1; p1 = q1;
                        sa0 += ga0; sa1 += ga1; sb0 += gb0; sb1 += gb1;
                        ma0 = fmax(ma0, fmax(fabs(ga0), fabs(ga1)));
                        mb0 = fmax(mb0, fmax(fabs(gb0), fabs(gb1)));
                    }
                    if (j < jhi) {
                        const double ga0 = xra * (cka[-j] * vs[j]) + xhsa * (c1a[-j] * vh[j]);
                        const double gb0 = xrb * (pk * vs[j]) + xhsb * (p1 * vh[j]);
                        sa0 += ga0; ma0 = fmax(ma0, fabs(ga0));
                        sb0 += gb0; mb0 = fmax(mb0, fabs(gb0));
                    }
                    lmax = fmax(lmax, ma0);
                    bsum[ia] = sa0 + sa1;
                    gdia[ia] = ((xra * ck00) * xia + ((xh[ia] * c100) * xh[ia]) / sig2) + beta;
                    if (hb) {
                        lmax = fmax(lmax, mb0);
                        bsum[ib] = sb0 + sb1;
                        gdia[ib] = ((xrb * ck00) * xib + ((xh[ib] * c100) * xh[ib]) / sig2) + beta;
                    }
                } else {
                    double sa0 = 0.0, sa1 = 0.0, sa2 = 0.0, sa3 = 0.0, sb0 = 0.0, sb1 = 0.0, sb2 = 0.0, sb3 = 0.0;
                    double ma0 = 0.0, mb0 = 0.0;          // (one running maximum per row: a maximum does not care about the grouping)
                    double pk = hb ? cka[1 - jlo] : cka[-jlo];
                    int j = jlo;
                    for (; j + 3 < jhi; j += 4) {
                        const double k0 = cka[-j], k1 = cka[-j - 1], k2 = cka[-j - 2], k3 = cka[-j - 3];
                        const double v0 = vs[j], v1 = vs[j + 1], v2 = vs[j + 2], v3 = vs[j + 3];
                        const double ta0 = k0 * v0, ta1 = k1 * v1, ta2 = k2 * v2, ta3 = k3 * v3;
                        const double tb0 = pk * v0, tb1 = k0 * v1, tb2 = k1 * v2, tb3 = k2 * v3;
                        pk = k3;
                        sa0 += ta0; sa1 += ta1; sa2 += ta2; sa3 += ta3;
                        sb0 += tb0; sb1 += tb1; sb2 += tb2; sb3 += tb3;
                        ma0 = fmax(ma0, fmax(fmax(fabs(ta0), fabs(ta1)), fmax(fabs(ta2), fabs(ta3))));
                        mb0 = fmax(mb0, fmax(fmax(fabs(tb0), fabs(tb1)), fmax(fabs(tb2), fabs(tb3))));
                    }
                    for (; j < jhi; ++j) {
                        const double k0 = cka[-j];
                        const double ta0 = k0 * vs[j], tb0 = pk * vs[j];
                        pk = k0;
                        sa0 += ta0; ma0 = fmax(ma0, fabs(ta0));
                        sb0 += tb0; mb0 = fmax(mb0, fabs(tb0));
                    }
                    lmax = fmax(lmax, fabs(xra) * ma0);
                    bsum[ia] = xra * ((sa0 + sa1) + (sa2 + sa3));
                    gdia[ia] = (xra * ck00) * xia + beta;
                    if (hb) {
                        lmax = fmax(lmax, fabs(xrb) * mb0);
                        bsum[ib] = xrb * ((sb0 + sb1) + (sb2 + sb3));
                        gdia[ib] = (xrb * ck00) * xib + beta;
                    }
                }
            }
            gmax = hw_max(lmax);
            __syncthreads();
            if (tid == 0) { ctp[k * cw + (nd - 1)] = ck00; if (use_g) ctp[cw + (nd - 1)] = c100; }
        } else
        for (int i = fresh_rows ? opaque_lane(wv) : wv; i < nd; i += HNW) {     // (fresh_rows: the row addresses are formed here, not carried in scratch)
            const double* row = Mk + (size_t)i * st.ldm;
            const double* row1 = M1 + (size_t)i * st.ldm;
            const double xi = xd[i], xhi = xh[i], xr = reff * xi;
            double sacc = 0.0, mxx = 0.0, dg = 0.0;
            for (int j = lane; j < nd; j += 64) {
                const double mij = row[j];
                double g = (xr * mij) * xd[j];
                if (use_g) g += ((xhi * row1[j]) * xh[j]) / sig2;
                if (j == i) dg = g + beta;
                else {
                    const double gu = g * sq[j];
                    sacc += gu;
                    mxx = fmax(mxx, fabs(gu));
                }
            }
            sacc = hw_sum(sacc);
            mxx = hw_max(mxx);
            dg = hw_sum(dg);      // exactly one lane holds the diagonal entry
            gmax = fmax(gmax, mxx);
            if (lane == 0) { bsum[i] = sacc; gdia[i] = dg; }
        }
        gmax = blk_max(gmax, red);   // also orders bsum/gdia writes before the reads below
        for (int i = tid; i < nd; i += HT) {
            const double gd = gdia[i];
            double sh;
            if (gmax > 1e-10) {
                const double bb = bsum[i];
                const double sgn = (bb > 0.0) ? 1.0 : ((bb < 0.0) ? -1.0 : 0.0);
                const double u = (-bb + sgn * sqrt(bb * bb + 4.0 * gd * (alpha - 1.0))) / (2.0 * gd);
                sh = u * u;
            } else {
                sh = (alpha - 1.0) / gd;
            }
            if (sh != sh) sh = 1.0;          // s_hat[isnan] = 1
            if (sh <= 0.0) sh = 1e-15;       // caller's floor (qphb.py:780)
            sk[i] = sh;
            tmp[i] = sqrt(sh) * xd[i];       // v = S^1/2 x for solve_rho
        }
        __syncthreads();
        if (tpl) toeplitz_matvec(ctp + k * cw + (nd - 1), nd, tmp, tmp2, st.toep_reach);
        else rows_matvec(Mk, st.ldm, nd, nd, tmp, tmp2);
        __syncthreads();
        double part = 0.0;
        for (int i = tid; i < nd; i += HT) part += tmp[i] * tmp2[i];
        const double xsmsx = blk_sum(part, red);
        if (tid == 0) {
            const double rb = ra / r0;
            rho_out[k] = ra / (xsmsx / xmx_in[k] + rb);
        }
        __syncthreads();
    };
    for (int k = 0; k < 3; ++k) {
        if (!(st.opts.derivative_weights[k] > 0.0)) continue;
        update_block(k, xs + ns, n - ns, ns, st.toeplitz_m != 0, k == 0, st.opts.s_alpha[k], st.opts.s_0[k],
                     st.opts.sigma_ds[k], st.opts.rho_alpha[k], st.opts.rho_0[k], st.rho + (size_t)b * 3,
                     st.xmx + (size_t)b * 3, st.opts.eff_hp ? 1.0 : st.rho[(size_t)b * 3 + k]);
        HPROF(1 + k);
    }
    const int tidw = opaque_lane(tid);      // (a fresh copy: the per-lane offsets of everything below are formed here, not carried in scratch)
    if (st.prepared && st.desc.dop_size > 0) {
        for (int k = 0; k < 3; ++k) {
            if (!(st.desc.dop_derivative_weights[k] > 0.0)) continue;
            update_block(k, xs + st.desc.dop_start, st.desc.dop_size, st.desc.dop_start, false, false,
                         st.desc.dop_s_alpha[k], st.desc.dop_s_0[k], 1.0, st.desc.dop_rho_alpha[k],
                         st.desc.dop_rho_0[k], st.dop_rho + (size_t)b * 3, st.dop_xmx + (size_t)b * 3,
                         st.opts.eff_hp ? 1.0 : st.dop_rho[(size_t)b * 3 + k], true);
        }
    }

    if (it == 0 && !st.continue_mode) {   // xmx_norms frozen after the first iteration (drt1d.py:946-951)
        for (int k = 0; k < 3; ++k) {
            const double* Mk = st.mk[k] + (size_t)ns * st.ldm + ns;
            if (tpl) toeplitz_matvec(ctp + k * cw + (nd - 1), nd, xd, tmp2, st.toep_reach);
            else rows_matvec(Mk, st.ldm, nd, nd, xd, tmp2);
            __syncthreads();
            double part = 0.0;
            for (int i = tidw; i < nd; i += HT) part += xd[i] * tmp2[i];
            const double v = blk_sum(part, red);
            if (tidw == 0) st.xmx[(size_t)b * 3 + k] = v;
            __syncthreads();
        }
        if (st.prepared && st.desc.dop_size > 0) {      // dop_xmx_norms (drt1d.py:953-960)
            const int d0 = st.desc.dop_start, dn = st.desc.dop_size;
            for (int k = 0; k < 3; ++k) {
                rows_matvec(st.mk[k] + (size_t)d0 * st.ldm + d0, st.ldm, dn, dn, xs + d0, tmp2);
                __syncthreads();
                double part = 0.0;
                for (int i = tidw; i < dn; i += HT) part += xs[d0 + i] * tmp2[i];
                const double v = blk_sum(part, red);
                if (tidw == 0) st.dop_xmx[(size_t)b * 3 + k] = v;
                __syncthreads();
            }
        }
    }

    HPROF(4);
    // weights
    double* wg = st.w + (size_t)b * m;
    double* tmp3 = ctp + 3 * cw;       // [2][m], only present (and only touched) when outlier_p is set
    estimate_weights_dev(st, b, st.vmm, xs, tmp, tmp2, tmp3, tmp3 + m, st.est_w + (size_t)b * m, wg, 0, -1, -1.0, st.premv);
    HPROF(5);

    // convergence (qphb.py:597-603, 969-970)
    double* xin = st.x_in + (size_t)b * n;
    double mrel = 0.0, mabs = 0.0, sx = 0.0;
    for (int i = tidw; i < n; i += HT) {
        const double xi0 = xin[i];
        const double dlt = xs[i] - xi0;
        mrel = fmax(mrel, fabs(dlt / (xi0 + 1e-15)));
        mabs = fmax(mabs, fabs(dlt));
        sx += xi0;
    }
    mrel = blk_max(mrel, red);
    mabs = blk_max(mabs, red);
    sx = blk_sum(sx, red);
    const double atol = sx / (double)n * 1e-3;
    const bool conv = (mrel <= st.opts.xtol) || (mabs <= atol);
    for (int i = tidw; i < n; i += HT) xin[i] = xs[i];
    if (st.hist_b == b && it < st.hist_cap) {
        for (int i = tidw; i < n; i += HT) st.hist_x[(size_t)it * n + i] = xs[i];
        for (int i = tidw; i < m; i += HT) st.hist_w[(size_t)it * m + i] = wg[i];
        if (tidw < 3) st.hist_rho[(size_t)it * 3 + tidw] = st.rho[(size_t)b * 3 + tidw];
        if (tidw < 3 && st.prepared && st.desc.dop_size > 0)
            st.hist_dop_rho[(size_t)it * 3 + tidw] = st.dop_rho[(size_t)b * 3 + tidw];
        if (tidw == 0) { st.hist_qp[it + 1] = st.qp_iters[b]; st.hist_rows[0] = it + 1; }
    }
    if (st.prepared && st.desc.vz_index >= 0 && st.continue_mode != 2) {
        // drt1d.py:973-979: the vz_offset column becomes the current prediction of the matrix without the baseline and
        // offset columns (rzm_vz was copied while the offset column was still zero), impedance rows negated, times the
        // strength vector.  The weights above were estimated with the previous column, as in iterate_qphb.
        __syncthreads();
        const int vz = st.desc.vz_index;
        for (int i = tidw; i < n; i += HT)
            if (i == vz || (i >= st.desc.vb_start && i < st.desc.vb_start + st.desc.vb_size)) xs[i] = 0.0;
        __syncthreads();
        double* rmb = st.rm_rw + (size_t)b * st.rm_stride;
        if (st.premv) { for (int i = tidw; i < m; i += HT) tmp[i] = st.premv[(2 * (size_t)gridDim.x + b) * m + i]; }
        else rows_matvec(rmb, st.ldrm, m, n, xs, tmp);
        __syncthreads();
        // warm restart (drt1d.py:1295-1298, 1353): rzm_vz is a copy made when _continue_from_init was entered, with the baseline
        // columns zeroed but the offset column as the previous loop left it -- its contribution comes on top
        const bool frozen = st.continue_mode == 1 && st.vz_entry != nullptr;
        const double xvz = frozen ? xg[vz] : 0.0;
        const double* vze = frozen ? st.vz_entry + (size_t)b * m : nullptr;
        for (int i = tidw; i < m; i += HT) {
            const double pred = frozen ? tmp[i] + vze[i] * xvz : tmp[i];
            rmb[(size_t)i * st.ldrm + vz] = (i < st.desc.num_chrono ? pred : -pred) * st.vz_strength[i];
        }
    }
    const bool stop = conv && it + 1 >= st.min_iter;          // `converged and it >= min_iter - 1` (drt1d.py:1356)
    if (st.opts.update_scale && st.opts.scale_data && it >= 1 && !stop && it + 1 < st.opts.max_iter && !st.continue_mode) {
        // drt1d.py:903-927, executed by the reference at the top of the NEXT iteration (it > 1 there): keep the apparent
        // polarisation resistance at rp_scale by rescaling the data and everything that carries its units
        __syncthreads();
        double part = 0.0;
        for (int i = ns + tidw; i < n; i += HT) part += fabs(xs[i]);
        const double rp = blk_sum(part, red) * st.basis_area;                 // predict_r_p(absolute=True, raw=True)
        const double sf = sqrt(st.opts.rp_scale / rp);                        // damped scale factor
        for (int i = tidw; i < n; i += HT) xin[i] *= sf;
        double* rvb = st.rv + (size_t)b * m;
        double* ewb = st.est_w + (size_t)b * m;
        for (int i = tidw; i < m; i += HT) { rvb[i] *= sf; ewb[i] /= sf; wg[i] /= sf; }
        if (tidw < 3) {
            st.xmx[(size_t)b * 3 + tidw] *= sqrt(sf);                          // as coded upstream (drt1d.py:918)
            if (st.prepared && st.desc.dop_size > 0) st.dop_xmx[(size_t)b * 3 + tidw] *= sqrt(sf);
        }
        if (tidw == 0) { st.coef_scale[b] /= sf; st.var_floor[b] *= sf * sf; }
    }
    HPROF(6);
    HPROF_COUNT(15);
    if (tidw == 0) {
        st.outer_iters[b] = it + 1;
        if (stop) { st.active[b] = 0; st.fit_status[b] = 0; }
        else if (it + 1 >= st.opts.max_iter) { st.active[b] = 0; st.fit_status[b] = 1; }
        if (!stop && it + 1 < st.opts.max_iter) atomicAdd(st.n_active, 1);
    }
}

__global__ void record_init_qp_kernel(FitState st) {
    if (st.hist_b >= 0 && threadIdx.x == 0) st.hist_qp[0] = st.qp_iters[st.hist_b];
}

// specials: diagonal penalties of the padded M_k (drt1d.py:5880-5896) and the special columns of the stacked
// response matrix rm = [Re; Im] (drt1d.py:5825-5857)
__global__ void assemble_rm_kernel(FitState st, const double* __restrict__ a_re, const double* __restrict__ a_im,
                                   const double* __restrict__ freq, double* __restrict__ rm, int idx_rinf,
                                   int idx_induc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;   // column of rm
    const int i = blockIdx.y;                              // row in [0, 2nf)
    const int n = st.n, nf = st.nf, ns = st.ns, ntau = n - ns;
    if (j >= n) return;
    const bool im = i >= nf;
    const int r = im ? i - nf : i;
    double v;
    if (j >= ns) v = im ? a_im[(size_t)r * ntau + (j - ns)] : a_re[(size_t)r * ntau + (j - ns)];
    else if (j == idx_rinf) v = im ? 0.0 : 1.0;
    else if (j == idx_induc) v = im ? (2.0 * 3.141592653589793 * freq[r]) * st.opts.inductance_scale : 0.0;
    else v = 0.0;
    rm[(size_t)i * st.ldrm + j] = v;
}

__global__ void special_penalty_kernel(double* m0, double* m1, double* m2, int ld, int idx_rinf, int idx_induc,
                                       double pen_r, double pen_l) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double* ms[3] = {m0, m1, m2};
        for (int k = 0; k < 3; ++k) {
            if (idx_rinf >= 0) ms[k][(size_t)idx_rinf * ld + idx_rinf] = pen_r;
            if (idx_induc >= 0) ms[k][(size_t)idx_induc * ld + idx_induc] = pen_l;
        }
    }
}

__global__ void make_h_kernel(double* h, int n, int ns, int nonneg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // make_h_constraint (qphb.py:521-557): R_inf / inductance are always non-negative specials
    h[i] = (nonneg || i < ns) ? 0.0 : 1e5;
}

size_t hyper_lds_bytes(int n, int m, int ns) {
    const int nd = n - ns;
    const int tl = m > nd ? m : nd;
    return ((size_t)n + 4 * (size_t)nd + 2 * (size_t)tl + 3 * (size_t)(2 * nd - 1)) * sizeof(double);
}
static constexpr size_t kLdsLimit = 160 * 1024 - 256;     // per-workgroup LDS of a gfx950 CU, less the static part

int launch_prep(hipStream_t s, const FitState& st, int B) {
    if (st.prepared) hipLaunchKernelGGL(prep_prepared_kernel, dim3(B), dim3(HT), 0, s, st);
    else hipLaunchKernelGGL(prep_kernel, dim3(B), dim3(HT), 0, s, st);
    return 0;
}

static int set_lds(const void* f, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return 0;
}

// w[b][i] = 1 inside [r0, r1), 0 elsewhere: the QP of init_weights_separately sees one data block at a time
__global__ void row_mask_kernel(int m, int r0, int r1, double* __restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) w[(size_t)blockIdx.y * m + i] = (i >= r0 && i < r1) ? 1.0 : 0.0;
}

void launch_row_mask(hipStream_t s, int B, int m, int r0, int r1, double* w) {
    hipLaunchKernelGGL(row_mask_kernel, dim3((m + 255) / 256, B), dim3(256), 0, s, m, r0, r1, w);
}

// hybrid_weight_factor_method='weight' (drt1d.py:748-760): chrono / EIS row factors from the ratio of the two blocks'
// weight scales mean(est_w^-2)^-1/2; a fixed value (> 0) overrides either.  wrow[b][m], wfac[b] = {chrono, eis}.  grid B
__global__ __launch_bounds__(HT) void weight_method_kernel(FitState st, double fixed_chrono, double fixed_eis,
                                                           double* __restrict__ wrow, double* __restrict__ wfac) {
    __shared__ double red[HNW];
    const int b = blockIdx.x, tid = threadIdx.x, m = st.m, nc = st.desc.num_chrono;
    const double* ew = st.est_w + (size_t)b * m;
    double sc = 0.0, se = 0.0;
    for (int i = tid; i < nc; i += HT) sc += 1.0 / (ew[i] * ew[i]);
    for (int i = nc + tid; i < m; i += HT) se += 1.0 / (ew[i] * ew[i]);
    sc = blk_sum(sc, red);
    se = blk_sum(se, red);
    const double cws = 1.0 / sqrt(sc / (double)nc), ews = 1.0 / sqrt(se / (double)(m - nc));
    const double ratio = sqrt(sqrt(ews / cws));
    const double cf = fixed_chrono > 0.0 ? fixed_chrono : ratio;
    const double ef = fixed_eis > 0.0 ? fixed_eis : 1.0 / ratio;
    for (int i = tid; i < m; i += HT) wrow[(size_t)b * m + i] = i < nc ? cf : ef;
    if (tid == 0) { wfac[2 * b] = cf; wfac[2 * b + 1] = ef; }
}

void launch_weight_method(hipStream_t s, const FitState& st, int B, double fixed_chrono, double fixed_eis, double* wrow,
                          double* wfac) {
    hipLaunchKernelGGL(weight_method_kernel, dim3(B), dim3(HT), 0, s, st, fixed_chrono, fixed_eis, wrow, wfac);
}

int launch_init_weights(hipStream_t s, const FitState& st, int B, int stage, int r0, int r1) {
    // x + two row vectors, two more only for the outlier branch
    const size_t lds = ((size_t)st.n + (st.opts.outlier_p > 0.0 ? 4 : 2) * (size_t)st.m) * sizeof(double);
    if (lds > kLdsLimit) { set_error("initialize_weights: m too large for the LDS-resident kernel"); return HIPDRT_E_INVALID; }
    if (int rc = set_lds(reinterpret_cast<const void*>(init_weights_kernel), lds)) return rc;
    hipLaunchKernelGGL(init_weights_kernel, dim3(B), dim3(HT), lds, s, st, stage, r0, r1);
    if (stage == 1 || stage == 3) hipLaunchKernelGGL(record_init_qp_kernel, dim3(1), dim3(64), 0, s, st);
    return 0;
}

// Ingredients of DRT.evaluate_llh(weights=estimate_weights(x), x) (drt1d.py:4457-4496, qphb.py:1347-1377) as the PFRT
// driver evaluates it after every step (drt1d.py:2618-2622): weights re-estimated from the current x alone
// (est_weights=None), then rss = x'(WR)'(WR)x - 2 (Wy)'(WR)x + (Wy)'(Wy) and sum(log w).  grid = B.
// stored = 1: the weights are the fit's own est_weights instead (DRT.evaluate_llh() / evaluate_rss() with weights=None);
// stored = 2: weights='uniform' (drt1d.py:4436-4441, 4465-4470): within each domain -- chrono rows [0, num_chrono), impedance
// rows after them -- every weight is the mean of that domain's est_weights, which is what DRTMD.fit_observation records per
// observation with its default llh_kw / rss_kw (drtmd.py:129-131, 259-260); stored = 3: one scalar weight for every row.
__global__ __launch_bounds__(HT) void llh_kernel(FitState st, double* __restrict__ rss, double* __restrict__ slw, int stored,
                                                 double scalar_w) {
    extern __shared__ double sm[];
    __shared__ double red[HNW];
    const int b = blockIdx.x, tid = threadIdx.x, n = st.n, m = st.m;
    double* xs = sm;
    double* yh = xs + n;        // rm @ x
    double* r2 = yh + m;        // residual^2
    double* sh = r2 + m;        // vmm @ residual^2
    const double* rv = st.rv + (size_t)b * m;
    for (int i = tid; i < n; i += HT) xs[i] = st.x[(size_t)b * n + i];
    __syncthreads();
    rows_matvec(st.rm + (size_t)b * st.rm_stride, st.ldrm, m, n, xs, yh);
    __syncthreads();
    for (int i = tid; i < m; i += HT) { const double r = yh[i] - rv[i]; r2[i] = r * r; }
    __syncthreads();
    if (!stored) rows_matvec(st.vmm, m, m, m, r2, sh);
    __syncthreads();
    const double vf = st.var_floor[b];
    const int nchr = st.prepared ? st.desc.num_chrono : 0;
    double wmean_c = scalar_w, wmean_e = scalar_w;
    if (stored == 2) {
        double sc = 0.0, se = 0.0;
        for (int i = tid; i < m; i += HT) {
            const double w = st.est_w[(size_t)b * m + i];
            if (i < nchr) sc += w; else se += w;
        }
        sc = blk_sum(sc, red); se = blk_sum(se, red);
        wmean_c = nchr > 0 ? sc / (double)nchr : 0.0;
        wmean_e = m > nchr ? se / (double)(m - nchr) : 0.0;
    }
    double a = 0.0, c2 = 0.0, d = 0.0, lw = 0.0;
    for (int i = tid; i < m; i += HT) {
        double w;
        if (stored == 1) {
            w = st.est_w[(size_t)b * m + i];
        } else if (stored >= 2) {
            w = i < nchr ? wmean_c : wmean_e;
        } else {
            double v = sh[i];
            if (v < vf) v = vf;
            w = fmax(1.0 / sqrt(v), 1e-10);
        }
        const double wy = w * yh[i], wr = w * rv[i];
        a += wy * wy; c2 += wr * wy; d += wr * wr; lw += log(w);
    }
    a = blk_sum(a, red); c2 = blk_sum(c2, red); d = blk_sum(d, red); lw = blk_sum(lw, red);
    if (tid == 0) { rss[b] = a - 2.0 * c2 + d; slw[b] = lw; }
}

int launch_llh(hipStream_t s, const FitState& st, int B, double* rss, double* slw, int stored, double scalar_w) {
    const size_t lds = (size_t)(st.n + 3 * st.m) * sizeof(double);
    if (int rc = set_lds(reinterpret_cast<const void*>(llh_kernel), lds)) return rc;
    hipLaunchKernelGGL(llh_kernel, dim3(B), dim3(HT), lds, s, st, rss, slw, stored, scalar_w);
    return 0;
}

// weights = weights * weight_factor at the top of every warm-restart iteration (drt1d.py:1322)
__global__ void scale_weights_kernel(FitState st, double factor) {
    const int b = blockIdx.y;
    if (!st.active[b]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < st.m) st.w[(size_t)b * st.m + i] *= factor;
}

// out[b][i] = w[b][i] * (rows ? rows[(batched ? b : 0)][i] : 1) * factor   (in place allowed); grid (ceil(m/256), B)
__global__ void scale_rows_kernel(int m, const double* __restrict__ w, const double* __restrict__ rows, int batched,
                                  double factor, const int* __restrict__ active, double* __restrict__ out) {
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double v = w[(size_t)b * m + i] * factor;
    if (rows) v *= rows[(size_t)(batched ? b : 0) * m + i];
    out[(size_t)b * m + i] = v;
}

// out[b][i] = rm_b[i][col]: the vz_offset column as a warm restart finds it; grid (ceil(m/256), B)
__global__ void copy_column_kernel(int m, const double* __restrict__ rm, long long rm_stride, int ldrm, int col,
                                   double* __restrict__ out) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[(size_t)b * m + i] = rm[(size_t)b * rm_stride + (size_t)i * ldrm + col];
}

void launch_copy_column(hipStream_t s, int B, int m, const double* rm, long long rm_stride, int ldrm, int col, double* out) {
    hipLaunchKernelGGL(copy_column_kernel, dim3((m + 255) / 256, B), dim3(256), 0, s, m, rm, rm_stride, ldrm, col, out);
}

void launch_scale_rows(hipStream_t s, int B, int m, const double* w, const double* rows, int batched, double factor,
                       const int* active, double* out) {
    hipLaunchKernelGGL(scale_rows_kernel, dim3((m + 255) / 256, B), dim3(256), 0, s, m, w, rows, batched, factor, active, out);
}

void launch_scale_weights(hipStream_t s, const FitState& st, int B, double factor) {
    hipLaunchKernelGGL(scale_weights_kernel, dim3((st.m + 255) / 256, B), dim3(256), 0, s, st, factor);
}

int launch_hyper(hipStream_t s, const FitState& st_in, int B, int it) {
    FitState st = st_in;
    const size_t extra = st.opts.outlier_p > 0.0 ? 2 * (size_t)st.m * sizeof(double) : 0;
    size_t lds = hyper_lds_bytes(st.n, st.m, st.ns) + extra;
    if (lds > kLdsLimit && st.toeplitz_m && extra == 0) {
        // large joint fits (config 5: m = 5120, n = 1078): no room for the mirrored Toeplitz columns NEXT to the two
        // m-vectors -- they share the second one's space (see hyper_kernel); if even that does not fit, the general
        // row-streaming form of the same updates reads the penalty blocks from L2 instead
        const size_t nd = (size_t)(st.n - st.ns), tl = (size_t)st.m > nd ? (size_t)st.m : nd, cols = 3 * (2 * nd - 1);
        const size_t second = tl > nd + cols ? tl : nd + cols;
        const size_t compact = ((size_t)st.n + 4 * nd + tl + second) * sizeof(double);
        if (compact <= kLdsLimit) { st.toeplitz_m = 2; lds = compact; }
        else { st.toeplitz_m = 0; lds -= cols * sizeof(double); }
    }
    if (lds > kLdsLimit) { set_error("hyper-parameter kernel: problem too large for LDS (m, n)"); return HIPDRT_E_INVALID; }
    if (int rc = set_lds(reinterpret_cast<const void*>(hyper_kernel), lds)) return rc;
    if (st.premv && st.premv_batched && st.opts.outlier_p <= 0.0) {
        // many fits sharing rm and vmm: both products of estimate_weights for the whole batch on the matrix pipe
        const dim3 grid((st.m + BG_TI - 1) / BG_TI, (B + BG_TB - 1) / BG_TB);
        hipLaunchKernelGGL((batch_products_kernel<0, false>), grid, dim3(256), 0, s, B, st.m, st.n, st.x, st.n, nullptr, st.rm,
                           st.ldrm, st.active, st.premv, nullptr, 0.0);
        hipLaunchKernelGGL((batch_products_kernel<1, false>), grid, dim3(256), 0, s, B, st.m, st.m, st.premv, st.m, st.rv, st.vmm,
                           st.m, st.active, st.premv + (size_t)B * st.m, nullptr, 0.0);
    } else if (st.premv && st.opts.outlier_p <= 0.0) {
        const size_t l0 = (size_t)st.n * sizeof(double), l1 = (size_t)st.m * sizeof(double);
        const dim3 grid((st.m + PREMV_ROWS - 1) / PREMV_ROWS, B);
        hipLaunchKernelGGL(premv_kernel, grid, dim3(HT), l0, s, st, 0, B);
        hipLaunchKernelGGL(premv_kernel, grid, dim3(HT), l1, s, st, 1, B);
    } else {
        st.premv = nullptr;
    }
    hipLaunchKernelGGL(hyper_kernel, dim3(B), dim3(HT), lds, s, st, it);
    return 0;
}

void launch_assemble_rm(hipStream_t s, const FitState& st, const double* a_re, const double* a_im, const double* freq,
                        double* rm, int idx_rinf, int idx_induc) {
    hipLaunchKernelGGL(assemble_rm_kernel, dim3((st.n + 255) / 256, st.m), dim3(256), 0, s, st, a_re, a_im, freq, rm,
                       idx_rinf, idx_induc);
}

void launch_special_penalty(hipStream_t s, double* m0, double* m1, double* m2, int ld, int idx_rinf, int idx_induc,
                            double pen_r, double pen_l) {
    hipLaunchKernelGGL(special_penalty_kernel, dim3(1), dim3(64), 0, s, m0, m1, m2, ld, idx_rinf, idx_induc, pen_r,
                       pen_l);
}

void launch_make_h(hipStream_t s, double* h, int n, int ns, int nonneg) {
    hipLaunchKernelGGL(make_h_kernel, dim3((n + 255) / 256), dim3(256), 0, s, h, n, ns, nonneg);
}

}  // namespace hipdrt

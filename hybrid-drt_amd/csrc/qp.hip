// Batched coneqp for gfx950: cvxopt.solvers.qp(P, q, G=-I, h) as called at hybdrt/models/qphb.py:512-519,
// one workgroup per problem, the whole interior-point trajectory inside one launch (no host round trips).
//
// Algorithm = cvxopt coneprog.coneqp restricted to one 'l' cone with G = -I (restated in oracle/coneqp.py,
// SURVEY.md Appendix A): default start point, Nesterov-Todd scaling W = diag(d), Mehrotra predictor-corrector
// (STEP 0.99, EXPON 3), KKT solves through the Cholesky factor of S = P + diag(d^-2) ('chol2' solver),
// cvxopt's stopping test.  Everything FP64.
//
// Work decomposition inside the 512-thread workgroup (8 wavefronts):
//   * the O(n) IPM vectors live in registers, element i owned by thread i % 512 (EPT elements per thread);
//     scalar reductions are wave shuffles + one LDS hop, in a fixed order (bit-reproducible run to run);
//   * S is never materialised: the left-looking blocked Cholesky (block 32) reads P, adds the diagonal on the
//     fly and writes L to a per-problem scratch matrix in HBM/L2.  Per block column: (1) every wavefront
//     computes 16x32 tiles  C = P - L[rows,:k] L[blk,:k]'  on v_mfma_f64_16x16x4_f64 with operands streamed
//     from L in 128-byte row segments, (2) the tiles are staged in an LDS panel, wavefront 0 factors the
//     32x32 diagonal block in registers (lane = row, pivots broadcast with v_readlane), (3) one thread per
//     panel row does the triangular solve against the LDS copy of L11, (4) rows are written back coalesced;
//   * triangular solves: per 32-block a register-resident substitution by wavefront 0 followed by a
//     thread-per-row (forward) / thread-per-column (backward) rank-32 update with coalesced reads of L;
//   * P x: one wavefront per row, coalesced, shuffle-reduced.
// LDS: panel (PR x 33 doubles) + L11 (32 x 33) + two length-n vectors  ~= 77 kB at PR = 224, so two
// workgroups share a CU and one's sequential phases overlap the other's MFMA phases.
#include "common.hpp"

namespace hipdrt {

typedef double v4d __attribute__((ext_vector_type(4)));

static constexpr int NB = 32;     // Cholesky block
static constexpr int PLD = 33;    // LDS panel row stride (doubles): odd => conflict-free row-per-lane access

__device__ __forceinline__ double bcast_lane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

template <int NW>
struct Reducer {
    double* buf;   // LDS [4][NW][4]
    int slot;
    __device__ Reducer(double* b) : buf(b), slot(0) {}
    // sums up to 4 values at once; every thread gets the totals
    template <int N>
    __device__ __forceinline__ void sum(double (&v)[N]) {
        static_assert(N <= 4, "");
        double* s = buf + (slot & 3) * NW * 4;
        ++slot;
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = wsum(v[i]);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) s[(threadIdx.x >> 6) * 4 + i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += s[w * 4 + i];
            v[i] = t;
        }
    }
    template <int N>
    __device__ __forceinline__ void max(double (&v)[N]) {
        static_assert(N <= 4, "");
        double* s = buf + (slot & 3) * NW * 4;
        ++slot;
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = wmax(v[i]);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) s[(threadIdx.x >> 6) * 4 + i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = s[i];
#pragma unroll
            for (int w = 1; w < NW; ++w) t = fmax(t, s[w * 4 + i]);
            v[i] = t;
        }
    }
};

struct QpSmem {
    double* panel;   // [PR][PLD]
    double* l11;     // [NB][PLD]   L11 (lower) of the current block column
    double* vec;     // [n]  rhs / solution of the triangular solves, x for the mat-vec
    double* dvec;    // [n]  diagonal shift d^-2 for the factorisation; P x result
    double* red;     // [4][NW][4]
    int* flag;       // [4]
};

// ---------------------------------------------------------------------------------------------------------
// 32x32 diagonal block: wavefront 0, lane r = row r (lanes >= 32 idle), right-looking in registers.
// On exit l11[r][c] (c <= r) holds L11; returns false on a non-positive pivot.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool factor_diag_block(double* __restrict__ panel, double* __restrict__ l11, int nv, int lane) {
    double a[NB];
    const int r = lane & 31;
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = panel[r * PLD + c];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (c < nv) {
            const double piv = bcast_lane(a[c], c);
            if (!(piv > 0.0)) ok = false;
            const double ljj = sqrt(piv);
            const double rinv = 1.0 / ljj;
            const double lrc = (r == c) ? ljj : a[c] * rinv;   // column c of L (valid for r >= c)
            a[c] = lrc;
#pragma unroll
            for (int k = c + 1; k < NB; ++k) {
                const double lkc = bcast_lane(lrc, k);
                a[k] -= lrc * lkc;                              // only k <= r is ever used
            }
        }
    }
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < NB; ++c) l11[r * PLD + c] = (c <= r) ? a[c] : 0.0;
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------
// Left-looking blocked Cholesky of S = P + diag(dvec): writes L (lower, row-major, ld = ldl).
// PR = panel rows per pass (multiple of 16, <= NW*MAXT*16).  Returns false on breakdown (uniform).
// ---------------------------------------------------------------------------------------------------------
template <int THREADS, int MAXT>
__device__ bool chol_factor(const double* __restrict__ P, int ldp, double* __restrict__ L, int ldl, int n, int PR,
                            const QpSmem& sm) {
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + NB - 1) / NB;
    for (int jb = 0; jb < nblk; ++jb) {
        const int j0 = jb * NB;
        const int nv = (n - j0) < NB ? (n - j0) : NB;
        const int R = n - j0;
        const int npass = (R + PR - 1) / PR;
        // balanced pass size (multiple of 16)
        int pr = ((R + npass - 1) / npass + 15) & ~15;
        if (pr > PR) pr = PR;
        for (int c0 = 0; c0 < R; c0 += pr) {
            const int cr = (R - c0) < pr ? (R - c0) : pr;      // valid rows in this pass
            const int ntile = (cr + 15) >> 4;
            const int rowbase = j0 + c0;
            // ---- (1) C = P - L[rows,:j0] L[blk,:j0]' on MFMA -------------------------------------------
            v4d acc[MAXT][2];
#pragma unroll
            for (int u = 0; u < MAXT; ++u) { acc[u][0] = (v4d){0, 0, 0, 0}; acc[u][1] = (v4d){0, 0, 0, 0}; }
            const int li = lane & 15, kq = lane >> 4;
            int brow0 = j0 + li;       if (brow0 > n - 1) brow0 = n - 1;
            int brow1 = j0 + 16 + li;  if (brow1 > n - 1) brow1 = n - 1;
            const double* pb0 = L + (size_t)brow0 * ldl + 4 * kq;
            const double* pb1 = L + (size_t)brow1 * ldl + 4 * kq;
            const double* pa[MAXT];
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                int ar = rowbase + (wv + u * NW) * 16 + li;
                if (ar > n - 1) ar = n - 1;
                pa[u] = L + (size_t)ar * ldl + 4 * kq;
            }
            if (wv < ntile) {
                for (int k0 = 0; k0 < j0; k0 += 16) {
                    const double2 b0a = *reinterpret_cast<const double2*>(pb0 + k0);
                    const double2 b0b = *reinterpret_cast<const double2*>(pb0 + k0 + 2);
                    const double2 b1a = *reinterpret_cast<const double2*>(pb1 + k0);
                    const double2 b1b = *reinterpret_cast<const double2*>(pb1 + k0 + 2);
#pragma unroll
                    for (int u = 0; u < MAXT; ++u) {
                        if (wv + u * NW < ntile) {
                            const double2 aa = *reinterpret_cast<const double2*>(pa[u] + k0);
                            const double2 ab = *reinterpret_cast<const double2*>(pa[u] + k0 + 2);
                            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa.x, b0a.x, acc[u][0], 0, 0, 0);
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa.x, b1a.x, acc[u][1], 0, 0, 0);
                            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa.y, b0a.y, acc[u][0], 0, 0, 0);
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa.y, b1a.y, acc[u][1], 0, 0, 0);
                            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab.x, b0b.x, acc[u][0], 0, 0, 0);
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab.x, b1b.x, acc[u][1], 0, 0, 0);
                            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab.y, b0b.y, acc[u][0], 0, 0, 0);
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab.y, b1b.y, acc[u][1], 0, 0, 0);
                        }
                    }
                }
            }
            // C/D map of v_mfma_f64_16x16x4: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                const int t = wv + u * NW;
                if (t < ntile) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int lr = t * 16 + (lane >> 4) + 4 * rg;   // row inside the pass
                            const int row = rowbase + lr;
                            const int cc = ct * 16 + (lane & 15);
                            const int col = j0 + cc;
                            double v = 0.0;
                            if (row < n && col < n) {
                                const int pr_ = row > col ? row : col, pc_ = row > col ? col : row;
                                v = P[(size_t)pr_ * ldp + pc_];
                                if (row == col) v += sm.dvec[row];
                                v -= acc[u][ct][rg];
                            }
                            sm.panel[lr * PLD + cc] = v;
                        }
                }
            }
            __syncthreads();
            // ---- (2) diagonal block ---------------------------------------------------------------------
            if (c0 == 0) {
                if (wv == 0) {
                    const bool ok = factor_diag_block(sm.panel, sm.l11, nv, lane);
                    const unsigned long long bad = __ballot(!ok);
                    if (lane == 0) sm.flag[0] = bad ? 1 : 0;
                }
                __syncthreads();
                if (sm.flag[0]) return false;
            }
            // ---- (3) panel rows: X L11' = C, one thread per row -----------------------------------------
            {
                const int rstart = (c0 == 0) ? NB : 0;
                for (int rr = rstart + tid; rr < cr; rr += THREADS) {
                    double v[NB];
                    double* prow = sm.panel + rr * PLD;
#pragma unroll
                    for (int c = 0; c < NB; ++c) v[c] = prow[c];
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        if (c < nv) {
                            double t = v[c];
#pragma unroll
                            for (int k = 0; k < c; ++k) t -= v[k] * sm.l11[c * PLD + k];
                            v[c] = t / sm.l11[c * PLD + c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NB; ++c) prow[c] = v[c];
                }
                if (c0 == 0) {   // diagonal block rows: copy L11 back into the panel for the coalesced store
                    for (int e = tid; e < NB * NB; e += THREADS) {
                        const int r = e >> 5, c = e & 31;
                        sm.panel[r * PLD + c] = sm.l11[r * PLD + c];
                    }
                }
            }
            __syncthreads();
            // ---- (4) coalesced write-back of the pass (32 columns = 256 B per row) ---------------------
            for (int e = tid; e < cr * NB; e += THREADS) {
                const int r = e >> 5, c = e & 31;
                if (c < nv) L[(size_t)(rowbase + r) * ldl + j0 + c] = sm.panel[r * PLD + c];
            }
            __syncthreads();
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// vec := S^-1 vec  with S = L L'
// ---------------------------------------------------------------------------------------------------------
template <int THREADS>
__device__ void chol_solve(const double* __restrict__ L, int ldl, int n, const QpSmem& sm) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + NB - 1) / NB;
    double* vec = sm.vec;
    // ---- forward: L y = b ---------------------------------------------------------------------------------
    for (int jb = 0; jb < nblk; ++jb) {
        const int j0 = jb * NB;
        const int nv = (n - j0) < NB ? (n - j0) : NB;
        if (wv == 0) {
            const int r = lane & 31;
            const int row = (j0 + r) < n ? (j0 + r) : (n - 1);
            double lr[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) lr[c] = (c <= r && j0 + c < n) ? L[(size_t)row * ldl + j0 + c] : 0.0;
            double bb = (j0 + r < n) ? vec[j0 + r] : 0.0;
            double rdiag = 1.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) if (c == r) rdiag = lr[c];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c < nv) {
                    const double yc = bcast_lane(bb / rdiag, c);
                    if (r > c) bb -= lr[c] * yc;
                    else if (r == c) bb = yc;
                }
            }
            if (lane < nv) vec[j0 + lane] = bb;
        }
        __syncthreads();
        for (int row = j0 + NB + tid; row < n; row += THREADS) {
            const double* lp = L + (size_t)row * ldl + j0;
            double t = vec[row];
#pragma unroll
            for (int c = 0; c < NB; c += 2) {
                const double2 l2 = *reinterpret_cast<const double2*>(lp + c);
                t -= l2.x * vec[j0 + c];
                t -= l2.y * vec[j0 + c + 1];
            }
            vec[row] = t;
        }
        __syncthreads();
    }
    // ---- backward: L' x = y -------------------------------------------------------------------------------
    for (int jb = nblk - 1; jb >= 0; --jb) {
        const int j0 = jb * NB;
        const int nv = (n - j0) < NB ? (n - j0) : NB;
        if (wv == 0) {
            const int c = lane & 31;     // lane = column c of the block: holds L[j0+r][j0+c], r >= c
            double lc[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) lc[r] = (r >= c && j0 + r < n && j0 + c < n) ? L[(size_t)(j0 + r) * ldl + j0 + c] : 0.0;
            double yy = (j0 + c < n) ? vec[j0 + c] : 0.0;
            double cdiag = 1.0;
#pragma unroll
            for (int r = 0; r < NB; ++r) if (r == c) cdiag = lc[r];
            if (j0 + c >= n) cdiag = 1.0;
#pragma unroll
            for (int r = NB - 1; r >= 0; --r) {
                if (r < nv) {
                    const double xr = bcast_lane(yy / cdiag, r);
                    if (c < r) yy -= lc[r] * xr;
                    else if (c == r) yy = xr;
                }
            }
            if (lane < nv) vec[j0 + lane] = yy;
        }
        __syncthreads();
        for (int col = tid; col < j0; col += THREADS) {
            double t = vec[col];
#pragma unroll 8
            for (int r = 0; r < NB; ++r) {
                if (r < nv) t -= L[(size_t)(j0 + r) * ldl + col] * vec[j0 + r];
            }
            vec[col] = t;
        }
        __syncthreads();
    }
}

// out[i] = sum_j P[i][j] * vec[j]   (P symmetric, full storage); one wavefront per row
template <int THREADS>
__device__ void matvec_P(const double* __restrict__ P, int ldp, int n, const double* __restrict__ xin,
                         double* __restrict__ out) {
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < n; i += NW) {
        const double* row = P + (size_t)i * ldp;
        double s = 0.0;
        for (int j = lane; j < n; j += 64) s += row[j] * xin[j];
        s = wsum(s);
        if (lane == 0) out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// the solver
// ---------------------------------------------------------------------------------------------------------
template <int THREADS, int MAXT, int EPT>
__global__ __launch_bounds__(THREADS) void qp_kernel(QpArgs a, int PR) {
    constexpr int NW = THREADS / 64;
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int n = a.n, tid = threadIdx.x;
    const double* P = a.P + (size_t)b * a.p_stride;
    const double* qg = a.q + (size_t)b * n;
    const double* hg = a.h + (size_t)b * a.h_stride;
    double* L = a.L + (size_t)b * a.l_stride;
    const int ldp = a.ldp, ldl = a.ldl;

    extern __shared__ double smem[];
    QpSmem sm;
    sm.panel = smem;
    sm.l11 = sm.panel + (size_t)PR * PLD;
    sm.vec = sm.l11 + NB * PLD;
    sm.dvec = sm.vec + n;
    sm.red = sm.dvec + n;
    sm.flag = reinterpret_cast<int*>(sm.red + 4 * NW * 4);
    Reducer<NW> red(sm.red);

    double x[EPT], z[EPT], s[EPT], d[EPT], di[EPT], lm[EPT], qv[EPT], hv[EPT];
#define FOR_E for (int e = 0, i = tid; e < EPT; ++e, i += THREADS)
#define VALID (i < n)
#pragma unroll
    FOR_E { qv[e] = VALID ? qg[i] : 0.0; hv[e] = VALID ? hg[i] : 0.0; x[e] = z[e] = 0.0; s[e] = lm[e] = 1.0; d[e] = di[e] = 1.0; }

    double nq[2] = {0.0, 0.0};
#pragma unroll
    FOR_E { nq[0] += qv[e] * qv[e]; nq[1] += hv[e] * hv[e]; }
    red.sum(nq);
    const double resx0 = fmax(1.0, sqrt(nq[0]));
    const double resz0 = fmax(1.0, sqrt(nq[1]));

    int status = HIPDRT_QP_MAXITER, iters = 0;
    double pcost = 0.0;
    bool done = false;

    // ---- start point: W = I ---------------------------------------------------------------------------------
#pragma unroll
    FOR_E if (VALID) { sm.dvec[i] = 1.0; sm.vec[i] = -qv[e] - hv[e]; }
    __syncthreads();
    if (!chol_factor<THREADS, MAXT>(P, ldp, L, ldl, n, PR, sm)) {
        status = HIPDRT_QP_SINGULAR;
        done = true;
    } else {
        chol_solve<THREADS>(L, ldl, n, sm);
        double st[2] = {0.0, 0.0}, mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
        FOR_E {
            if (VALID) {
                x[e] = sm.vec[i];
                z[e] = -x[e] - hv[e];
                s[e] = -z[e];
                st[0] += s[e] * s[e]; st[1] += z[e] * z[e];
                mx[0] = fmax(mx[0], -s[e]); mx[1] = fmax(mx[1], -z[e]);
            }
        }
        red.sum(st);
        red.max(mx);
        const double nrms = sqrt(st[0]), nrmz = sqrt(st[1]);
        if (mx[0] >= -1e-8 * fmax(nrms, 1.0)) {
#pragma unroll
            FOR_E s[e] += 1.0 + mx[0];
        }
        if (mx[1] >= -1e-8 * fmax(nrmz, 1.0)) {
#pragma unroll
            FOR_E z[e] += 1.0 + mx[1];
        }
    }
    double gp[1] = {0.0};
#pragma unroll
    FOR_E if (VALID) gp[0] += s[e] * z[e];
    red.sum(gp);
    double gap = gp[0];

    double rx[EPT], rz[EPT];
    while (!done) {
        // ---- residuals, costs, stopping test -------------------------------------------------------------
        __syncthreads();
#pragma unroll
        FOR_E if (VALID) sm.vec[i] = x[e];
        __syncthreads();
        matvec_P<THREADS>(P, ldp, n, sm.vec, sm.dvec);
        __syncthreads();
        double t4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        FOR_E {
            rx[e] = rz[e] = 0.0;
            if (VALID) {
                double r = sm.dvec[i] + qv[e];          // P x + q
                t4[0] += x[e] * r;                      // x'(Px+q)
                t4[1] += x[e] * qv[e];                  // x'q
                r -= z[e];                              // + G'z
                rx[e] = r;
                t4[2] += r * r;
                const double rzz = s[e] - hv[e] - x[e]; // s + Gx - h
                rz[e] = rzz;
                t4[3] += rzz * rzz;
            }
        }
        red.sum(t4);
        double zr[1] = {0.0};
#pragma unroll
        FOR_E if (VALID) zr[0] += z[e] * rz[e];
        red.sum(zr);
        const double f0 = 0.5 * (t4[0] + t4[1]);
        const double resx = sqrt(t4[2]), resz = sqrt(t4[3]);
        pcost = f0;
        const double dcost = f0 + zr[0] - gap;
        bool has_rel = false;
        double relgap = 0.0;
        if (pcost < 0.0) { relgap = gap / -pcost; has_rel = true; }
        else if (dcost > 0.0) { relgap = gap / dcost; has_rel = true; }
        const double pres = resz / resz0, dres = resx / resx0;
        const bool conv = pres <= a.opts.feastol && dres <= a.opts.feastol &&
                          (gap <= a.opts.abstol || (has_rel && relgap <= a.opts.reltol));
        if (conv) { status = HIPDRT_QP_OPTIMAL; break; }
        if (iters == a.opts.maxiters) { status = HIPDRT_QP_MAXITER; break; }

        // ---- scaling --------------------------------------------------------------------------------------
        if (iters == 0) {
#pragma unroll
            FOR_E if (VALID) { d[e] = sqrt(s[e] / z[e]); di[e] = 1.0 / d[e]; lm[e] = sqrt(s[e] * z[e]); }
        }
        __syncthreads();
#pragma unroll
        FOR_E if (VALID) sm.dvec[i] = di[e] * di[e];
        __syncthreads();
        if (!chol_factor<THREADS, MAXT>(P, ldp, L, ldl, n, PR, sm)) {
            status = (iters == 0) ? HIPDRT_QP_SINGULAR : HIPDRT_QP_SINGULAR_LATE;
            break;
        }

        const double mu = gap / (double)n;
        double sigma = 0.0, step = 1.0;
        double dx[EPT], ds[EPT], dz[EPT], ws3[EPT];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            double zz[EPT], sv[EPT];
#pragma unroll
            FOR_E {
                if (VALID) {
                    double t = (pc == 1) ? (-ws3[e] - lm[e] * lm[e]) : (-(lm[e] * lm[e]));
                    t += sigma * mu;
                    sv[e] = t / lm[e];
                    const double bz = -rz[e] - d[e] * sv[e];
                    zz[e] = bz * di[e];
                    sm.vec[i] = -rx[e] - di[e] * zz[e];
                }
            }
            __syncthreads();
            chol_solve<THREADS>(L, ldl, n, sm);
            double dd[1] = {0.0}, mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
            FOR_E {
                if (VALID) {
                    dx[e] = sm.vec[i];
                    dz[e] = -di[e] * dx[e] - zz[e];
                    ds[e] = sv[e] - dz[e];
                    dd[0] += ds[e] * dz[e];
                    if (pc == 0) ws3[e] = ds[e] * dz[e];
                    ds[e] /= lm[e];
                    dz[e] /= lm[e];
                    mx[0] = fmax(mx[0], -ds[e]);
                    mx[1] = fmax(mx[1], -dz[e]);
                }
            }
            red.sum(dd);
            red.max(mx);
            const double t = fmax(0.0, fmax(mx[0], mx[1]));
            if (t == 0.0) step = 1.0;
            else if (pc == 0) step = fmin(1.0, 1.0 / t);
            else step = fmin(1.0, 0.99 / t);
            if (pc == 0) {
                const double sg = fmin(1.0, fmax(0.0, 1.0 - step + dd[0] / gap * (step * step)));
                sigma = sg * sg * sg;
            }
        }
        // ---- update ---------------------------------------------------------------------------------------
        double g2[1] = {0.0};
#pragma unroll
        FOR_E {
            if (VALID) {
                x[e] += step * dx[e];
                const double dss = (1.0 + step * ds[e]) * lm[e];
                const double dzz = (1.0 + step * dz[e]) * lm[e];
                const double sqs = sqrt(dss), sqz = sqrt(dzz);
                d[e] = d[e] * sqs / sqz;
                di[e] = 1.0 / d[e];
                lm[e] = sqs * sqz;
                s[e] = lm[e] * d[e];
                z[e] = lm[e] * di[e];
                g2[0] += lm[e] * lm[e];
            }
        }
        red.sum(g2);
        gap = g2[0];
        ++iters;
    }

#pragma unroll
    FOR_E if (VALID) a.x[(size_t)b * n + i] = x[e];
    if (tid == 0) {
        if (a.iters) a.iters[b] = iters;
        if (a.pcost) a.pcost[b] = pcost;
        a.status[b] = status;
        if (a.iters_accum) a.iters_accum[b] += iters;
    }
#undef FOR_E
#undef VALID
}

size_t qp_scratch_ld(int n) { return (size_t)round_up(n, 16); }

static constexpr int QP_THREADS = 512;
static constexpr int QP_MAXT = 2;

template <int EPT>
static int launch_qp_ept(hipStream_t st, const QpArgs& a, int PR, size_t lds) {
    auto kern = qp_kernel<QP_THREADS, QP_MAXT, EPT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(QP_THREADS), lds, st, a, PR);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("qp launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

int launch_qp(hipStream_t st, const QpArgs& a) {
    const int n = a.n;
    constexpr int NW = QP_THREADS / 64;
    // panel rows: as many as keep two workgroups per CU (<= 80 kB each), at most NW*MAXT*16
    const size_t fixed = ((size_t)NB * PLD + 2 * (size_t)n + 4 * NW * 4) * sizeof(double) + 64;
    int PR = NW * QP_MAXT * 16;
    const size_t budget = 80 * 1024;
    while (PR > 32 && fixed + (size_t)PR * PLD * sizeof(double) > budget) PR -= 16;
    if (PR > round_up(n, 16)) PR = round_up(n, 16);
    if (PR < 32) PR = 32;
    const size_t lds = fixed + (size_t)PR * PLD * sizeof(double);
    if (lds > 160 * 1024) { set_error("qp: problem too large for LDS"); return HIPDRT_E_INVALID; }
    const int ept = (n + QP_THREADS - 1) / QP_THREADS;
    switch (ept) {
        case 1: return launch_qp_ept<1>(st, a, PR, lds);
        case 2: return launch_qp_ept<2>(st, a, PR, lds);
        case 3: return launch_qp_ept<3>(st, a, PR, lds);
        case 4: return launch_qp_ept<4>(st, a, PR, lds);
        default: set_error("qp: n > 2048 not supported"); return HIPDRT_E_INVALID;
    }
}

}  // namespace hipdrt

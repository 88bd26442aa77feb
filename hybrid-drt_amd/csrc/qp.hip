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
#include <mutex>
#include <cstdlib>

#include "qp_common.hpp"
#include "qp_resident.hpp"
#include "qp_super.hpp"

namespace hipdrt {

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
__device__ __forceinline__ bool chol_factor(const double* __restrict__ P, int ldp, double* __restrict__ L, int ldl, int n, int PR,
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
            PROF_DECL
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
            PROF(0);
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
            PROF(1);
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
            PROF(2);
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
            PROF(3);
            // ---- (4) coalesced write-back of the pass (32 columns = 256 B per row) ---------------------
            for (int e = tid; e < cr * NB; e += THREADS) {
                const int r = e >> 5, c = e & 31;
                if (c < nv) L[(size_t)(rowbase + r) * ldl + j0 + c] = sm.panel[r * PLD + c];
            }
            __syncthreads();
            PROF(4);
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// One large problem on G co-resident workgroups (launch_qp forms groups only when B*G <= number of CUs).
// Barrier among the G workgroups of a problem: monotonic counter in global memory; the fences make the stores to L
// visible across CUs and XCDs (release: L2 write-back, acquire: L1 / non-local L2 invalidate).
// ---------------------------------------------------------------------------------------------------------
// The wait is bounded: group mode needs all G workgroups of a problem resident at the same time, which launch_qp
// arranges on a device it has to itself (B*G <= CUs, one workgroup per CU, group launches chained per device); if
// something else keeps a partner off the device for GROUP_WAIT_TICKS the waiting workgroup poisons the counter (every
// later wait of the group then falls through at once) and the problem ends with status HIPDRT_QP_ABORTED.
static constexpr int GROUP_POISON = 1 << 30;
static constexpr unsigned long long GROUP_WAIT_TICKS = 400000000ull;     // 4 s of the 100 MHz s_memrealtime clock
__device__ __forceinline__ void group_barrier(int* ctr, int G, int& epoch) {
    __syncthreads();                       // all stores of this workgroup issued and acknowledged (vmcnt(0))
    if (threadIdx.x == 0) {
        ++epoch;
        __threadfence();
        atomicAdd(ctr, 1);
        const int target = G * epoch;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t0 > GROUP_WAIT_TICKS) { atomicOr(ctr, GROUP_POISON); break; }
        }
        __threadfence();
    }
    __syncthreads();
}

// Left-looking blocked Cholesky as chol_factor, the rows below the diagonal block of every block column dealt out to
// the G workgroups in chunks.  Every workgroup recomputes and factors the 32x32 diagonal block itself (a 2-tile rank-k
// update: cheaper than publishing it and waiting), so one barrier per block column suffices; tiles, the per-row
// substitution and the diagonal factorisation are computed exactly as in chol_factor, i.e. the factor is bit-identical.
template <int THREADS, int MAXT>
__device__ __forceinline__ bool chol_factor_group(const double* __restrict__ P, int ldp, double* __restrict__ L, int ldl,
                                                  int n, int PR, const QpSmem& sm, int G, int g, int* ctr, int& epoch) {
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + NB - 1) / NB;
    const int li = lane & 15, kq = lane >> 4;
    bool ok_all = true;
    for (int jb = 0; jb < nblk; ++jb) {
        const int j0 = jb * NB;
        const int nv = (n - j0) < NB ? (n - j0) : NB;
        // one pass = (optionally) the 32 rows of the diagonal block in panel rows 0..31, followed by `cr` panel rows starting
        // at `chunkbase`: rank-k update on MFMA into the LDS panel -- the two diagonal tiles on wavefronts 0 and 1 while the
        // others already work on the chunk's tiles --, factorisation of the diagonal block, forward substitution of the
        // chunk rows against L11 one thread per row, coalesced write-back
        auto pass = [&](const bool with_diag, const int chunkbase, const int cr, const bool store_diag) -> bool {
            const int off = with_diag ? NB : 0;                       // panel row of the chunk's first row
            const int ntile = (off >> 4) + ((cr + 15) >> 4);
            auto tile_row = [&](int t) { return (with_diag && t < 2) ? j0 + 16 * t : chunkbase + 16 * (t - (off >> 4)); };
            v4d acc[MAXT][2];
#pragma unroll
            for (int u = 0; u < MAXT; ++u) { acc[u][0] = (v4d){0, 0, 0, 0}; acc[u][1] = (v4d){0, 0, 0, 0}; }
            int brow0 = j0 + li;       if (brow0 > n - 1) brow0 = n - 1;
            int brow1 = j0 + 16 + li;  if (brow1 > n - 1) brow1 = n - 1;
            const double* pb0 = L + (size_t)brow0 * ldl + 4 * kq;
            const double* pb1 = L + (size_t)brow1 * ldl + 4 * kq;
            const double* pa[MAXT];
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                int ar = tile_row(wv + u * NW) + li;
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
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                const int t = wv + u * NW;
                if (t < ntile) {
                    const int trow = tile_row(t);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int lr = t * 16 + (lane >> 4) + 4 * rg;
                            const int row = trow + (lane >> 4) + 4 * rg;
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
            if (with_diag) {
                if (wv == 0) {
                    const bool ok = factor_diag_block(sm.panel, sm.l11, nv, lane);
                    const unsigned long long bad = __ballot(!ok);
                    if (lane == 0) sm.flag[0] = bad ? 1 : 0;
                }
                __syncthreads();
                if (sm.flag[0]) return false;
            }
            for (int rr = tid; rr < cr; rr += THREADS) {
                double v[NB];
                double* prow = sm.panel + (off + rr) * PLD;
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
            __syncthreads();
            if (with_diag && store_diag) {
                for (int e = tid; e < nv * NB; e += THREADS) {
                    const int r = e >> 5, c = e & 31;
                    if (c < nv) L[(size_t)(j0 + r) * ldl + j0 + c] = sm.l11[r * PLD + c];
                }
            }
            for (int e = tid; e < cr * NB; e += THREADS) {
                const int r = e >> 5, c = e & 31;
                if (c < nv) L[(size_t)(chunkbase + r) * ldl + j0 + c] = sm.panel[(off + r) * PLD + c];
            }
            __syncthreads();
            return true;
        };
        // This workgroup's first chunk of the rows below the diagonal block shares a pass with the (redundant) diagonal
        // block; further chunks (only when the column is taller than G full panels) follow on their own
        const int rem = n - (j0 + NB);
        int pr = rem > 0 ? (((rem + G - 1) / G) + 15) & ~15 : 16;
        if (pr > PR - NB) pr = PR - NB;
        const int nchunk = rem > 0 ? (rem + pr - 1) / pr : 0;
        {
            const int c0 = g * pr;
            const int cr = g < nchunk ? ((rem - c0) < pr ? (rem - c0) : pr) : 0;
            if (!pass(true, j0 + NB + c0, cr, g == 0)) { ok_all = false; break; }     // same outcome in every workgroup
        }
        for (int c = g + G; c < nchunk; c += G) {
            const int c0 = c * pr;
            pass(false, j0 + NB + c0, (rem - c0) < pr ? (rem - c0) : pr, false);
        }
        group_barrier(ctr, G, epoch);          // column block complete before anyone's next rank-k update reads it
    }
    return ok_all;
}

// ---------------------------------------------------------------------------------------------------------
// vec := S^-1 vec  with S = L L'
// ---------------------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void chol_solve(const double* __restrict__ L, int ldl, int n, const QpSmem& sm) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = (n + NB - 1) / NB;
    double* vec = sm.vec;
    // ---- forward: L y = b ---------------------------------------------------------------------------------
    PROF_DECL
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
            rdiag = 1.0 / rdiag;             // one division per lane, not one per step of the dependent chain
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c < nv) {
                    const double yc = bcast_lane(bb * rdiag, c);
                    if (r > c) bb -= lr[c] * yc;
                    else if (r == c) bb = yc;
                }
            }
            if (lane < nv) vec[j0 + lane] = bb;
        }
        __syncthreads();
        PROF(5);
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
        PROF(6);
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
            cdiag = 1.0 / cdiag;
#pragma unroll
            for (int r = NB - 1; r >= 0; --r) {
                if (r < nv) {
                    const double xr = bcast_lane(yy * cdiag, r);
                    if (c < r) yy -= lc[r] * xr;
                    else if (c == r) yy = xr;
                }
            }
            if (lane < nv) vec[j0 + lane] = yy;
        }
        __syncthreads();
        PROF(7);
        for (int col = tid; col < j0; col += THREADS) {
            double t = vec[col];
#pragma unroll 8
            for (int r = 0; r < NB; ++r) {
                if (r < nv) t -= L[(size_t)(j0 + r) * ldl + col] * vec[j0 + r];
            }
            vec[col] = t;
        }
        __syncthreads();
        PROF(8);
    }
}

// out[i] = sum_j P[i][j] * vec[j]   (P symmetric, full storage); one wavefront per row
template <int THREADS>
__device__ __forceinline__ void matvec_P(const double* __restrict__ P, int ldp, int n, const double* __restrict__ xin,
                         double* __restrict__ out, int r0 = 0, int r1 = -1) {
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (r1 < 0) r1 = n;
    for (int i = r0 + wv; i < r1; i += NW) {
        const double* row = P + (size_t)i * ldp;
        double s = 0.0;
        for (int j = lane; j < n; j += 64) s += row[j] * xin[j];
        s = wsum(s);
        if (lane == 0) out[i] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// multi-pass kernel: any n <= 2048 (panel of PR rows cycled through LDS, L11 copy, strided solves)
// ---------------------------------------------------------------------------------------------------------
template <int THREADS, int MAXT>
struct OpsMultipass {
    const double* P; int ldp; double* L; int ldl; int n; int PR; QpSmem sm;
    static constexpr bool kFusedForward = false;
    __device__ __forceinline__ bool factor() { return chol_factor<THREADS, MAXT>(P, ldp, L, ldl, n, PR, sm); }
    __device__ __forceinline__ void solve() { chol_solve<THREADS>(L, ldl, n, sm); }
    __device__ __forceinline__ void backward() {}
    __device__ __forceinline__ void matvec() { matvec_P<THREADS>(P, ldp, n, sm.vec, sm.dvec); }
};

template <int THREADS, int MAXT>
struct OpsGroup {
    const double* P; int ldp; double* L; int ldl; int n; int PR; QpSmem sm;
    int G, g; int* ctr; int epoch; double* gvec;
    static constexpr bool kFusedForward = false;
    __device__ __forceinline__ bool factor() {
        // every partner must be through with the previous factor (its triangular sweeps read all of L) before anybody
        // overwrites L: the iterations no longer meet in a P x product (qp_common.hpp: P x recurrence)
        group_barrier(ctr, G, epoch);
        return chol_factor_group<THREADS, MAXT>(P, ldp, L, ldl, n, PR, sm, G, g, ctr, epoch);
    }
    __device__ __forceinline__ void solve() { chol_solve<THREADS>(L, ldl, n, sm); }
    __device__ __forceinline__ void backward() {}
    // P x: every workgroup does its share of the rows (same per-row arithmetic as matvec_P, so the same bits), the
    // pieces meet in a global vector
    __device__ __forceinline__ void matvec() {
        const int per = (n + G - 1) / G;
        const int r0 = g * per, r1 = (r0 + per) < n ? (r0 + per) : n;
        matvec_P<THREADS>(P, ldp, n, sm.vec, gvec, r0, r1);
        group_barrier(ctr, G, epoch);
        for (int i = threadIdx.x; i < n; i += THREADS) sm.dvec[i] = gvec[i];
    }
};

// G workgroups per problem: the factorisation is shared, everything else (O(n^2) sweeps, O(n) vector work, all
// decisions) runs redundantly and identically in each, so no further communication is needed
template <int THREADS, int MAXT, int EPT>
__global__ __launch_bounds__(THREADS) void qp_kernel_group(QpArgs a, int PR, int G) {
    constexpr int NW = THREADS / 64;
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = a.order ? a.order[slot] : slot;
    if (a.active && !a.active[b]) return;
    const int n = a.n;
    extern __shared__ double smem[];
    OpsGroup<THREADS, MAXT> ops;
    ops.P = a.P + (size_t)b * a.p_stride; ops.ldp = a.ldp;
    ops.L = a.L + (size_t)b * a.l_stride; ops.ldl = a.ldl; ops.n = n; ops.PR = PR;
    ops.G = G; ops.g = g; ops.ctr = a.gsync + slot; ops.epoch = 0; ops.gvec = a.gvec + (size_t)slot * a.state_ld;
    ops.sm.panel = smem;
    ops.sm.l11 = ops.sm.panel + (size_t)PR * PLD;
    ops.sm.vec = ops.sm.l11 + NB * PLD;
    ops.sm.dvec = ops.sm.vec + n;
    ops.sm.red = ops.sm.dvec + n;
    ops.sm.flag = reinterpret_cast<int*>(ops.sm.red + 4 * NW * 4);
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    QpArgs ag = a;
    ag.state = a.gstate;
    ipm_solve<THREADS, EPT>(ag, b, ops, is, (int)blockIdx.x, g == 0);
    if (g == 0 && threadIdx.x == 0 &&
        (__hip_atomic_load(ops.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & GROUP_POISON))
        a.status[b] = HIPDRT_QP_ABORTED;
}

template <int THREADS, int MAXT, int EPT>
__global__ __launch_bounds__(THREADS) void qp_kernel(QpArgs a, int PR) {
    constexpr int NW = THREADS / 64;
    const int b = a.order ? a.order[blockIdx.x] : blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int n = a.n;
    extern __shared__ double smem[];
    OpsMultipass<THREADS, MAXT> ops;
    ops.P = a.P + (size_t)b * a.p_stride; ops.ldp = a.ldp;
    ops.L = a.L + (size_t)b * a.l_stride; ops.ldl = a.ldl; ops.n = n; ops.PR = PR;
    ops.sm.panel = smem;
    ops.sm.l11 = ops.sm.panel + (size_t)PR * PLD;
    ops.sm.vec = ops.sm.l11 + NB * PLD;
    ops.sm.dvec = ops.sm.vec + n;
    ops.sm.red = ops.sm.dvec + n;
    ops.sm.flag = reinterpret_cast<int*>(ops.sm.red + 4 * NW * 4);
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    ipm_solve<THREADS, EPT>(a, b, ops, is);
}

size_t qp_scratch_ld(int n) { return (size_t)round_up(n, 16); }

// doubles of factor scratch per problem (covers both the row-major multipass and the tile-packed resident layout)
size_t qp_scratch_doubles(int n) {
    size_t d = (size_t)n * qp_scratch_ld(n);
    if (n <= RNP_MAX) { const size_t r = resident_l_doubles(n); if (r > d) d = r; const size_t s_ = super_l_doubles(n); if (s_ > d) d = s_; }
    return d;
}

int qp_profile_read(unsigned long long* out, int n, int reset) {
#ifdef HIPDRT_QP_PROFILE
    unsigned long long h[QP_PROF_SLOTS];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qp_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n; ++i) out[i] = i < QP_PROF_SLOTS ? h[i] : 0;
    if (reset) { unsigned long long z[QP_PROF_SLOTS] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qp_prof), z, sizeof(z)); }
    return 1;
#else
    for (int i = 0; i < n; ++i) out[i] = 0;
    return 0;
#endif
}

static constexpr int QP_THREADS = 512;
static constexpr int QP_MAXT = 2;

template <int EPT>
static int launch_qp_group_ept(hipStream_t st, const QpArgs& a, int PR, size_t lds, int G) {
    auto kern = qp_kernel_group<QP_THREADS, QP_MAXT, EPT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp group): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    e = hipMemsetAsync(a.gsync, 0, (size_t)a.B * sizeof(int), st);
    if (e != hipSuccess) { set_error(std::string("qp group sync reset: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    // The workgroups of a group spin on a barrier, so a group launch must become fully resident.  Two group launches from
    // different streams could each grab part of the CUs and wait for the rest forever; they are therefore chained through
    // an event per device (stream-ordered: the host never blocks, other kernels still overlap freely).
    static std::mutex mtx;
    static hipEvent_t last[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    hipEvent_t& ev = last[dev & 63];
    if (ev) {
        e = hipStreamWaitEvent(st, ev, 0);
        if (e != hipSuccess) { set_error(std::string("qp group chain: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    } else {
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) { ev = nullptr; set_error(std::string("qp group event: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    }
    hipLaunchKernelGGL(kern, dim3(a.B * G), dim3(QP_THREADS), lds, st, a, PR, G);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("qp group launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    e = hipEventRecord(ev, st);
    if (e != hipSuccess) { set_error(std::string("qp group record: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

// CUs of the CURRENT device (cached per device ordinal; a group launch needs one CU per workgroup: 512 threads at
// ~170 VGPRs and 80 kB of LDS leave no room for a second one)
static int device_cus() {
    static std::mutex mtx;
    static int cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mtx);
    int& v = cache[dev & 63];
    if (v <= 0 && (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)) v = 256;
    return v;
}

// Workgroups per problem for the multi-pass kernel: up to 16 when the launch would otherwise leave most CUs idle and the
// problem is big enough for the split to pay (chunks of at least 64 rows below the first diagonal block)
int qp_group_size(int B, int n) {
    if (qp_packed_only(n) || getenv("HIPDRT_QP_NOGROUP")) return 1;
    int G = device_cus() / (B > 0 ? B : 1);
    int gmax = 16;
    if (const char* e = getenv("HIPDRT_QP_GROUP")) { const int v = atoi(e); if (v >= 1 && v <= 32) gmax = v; }   // tuning knob
    if (G > gmax) G = gmax;
    if (G > (n - NB) / 64) G = (n - NB) / 64;          // keep at least 64 panel rows per workgroup in the first column
    if (B * G > qp_group_slots()) G = qp_group_slots() / B;
    return G >= 2 ? G : 1;
}

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

// Longest-processing-time-first dispatch: the iteration count of a spectrum's previous QP predicts the next one's,
// and workgroups are dispatched in blockIdx order, so starting the long problems first trims the tail of the launch
// (4 problems per CU with 2..9 iterations each otherwise leave a third of the CUs idle at the end).  Rank by brute
// force (B^2 comparisons, B ~ 1e3); ties keep index order, so the permutation is deterministic.
__global__ __launch_bounds__(256) void lpt_order_kernel(int B, const int* __restrict__ iters,
                                                        const int* __restrict__ active, int* __restrict__ order) {
    extern __shared__ int keys[];
    for (int i = threadIdx.x; i < B; i += blockDim.x) keys[i] = (active && !active[i]) ? -1 : iters[i];
    __syncthreads();
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const int kb = keys[b];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int kj = keys[j];
            rank += (kj > kb) || (kj == kb && j < b);
        }
        order[rank] = b;
    }
}

void launch_lpt_order(hipStream_t st, int B, const int* iters, const int* active, int* order) {
    const int blocks = (B + 255) / 256;
    hipLaunchKernelGGL(lpt_order_kernel, dim3(blocks), dim3(256), (size_t)B * sizeof(int), st, B, iters, active, order);
}

// which kernel serves n <= 528: the 32-column one (default) or the experimental 64-wide super-column kernel
// (HIPDRT_QP_KERNEL=super; parity-green but slower so far: its diagonal-block chain is the critical path)
static bool use_super() {
    static const bool v = [] { const char* e = getenv("HIPDRT_QP_KERNEL"); return e && std::string(e) == "super"; }();
    return v;
}

static int launch_qp_super(hipStream_t st, const QpArgs& a) {
    const int NP64 = round_up(a.n, 64);
    if (!a.Ppk) { set_error("qp super: packed copy of P missing"); return HIPDRT_E_INVALID; }
    const size_t lds = super_lds_bytes(a.n);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qp_kernel_super),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp super): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    hipLaunchKernelGGL(qp_kernel_super, dim3(a.B), dim3(ST), lds, st, a, NP64);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("qp launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

static int launch_qp_resident(hipStream_t st, const QpArgs& a) {
    const int NP = round_up(a.n, 32);
    if (!a.Ppk) { set_error("qp resident: packed copy of P missing"); return HIPDRT_E_INVALID; }
    const size_t lds = resident_lds_bytes(NP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qp_kernel_resident),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp resident): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    hipLaunchKernelGGL(qp_kernel_resident, dim3(a.B), dim3(RT), lds, st, a, NP);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("qp launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

int launch_dist_var(hipStream_t st, int B, int n, const double* Ppk, long long ppk_stride, const double* Bex, int nex,
                    double* L, long long l_stride, double* out, long long out_stride, int* status) {
    if (n > RNP_MAX) { set_error("posterior variance: only built for n <= 528 unknowns"); return HIPDRT_E_INVALID; }
    if (use_super()) {
        const size_t lds = super_lds_bytes(n);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_kernel_super),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(cov): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
        CovArgs a;
        a.B = B; a.n = n; a.Ppk = Ppk; a.ppk_stride = ppk_stride; a.nchp = qp_nchp(n); a.Bex = Bex; a.nex = nex;
        a.L = L; a.l_stride = l_stride; a.out = out; a.out_stride = out_stride; a.status = status;
        hipLaunchKernelGGL(cov_kernel_super, dim3(B), dim3(ST), lds, st, a, round_up(n, 64));
        e = hipGetLastError();
        if (e != hipSuccess) { set_error(std::string("cov launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
        return HIPDRT_OK;
    }
    const int NP = round_up(n, 32);
    const size_t lds = resident_lds_bytes(NP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_kernel_resident),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(cov): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    CovArgs a;
    a.B = B; a.n = n; a.Ppk = Ppk; a.ppk_stride = ppk_stride; a.nchp = qp_nchp(n); a.Bex = Bex; a.nex = nex;
    a.L = L; a.l_stride = l_stride; a.out = out; a.out_stride = out_stride; a.status = status;
    hipLaunchKernelGGL(cov_kernel_resident, dim3(B), dim3(RT), lds, st, a, NP);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("cov launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

// doubles of factor scratch per spectrum for the posterior-variance kernel: (nch + nex) x nch tiles
size_t dist_var_scratch_doubles(int n, int nex) {
    const size_t nch = (size_t)round_up(n, 64) / 16;      // the super-column layout (the wider of the two)
    return (nch + (size_t)nex) * nch * TSZ;
}

bool qp_packed_only(int n) { return n <= RNP_MAX && !getenv("HIPDRT_QP_MULTIPASS"); }

int launch_qp(hipStream_t st, const QpArgs& a) {
    const int n = a.n;
    if (qp_packed_only(n)) return use_super() ? launch_qp_super(st, a) : launch_qp_resident(st, a);
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
    const int G = (a.gstate && a.gsync && a.gvec) ? qp_group_size(a.B, n) : 1;
    if (G > 1) {
        switch (ept) {
            case 2: return launch_qp_group_ept<2>(st, a, PR, lds, G);
            case 3: return launch_qp_group_ept<3>(st, a, PR, lds, G);
            case 4: return launch_qp_group_ept<4>(st, a, PR, lds, G);
            default: break;
        }
    }
    switch (ept) {
        case 1: return launch_qp_ept<1>(st, a, PR, lds);
        case 2: return launch_qp_ept<2>(st, a, PR, lds);
        case 3: return launch_qp_ept<3>(st, a, PR, lds);
        case 4: return launch_qp_ept<4>(st, a, PR, lds);
        default: set_error("qp: n > 2048 not supported"); return HIPDRT_E_INVALID;
    }
}

}  // namespace hipdrt

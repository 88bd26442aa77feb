// Linear-algebra services of the coneqp kernel for n <= 528 unknowns (C1-C4 sizes: n = 93 ... 514), built around
// memory-level parallelism: one 1024-thread workgroup (16 wavefronts) per problem, one per CU.
//
//  * LDS array U[NP][33] (NP = n rounded up to 32): while block column j of the left-looking Cholesky is
//    processed it is the panel for rows >= 32 j; rows of already finished block columns keep their 32x32 diagonal
//    block L_jj there.  After the factorisation all diagonal blocks are LDS resident, so the triangular solves
//    never fetch them from HBM, and their reciprocal diagonals sit in the pad column U[i][32].
//  * Block column j in ONE pass: every wavefront owns up to 3 row tiles (16 rows x 32 cols, two MFMA
//    accumulators each) initialised with -(P + diag) and accumulating +L L' over the finished columns with
//    v_mfma_f64_16x16x4_f64; operands come straight from L in HBM/L2 as 128-byte row segments; 4 wavefronts
//    per SIMD hide the load latency.
//  * Diagonal block: wavefront 0, lane = row, rows in registers, column broadcast through a 32-double LDS
//    buffer (one ds_write + b128 broadcast reads per step) instead of v_readlane chains.
//  * Panel rows (X L11' = C): thread per row, right-looking substitution (independent FMAs per step, reciprocal
//    pivots) against the LDS-resident L11.
//  * Solves: per 32-block a register/LDS substitution by wavefront 0, then the rank-32 update with fully
//    coalesced reads of L: forward = 4 rows per wave instruction + 16-lane shuffle reduction, backward = thread
//    per column with 32 independent loads.
//  * P x: two rows x 5 column chunks of 16-byte loads in flight per lane.
#pragma once
#include "qp_common.hpp"

namespace hipdrt {

static constexpr int RT = 1024;          // threads
static constexpr int RNW = RT / 64;      // 16 wavefronts
static constexpr int RMAXT = 2;          // row tiles per wavefront for block columns j >= 1 (<= 31 tiles)
static constexpr int RNP_MAX = 528;

struct ResSmem {
    double* U;       // [NP][PLD]
    double* vec;     // [NP]
    double* dvec;    // [NP]
    double* colbuf;  // [64]
    double* red;     // [4][RNW][4]
    int* flag;       // [4]
};

// reduction over the 16 lanes of a DPP row (lanes sharing lane>>4)
__device__ __forceinline__ double row16_sum(double v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

struct OpsResident {
    const double* P; int ldp; double* L; int ldl; int n; ResSmem sm;

    // -----------------------------------------------------------------------------------------------------
    __device__ __forceinline__ bool factor() {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int nblk = (n + NB - 1) / NB;
        double* U = sm.U;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int nv = (n - j0) < NB ? (n - j0) : NB;
            const int R = n - j0;
            const int ntile = (R + 15) >> 4;
            PROF_DECL
            // ---- (1) tiles: acc = -(P + diag) + L[rows,:j0] L[blk,:j0]' -----------------------------------
            if (j0 == 0) {
                // first block column: nothing to subtract, the panel is P + diag itself
                for (int e = tid; e < n * NB; e += RT) {
                    const int r = e >> 5, c = e & 31;
                    double v = 0.0;
                    if (c < n) {
                        v = (r >= c) ? P[(size_t)r * ldp + c] : P[(size_t)c * ldp + r];
                        if (r == c) v += sm.dvec[r];
                    }
                    U[r * PLD + c] = v;
                }
                PROF(0);
            } else {
                v4d acc[RMAXT][2];
                const int li = lane & 15, kq = lane >> 4;
#pragma unroll
                for (int u = 0; u < RMAXT; ++u) {
                    const int t = wv + u * RNW;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            double v = 0.0;
                            if (t < ntile) {
                                const int row = j0 + t * 16 + kq + 4 * rg;
                                const int col = j0 + ct * 16 + li;
                                if (row < n && col < n) {
                                    const int pr_ = row > col ? row : col, pc_ = row > col ? col : row;
                                    v = -P[(size_t)pr_ * ldp + pc_];
                                    if (row == col) v -= sm.dvec[row];
                                }
                            }
                            acc[u][ct][rg] = v;
                        }
                }
                if (wv < ntile) {
                    int brow0 = j0 + li;       if (brow0 > n - 1) brow0 = n - 1;
                    int brow1 = j0 + 16 + li;  if (brow1 > n - 1) brow1 = n - 1;
                    const double* pb0 = L + (size_t)brow0 * ldl + 4 * kq;
                    const double* pb1 = L + (size_t)brow1 * ldl + 4 * kq;
                    const double* pa[RMAXT];
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        int ar = j0 + (wv + u * RNW) * 16 + li;
                        if (ar > n - 1) ar = n - 1;
                        pa[u] = L + (size_t)ar * ldl + 4 * kq;
                    }
                    // ping-pong prefetch: the next 16-deep operand slab is in flight while the current one is
                    // multiplied (j0 is a multiple of 32, so slabs come in pairs)
                    struct Slab { double2 b0a, b0b, b1a, b1b, aa[RMAXT], ab[RMAXT]; };
                    auto load = [&](Slab& s_, int k0) {
                        s_.b0a = *reinterpret_cast<const double2*>(pb0 + k0);
                        s_.b0b = *reinterpret_cast<const double2*>(pb0 + k0 + 2);
                        s_.b1a = *reinterpret_cast<const double2*>(pb1 + k0);
                        s_.b1b = *reinterpret_cast<const double2*>(pb1 + k0 + 2);
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) {
                            s_.aa[u] = *reinterpret_cast<const double2*>(pa[u] + k0);
                            s_.ab[u] = *reinterpret_cast<const double2*>(pa[u] + k0 + 2);
                        }
                    };
                    auto mult = [&](const Slab& s_) {
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) {
                            if (wv + u * RNW < ntile) {
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
                    load(sa, 0);
                    for (int k0 = 0; k0 < j0; k0 += 32) {
                        load(sb, k0 + 16);
                        mult(sa);
                        if (k0 + 32 < j0) load(sa, k0 + 32);
                        mult(sb);
                    }
                }
                PROF(0);
                // C/D map of v_mfma_f64_16x16x4: col = lane&15, row = (lane>>4) + 4*reg
#pragma unroll
                for (int u = 0; u < RMAXT; ++u) {
                    const int t = wv + u * RNW;
                    if (t < ntile) {
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg)
                                U[(j0 + t * 16 + kq + 4 * rg) * PLD + ct * 16 + li] = -acc[u][ct][rg];
                    }
                }
            }
            __syncthreads();
            PROF(1);
            // ---- (2) diagonal block, in place in U rows j0..j0+31 -------------------------------------------
            if (wv == 0) {
                const int r = lane & 31;
                double* Ub = U + (size_t)j0 * PLD;
                double a[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) a[c] = Ub[r * PLD + c];
                bool ok = true;
                double* cb = sm.colbuf;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    if (c < nv) {
                        double* col = cb + (c & 1) * 32;
                        if (lane < 32) col[r] = a[c];                         // column c before scaling
                        __builtin_amdgcn_wave_barrier();
                        double cv[NB];
#pragma unroll
                        for (int k = c; k < NB; ++k) cv[k] = col[k];           // broadcast reads, issued together
                        const double piv = cv[c];
                        if (!(piv > 0.0)) ok = false;
                        const double rinv = rsqrt(piv);                        // 1 / L_cc
                        const double ljj = piv * rinv;                         // L_cc
                        const double lrc = (r == c) ? ljj : a[c] * rinv;       // L_rc
                        const double lrs = lrc * rinv;                         // L_rc / L_cc
                        a[c] = lrc;
                        if (lane == c) U[(size_t)(j0 + c) * PLD + NB] = rinv;  // reciprocal pivot lives in the pad column
#pragma unroll
                        for (int k = c + 1; k < NB; ++k) a[k] -= lrs * cv[k];  // a_rk -= L_rc * L_kc
                    }
                }
                if (lane < 32) {
#pragma unroll
                    for (int c = 0; c < NB; ++c) Ub[r * PLD + c] = (c <= r) ? a[c] : 0.0;
                }
                const unsigned long long bad = __ballot(!ok);
                if (lane == 0) sm.flag[0] = bad ? 1 : 0;
            }
            __syncthreads();
            if (sm.flag[0]) return false;
            PROF(2);
            // ---- (3) panel rows: X L11' = C, thread per row, right-looking ------------------------------------
            {
                const int rr = j0 + NB + tid;
                if (rr < n) {
                    double v[NB];
                    double* prow = U + (size_t)rr * PLD;
                    const double* Ub = U + (size_t)j0 * PLD;
#pragma unroll
                    for (int c = 0; c < NB; ++c) v[c] = prow[c];
#pragma unroll
                    for (int c = 0; c < NB; ++c) {
                        if (c < nv) {
                            const double xc = v[c] * U[(size_t)(j0 + c) * PLD + NB];
                            v[c] = xc;
#pragma unroll
                            for (int k = c + 1; k < NB; ++k) v[k] -= xc * Ub[k * PLD + c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NB; ++c) prow[c] = v[c];
                }
            }
            __syncthreads();
            PROF(3);
            // ---- (4) coalesced write-back: 16 x 16-byte pieces per row ---------------------------------------
            for (int e = tid; e < R * 16; e += RT) {
                const int r = e >> 4, p2 = (e & 15) * 2;
                const double* src = U + (size_t)(j0 + r) * PLD + p2;
                double* dst = L + (size_t)(j0 + r) * ldl + j0 + p2;
                if (p2 + 1 < nv) *reinterpret_cast<double2*>(dst) = make_double2(src[0], src[1]);
                else if (p2 < nv) dst[0] = src[0];
            }
            __syncthreads();
            PROF(4);
        }
        return true;
    }

    // -----------------------------------------------------------------------------------------------------
    // vec := S^-1 vec.  Wavefront 0 solves the 32x32 diagonal systems (L_jj and 1/L_ii are LDS resident);
    // wavefronts 1..15 apply the rank-32 updates.  The update operands (rows/columns of L in HBM) do not
    // depend on the running solution, so they are fetched BEFORE the diagonal solve of the same block and
    // are in flight while wavefront 0 substitutes.
    __device__ __forceinline__ void solve() {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int nblk = (n + NB - 1) / NB;
        double* vec = sm.vec;
        const double* U = sm.U;
        constexpr int UW = RNW - 1;                 // updater wavefronts
        PROF_DECL
        // ---- forward: L y = b -----------------------------------------------------------------------------
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int nv = (n - j0) < NB ? (n - j0) : NB;
            const int rows_below = n - (j0 + NB);
            const int li = lane & 15, rq = lane >> 4;
            constexpr int FG = 9;                     // 4-row groups per updater wavefront: 15*9*4 = 540 rows
            double2 lv[FG];
            if (wv > 0 && rows_below > 0) {
#pragma unroll
                for (int u = 0; u < FG; ++u) {
                    int row = j0 + NB + ((wv - 1) + u * UW) * 4 + rq;
                    if (row > n - 1) row = n - 1;
                    lv[u] = *reinterpret_cast<const double2*>(L + (size_t)row * ldl + j0 + 2 * li);
                }
            }
            if (wv == 0) {
                const int r = lane & 31;
                const double* Ub = U + (size_t)(j0 + r) * PLD;
                double lr[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) lr[c] = Ub[c];
                const bool rv_ = (j0 + r) < n;
                double bb = rv_ ? vec[j0 + r] : 0.0;
                const double rinv = rv_ ? Ub[NB] : 0.0;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    if (c < nv) {
                        const double yc = bcast_lane(bb * rinv, c);
                        if (r > c) bb -= lr[c] * yc;
                        else if (r == c) bb = yc;
                    }
                }
                if (lane < nv) vec[j0 + lane] = bb;
            }
            __syncthreads();
            PROF(5);
            if (wv > 0 && rows_below > 0) {
                const double y0 = vec[j0 + 2 * li], y1 = vec[j0 + 2 * li + 1];
#pragma unroll
                for (int u = 0; u < FG; ++u) {
                    const int row = j0 + NB + ((wv - 1) + u * UW) * 4 + rq;
                    double pv = lv[u].x * y0 + lv[u].y * y1;
                    pv = row16_sum(pv);
                    if (li == 0 && row < n) vec[row] -= pv;
                }
            }
            __syncthreads();
            PROF(6);
        }
        // ---- backward: L' x = y ---------------------------------------------------------------------------
        for (int jb = nblk - 1; jb >= 0; --jb) {
            const int j0 = jb * NB;
            const int nv = (n - j0) < NB ? (n - j0) : NB;
            const int col = tid - 64;                 // updater thread -> column
            double lvb[NB];
            if (wv > 0 && col < j0) {
                const double* lp = L + (size_t)j0 * ldl + col;
#pragma unroll
                for (int r = 0; r < NB; ++r) lvb[r] = (r < nv) ? lp[(size_t)r * ldl] : 0.0;
            }
            if (wv == 0) {
                const int c = lane & 31;     // lane = column c of the block: needs L[j0+r][j0+c], r >= c
                double lc[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r) lc[r] = U[(size_t)(j0 + r) * PLD + c];
                const bool cv_ = (j0 + c) < n;
                double yy = cv_ ? vec[j0 + c] : 0.0;
                const double rinv = cv_ ? U[(size_t)(j0 + c) * PLD + NB] : 0.0;
#pragma unroll
                for (int r = NB - 1; r >= 0; --r) {
                    if (r < nv) {
                        const double xr = bcast_lane(yy * rinv, r);
                        if (c < r) yy -= lc[r] * xr;
                        else if (c == r) yy = xr;
                    }
                }
                if (lane < nv) vec[j0 + lane] = yy;
            }
            __syncthreads();
            PROF(7);
            if (wv > 0 && col < j0) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int r = 0; r < NB; r += 2) {
                    t0 += lvb[r] * vec[j0 + r];          // vec beyond n is zero padding
                    t1 += lvb[r + 1] * vec[j0 + r + 1];
                }
                vec[col] -= (t0 + t1);
            }
            __syncthreads();
            PROF(8);
        }
    }

    // -----------------------------------------------------------------------------------------------------
    // dvec = P * vec ; rows in pairs per wavefront, 16-byte loads, up to 5 column chunks of 128
    __device__ __forceinline__ void matvec() {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const double* xin = sm.vec;
        double* out = sm.dvec;
        const int c0 = 2 * lane;
        double2 xv[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int c = c0 + 128 * t;
            xv[t] = make_double2(c < n ? xin[c] : 0.0, c + 1 < n ? xin[c + 1] : 0.0);
        }
        for (int i0 = 2 * wv; i0 < n; i0 += 2 * RNW) {
            const int i1 = (i0 + 1 < n) ? i0 + 1 : i0;
            const double* r0 = P + (size_t)i0 * ldp;
            const double* r1 = P + (size_t)i1 * ldp;
            double2 a0[5], a1[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                int c = c0 + 128 * t;
                if (c + 1 >= ldp) c = 0;                       // clamp (value is multiplied by xv = 0)
                a0[t] = *reinterpret_cast<const double2*>(r0 + c);
                a1[t] = *reinterpret_cast<const double2*>(r1 + c);
            }
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                s0 += a0[t].x * xv[t].x + a0[t].y * xv[t].y;
                s1 += a1[t].x * xv[t].x + a1[t].y * xv[t].y;
            }
            s0 = wsum(s0);
            s1 = wsum(s1);
            if (lane == 0) { out[i0] = s0; if (i0 + 1 < n) out[i0 + 1] = s1; }
        }
    }
};

__global__ __launch_bounds__(RT) void qp_kernel_resident(QpArgs a, int NP) {
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    extern __shared__ double smem[];
    OpsResident ops;
    ops.P = a.P + (size_t)b * a.p_stride; ops.ldp = a.ldp;
    ops.L = a.L + (size_t)b * a.l_stride; ops.ldl = a.ldl; ops.n = a.n;
    ops.sm.U = smem;
    ops.sm.vec = ops.sm.U + (size_t)NP * PLD;
    ops.sm.dvec = ops.sm.vec + NP + 32;
    ops.sm.colbuf = ops.sm.dvec + NP + 32;
    ops.sm.red = ops.sm.colbuf + 64;
    ops.sm.flag = reinterpret_cast<int*>(ops.sm.red + 4 * RNW * 4);
    // zero U (rows >= n are read, never used) and the padding of vec (read by the backward update of the
    // last, partial block)
    for (int i = threadIdx.x; i < NP * PLD; i += RT) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < NP + 32; i += RT) ops.sm.vec[i] = 0.0;
    __syncthreads();
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    ipm_solve<RT, 1>(a, b, ops, is);
}

static size_t resident_lds_bytes(int NP) {
    return ((size_t)NP * PLD + 2 * (size_t)(NP + 32) + 64 + 4 * RNW * 4) * sizeof(double) + 64;
}

}  // namespace hipdrt

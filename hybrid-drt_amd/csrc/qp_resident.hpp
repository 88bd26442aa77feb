// Linear-algebra services of the coneqp kernel for n <= 528 unknowns (C1-C4 sizes: n = 93 ... 514), built around
// memory-level parallelism: one 1024-thread workgroup (16 wavefronts) per problem, one per CU.
//
//  * The Cholesky factor L lives in HBM in a TILE-PACKED layout: 16x16 tiles, each one contiguous 2 KB block
//    stored as [k-half][row][8 doubles] so that a wavefront's MFMA fragment load covers one contiguous 1 KB,
//    tiles of one tile-row adjacent ([tile_row][k_chunk][2][16][8]).  An MFMA operand fragment (16 rows x 16 k),
//    a block-column panel of a tile-row (2 adjacent tiles) and a block-row of finished columns are all
//    contiguous byte ranges, so every phase reads and writes long coalesced runs (1 KB per wave instruction)
//    instead of 128-byte row segments 4 KB apart.
//  * LDS array U[NP][33] (NP = n rounded up to 32): while block column j of the left-looking Cholesky is
//    processed it is the panel for rows >= 32 j; rows of already finished block columns keep their 32x32 diagonal
//    block L_jj there, with 1/L_ii in the pad column U[i][32].  After the factorisation all diagonal blocks are
//    LDS resident, so the triangular solves never fetch them from HBM.
//  * Block column j in ONE pass: every wavefront owns up to 2 row tiles (16 rows x 32 cols, two MFMA
//    accumulators each) initialised with -(P + diag) and accumulating +L L' over the finished columns with
//    v_mfma_f64_16x16x4_f64; operand slabs are double buffered in registers (the next 16-deep slab is in flight
//    while the current one is multiplied).
//  * Diagonal block: wavefront 0, lane = row, rows in registers, column broadcast through a 32-double LDS
//    buffer, reciprocal pivots from rsqrt.
//  * Panel rows (X L11' = C): thread per row, right-looking substitution against the LDS-resident L11.
//  * Solves: per 32-block a substitution by wavefront 0; wavefronts 1..15 apply the rank-32 updates from
//    operands that were fetched (contiguous tiles) before the diagonal solve started.
//  * P x: two rows x 5 column chunks of 16-byte loads in flight per lane.
#pragma once
#include "qp_common.hpp"

namespace hipdrt {

static constexpr int RT = 512;           // threads
static constexpr int RNW = RT / 64;      // 16 wavefronts
static constexpr int RMAXT = 32 / RNW;   // row tiles per wavefront for block columns j >= 1 (<= 31 tiles)
static constexpr int RNP_MAX = 528;
static constexpr int TSZ = 256;          // doubles per 16x16 tile

struct ResSmem {
    double* U;       // [NP][PLD]
    double* vec;     // [NP + 32]
    double* dvec;    // [NP + 32]
    double* colbuf;  // [64]
    double* red;     // [4][RNW][4]
    int* flag;       // [4]
};

struct OpsResident {
    const double* P; int ldp; double* L; int nch; int n; ResSmem sm;   // nch = tiles per tile-row (NP/16)
    const double* Ppk; int nchp;                                       // optional accumulator-native copy of P

    // tile (t, c) starts at ((t*nch + c) * TSZ) doubles; returned in double2 units
    __device__ __forceinline__ const double2* tile2(int t, int c) const {
        return reinterpret_cast<const double2*>(L) + (size_t)((t * nch + c) * (TSZ / 2));
    }

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
            const int tb = j0 >> 4;                 // first tile-row of the block
            PROF_DECL
            // ---- (1) tiles: acc = -(P + diag) + L[rows,:j0] L[blk,:j0]' -----------------------------------
            // panel <- P + diag with coalesced 16-byte loads (16 lanes per 256-byte row segment, 4 rows = 1 KB per
            // wave instruction; the accumulator-shaped 8-byte gather used before cost ~45 % of the GEMM phase).
            // Inside the diagonal block the upper part is taken from the lower triangle (only P's lower triangle is
            // ever read there).
            const bool packed = (Ppk != nullptr) && j0 > 0;
            if (!packed)
            for (int e = tid; e < ntile * 256; e += RT) {
                const int rr = e >> 4, pc = (e & 15) * 2;
                const int r = j0 + rr, cc = j0 + pc;
                double2 v = make_double2(0.0, 0.0);
                if (r < n) {
                    if (cc + 1 <= r || rr >= NB) {
                        if (cc + 1 < n) v = *reinterpret_cast<const double2*>(P + (size_t)r * ldp + cc);
                        else if (cc < n) v.x = P[(size_t)r * ldp + cc];
                    } else {
                        if (cc < n) v.x = (r >= cc) ? P[(size_t)r * ldp + cc] : P[(size_t)cc * ldp + r];
                        if (cc + 1 < n) v.y = (r >= cc + 1) ? P[(size_t)r * ldp + cc + 1] : P[(size_t)(cc + 1) * ldp + r];
                    }
                    if (r == cc) v.x += sm.dvec[r];
                    if (r == cc + 1) v.y += sm.dvec[r];
                }
                U[r * PLD + pc] = v.x;
                U[r * PLD + pc + 1] = v.y;
            }
            if (!packed) __syncthreads();
            if (j0 == 0) {
                PROF(0);
            } else {
                v4d acc[RMAXT][2];
                const int li = lane & 15, kq = lane >> 4;
                if (packed) {
                    // accumulators straight from the accumulator-native copy of P the Gram kernel wrote: two
                    // contiguous 1 KB loads per 16x16 tile, no LDS hop, no extra barrier
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        const int t = wv + u * RNW;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) {
                            v4d a_ = (v4d){0, 0, 0, 0};
                            const int T = tb + t, Cc = 2 * jb + ct;
                            if (t < ntile && T >= Cc) {
                                const double2* tile = reinterpret_cast<const double2*>(Ppk + ((size_t)T * nchp + Cc) * 256);
                                const double2 d0 = tile[lane], d1 = tile[64 + lane];
                                a_ = (v4d){-d0.x, -d0.y, -d1.x, -d1.y};
                                if (T == Cc) {
#pragma unroll
                                    for (int rg = 0; rg < 4; ++rg)
                                        if (kq + 4 * rg == li) a_[rg] -= sm.dvec[T * 16 + li];
                                }
                            }
                            acc[u][ct] = a_;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        const int t = wv + u * RNW;
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg)
                                acc[u][ct][rg] = (t < ntile) ? -U[(j0 + t * 16 + kq + 4 * rg) * PLD + ct * 16 + li] : 0.0;
                    }
                }
                if (wv < ntile) {
                    // Tile-internal layout [k-half h][row i][8 doubles]: lane (i = lane&15, kq = lane>>4) takes the
                    // double2 at h*64 + i*4 + kq, i.e. k = 8h + 2kq, +1 -- each wave instruction covers one
                    // contiguous 1 KB (full 128-byte lines; the half-line pattern of a row-major tile halves the
                    // per-CU load throughput).  A and B use the same k assignment, so the MFMA sums are complete.
                    const int fo = li * 4 + kq;
                    const double2* pb0 = tile2(tb, 0) + fo;
                    const double2* pb1 = tile2(tb + 1 < nch ? tb + 1 : tb, 0) + fo;
                    const double2* pa[RMAXT];
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        int t = tb + wv + u * RNW;
                        if (t > nch - 1) t = nch - 1;
                        pa[u] = tile2(t, 0) + fo;
                    }
                    struct Slab { double2 b0a, b0b, b1a, b1b, aa[RMAXT], ab[RMAXT]; };
                    auto load = [&](Slab& s_, int c) {          // c = k-chunk index (16 columns)
                        const int o = c * (TSZ / 2);
                        s_.b0a = pb0[o]; s_.b0b = pb0[o + 64];
                        s_.b1a = pb1[o]; s_.b1b = pb1[o + 64];
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) { s_.aa[u] = pa[u][o]; s_.ab[u] = pa[u][o + 64]; }
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
                    // ping-pong prefetch over the 2*jb finished 16-column chunks (always an even count)
                    Slab sa, sb;
                    const int nc = 2 * jb;
                    load(sa, 0);
                    for (int c = 0; c < nc; c += 2) {
                        load(sb, c + 1);
                        mult(sa);
                        if (c + 2 < nc) load(sa, c + 2);
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
            // ---- (2)+(3) 32-wide panel in two 16-wide halves (halves the sequential diagonal work and the
            //      substitution work per panel row; the rank-16 coupling between the halves runs on MFMA) -------
            // half h: wavefront 0 factors the 16x16 diagonal block D_h in place (lane = row, column broadcast via
            // LDS, rsqrt pivots) -- for h = 0 the lanes 16..31 carry the rows of the lower-left block L21 along;
            // then every panel row (thread per row) is substituted against D_h.
            bool failed = false;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int c0 = 16 * h;                               // first column of the half
                const int nvh = nv - c0 < 16 ? nv - c0 : 16;          // valid columns in this half (may be <= 0)
                if (nvh <= 0) break;
                if (wv == 0) {
                    const int r = lane & 31;                          // row j0 + c0 + r  (h = 1: only r < 16 matter)
                    double* Ub = U + (size_t)(j0 + c0) * PLD + c0;    // D_h origin
                    double a[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) a[c] = Ub[r * PLD + c];
                    bool ok = true;
                    double* cb = sm.colbuf;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        if (c < nvh) {
                            double* col = cb + (c & 1) * 32;
                            if (lane < 32) col[r] = a[c];
                            __builtin_amdgcn_wave_barrier();
                            const double piv = col[c];
                            if (!(piv > 0.0)) ok = false;
                            const double rinv = rsqrt(piv);            // 1 / L_cc
                            const double ljj = piv * rinv;             // L_cc
                            const double lrc = (r == c) ? ljj : a[c] * rinv;
                            const double lrs = lrc * rinv;
                            a[c] = lrc;
                            if (lane == c) U[(size_t)(j0 + c0 + c) * PLD + NB] = rinv;   // reciprocal pivot -> pad column
#pragma unroll
                            for (int k = c + 1; k < 16; ++k) a[k] -= lrs * col[k];
                        }
                    }
                    const int rows_here = (h == 0) ? 32 : 16;
                    if (lane < rows_here) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) Ub[r * PLD + c] = (c <= r) ? a[c] : 0.0;
                    } else if (h == 1 && lane < 32) {
                        // upper-right 16x16 of the diagonal block: zero (never referenced, kept clean)
#pragma unroll
                        for (int c = 0; c < 16; ++c) U[(size_t)(j0 + (lane - 16)) * PLD + 16 + c] = 0.0;
                    }
                    const unsigned long long bad = __ballot(!ok);
                    if (lane == 0) sm.flag[0] = bad ? 1 : 0;
                }
                __syncthreads();
                if (sm.flag[0]) { failed = true; break; }
                if (h == 0) PROF(2);
                // panel rows below the diagonal block: x = v D_h^-T, right-looking, reciprocal pivots
                {
                    for (int rr = j0 + NB + tid; rr < n; rr += RT) {
                        double v[16];
                        double* prow = U + (size_t)rr * PLD + c0;
                        const double* Ub = U + (size_t)(j0 + c0) * PLD + c0;
#pragma unroll
                        for (int c = 0; c < 16; ++c) v[c] = prow[c];
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            if (c < nvh) {
                                const double xc = v[c] * U[(size_t)(j0 + c0 + c) * PLD + NB];
                                v[c] = xc;
#pragma unroll
                                for (int k = c + 1; k < 16; ++k) v[k] -= xc * Ub[k * PLD + c];
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 16; ++c) prow[c] = v[c];
                    }
                }
                __syncthreads();
                if (h == 0 && nv > 16) {
                    // rank-16 coupling on MFMA: C[:, 16:32] -= X[:, 0:16] * L21' for every row >= j0+16
                    // (tile 0 = the lower half of the diagonal block itself); operands straight from the LDS panel
                    const int li = lane & 15, kq = lane >> 4;
                    const int nt3 = (n - (j0 + 16) + 15) >> 4;
                    const double* Bp = U + (size_t)(j0 + 16 + li) * PLD + 4 * kq;
                    const double b0 = Bp[0], b1 = Bp[1], b2 = Bp[2], b3 = Bp[3];
                    for (int t = wv; t < nt3; t += RNW) {
                        const int row0 = j0 + 16 + 16 * t;
                        const double* Ap = U + (size_t)(row0 + li) * PLD + 4 * kq;
                        double* Cp = U + (size_t)(row0 + kq) * PLD + 16 + li;
                        v4d acc;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) acc[rg] = Cp[(size_t)(4 * rg) * PLD];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[0], b0, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[1], b1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[2], b2, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[3], b3, acc, 0, 0, 0);
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) Cp[(size_t)(4 * rg) * PLD] = acc[rg];
                    }
                    __syncthreads();
                }
            }
            if (failed) return false;
            PROF(3);
            // ---- (4) write-back: per tile-row 2 adjacent tiles = 4 KB contiguous ---------------------------
            // e -> (tile t, chunk ch, half h, row i, 16-byte piece p): consecutive threads write consecutive 16 B
            for (int e = tid; e < ntile * 256; e += RT) {
                const int pc = e & 3, i = (e >> 2) & 15, h = (e >> 6) & 1, ch = (e >> 7) & 1, t = e >> 8;
                const double* src = U + (size_t)(j0 + t * 16 + i) * PLD + ch * 16 + 8 * h + 2 * pc;
                double2* dst = reinterpret_cast<double2*>(L) +
                               (size_t)(((tb + t) * nch + 2 * jb + ch) * (TSZ / 2) + h * 64 + i * 4 + pc);
                *dst = make_double2(src[0], src[1]);
            }
            __syncthreads();
            PROF(4);
        }
        return true;
    }

    // -----------------------------------------------------------------------------------------------------
    // vec := S^-1 vec.  Wavefront 0 solves the 32x32 diagonal systems (L_jj and 1/L_ii are LDS resident);
    // wavefronts 1..15 apply the rank-32 updates.  The update operands (tiles of L in HBM) do not depend on the
    // running solution, so they are fetched BEFORE the diagonal solve of the same block and are in flight while
    // wavefront 0 substitutes.
    //   tile load map: instruction h (k-half) of a 2 KB tile covers double2 index h*64 + lane  ->  row i = lane/4,
    //   columns 8h + 2*(lane%4), +1.
    __device__ __forceinline__ void solve() {
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        const int nblk = (n + NB - 1) / NB;
        double* vec = sm.vec;
        const double* U = sm.U;
        constexpr int UW = RNW - 1;                 // updater wavefronts
        const int l4 = lane & 3, g4 = lane >> 2;
        PROF_DECL
        // ---- forward: L y = b -----------------------------------------------------------------------------
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int nv = (n - j0) < NB ? (n - j0) : NB;
            const int tb = j0 >> 4;
            const int tbelow = ((n + 15) >> 4) - (tb + 2);     // row tiles below the block
            constexpr int FT = (31 + UW - 1) / UW;              // tiles per updater wavefront
            double2 lv[FT][4];                                  // [tile][chunk*2 + half]
            if (wv > 0 && tbelow > 0) {
#pragma unroll
                for (int u = 0; u < FT; ++u) {
                    const int tt = (wv - 1) + u * UW;
                    if (tt < tbelow) {
                        const double2* p = tile2(tb + 2 + tt, 2 * jb) + lane;   // chunks 2jb, 2jb+1 are adjacent
#pragma unroll
                        for (int q = 0; q < 4; ++q) lv[u][q] = p[q * 64];
                    }
                }
            }
            if (wv == 0) {
                const int r = lane & 31;
                const double* Ub = U + (size_t)(j0 + r) * PLD;
                // minimal dependency chain per step: mul, broadcast, fma.  Entries with c >= r are zeroed at load
                // time, so bb stops changing after step r-1 and y_r = bb * rinv falls out at the end.
                double lr[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) lr[c] = (c < r) ? Ub[c] : 0.0;
                const bool rv_ = (j0 + r) < n;
                double bb = rv_ ? vec[j0 + r] : 0.0;
                const double rinv = rv_ ? Ub[NB] : 0.0;
#pragma unroll
                for (int c = 0; c < NB - 1; ++c) {
                    const double yc = bcast_lane(bb * rinv, c);
                    bb -= lr[c] * yc;
                }
                if (lane < nv) vec[j0 + lane] = bb * rinv;
            }
            __syncthreads();
            PROF(5);
            if (wv > 0 && tbelow > 0) {
                // q = 2*chunk + half: lane holds columns 16*chunk + 8*half + 2*l4, +1 of tile row g4
                double ya[4], yb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ya[q] = vec[j0 + 8 * q + 2 * l4]; yb[q] = vec[j0 + 8 * q + 2 * l4 + 1]; }
#pragma unroll
                for (int u = 0; u < FT; ++u) {
                    const int tt = (wv - 1) + u * UW;
                    if (tt < tbelow) {
                        double pv = 0.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) pv += lv[u][q].x * ya[q] + lv[u][q].y * yb[q];
                        pv += __shfl_xor(pv, 1, 64);
                        pv += __shfl_xor(pv, 2, 64);
                        if (l4 == 0) {
                            const int row = (tb + 2 + tt) * 16 + g4;
                            if (row < n) vec[row] -= pv;
                        }
                    }
                }
            }
            __syncthreads();
            PROF(6);
        }
        // ---- backward: L' x = y ---------------------------------------------------------------------------
        for (int jb = nblk - 1; jb >= 0; --jb) {
            const int j0 = jb * NB;
            const int nv = (n - j0) < NB ? (n - j0) : NB;
            const int tb = j0 >> 4;
            const int nc = 2 * jb;                              // 16-column chunks left of the block
            const bool two = (tb + 1) * 16 < n;                 // second tile-row of the block holds valid rows
            constexpr int BC = (32 + UW - 1) / UW;              // chunks per updater wavefront
            double2 lb[BC][4];                                  // [chunk][tile*2 + half]
            if (wv > 0) {
#pragma unroll
                for (int u = 0; u < BC; ++u) {
                    const int c = (wv - 1) + u * UW;
                    if (c < nc) {
                        const double2* p0 = tile2(tb, c) + lane;
                        const double2* p1 = tile2(two ? tb + 1 : tb, c) + lane;
                        lb[u][0] = p0[0]; lb[u][1] = p0[64];
                        lb[u][2] = p1[0]; lb[u][3] = p1[64];
                    }
                }
            }
            if (wv == 0) {
                const int c = lane & 31;     // lane = column c of the block: needs L[j0+r][j0+c], r >= c
                double lc[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r) lc[r] = (r > c && r < nv) ? U[(size_t)(j0 + r) * PLD + c] : 0.0;
                const bool cv_ = (j0 + c) < n;
                double yy = cv_ ? vec[j0 + c] : 0.0;
                const double rinv = cv_ ? U[(size_t)(j0 + c) * PLD + NB] : 0.0;
#pragma unroll
                for (int r = NB - 1; r > 0; --r) {
                    const double xr = bcast_lane(yy * rinv, r);    // lanes >= nv carry yy = 0, rinv = 0
                    yy -= lc[r] * xr;
                }
                if (lane < nv) vec[j0 + lane] = yy * rinv;
            }
            __syncthreads();
            PROF(7);
            if (wv > 0 && nc > 0) {
                // x of the block: row g4 of tile tb and of tile tb+1 (zero padding beyond n)
                const double x0 = vec[j0 + g4];
                const double x1 = two ? vec[j0 + 16 + g4] : 0.0;
#pragma unroll
                for (int u = 0; u < BC; ++u) {
                    const int c = (wv - 1) + u * UW;
                    if (c < nc) {
                        // lb[u][0..1] = tile tb halves 0,1 ; lb[u][2..3] = tile tb+1 halves 0,1
                        double s0 = lb[u][0].x * x0 + lb[u][2].x * x1;     // column 16c + 2*l4
                        double s1 = lb[u][0].y * x0 + lb[u][2].y * x1;     // column 16c + 2*l4 + 1
                        double s2 = lb[u][1].x * x0 + lb[u][3].x * x1;     // column 16c + 8 + 2*l4
                        double s3 = lb[u][1].y * x0 + lb[u][3].y * x1;     // column 16c + 8 + 2*l4 + 1
#pragma unroll
                        for (int off = 4; off < 64; off <<= 1) {
                            s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64);
                            s2 += __shfl_xor(s2, off, 64); s3 += __shfl_xor(s3, off, 64);
                        }
                        if (g4 == 0) {
                            vec[c * 16 + 2 * l4] -= s0;     vec[c * 16 + 2 * l4 + 1] -= s1;
                            vec[c * 16 + 8 + 2 * l4] -= s2; vec[c * 16 + 8 + 2 * l4 + 1] -= s3;
                        }
                    }
                }
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
    ops.L = a.L + (size_t)b * a.l_stride; ops.nch = NP / 16; ops.n = a.n;
    ops.Ppk = a.Ppk ? a.Ppk + (size_t)b * a.ppk_stride : nullptr; ops.nchp = a.nchp;
    ops.sm.U = smem;
    ops.sm.vec = ops.sm.U + (size_t)NP * PLD;
    ops.sm.dvec = ops.sm.vec + NP + 32;
    ops.sm.colbuf = ops.sm.dvec + NP + 32;
    ops.sm.red = ops.sm.colbuf + 64;
    ops.sm.flag = reinterpret_cast<int*>(ops.sm.red + 4 * RNW * 4);
    // zero U (rows >= n are read, never used) and the padding of vec (read by the updates of the last,
    // partial block)
    for (int i = threadIdx.x; i < NP * PLD; i += RT) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < NP + 32; i += RT) ops.sm.vec[i] = 0.0;
    __syncthreads();
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    ipm_solve<RT, (RNP_MAX + RT - 1) / RT>(a, b, ops, is);
}

static size_t resident_lds_bytes(int NP) {
    return ((size_t)NP * PLD + 2 * (size_t)(NP + 32) + 64 + 4 * RNW * 4) * sizeof(double) + 64;
}

// scratch doubles per problem for the tile-packed factor: (NP/16)^2 tiles of 256 doubles
static size_t resident_l_doubles(int n) {
    const size_t nt = (size_t)round_up(n, 32) / 16;
    return nt * nt * TSZ;
}

}  // namespace hipdrt

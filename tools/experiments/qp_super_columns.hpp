// Linear-algebra services of the coneqp kernel for n <= 528 unknowns (C1-C4 sizes: n = 93 ... 514): one
// 512-thread workgroup (8 wavefronts) per problem, one per CU.  Left-looking blocked Cholesky with 64-wide
// SUPER-COLUMNS: a pass over the finished part of L updates four 16-column tile columns at once, so the factor is
// re-read from HBM half as often as with 32-wide block columns (n^3 / (6 * 64) * 8 bytes per factorisation).
//
//  * L lives in HBM in the TILE-PACKED layout: 16x16 tiles of 2 KB, tiles of one tile row adjacent
//    ([tile_row][tile_col][256], nch = NP64 / 16 tile columns).  Inside a tile the double2 with index
//    h*64 + i*4 + q (h = k-half, i = row, q = 0..3) holds columns q + 8h and q + 8h + 4 of row i: with lane
//    (i = lane & 15, q = lane >> 4) a wavefront's MFMA operand load is one contiguous 1 KB run, and the same
//    double2 pair is the register image in which a tile is produced (accumulator layout: lane (i, q) register rg
//    <-> row i, column q + 4 rg), so tiles are stored without a shuffle.  P arrives in the same tile layout (Ppk).
//  * Roles.  Wavefront 0 ("chain"): Cholesky + inverse of the 64x64 diagonal block of the super-column, 16x16
//    tile by tile (register-only Gauss-Jordan, lane = row), everything sequential about the factorisation.
//    Wavefront 1 ("look-ahead"): the complete rank-k update of the NEXT 64x64 diagonal block (ten lower tiles), so
//    that the chain of super-column p+1 runs beside everybody else's rank-k update of super-column p+1.
//    Wavefronts 2..7 ("rows"): all tile rows below the diagonal block, dealt round robin, up to 4 tile rows x 4
//    tile columns of accumulators each per pass (v_mfma_f64_16x16x4_f64, operand half-chunks ping-pong prefetched).
//  * Panel solve on the matrix pipe from registers, right-looking over the four tile columns:
//        X_c = C_c W_c' ;  C_c2 -= X_c L(c2, c)'  (c2 > c)        (W_c = inverse of the c-th diagonal Cholesky tile)
//    The four tile rows next to the diagonal block go first and are published (LDS epoch flags) so that the
//    look-ahead wavefront can finish the next diagonal block while the others still solve their rows.
//  * LDS: U = inverses of the 32x32 diagonal blocks for the triangular sweeps, stored as the three 16x16 tiles
//    [[Wa, 0], [Wba, Wb]] (111 kB for n = 514); Lt = the six off-diagonal tiles of the current 64x64 block;
//    In = register images of the next diagonal block.
//  * The predictor's forward substitution is fused into the factorisation (a super-column's tiles update the
//    right-hand side while they sit in registers).
#pragma once
#include "qp_common.hpp"
#include "qp_resident.hpp"   // CovArgs

namespace hipdrt {

static constexpr int ST = 512;            // threads
static constexpr int SNW = ST / 64;       // 8 wavefronts
static constexpr int SRW = SNW - 2;       // row wavefronts
static constexpr int SMAXT = 4;           // tile rows per row wavefront and pass
static constexpr int SNP_MAX = 528;
static constexpr int STSZ = 256;          // doubles per 16x16 tile
static constexpr int SLD = 17;            // row stride of 16x16 LDS tiles
static constexpr int STL = 16 * SLD;      // doubles per padded LDS tile (272)
static constexpr int SUB = 3 * STL;       // doubles per inverse 32x32 block (Wa | Wba | Wb)

struct SupSmem {
    double* red;     // [4][SNW][4]
    double* dsc;     // [16][SLD]      diagonal tile being factored
    int* flag;       // [0] failure, [1..4] epochs of the published next-diagonal rows
    double* Lt;      // [6][STL]       L10, L20, L21, L30, L31, L32 of the current 64x64 diagonal block
    v4d* In;         // [10][64]       images of the next diagonal block, tile (i, j) at i(i+1)/2 + j
    double* vec;     // [VEC]
    double* dvec;    // [VEC]
    double* U;       // [nblk32][SUB]

    static constexpr int VEC = 576 + 32;
    static constexpr int FIXED = 4 * SNW * 4 + STL + 8 + 6 * STL + 10 * 64 * 4 + 2 * VEC;   // doubles before U
    __device__ __forceinline__ void carve(double* smem) {
        red = smem;
        dsc = red + 4 * SNW * 4;
        flag = reinterpret_cast<int*>(dsc + STL);
        Lt = dsc + STL + 8;
        In = reinterpret_cast<v4d*>(Lt + 6 * STL);
        vec = Lt + 6 * STL + 10 * 64 * 4;
        dvec = vec + VEC;
        U = dvec + VEC;
    }
};
static_assert((4 * SNW * 4 + STL + 8 + 6 * STL) % 4 == 0, "In must be 32-byte aligned");

struct OpsSuper {
    double* L; int nch; int n; SupSmem sm;                             // nch = tile columns of L (multiple of 4)
    const double* Ppk; int nchp;                                       // P in L's tile layout (lower tiles)
    // Optional extra tile rows appended below the square matrix (tile rows nch .. nch+nex-1 of L, source tiles
    // Bex[nex][nchp][256]): they ride through the factorisation as more panel rows and come out as Bex * L^-T.
    int nex = 0; const double* Bex = nullptr;
    static constexpr bool kFusedForward = true;
    bool fwd = true;

    __device__ __forceinline__ const double2* tile2(int t, int c) const {
        return reinterpret_cast<const double2*>(L) + (size_t)((t * nch + c) * (STSZ / 2));
    }
    static __device__ __forceinline__ int fresh_lane() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    }
    static __device__ __forceinline__ int dtile(int i, int j) { return i * (i + 1) / 2 + j; }        // In slot
    static __device__ __forceinline__ int ltile(int i, int j) { return i * (i - 1) / 2 + j; }        // Lt slot, i > j
    // inverse diagonal tile c (0..3) of super-column p inside U
    __device__ __forceinline__ double* wtile(int p, int c) const {
        return sm.U + (size_t)(2 * p + (c >> 1)) * SUB + ((c & 1) ? 2 * STL : 0);
    }

    // ---- wavefront 0: Cholesky factor of a 16x16 tile and its inverse ------------------------------------------
    // at = -D' in accumulator layout -> D row-major in the LDS scratch tile
    __device__ __forceinline__ void stage_dsc(const v4d& at) const {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sm.dsc[li * SLD + kq + 4 * rg] = -at[rg];
    }
    // W = inverse of the Cholesky factor of the tile in dsc -> Wd[16][SLD].  Register-only Gauss-Jordan on [D | I]:
    // lane r holds row r of D and of W; pivot, multipliers and the finished row travel by v_readlane -- no LDS round
    // trip and no barrier on the dependency chain (pivot -> rsqrt -> multiplier -> next pivot).
    __device__ __forceinline__ bool cholinv16(const v4d& at, double* Wd) const {
        stage_dsc(at);
        __builtin_amdgcn_wave_barrier();
        const int lane = fresh_lane(), r = lane & 15;
        const double* D = sm.dsc;
        double a[16], w[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { a[c] = D[r * SLD + c]; w[c] = (c == r) ? 1.0 : 0.0; }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double piv = bcast_lane(a[c], c);
            if (!(piv > 0.0)) ok = false;
            const double rinv = rsqrt(piv);              // 1 / L_cc
            const double lrc = a[c] * rinv;              // L_rc for r >= c
            const double lm = (r > c) ? lrc : 0.0;
            const double sc = (r == c) ? rinv : 1.0;
#pragma unroll
            for (int j = 0; j <= c; ++j) {
                const double wcj = bcast_lane(w[j], c) * rinv;
                w[j] = w[j] * sc - lm * wcj;
                asm volatile("" : "+v"(w[j]));           // pinned: otherwise the optimiser sinks the updates and spills
            }
#pragma unroll
            for (int k = c + 1; k < 16; ++k) {
                const double lkc = bcast_lane(lrc, k);
                a[k] -= lrc * lkc;
                asm volatile("" : "+v"(a[k]));
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) Wd[r * SLD + j] = (j <= r) ? w[j] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        return ok;
    }

    // accumulator image of -(S tile (T, Cc)): lane (li, kq) register rg <-> row li, column kq + 4 rg
    __device__ __forceinline__ v4d init_tile(int T, int Cc, int ntr, int fo, int li, int kq) const {
        v4d a_ = (v4d){0, 0, 0, 0};
        if (Cc < ntr) {
            if (T >= nch) {
                const double2* tile = reinterpret_cast<const double2*>(Bex + ((size_t)(T - nch) * nchp + Cc) * 256);
                const double2 d0 = tile[fo], d1 = tile[64 + fo];
                a_ = (v4d){-d0.x, -d0.y, -d1.x, -d1.y};
            } else if (T < ntr) {
                const double2* tile = reinterpret_cast<const double2*>(Ppk + ((size_t)T * nchp + Cc) * 256);
                const double2 d0 = tile[fo], d1 = tile[64 + fo];
                a_ = (v4d){-d0.x, -d0.y, -d1.x, -d1.y};
            }
        }
        if (T == Cc) {
            const int row = T * 16 + li;
            const double dg = row < n ? sm.dvec[row] : 1.0;      // diagonal shift; identity beyond n
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                if (kq + 4 * rg == li) a_[rg] -= dg;
        }
        return a_;
    }

    // x = W C' on the matrix pipe: image of the solved tile from the image c_ of -(C) and the LDS tile Wd
    __device__ __forceinline__ v4d trsm_tile(const double* Wd, const v4d& c_, int li, int kq) const {
        v4d x = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[li * SLD + 4 * s_ + kq], c_[s_], x, 0, 0, 0);
        return x;
    }
    // e += X Lr'  with Lr an LDS tile (row operand) and X a register image
    __device__ __forceinline__ void syrk_tile(v4d& e, const double* Lr, const v4d& x, int li, int kq) const {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(Lr[li * SLD + 4 * s_ + kq], x[s_], e, 0, 0, 0);
    }
    __device__ __forceinline__ void put_lt(int slot, const v4d& x, int li, int kq) const {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sm.Lt[slot * STL + li * SLD + kq + 4 * rg] = x[rg];
    }
    __device__ __forceinline__ void store_tile(int T, int c, const v4d& x, int fo) const {
        double2* d0 = const_cast<double2*>(tile2(T, c)) + fo;
        d0[0] = make_double2(x[0], x[1]);
        d0[64] = make_double2(x[2], x[3]);
    }
    // y_r = row r (0..31) of the inverse 32x32 block Ub times v[0..31]
    static __device__ __forceinline__ double inv_block_row(const double* Ub, const double* v, int r) {
        const int rr = r & 15;
        const bool hi = r >= 16;
        const double* A_ = Ub + (hi ? STL : 0) + rr * SLD;        // Wa or Wba: columns 0..15
        const double* B_ = Ub + 2 * STL + rr * SLD;               // Wb: columns 16..31 (rows 16..31 only)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int c = 0; c < 16; c += 4) {
            s0 += A_[c] * v[c]; s1 += A_[c + 1] * v[c + 1]; s2 += A_[c + 2] * v[c + 2]; s3 += A_[c + 3] * v[c + 3];
        }
        if (hi) {
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                s0 += B_[c] * v[16 + c]; s1 += B_[c + 1] * v[17 + c]; s2 += B_[c + 2] * v[18 + c]; s3 += B_[c + 3] * v[19 + c];
            }
        }
        return (s0 + s1) + (s2 + s3);
    }
    // x_c = column c (0..31) of the inverse 32x32 block times y[0..31]  (transposed product)
    static __device__ __forceinline__ double inv_block_col(const double* Ub, const double* y, int c) {
        const int cc = c & 15;
        const bool hi = c >= 16;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (!hi) {
            const double* A_ = Ub + cc;                 // Wa[r][cc]
            const double* B_ = Ub + STL + cc;           // Wba[r][cc]
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                s0 += A_[r * SLD] * y[r] + B_[r * SLD] * y[16 + r];
                s1 += A_[(r + 1) * SLD] * y[r + 1] + B_[(r + 1) * SLD] * y[17 + r];
                s2 += A_[(r + 2) * SLD] * y[r + 2] + B_[(r + 2) * SLD] * y[18 + r];
                s3 += A_[(r + 3) * SLD] * y[r + 3] + B_[(r + 3) * SLD] * y[19 + r];
            }
        } else {
            const double* C_ = Ub + 2 * STL + cc;       // Wb[r][cc]
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                s0 += C_[r * SLD] * y[16 + r]; s1 += C_[(r + 1) * SLD] * y[17 + r];
                s2 += C_[(r + 2) * SLD] * y[18 + r]; s3 += C_[(r + 3) * SLD] * y[19 + r];
            }
        }
        return (s0 + s1) + (s2 + s3);
    }

    // ---- wavefront 0: the 64x64 diagonal block of super-column p (ncol = 1..4 valid tile columns) ----------------
    __device__ __forceinline__ bool chain(int p, int ncol, int lane) {
        const int li = lane & 15, kq = lane >> 4, fo = li * 4 + kq;
        const int tb = 4 * p;
        bool ok = true;
        PROF_DECL
        // One ROLLED loop over the tile columns, every tile of the block living in LDS (In: register images, Lt: the
        // solved tiles row-major as MFMA row operands): the 16x16 Gauss-Jordan is 12 kB of straight-line code, and four
        // copies of it plus unrolled tile arithmetic made wavefront 0 instruction-fetch bound (88 k cycles per block).
#pragma unroll 1
        for (int c = 0; c < ncol; ++c) {
            double* Wc = wtile(p, c);
            ok = cholinv16(sm.In[dtile(c, c) * 64 + lane], Wc) && ok;
            PROFW(44);
            // tiles below the diagonal tile: X_ic = E_ic Wc'
#pragma unroll 1
            for (int i = c + 1; i < ncol; ++i) {
                const v4d x = trsm_tile(Wc, sm.In[dtile(i, c) * 64 + lane], li, kq);
                put_lt(ltile(i, c), x, li, kq);
                sm.In[dtile(i, c) * 64 + lane] = x;
            }
            __builtin_amdgcn_wave_barrier();
            // E_ij += X_ic X_jc'  for c < j <= i
#pragma unroll 1
            for (int i = c + 1; i < ncol; ++i) {
                const v4d x = sm.In[dtile(i, c) * 64 + lane];
#pragma unroll 1
                for (int j = c + 1; j <= i; ++j) {
                    v4d e = sm.In[dtile(i, j) * 64 + lane];
                    syrk_tile(e, sm.Lt + ltile(j, c) * STL, x, li, kq);
                    sm.In[dtile(i, j) * 64 + lane] = e;
                }
            }
            __builtin_amdgcn_wave_barrier();
            PROFW(45);
        }
        if (ncol > 2) {
            // the tiles that couple the two 32x32 blocks of the super-column go to HBM for the triangular sweeps
            store_tile(tb + 2, tb, sm.In[dtile(2, 0) * 64 + lane], fo);
            store_tile(tb + 2, tb + 1, sm.In[dtile(2, 1) * 64 + lane], fo);
            if (ncol > 3) {
                store_tile(tb + 3, tb, sm.In[dtile(3, 0) * 64 + lane], fo);
                store_tile(tb + 3, tb + 1, sm.In[dtile(3, 1) * 64 + lane], fo);
            }
        }
        // lower-left tiles of the inverse 32x32 blocks: Wba = -Wb (L_ba Wa)
        PROFW(45);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if (2 * hb + 1 < ncol) {
                const double* Lba = sm.Lt + (hb == 0 ? ltile(1, 0) : ltile(3, 2)) * STL;
                const double* Wa = wtile(p, 2 * hb);
                const double* Wb = wtile(p, 2 * hb + 1);
                v4d y = (v4d){0, 0, 0, 0};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    y = __builtin_amdgcn_mfma_f64_16x16x4f64(Lba[li * SLD + 4 * s_ + kq], Wa[(4 * s_ + kq) * SLD + li], y, 0, 0, 0);
                v4d w21 = (v4d){0, 0, 0, 0};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    w21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wb[li * SLD + 4 * s_ + kq], y[s_], w21, 0, 0, 0);
                double* Wba = sm.U + (size_t)(2 * p + hb) * SUB + STL;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) Wba[(kq + 4 * rg) * SLD + li] = w21[rg];
            } else if (2 * hb < ncol) {
                // a lone diagonal tile: its partner rows are padding -> zero coupling, unit... never read (no rows there)
                double* Wba = sm.U + (size_t)(2 * p + hb) * SUB + STL;
                double* Wb = Wba + STL;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) { Wba[(kq + 4 * rg) * SLD + li] = 0.0; Wb[(kq + 4 * rg) * SLD + li] = 0.0; }
            }
        }
        PROFW(46);
        if (fwd) {
            // fused forward substitution of the super-column's own 64 entries: y = M b per 32x32 block, the second block
            // after the first one's contribution through the coupling tiles
            __builtin_amdgcn_wave_barrier();
            const int r = lane & 31;
            double* v = sm.vec + tb * 16;
            const double ya = inv_block_row(sm.U + (size_t)(2 * p) * SUB, v, r);
            __builtin_amdgcn_wave_barrier();
            if (lane < 32) v[lane] = ya;
            __builtin_amdgcn_wave_barrier();
            if (ncol > 2) {
                // b_b -= [L20 L21; L30 L31] y_a
                const int rr = r & 15;
                const double* La = sm.Lt + (r < 16 ? ltile(2, 0) : ltile(3, 0)) * STL + rr * SLD;
                const double* Lb = sm.Lt + (r < 16 ? ltile(2, 1) : ltile(3, 1)) * STL + rr * SLD;
                double s0 = 0.0, s1 = 0.0;
                if (r < 16 || ncol > 3) {
#pragma unroll
                    for (int c = 0; c < 16; c += 2) {
                        s0 += La[c] * v[c] + Lb[c] * v[16 + c];
                        s1 += La[c + 1] * v[c + 1] + Lb[c + 1] * v[17 + c];
                    }
                }
                if (lane < 32) v[32 + lane] -= s0 + s1;
                __builtin_amdgcn_wave_barrier();
                const double yb = inv_block_row(sm.U + (size_t)(2 * p + 1) * SUB, v + 32, r);
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) v[32 + lane] = yb;
            }
        }
        PROFW(47);
        return ok;
    }

    // ---- the factorisation ----------------------------------------------------------------------------------------
    __device__ __forceinline__ bool factor() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;              // this lane's double2 inside a 1 KB half tile
        const int ntr = (n + 15) >> 4;           // tile rows that hold valid rows
        const int nsc = (ntr + 3) >> 2;          // super-columns
        volatile int* vflag = sm.flag;
        if (wv == 1) {
            // prologue: the first diagonal block straight from P
            const int ne = ntr < 4 ? ntr : 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j)
                    if (i < ne) sm.In[dtile(i, j) * 64 + lane] = init_tile(i, j, ntr, fo, li, kq);
        }
        if (tid < 8) sm.flag[tid] = 0;
        __syncthreads();
        for (int p = 0; p < nsc; ++p) {
            const int tb = 4 * p;                    // first tile row / tile column of the super-column
            const int nk2 = 8 * p;                   // finished half-chunks (8 columns each)
            const int ncol = (ntr - tb) < 4 ? (ntr - tb) : 4;
            const int epoch = p + 1;
            if (wv == 0) {
                // ======== wavefront 0: the diagonal block =========================================================
                PROF_DECL
                const bool ok = chain(p, ncol, lane);
                PROFW(16);
                if (lane == 0) sm.flag[0] = ok ? 0 : 1;
                __syncthreads();                                    // (A) inverse tiles, coupling tiles, y published
                PROFW(17);
                if (vflag[0]) return false;
                __syncthreads();                                    // (B)
                PROFW(18);
                continue;
            } else if (wv == 1) {
                // ======== wavefront 1: rank-k update of the next 64x64 diagonal block ================================
                const int eb = tb + 4;
                const int ne = (ntr - eb) < 4 ? (ntr - eb) : 4;     // its valid tile rows (<= 0: none)
                v4d e[10];
                const char* pa[4] = {nullptr, nullptr, nullptr, nullptr};
                const unsigned voff = (unsigned)fo * 16u;
                if (ne > 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j <= i; ++j)
                            e[dtile(i, j)] = (i < ne) ? init_tile(eb + i, eb + j, ntr, fo, li, kq) : (v4d){0, 0, 0, 0};
                        pa[i] = uniform_ptr(tile2(i < ne ? eb + i : eb, 0));
                    }
                }
                struct Fr { v2d a[4]; };
                auto loadf = [&](Fr& f_, int k2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) f_.a[i] = gload16(pa[i] + (size_t)k2 * 1024, voff);
                };
                auto multf = [&](const Fr& f_) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j)
                            e[dtile(i, j)] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.a[j].x, f_.a[i].x, e[dtile(i, j)], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j)
                            e[dtile(i, j)] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.a[j].y, f_.a[i].y, e[dtile(i, j)], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto accumulate = [&](int k0, int k1) {             // half-chunks k0 .. k1-1 (a multiple of 4)
                    // hand-pipelined like the row wavefronts' loop: requests run three half-chunks ahead (4 loads per
                    // step, so "this step's operands have arrived" is vmcnt(12)); indices past the end are clamped
                    Fr f0, f1, f2, f3;
                    const int kl = k1 - 1;
                    auto ld = [&](Fr& f_, int k2) { loadf(f_, k2 < kl ? k2 : kl); };
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    ld(f0, k0); ld(f1, k0 + 1); ld(f2, k0 + 2);
                    for (int k2 = k0; k2 < k1; k2 += 4) {
                        ld(f3, k2 + 3); vm_wait<12>(); multf(f0);
                        ld(f0, k2 + 4); vm_wait<12>(); multf(f1);
                        ld(f1, k2 + 5); vm_wait<12>(); multf(f2);
                        ld(f2, k2 + 6); vm_wait<12>(); multf(f3);
                    }
                    vm_wait<0>();
                };
                PROF_DECL
                if (ne > 0 && nk2 > 0) accumulate(0, nk2);
                PROFW(20);
                __syncthreads();                                    // (A)
                PROFW(21);
                if (vflag[0]) return false;
                if (ne > 0) {
                    // the four tile columns of this super-column, as soon as their owners have stored them
                    int spins = 0;
                    for (int i = 0; i < ne; ++i)
                        while (vflag[1 + i] != epoch && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(2);
                    if (spins >= (1 << 24)) vflag[0] = 2;           // never expected: give up instead of hanging
                    asm volatile("" ::: "memory");
                    PROFW(22);
                    accumulate(nk2, nk2 + 8);
                    // non-existing rows of the block: identity diagonal, nothing else (they are never referenced)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j)
                            if (i < ne) sm.In[dtile(i, j) * 64 + lane] = e[dtile(i, j)];
                }
                PROFW(23);
                __syncthreads();                                    // (B)
                PROFW(24);
                continue;
            } else {
                // ======== wavefronts 2..7: the tile rows below the diagonal block ==================================
                const int fr = tb + 4;
                const int nsq = ntr - fr > 0 ? ntr - fr : 0;                 // rows of the square matrix below the block
                const int nothers = nsq + nex;                               // ... followed by the appended rows
                const int npass = nothers > SRW * SMAXT ? (nothers + SRW * SMAXT - 1) / (SRW * SMAXT) : 1;
                PROF_DECL
                const int pslot = wv == 2 ? 26 : (wv == 7 ? 32 : 38);     // profile build: wave 2, wave 7, the other four
                (void)pslot;
#pragma unroll 1
                for (int ps = 0; ps < npass; ++ps) {
                    int T[SMAXT];
                    bool act[SMAXT];
#pragma unroll
                    for (int u = 0; u < SMAXT; ++u) {
                        const int slot = (wv - 2) + SRW * (u + SMAXT * ps);
                        T[u] = slot < nsq ? fr + slot : nch + (slot - nsq);
                        act[u] = slot < nothers;
                    }
                    // ---- (1) acc = -(S tile) + sum_c L(T, c) L(tb + ct, c)' ------------------------------------
                    v4d acc[SMAXT][4];
#pragma unroll
                    for (int u = 0; u < SMAXT; ++u)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[u][ct] = (act[u] && ct < ncol) ? init_tile(T[u], tb + ct, ntr, fo, li, kq) : (v4d){0, 0, 0, 0};
                    if (nk2 > 0 && act[0]) {
                        // Operand ring, software-pipelined by hand: the A tiles (this wavefront's own rows, from HBM) are
                        // requested THREE half-chunks ahead, the B tiles (the block's rows, shared by all wavefronts: L1 /
                        // L2) one ahead.  Per step 4 B + 4 A loads are issued, B first, so "B of this step has arrived"
                        // is vmcnt(12): A(k+2), B(k+1), A(k+3) may still be in flight.  Indices past the end are clamped
                        // (a redundant load of the last half-chunk) so that the counts stay uniform.
                        const char* rb[4];
                        const char* ra[SMAXT];
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) rb[ct] = uniform_ptr(tile2(ct < ncol ? tb + ct : tb, 0));   // stand-in for padding
#pragma unroll
                        for (int u = 0; u < SMAXT; ++u) ra[u] = uniform_ptr(tile2(act[u] ? T[u] : tb, 0));
                        const unsigned voff = (unsigned)fo * 16u;
                        struct SlA { v2d a[SMAXT]; };
                        struct SlB { v2d b[4]; };
                        const int klast = nk2 - 1;
                        auto loadA = [&](SlA& s_, int k2) {
                            const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
#pragma unroll
                            for (int u = 0; u < SMAXT; ++u) s_.a[u] = gload16(ra[u] + o, voff);
                        };
                        auto loadB = [&](SlB& s_, int k2) {
                            const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) s_.b[ct] = gload16(rb[ct] + o, voff);
                        };
                        auto mult = [&](const SlA& a_, const SlB& b_) {
#pragma unroll
                            for (int u = 0; u < SMAXT; ++u)
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct)
                                    acc[u][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b[ct].x, a_.a[u].x, acc[u][ct], 0, 0, 0);
#pragma unroll
                            for (int u = 0; u < SMAXT; ++u)
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct)
                                    acc[u][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b[ct].y, a_.a[u].y, acc[u][ct], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        SlA a0, a1, a2, a3;
                        SlB b0, b1;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile loads above: from here on the count is ours
                        __builtin_amdgcn_sched_barrier(0);
                        loadB(b0, 0); loadA(a0, 0); loadA(a1, 1); loadA(a2, 2);
                        for (int k2 = 0; k2 < nk2; k2 += 4) {       // nk2 = 8p: a multiple of 4
                            loadB(b1, k2 + 1); loadA(a3, k2 + 3); vm_wait<12>(); mult(a0, b0);
                            loadB(b0, k2 + 2); loadA(a0, k2 + 4); vm_wait<12>(); mult(a1, b1);
                            loadB(b1, k2 + 3); loadA(a1, k2 + 5); vm_wait<12>(); mult(a2, b0);
                            loadB(b0, k2 + 4); loadA(a2, k2 + 6); vm_wait<12>(); mult(a3, b1);
                        }
                        vm_wait<0>();                               // the clamped requests past the end
                    }
                    PROFW(pslot);                                   // initial tiles + rank-k update
                    if (ps == 0) {
                        __syncthreads();                            // (A) published by wavefront 0
                        PROFW(pslot + 1);
                        if (vflag[0]) return false;
                    }
                    // ---- (2) panel solve from registers, right-looking over the tile columns; row slot 0 first: in pass
                    //          0 it is one of the rows the look-ahead wavefront is waiting for ---------------------------
                    if (act[0]) {
                        const bool publish = ps == 0 && (wv - 2) < 4 && (wv - 2) < nsq;
#pragma unroll
                        for (int grp = 0; grp < 2; ++grp) {
                            const int u0 = grp == 0 ? 0 : 1, u1 = grp == 0 ? 1 : SMAXT;
                            if (grp == 1 && !act[1]) break;
                            double pf[SMAXT];
#pragma unroll
                            for (int u = 0; u < SMAXT; ++u) pf[u] = 0.0;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (c < ncol) {
                                    const double* Wd = wtile(p, c);
                                    double wn[4];
#pragma unroll
                                    for (int s_ = 0; s_ < 4; ++s_) wn[s_] = -Wd[li * SLD + 4 * s_ + kq];
                                    v4d x[SMAXT];
#pragma unroll
                                    for (int u = 0; u < SMAXT; ++u) x[u] = (v4d){0, 0, 0, 0};
#pragma unroll
                                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                                        for (int u = 0; u < SMAXT; ++u)
                                            if (u >= u0 && u < u1)
                                                x[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn[s_], acc[u][c][s_], x[u], 0, 0, 0);
#pragma unroll
                                    for (int u = 0; u < SMAXT; ++u)
                                        if (u >= u0 && u < u1 && act[u]) store_tile(T[u], tb + c, x[u], fo);
                                    if (fwd) {
                                        const double* yv = sm.vec + (tb + c) * 16 + kq;
#pragma unroll
                                        for (int u = 0; u < SMAXT; ++u)
                                            if (u >= u0 && u < u1)
                                                pf[u] += (x[u][0] * yv[0] + x[u][1] * yv[4]) + (x[u][2] * yv[8] + x[u][3] * yv[12]);
                                    }
#pragma unroll
                                    for (int c2 = c + 1; c2 < 4; ++c2) {
                                        if (c2 < ncol) {
                                            const double* Lr = sm.Lt + ltile(c2, c) * STL;
                                            double lf[4];
#pragma unroll
                                            for (int s_ = 0; s_ < 4; ++s_) lf[s_] = Lr[li * SLD + 4 * s_ + kq];
#pragma unroll
                                            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                                                for (int u = 0; u < SMAXT; ++u)
                                                    if (u >= u0 && u < u1)
                                                        acc[u][c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(lf[s_], x[u][s_], acc[u][c2], 0, 0, 0);
                                        }
                                    }
                                }
                            }
                            if (grp == 0 && publish) {
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the row's four tiles have reached L2
                                if (lane == 0) vflag[1 + (wv - 2)] = epoch;
                            }
                            if (grp == 0) PROFW(pslot + 2); else PROFW(pslot + 3);
                            if (fwd) {
#pragma unroll
                                for (int u = 0; u < SMAXT; ++u) {
                                    if (u >= u0 && u < u1) {
                                        double p_ = pf[u];
                                        p_ += __shfl_xor(p_, 16, 64);
                                        p_ += __shfl_xor(p_, 32, 64);
                                        if (kq == 0 && act[u] && T[u] < nch) sm.vec[T[u] * 16 + li] -= p_;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();                                        // (B) super-column visible to everyone
#ifdef HIPDRT_QP_PROFILE
            if (wv >= 2 && (threadIdx.x & 63) == 0 && blockIdx.x == 0)
                atomicAdd(&g_qp_prof[(wv == 2 ? 26 : (wv == 7 ? 32 : 38)) + 4], 1ull);   // (rows' wait at B is 18/24's mirror)
#endif
        }
        return true;
    }

    // -----------------------------------------------------------------------------------------------------
    // vec := S^-1 vec by 32-wide blocks.  Wavefront 0 multiplies by the inverse 32x32 diagonal blocks (LDS resident);
    // wavefronts 1..7 apply the rank-32 updates from tiles fetched TWO blocks ahead into alternating register buffers;
    // the barriers order LDS traffic only, so those loads stay in flight across them.
    //   tile load map: instruction h (k-half) of a 2 KB tile covers double2 index h*64 + lane  ->  row i = lane/4,
    //   columns 8h + (lane%4) and 8h + (lane%4) + 4.
    __device__ __forceinline__ void forward() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* vec = sm.vec;
        constexpr int UW = SNW - 1;                 // updater wavefronts
        constexpr int FT = (31 + UW - 1) / UW;      // forward: tiles per updater wavefront
        constexpr int BC = (32 + UW - 1) / UW;      // backward: chunks per updater wavefront
        if (wv == 0) {
            const int r = lane & 31;
            for (int jb = 0; jb < nblk; ++jb) {
                const int j0 = jb * NB;
                const double y = inv_block_row(sm.U + (size_t)jb * SUB, vec + j0, r);
                __builtin_amdgcn_wave_barrier();
                if (lane < NB) vec[j0 + lane] = y;
                lds_barrier();
                lds_barrier();
            }
        } else {
            const int l4 = lane & 3, g4 = lane >> 2;
            struct Buf { double2 t[FT > BC ? FT : BC][4]; };
            auto fpre = [&](Buf& B_, int jb) {          // tiles (tb+2+tt, 2jb..2jb+1): [tile][chunk*2 + half]
                if (jb < nblk) {
                    const int tb = 2 * jb, tbelow = ntr - (tb + 2);
#pragma unroll
                    for (int u = 0; u < FT; ++u) {
                        const int tt = (wv - 1) + u * UW;
                        if (tt < tbelow) {
                            const double2* p = tile2(tb + 2 + tt, 2 * jb) + lane;   // chunks 2jb, 2jb+1 are adjacent
#pragma unroll
                            for (int q = 0; q < 4; ++q) B_.t[u][q] = p[q * 64];
                        }
                    }
                }
            };
            auto fstep = [&](Buf& B_, int jb) {
                const int j0 = jb * NB, tb = 2 * jb, tbelow = ntr - (tb + 2);
                lds_barrier();
                if (tbelow > 0) {
                    double ya[4], yb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ya[q] = vec[j0 + 8 * q + l4]; yb[q] = vec[j0 + 8 * q + l4 + 4]; }
#pragma unroll
                    for (int u = 0; u < FT; ++u) {
                        const int tt = (wv - 1) + u * UW;
                        if (tt < tbelow) {
                            double pv = 0.0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) pv += B_.t[u][q].x * ya[q] + B_.t[u][q].y * yb[q];
                            pv = quad_sum(pv);
                            if (l4 == 0) {
                                const int row = (tb + 2 + tt) * 16 + g4;
                                if (row < n) vec[row] -= pv;
                            }
                        }
                    }
                }
                fpre(B_, jb + 2);
                lds_barrier();
            };
            Buf fa, fb;
            fpre(fa, 0);
            fpre(fb, 1);
            for (int jb = 0; jb < nblk; jb += 2) {
                fstep(fa, jb);
                if (jb + 1 < nblk) fstep(fb, jb + 1);
            }
        }
    }

    __device__ __forceinline__ void backward() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* vec = sm.vec;
        constexpr int UW = SNW - 1;
        constexpr int FT = (31 + UW - 1) / UW;
        constexpr int BC = (32 + UW - 1) / UW;
        if (wv == 0) {
            const int c = lane & 31;
            for (int jb = nblk - 1; jb >= 0; --jb) {     // x = M' y, lane = column of M
                const int j0 = jb * NB;
                const double xv = inv_block_col(sm.U + (size_t)jb * SUB, vec + j0, c);
                __builtin_amdgcn_wave_barrier();
                if (lane < NB) vec[j0 + lane] = xv;
                lds_barrier();
                lds_barrier();
            }
        } else {
            const int l4 = lane & 3, g4 = lane >> 2;
            struct Buf { double2 t[FT > BC ? FT : BC][4]; };
            auto bpre = [&](Buf& B_, int jb) {          // tiles (tb..tb+1, c): [chunk][tile*2 + half]
                if (jb >= 0) {
                    const int tb = 2 * jb, nc = 2 * jb;
                    const bool two = (tb + 1) < ntr;
#pragma unroll
                    for (int u = 0; u < BC; ++u) {
                        const int c = (wv - 1) + u * UW;
                        if (c < nc) {
                            const double2* p0 = tile2(tb, c) + lane;
                            const double2* p1 = tile2(two ? tb + 1 : tb, c) + lane;
                            B_.t[u][0] = p0[0]; B_.t[u][1] = p0[64];
                            B_.t[u][2] = p1[0]; B_.t[u][3] = p1[64];
                        }
                    }
                }
            };
            auto bstep = [&](Buf& B_, int jb) {
                const int j0 = jb * NB, tb = 2 * jb, nc = 2 * jb;
                const bool two = (tb + 1) < ntr;
                lds_barrier();
                if (nc > 0) {
                    const double x0 = vec[j0 + g4];
                    const double x1 = two ? vec[j0 + 16 + g4] : 0.0;
#pragma unroll
                    for (int u = 0; u < BC; ++u) {
                        const int c = (wv - 1) + u * UW;
                        if (c < nc) {
                            double s0 = B_.t[u][0].x * x0 + B_.t[u][2].x * x1;     // column 16c + l4
                            double s1 = B_.t[u][0].y * x0 + B_.t[u][2].y * x1;     // column 16c + l4 + 4
                            double s2 = B_.t[u][1].x * x0 + B_.t[u][3].x * x1;     // column 16c + 8 + l4
                            double s3 = B_.t[u][1].y * x0 + B_.t[u][3].y * x1;     // column 16c + 12 + l4
                            const double f = colsum4(s0, s1, s2, s3, lane);
                            if ((lane & 12) == 0) vec[c * 16 + l4 + 4 * (lane >> 4)] -= f;
                        }
                    }
                }
                bpre(B_, jb - 2);
                lds_barrier();
            };
            Buf ba, bb;
            bpre(ba, nblk - 1);
            bpre(bb, nblk - 2);
            for (int jb = nblk - 1; jb >= 0; jb -= 2) {
                bstep(ba, jb);
                if (jb - 1 >= 0) bstep(bb, jb - 1);
            }
        }
    }

    __device__ __forceinline__ void solve() {
        forward();
        backward();
    }

    // -----------------------------------------------------------------------------------------------------
    // dvec = P * vec from the packed lower tiles: every tile is read once, as two contiguous 1 KB loads, and used for
    // both  y_T += tile x_C  and  y_C += tile' x_T.  The tiles, in column-major order, are dealt to the wavefronts in
    // equal contiguous runs; every wavefront adds into its OWN partial result vector (the array of inverse blocks is
    // dead between two factorisations and serves as scratch), so the final sum has a fixed order.
    __device__ __forceinline__ void matvec() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int l4 = lane & 3, g4 = lane >> 2;
        const int ntr = (n + 15) >> 4;
        const int NPd = nch * 16;
        const double* xin = sm.vec;
        double* yw = sm.U + (size_t)wv * NPd;               // this wavefront's partial result
        for (int i = lane; i < NPd; i += 64) yw[i] = 0.0;
        const int ntile = ntr * (ntr + 1) / 2;
        const int per = (ntile + SNW - 1) / SNW;
        const int t0 = wv * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
        int C = 0, rem = t0;
        while (C < ntr && rem >= ntr - C) { rem -= ntr - C; ++C; }
        int T = C + rem;
        struct TileR { double2 d0, d1; };
        auto tload = [&](TileR& r_, int T_, int C_) {
            const double2* tile = reinterpret_cast<const double2*>(Ppk + ((size_t)T_ * nchp + C_) * 256);
            r_.d0 = tile[lane]; r_.d1 = tile[64 + lane];
        };
        auto advance = [&](int& T_, int& C_) { if (++T_ == ntr) { ++C_; T_ = C_; } };
        constexpr int PDM = 4;
        TileR ring[PDM];
        int Tp = T, Cp = C;
        int tp = t0;
#pragma unroll
        for (int k = 0; k < PDM; ++k) { if (tp < t1) { tload(ring[k], Tp, Cp); advance(Tp, Cp); ++tp; } }
        double xc0 = 0.0, xc1 = 0.0, xc2 = 0.0, xc3 = 0.0, ca0 = 0.0, ca1 = 0.0, ca2 = 0.0, ca3 = 0.0;
        int Ccur = -1;
        auto flush = [&]() {
            if (Ccur >= 0) {
                const double f = colsum4(ca0, ca1, ca2, ca3, lane);
                if ((lane & 12) == 0) yw[Ccur * 16 + l4 + 4 * (lane >> 4)] += f;
            }
        };
        for (int t = t0; t < t1; t += PDM) {
#pragma unroll
            for (int k = 0; k < PDM; ++k) {
                if (t + k < t1) {
                    const TileR cur = ring[k];
                    if (tp < t1) { tload(ring[k], Tp, Cp); advance(Tp, Cp); ++tp; }
                    if (C != Ccur) {
                        flush();
                        Ccur = C;
                        const double* xc = xin + C * 16 + l4;
                        xc0 = xc[0]; xc1 = xc[4]; xc2 = xc[8]; xc3 = xc[12];
                        ca0 = ca1 = ca2 = ca3 = 0.0;
                    }
                    double pr = cur.d0.x * xc0 + cur.d0.y * xc1 + cur.d1.x * xc2 + cur.d1.y * xc3;
                    pr = quad_sum(pr);
                    if (l4 == 0) yw[T * 16 + g4] += pr;
                    if (T != C) {                              // the diagonal tile is stored in full
                        const double xt = xin[T * 16 + g4];
                        ca0 += cur.d0.x * xt; ca1 += cur.d0.y * xt; ca2 += cur.d1.x * xt; ca3 += cur.d1.y * xt;
                    }
                    advance(T, C);
                }
            }
        }
        flush();
        __syncthreads();
        for (int i = tid; i < n; i += ST) {
            double s_ = 0.0;
#pragma unroll
            for (int w = 0; w < SNW; ++w) s_ += sm.U[(size_t)w * NPd + i];
            sm.dvec[i] = s_;
        }
        __syncthreads();
        for (int i = tid; i < SNW * NPd; i += ST) sm.U[i] = 0.0;     // the scratch goes back zeroed
    }
};

// ---------------------------------------------------------------------------------------------------------
// Posterior variance of the distribution on an evaluation grid, the diagonal of
// drt1d.estimate_distribution_cov (hybdrt/models/drt1d.py:3063-3151, 4116-4138):
//     var_i = b_i' P^-1 b_i = || L^-1 b_i ||^2 ,   P = L L' ,  b_i = row i of the basis-evaluation matrix
// (zero in the special-parameter slots).  The rows b_i ride through the factorisation as appended panel rows.
// ---------------------------------------------------------------------------------------------------------
static __host__ __device__ inline int super_nblk32(int n) { return (n + 31) / 32; }

__global__ __launch_bounds__(ST) void cov_kernel_super(CovArgs a, int NP64) {
    const int b = blockIdx.x;
    extern __shared__ double smem[];
    OpsSuper ops;
    ops.L = a.L + (size_t)b * a.l_stride; ops.nch = NP64 / 16; ops.n = a.n;
    ops.Ppk = a.Ppk + (size_t)b * a.ppk_stride; ops.nchp = a.nchp;
    ops.nex = a.nex; ops.Bex = a.Bex; ops.fwd = false;
    ops.sm.carve(smem);
    for (int i = threadIdx.x; i < super_nblk32(a.n) * SUB; i += ST) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < SupSmem::VEC; i += ST) { ops.sm.vec[i] = 0.0; ops.sm.dvec[i] = 0.0; }   // no diagonal shift
    __syncthreads();
    const bool ok = ops.factor();
    double* out = a.out + (size_t)b * a.out_stride;
    if (threadIdx.x == 0) a.status[b] = ok ? 0 : -1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l4 = lane & 3, g4 = lane >> 2;
    const int ntr = (a.n + 15) >> 4;
    for (int e = wv; e < a.nex; e += SNW) {
        double s_ = 0.0;
        if (ok) {
            for (int c = 0; c < ntr; ++c) {
                const double2* t = ops.tile2(ops.nch + e, c) + lane;      // row lane/4, 4 of its 16 columns
                const double2 d0 = t[0], d1 = t[64];
                s_ += d0.x * d0.x + d0.y * d0.y + d1.x * d1.x + d1.y * d1.y;
            }
            s_ = quad_sum(s_);
        } else {
            s_ = __builtin_nan("");
        }
        if (l4 == 0) out[e * 16 + g4] = s_;
    }
}

__global__ __launch_bounds__(ST) void qp_kernel_super(QpArgs a, int NP64) {
    const int b = a.order ? a.order[blockIdx.x] : blockIdx.x;
    if (a.active && !a.active[b]) return;
    extern __shared__ double smem[];
    OpsSuper ops;
    ops.L = a.L + (size_t)b * a.l_stride; ops.nch = NP64 / 16; ops.n = a.n;
    ops.Ppk = a.Ppk ? a.Ppk + (size_t)b * a.ppk_stride : nullptr; ops.nchp = a.nchp;
    ops.sm.carve(smem);
    for (int i = threadIdx.x; i < super_nblk32(a.n) * SUB; i += ST) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < SupSmem::VEC; i += ST) ops.sm.vec[i] = 0.0;
    __syncthreads();
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    ipm_solve<ST, (SNP_MAX + ST - 1) / ST>(a, b, ops, is);
}

static size_t super_lds_bytes(int n) {
    return ((size_t)super_nblk32(n) * SUB + SupSmem::FIXED) * sizeof(double);
}

// scratch doubles per problem for the tile-packed factor: (NP64/16)^2 tiles of 256 doubles
static size_t super_l_doubles(int n) {
    const size_t nt = (size_t)round_up(n, 64) / 16;
    return nt * nt * STSZ;
}

}  // namespace hipdrt

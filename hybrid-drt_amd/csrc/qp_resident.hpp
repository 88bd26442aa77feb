// Linear-algebra services of the coneqp kernel for n <= 528 unknowns (C1-C4 sizes: n = 93 ... 514): one
// 512-thread workgroup (8 wavefronts) per problem, one per CU.
//
//  * The Cholesky factor L lives in HBM in a TILE-PACKED layout: 16x16 tiles, each one contiguous 2 KB block,
//    tiles of one tile-row adjacent ([tile_row][k_chunk][256]).  Inside a tile the double2 with index
//    h*64 + i*4 + q (h = k-half, i = row, q = 0..3) holds columns q + 8h and q + 8h + 4 of row i.  With lane
//    (i = lane&15, q = lane>>4) reading that double2, a wavefront's MFMA operand load covers one contiguous 1 KB
//    (full 128-byte lines), and it is also exactly the register image in which the factorisation produces a tile
//    (below), so tiles are stored with two 16-byte-per-lane instructions and no shuffle.  The accumulator-native
//    copy of P (Ppk, written by the Gram kernel) uses the same tile layout.
//  * Left-looking blocked Cholesky, block 32.  Every wavefront owns up to 4 tile rows x 2 tile columns of the block
//    column and accumulates the TRANSPOSED tiles  accT = -(S - L L')'  with v_mfma_f64_16x16x4_f64 (operand slabs
//    ping-pong prefetched in registers).  In that register image a tile is directly the B operand of the next
//    MFMA, so the triangular solve against the diagonal block runs on the matrix pipe from registers:
//        X1' = W1 C1' ,  C2' -= L21 X1' ,  X2' = W2 C2'      (W = inverse of a 16x16 diagonal Cholesky block)
//    -- no LDS panel, no thread-per-row substitution.
//  * Wave roles.  Wavefront 0 does nothing but factor and invert the two 16x16 diagonal blocks of a block column
//    (register-only Gauss-Jordan on [D | I], lane = row, pivots / multipliers / finished rows broadcast with
//    v_readlane).  Wavefront 1 owns the two tile rows of the NEXT diagonal block: their panel tiles in the current
//    column and, one column ahead, the whole rank-k update of the next diagonal block, handed to wavefront 0 through
//    LDS -- so the sequential diagonal work of column j+1 overlaps everybody else's rank-k update of column j+1.
//    Wavefronts 2..7 own all rows below.
//  * LDS array U[NP][33] keeps, for every finished block, the INVERSE of its 32x32 diagonal Cholesky block
//    [[W1, 0], [-W2 L21 W1, W2]], so the triangular solves do one 32x32 mat-vec per block instead of a 32-step
//    substitution chain and never fetch diagonal blocks from HBM (the diagonal-block tiles of L are not even
//    written to HBM).
//  * The predictor's forward substitution is fused into the factorisation (a block column's tiles update the
//    right-hand side while they are still in registers).  Separate solves: per 32-block the mat-vec by wavefront 0, which
//    also applies the two tiles of the rank-32 update its next step needs; wavefronts 1..7 apply the rest meanwhile (tiles
//    fetched two blocks ahead, one LDS-only barrier per block: forward() / backward()).
//  * P x from the packed lower tiles: every tile is read once and used for y_T += tile x_C and y_C += tile' x_T.
//  * Optional extra tile rows appended below the matrix turn the same factorisation into a multi-right-hand-side
//    triangular solve (posterior variance, cov_kernel_resident).
#pragma once
#include "qp_common.hpp"


namespace hipdrt {

static constexpr int RT = 512;           // threads
static constexpr int RNW = RT / 64;      // 8 wavefronts
// (non-temporal loads for the source tiles of P -- read once per factorisation -- were tried: __builtin_nontemporal_load in
// tile_src makes hipcc 7.2's simplifycfg pass crash on this translation unit, in either coneqp kernel)
// Tunables that were swept on the device and are settled (the sweeps: DESIGN.md section 6; the variants that lost -- register-only
// Gauss-Jordan chain, two workgroups per CU, row prefetch before barrier (B), look-ahead helper, un-split barrier (B), 4 rows per
// pass -- are kept as patches under tools/experiments/, not as dead branches here):
static constexpr int RMAXT = 4;          // factor(): tile rows per wavefront and pass
static constexpr int kSweepCap = 5;      // register-buffered tiles per trailing-update wavefront in the sweeps (vm_wait_tiles covers <= 5)
static constexpr int kLa1Load = 14;      // factor64 row schedule: MFMAs per half-chunk booked on SIMD 0 for wavefront 0's look-ahead
static constexpr int kLa2Load = 22;      //                        ... on SIMD 1 for wavefront 1's (0 / 7 / 14 / 21 / 28 and 16 / 22 / 30 measured)
static constexpr int RNP_MAX = 528;
static constexpr int TSZ = 256;          // doubles per 16x16 tile
static constexpr int DLD = 17;           // row stride of the 16x16 LDS scratch blocks

// GU = the inverse diagonal blocks U live in GLOBAL memory (per-problem scratch behind the factor, L2 resident) instead of
// LDS: the only structure of this kernel whose size grows with n^1 x 33, i.e. what limits the LDS-resident form to
// n <= 528.  With U outside, the same kernel serves every n <= 2048 (LDS then holds the two n-vectors and the small
// fixed buffers: 39 kB).
#ifndef HIPDRT_QP_PANEL64
#define HIPDRT_QP_PANEL64 1      // the QP kernel's factorisation walks the finished part of L once per 64 columns (factor64)
#endif
// P64 = LDS layout of the 64-column factorisation (factor64, the QP kernel with 8 wavefronts): L21 of BOTH 32-blocks of a super
// column in operand-fragment layout (t21: [2][4][64]), two reduction slots instead of four, dvec without the 32 padding
// entries, a schedule table per super column -- at n = 514 the workgroup's 160 KB are used to the last 32 bytes.
// VT = virtual threads per thread of the interior-point driver (qp_common.hpp: Reducer): the reduction slots are per VIRTUAL wavefront
template <bool GU, int RTT = 512, bool P64 = false, int VT = 1>
struct ResSmemT {
    double* U;       // [NP][PLD]   inverse diagonal blocks (LDS, or global when GU)
    double* vec;     // [NP + 32]
    double* dvec;    // [NP + 32]   (P64: [NP])
    double* red;     // [4][RNW][4] (P64: [2][RNW][4])
    double* t21;     // [16][DLD]   L21 of the current block (P64: [2][4][64], register images of L21 of blocks a and b)
    double* dsc;     // [16][DLD]   diagonal block being factored
    double* img;     // [2][64][4]  register images of -D21', -D22' of the next diagonal block
    int* flag;       // [4]
    unsigned char* sched;   // [nblk][SROW] owner wavefront of every tile row below a block column (build_schedule)

    // fixed offsets for everything but U, so that the small buffers have compile-time LDS addresses
    // NP + 32: NP <= 544 with U in LDS; with U outside the QP kernel (P64: n <= 2048) needs 2112, the posterior-variance kernel
    // (n <= 4096, as far as the QP entry point reaches with the group kernel) 4160 -- per instantiation, so that the QP kernel's
    // workgroup does not hold 33 kB of LDS it never touches away from the Gram / hyper kernels of other streams on its CU
    static constexpr int VEC = GU ? (P64 ? 2048 + 32 + 32 : 4096 + 32 + 32) : 528 + 16 + 32;
    static constexpr int DVEC = P64 ? VEC - 32 : VEC;
    static constexpr int SROW = GU ? 128 : 32;                           // table row: tile rows below a block column (<= 124 | 29)
    static constexpr int SCHED = GU ? 64 * 128 / 8 : (P64 ? 9 * 32 / 8 : 17 * 32 / 8 + 4);    // doubles: nblk <= 64 | 17 (9 super columns) rows of SROW bytes
    static constexpr int RED = (P64 ? 2 : 4) * (RTT / 64) * VT * 4;
    static constexpr int T21 = P64 ? 2 * 256 : 16 * 17;
    static constexpr int FIXED = RED + T21 + 16 * 17 + 8 + 512 + SCHED + VEC + DVEC;   // doubles before U
    __device__ __forceinline__ void carve(double* smem) {
        red = smem;
        t21 = red + RED;
        dsc = t21 + T21;
        flag = reinterpret_cast<int*>(dsc + 16 * 17);
        img = dsc + 16 * 17 + 8;
        sched = reinterpret_cast<unsigned char*>(img + 512);
        vec = img + 512 + SCHED;
        dvec = vec + VEC;
        U = dvec + DVEC;
    }
};
static_assert((ResSmemT<false, 512, true>::FIXED + 544 * 33) * 8 <= 160 * 1024, "n = 514 must fit one CU's LDS");
static_assert((ResSmemT<false, 512, true>::RED + ResSmemT<false, 512, true>::T21 + 16 * 17 + 8) % 4 == 0, "img must be 32-byte aligned");

using ResSmem = ResSmemT<false>;

// RTT = threads per workgroup.  512 (8 wavefronts: chain, look-ahead, six row wavefronts; one workgroup per CU) is what
// is built.  256 (chain, look-ahead, two row wavefronts, U in global memory, two workgroups per CU so that one's sequential
// phases overlap the other's matrix work) compiles to 256 VGPRs and is parity-green, but measured 12.5 ms per launch against
// 10.6: both workgroups put their row wavefronts on the same two SIMDs and every block column needs three to four passes.
// Two 512-THREAD workgroups per CU (128 VGPRs each: two rows per pass, ring depth 2, two buffered sweep tiles, U in global memory
// so that LDS admits two; round 2, the knobs went with tools/experiments/qp_resident_retired_knobs.patch) was parity-green as
// well -- the ~200 spilled registers stayed outside the operand rings -- and measured 14.7 ms per launch against 11.3 for one
// workgroup per CU in the same U-outside form (10.65 with U in LDS): halving every wavefront's rows per pass, ring depth and
// sweep buffers costs more than the second workgroup's overlap returns.
template <bool GU, int RTT = 512, bool P64 = false, int VT = 1>
struct OpsResidentT {
    static constexpr int RT = RTT, RNW = RTT / 64;                     // (shadow the namespace-level defaults)
    static_assert(!P64 || RTT == 512 || RTT == 256, "factor64 is written for eight wavefronts, or four fat ones");
    using Smem = ResSmemT<GU, RTT, P64, VT>;
    static constexpr int kVT = VT;
    static constexpr int kRedSlots = P64 ? 2 : 4;
    double* L; int nch; int n; Smem sm;                                // nch = tiles per tile-row (NP/16)
    const double* Ppk; int nchp;                                       // P in L's tile layout (lower tiles)
    // Optional extra tile rows appended below the square matrix (tile rows nch .. nch+nex-1 of L, source tiles
    // Bex[nex][nchp][256]): the factorisation treats them as more panel rows, so they come out as Bex * L^-T --
    // the multi-right-hand-side triangular solve of the posterior-variance kernel at the price of a taller panel.
    int nex = 0; const double* Bex = nullptr;
    // The factorisation also forward-substitutes the right-hand side waiting in sm.vec (column by column, as soon as
    // a block column's tiles sit in registers): the predictor's forward sweep costs no pass over L in HBM.
    static constexpr bool kFusedForward = true;
    bool fwd = true;

    // Which row wavefront owns which tile row below block column jb.  The rank-k phase is bound by the busiest SIMD's FP64
    // MFMA pipe: wavefront w sits on SIMD w % 4, the look-ahead wavefront (1) carries a fixed 14 MFMAs per half-chunk
    // whatever the number of rows left, and a plain round-robin gives its SIMD partner (5) as many rows as everyone else
    // (sum over the block columns of the busiest SIMD at n = 514: 657k cycles, 417k if perfectly balanced).  The table gives
    // every row to the least-loaded SIMD instead -- one thread per block column, once per launch; results do not depend on
    // who computes a row.  Only for the plain factorisation (no appended rows) with 8 wavefronts.
    bool balanced = false;
    __device__ __forceinline__ void build_schedule() {
        if constexpr (P64) { build_schedule64(); return; }
        balanced = (RNW == 8) && nex == 0;
        if (!balanced) return;
        const int ntr = (n + 15) >> 4, nblk = (n + NB - 1) / NB;
        constexpr int SROW = Smem::SROW;
        for (int jb = threadIdx.x; jb < nblk; jb += RT) {
            const int tb = 2 * jb, nk2 = 4 * jb;
            const int nsq = ntr - (tb + 4) > 0 ? ntr - (tb + 4) : 0;
            const int c = 4 * nk2 + 12;                               // MFMAs of one row: rank-k + triangular solve
            // no more passes than the round-robin needs: a wavefront's rows beyond its first RMAXT are streamed after
            // barrier (A), i.e. in series with the diagonal chain
            const int cap = RMAXT * (nsq > 6 * RMAXT ? (nsq + 6 * RMAXT - 1) / (6 * RMAXT) : 1);
            // SIMD 0 also runs the diagonal chain (~21k cycles per block column = ~300 MFMA slots of FP64 vector work that a
            // partner's FP64 MFMAs slow down by 40 %, profiles/r02d_chain_partner.txt): wavefront 4 is charged with it
            int l0 = 150, l1 = (tb + 2 < ntr) ? 14 * nk2 + 24 : 0, l2 = 0, l3 = 0;      // MFMAs per SIMD
            int c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;                          // rows per wavefront
            for (int r = 0; r < nsq; ++r) {
                // candidate of each SIMD: its row wavefront with fewer rows (SIMD 0: 4, SIMD 1: 5, SIMD 2: 2|6, SIMD 3: 3|7)
                const int w2 = c2 <= c6 ? 2 : 6, n2 = c2 <= c6 ? c2 : c6;
                const int w3 = c3 <= c7 ? 3 : 7, n3 = c3 <= c7 ? c3 : c7;
                int best = -1, bl = 0x7fffffff;
                if (c4 < cap && l0 < bl) { best = 4; bl = l0; }
                if (n2 < cap && l2 < bl) { best = w2; bl = l2; }
                if (n3 < cap && l3 < bl) { best = w3; bl = l3; }
                if (c5 < cap && l1 < bl) { best = 5; bl = l1; }
                sm.sched[jb * SROW + r] = (unsigned char)best;
                if (best == 4) { ++c4; l0 += c; }
                else if (best == 5) { ++c5; l1 += c; }
                else if (best == 2) { ++c2; l2 += c; }
                else if (best == 6) { ++c6; l2 += c; }
                else if (best == 3) { ++c3; l3 += c; }
                else { ++c7; l3 += c; }
            }
        }
    }

    // tile (t, c) starts at ((t*nch + c) * TSZ) doubles; returned in double2 units
    __device__ __forceinline__ const double2* tile2(int t, int c) const {
        return reinterpret_cast<const double2*>(L) + (size_t)((t * nch + c) * (TSZ / 2));
    }

    // lane id recomputed on the spot (two VALU instructions) instead of carried in a register across the kernel
    static __device__ __forceinline__ int fresh_lane() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    }

    // Wavefront 0: W = inverse of the Cholesky factor of the 16x16 block D (given as at = -D' in accumulator layout)
    // -> U[(r0+i)*PLD + c0 + j].  Register-only Gauss-Jordan on [D | I]: lane r holds row r of D and of W, the pivot,
    // the column of multipliers and the finished row of W travel by v_readlane -- no LDS round trip and no barrier on
    // the dependency chain (pivot -> rsqrt -> multiplier -> next pivot).  Only the lower triangle of D is referenced.
    // at = -D' (accumulator layout) -> D row-major in the LDS scratch block
    __device__ __forceinline__ void stage_dsc(const v4d& at) const {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sm.dsc[li * DLD + kq + 4 * rg] = -at[rg];
    }
    __device__ __forceinline__ bool cholinv16(const v4d& at, int r0, int c0) const {
        stage_dsc(at);
        __builtin_amdgcn_wave_barrier();
        return cholinv16_dsc(r0, c0);
    }
    __device__ __forceinline__ bool cholinv16_dsc(int r0, int c0) const { return cholinv16_blocked(r0, c0); }
    // 1 / sqrt(x) as the library's rsqrt() computes it for a finite x > 0 -- v_rsq_f64 and one refinement, the same five
    // operations -- without its select that keeps the raw result for 0 / inf / nan (three more dependent vector instructions per
    // pivot: the pivots are checked for > 0 below anyway).  Every vector instruction of this chain counts twice: it is latency
    // on the path everybody waits for, and beside a SIMD partner that streams v_mfma_f64 it gets one issue slot per MFMA
    // (tools/cholinv16_bench.hip: 4592 -> 3836 cycles per call alone, 102.6 k -> 69.2 k beside a saturating partner; same bits).
    static __device__ __forceinline__ double rsq_pivot(double x) {
        const double y0 = __builtin_amdgcn_rsq(x);
        const double e = __builtin_fma(y0 * -x, y0, 1.0);
        return __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
    }
    // The same in four block steps of four pivots on the matrix pipe: forward elimination of [D | I] to [L' | W], the two
    // halves kept as two MFMA accumulators (lane (li, kq) register rg <-> row kq + 4 rg, column li).  Step k reads its 4 x 4
    // diagonal block (ten v_readlane pairs), factors and inverts it in uniform arithmetic (every lane the same values: four
    // dependent rsqrt instead of sixteen pivots with a cross-lane broadcast per update), and then needs
    //     X = Wkk D[block k, :]      (the scaled pivot rows; by symmetry also the panel L[:, block k] transposed)
    //     Y = Wkk W[block k, :]      (rows 4k .. 4k+3 of the result)
    //     D -= L[:, block k] X,  W -= L[:, block k] Y   for the rows below the block
    // -- four v_mfma_f64_16x16x4: register k of an accumulator IS the B operand "rows of block k", and register 0 of X is at
    // once B operand (X) and, negated and masked to the rows below, A operand (the panel), so nothing moves between lanes.
    __device__ __forceinline__ bool cholinv16_blocked(int r0, int c0) const {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const double* D = sm.dsc;
        v4d aA, aW;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int i = kq + 4 * rg;
            aA[rg] = D[(i > li ? i : li) * DLD + (i > li ? li : i)];      // lower triangle mirrored
            aW[rg] = (i == li) ? 1.0 : 0.0;
        }
        double pmin = 1.0, plast = 0.0;          // smallest pivot so far (one v_min per pivot instead of a compare + scalar and)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double tA = aA[k], tW = aW[k];
            // element (a, b) of the diagonal block: row 4k + a, column 4k + b -> lane (li = 4k + b, kq = a)
            auto dg = [&](int a_, int b_) { return bcast_lane(tA, 4 * k + b_ + 16 * a_); };
            const double d00 = dg(0, 0), d10 = dg(1, 0), d20 = dg(2, 0), d30 = dg(3, 0);
            const double d11 = dg(1, 1), d21 = dg(2, 1), d31 = dg(3, 1), d22 = dg(2, 2), d32 = dg(3, 2), d33 = dg(3, 3);
            const double i0 = rsq_pivot(d00);
            const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
            const double p1 = d11 - l10 * l10;
            const double i1 = rsq_pivot(p1);
            const double l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
            const double p2 = d22 - l20 * l20 - l21 * l21;
            const double i2 = rsq_pivot(p2);
            const double l32 = (d32 - l30 * l20 - l31 * l21) * i2;
            const double p3 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
            const double i3 = rsq_pivot(p3);
            // (fmin drops NaNs, so they are caught at the end: a NaN anywhere in the block -- in the input, or bred by a pivot
            // <= 0 -- travels through the multipliers and the panel into every later pivot, the very last one included)
            pmin = fmin(fmin(pmin, fmin(d00, p1)), fmin(p2, p3));
            plast = p3;
            // Wkk = inverse of the 4 x 4 factor
            const double w10 = -(l10 * i0) * i1;
            const double w21 = -(l21 * i1) * i2, w20 = -(l20 * i0 + l21 * w10) * i2;
            const double w32 = -(l32 * i2) * i3, w31 = -(l31 * i1 + l32 * w21) * i3, w30 = -(l30 * i0 + l31 * w10 + l32 * w20) * i3;
            // A operand: lane (li < 4, kq <= li) holds Wkk[li][kq], every other lane zero -- selected column by column (ten
            // selects; the row-by-row form compiled to fourteen and a nest of exec-mask branches)
            const double c0_ = li == 0 ? i0 : li == 1 ? w10 : li == 2 ? w20 : w30;
            const double c1_ = li == 1 ? i1 : li == 2 ? w21 : w31;
            const double c2_ = li == 2 ? i2 : w32;
            double wsel = kq == 0 ? c0_ : kq == 1 ? c1_ : kq == 2 ? c2_ : i3;
            wsel = (li < 4 && kq <= li) ? wsel : 0.0;
            const v4d z4 = (v4d){0, 0, 0, 0};
            const v4d X = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, tA, z4, 0, 0, 0);
            const v4d Y = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, tW, z4, 0, 0, 0);
            // rows 4k .. 4k+3 of the inverse: Y[0] at lane (li, kq) = W[4k + kq][li]
            sm.U[(size_t)(r0 + 4 * k + kq) * PLD + c0 + li] = (li <= 4 * k + kq) ? Y[0] : 0.0;
            if (k < 3) {
                const double pan = (li > 4 * k + 3) ? -X[0] : 0.0;      // -L[li][4k + kq] below the block, nothing above
                aA = __builtin_amdgcn_mfma_f64_16x16x4f64(pan, X[0], aA, 0, 0, 0);
                aW = __builtin_amdgcn_mfma_f64_16x16x4f64(pan, Y[0], aW, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        return pmin > 0.0 && plast == plast;
    }

    // accumulator image of -(S tile (T, Cc))': lane (li, kq) register rg <-> row li, column kq + 4 rg.  In two steps so that
    // the loads can be issued a block column ahead (factor(): the source tiles of column jb + 1 are requested before barrier
    // (B) of column jb and travel while the stores drain): tile_src() = the two 16-byte halves as they lie in memory,
    // tile_image() = negation + diagonal shift.
    struct TileSrc { double2 d0, d1; };
    __device__ __forceinline__ TileSrc tile_src(int T, int Cc, int ntr, int fo) const {
        TileSrc r_;
        r_.d0 = make_double2(0.0, 0.0); r_.d1 = r_.d0;
        if (T >= nch) {
            const double2* tile = reinterpret_cast<const double2*>(Bex + ((size_t)(T - nch) * nchp + Cc) * 256);
            r_.d0 = tile[fo]; r_.d1 = tile[64 + fo];
        } else if (T < ntr) {
            const double2* tile = reinterpret_cast<const double2*>(Ppk + ((size_t)T * nchp + Cc) * 256);
            r_.d0 = tile[fo]; r_.d1 = tile[64 + fo];
        }
        return r_;
    }
    __device__ __forceinline__ v4d tile_image(const TileSrc& r_, int T, int Cc, int li, int kq) const {
        v4d a_ = (v4d){-r_.d0.x, -r_.d0.y, -r_.d1.x, -r_.d1.y};
        if (T == Cc) {
            // diagonal shift; identity beyond n
            const int row = T * 16 + li;
            const double dg = row < n ? sm.dvec[row] : 1.0;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                if (kq + 4 * rg == li) a_[rg] -= dg;
        }
        return a_;
    }
    __device__ __forceinline__ v4d init_tile(int T, int Cc, int ntr, int fo, int li, int kq) const {
        return tile_image(tile_src(T, Cc, ntr, fo), T, Cc, li, kq);
    }

    // -----------------------------------------------------------------------------------------------------
    // fused forward substitution, update of tile row T by block column j0: b_T -= X_T y_j with X_T = (x1 | x2) in
    // its register image (lane (li, kq): row li, columns kq + 4 rg of each 16-column half)
    __device__ __forceinline__ void fwd_update(const v4d& x1, const v4d& x2, int T, int j0, int li, int kq) const {
        double p_ = 0.0;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            p_ += x1[rg] * sm.vec[j0 + kq + 4 * rg] + x2[rg] * sm.vec[j0 + 16 + kq + 4 * rg];
        p_ = swap_add16(p_, p_);                 // p + its neighbour 16 lanes away, in every lane
        p_ = swap_add32(p_, p_);
        if (kq == 0) sm.vec[T * 16 + li] -= p_;
    }

    // s_waitcnt vmcnt(4 k) for a wave-uniform k <= 7: the sweeps leave the k tiles (4 loads each) of the next block in flight
    static __device__ __forceinline__ void vm_wait_tiles(int k) {
        switch (k) {
            case 0: vm_wait<0>(); break;
            case 1: vm_wait<4>(); break;
            case 2: vm_wait<8>(); break;
            case 3: vm_wait<12>(); break;
            case 4: vm_wait<16>(); break;
            case 5: vm_wait<20>(); break;
            case 6: vm_wait<24>(); break;
            default: vm_wait<28>(); break;
        }
    }

    // operand fragments of a tile held in its register image (rg <-> column kq + 4 rg): k-half h = (x[2h], x[2h+1])
    //
    // One loop over the block columns PER ROLE (the barrier sequences of the three loops match: (A), (A2) when the forward
    // substitution is fused, (B)): state carried from one block column to the next -- the source tiles requested a column
    // ahead -- is then live in its own role's loop only and costs the other roles no registers.
    __device__ __forceinline__ bool factor() {
        if constexpr (P64) return factor64();
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ntr = (n + 15) >> 4;           // tile rows that hold valid rows
        if (wv == 1) {
            // prologue: diagonal block of column 0 straight from P
            const int li = lane & 15, kq = lane >> 4, fo = li * 4 + kq;
            v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
            const v4d d11 = init_tile(0, 0, ntr, fo, li, kq);
            const v4d d21 = init_tile(1, 0, ntr, fo, li, kq);
            const v4d d22 = init_tile(1, 1, ntr, fo, li, kq);
            stage_dsc(d11);
            img21[lane] = d21;
            img21[64 + lane] = d22;
        }
        __syncthreads();
        if (wv == 0) return factor_chain();
        if (wv == 1) return factor_lookahead();
        return factor_rows(wv);
    }

    // ======== wavefront 0: factor + invert the diagonal blocks (no global traffic at all) ============================
    __device__ __forceinline__ bool factor_chain() {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int nblk = (n + NB - 1) / NB;
        double* U = sm.U;
        v4d* const img21 = reinterpret_cast<v4d*>(sm.img);        // register images of -D21', -D22' of the next block
        v4d* const img22 = img21 + 64;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            PROF_DECL
            bool ok = cholinv16_dsc(j0, 0);
            PROF(12);
            // L21' = W1 * C21'  (image = -C21', operand -W1)
            const v4d d21 = img21[lane];
            v4d d22 = img22[lane];
            v4d x21 = (v4d){0, 0, 0, 0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                x21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(size_t)(j0 + li) * PLD + 4 * s_ + kq], d21[s_], x21, 0, 0, 0);
            // lane (li, kq) holds L21[li][kq + 4 rg]
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) sm.t21[li * DLD + kq + 4 * rg] = x21[rg];
            __builtin_amdgcn_wave_barrier();
            // -D2' += L21 * L21'
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                d22 = __builtin_amdgcn_mfma_f64_16x16x4f64(sm.t21[li * DLD + 4 * s_ + kq], x21[s_], d22, 0, 0, 0);
            PROF(14);
            ok = cholinv16(d22, j0 + 16, 16) && ok;
            PROF(15);
            if (lane == 0) sm.flag[0] = ok ? 0 : 1;
            __syncthreads();                                    // (A) W1, L21, W2 published
            PROF2(1, 16 + (jb < 31 ? jb : 31));                 // (slots 16..: the wait at (A) by block column)
            if (sm.flag[0]) return false;
            // lower-left block of the inverse for the solves: W21 = -W2 (L21 W1)
            v4d y = (v4d){0, 0, 0, 0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                y = __builtin_amdgcn_mfma_f64_16x16x4f64(sm.t21[li * DLD + 4 * s_ + kq],
                                                         U[(size_t)(j0 + 4 * s_ + kq) * PLD + li], y, 0, 0, 0);
            v4d w21 = (v4d){0, 0, 0, 0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                w21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(size_t)(j0 + 16 + li) * PLD + 16 + 4 * s_ + kq], y[s_], w21, 0, 0, 0);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) U[(size_t)(j0 + 16 + kq + 4 * rg) * PLD + li] = w21[rg];
            if (fwd) {
                // fused forward substitution: y_j = M_j b_j (b_j has received every earlier column's update)
                __builtin_amdgcn_wave_barrier();
                const int r = lane & 31;
                const double* Mr = U + (size_t)(j0 + r) * PLD;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int c = 0; c < NB; c += 4) {
                    s0 += Mr[c] * sm.vec[j0 + c];
                    s1 += Mr[c + 1] * sm.vec[j0 + c + 1];
                    s2 += Mr[c + 2] * sm.vec[j0 + c + 2];
                    s3 += Mr[c + 3] * sm.vec[j0 + c + 3];
                }
                const double yv = (s0 + s1) + (s2 + s3);
                __builtin_amdgcn_wave_barrier();
                if (lane < NB) sm.vec[j0 + lane] = yv;
                lds_barrier();                                  // (A2) y_j published
            }
            PROF(3);
            __syncthreads();                                    // (B) block column visible to everyone
            PROF(4);
        }
        return true;
    }

    // ======== wavefront 1: the two tile rows R2 = tb+2, R3 = tb+3 of the NEXT diagonal block ==========================
    // their panel tiles in this block column plus, one column ahead, the rank-k update of the next diagonal block, which it
    // hands to wavefront 0 through LDS: the sequential diagonal work of column j+1 overlaps everybody else's rank-k update
    // of column j+1.
    __device__ __forceinline__ bool factor_lookahead() {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;              // this lane's double2 inside a 1 KB half tile
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* U = sm.U;
        v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
        v4d* const img22 = img21 + 64;
        // source tiles of the NEXT block column, requested before barrier (B) (tile_src): (R2|R3, tb|tb+1), (R2,R2), (R3,R2), (R3,R3)
        TileSrc pre[7];
        bool have_pre = false;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int tb = j0 >> 4;                 // first tile-row of the block
            const int nc = 2 * jb;                  // finished 16-column chunks
            const int R2 = tb + 2, R3 = tb + 3;
            const bool v2 = R2 < ntr, v3 = R3 < ntr;
            v4d p20, p21, p30, p31, e11, e21, e22;   // panel tiles (R2|R3, 2jb|2jb+1) and the next diagonal block
            if (v2) {
                if (have_pre) {
                    p20 = tile_image(pre[0], R2, tb, li, kq);      p21 = tile_image(pre[1], R2, tb + 1, li, kq);
                    p30 = tile_image(pre[2], R3, tb, li, kq);      p31 = tile_image(pre[3], R3, tb + 1, li, kq);
                    e11 = tile_image(pre[4], R2, R2, li, kq);      e21 = tile_image(pre[5], R3, R2, li, kq);
                    e22 = tile_image(pre[6], R3, R3, li, kq);
                } else {
                    p20 = init_tile(R2, tb, ntr, fo, li, kq);      p21 = init_tile(R2, tb + 1, ntr, fo, li, kq);
                    p30 = init_tile(R3, tb, ntr, fo, li, kq);      p31 = init_tile(R3, tb + 1, ntr, fo, li, kq);
                    e11 = init_tile(R2, R2, ntr, fo, li, kq);      e21 = init_tile(R3, R2, ntr, fo, li, kq);
                    e22 = init_tile(R3, R3, ntr, fo, li, kq);
                }
                if (jb > 0) {
                    // hand-pipelined operand ring (qp_common.hpp: gload16 / vm_wait): half-chunks of 8 columns, the
                    // four operand tiles requested three half-chunks ahead (4 loads per step -> vmcnt(12))
                    const char* q0 = uniform_ptr(tile2(tb, 0));
                    const char* q1 = uniform_ptr(tile2(tb + 1, 0));
                    const char* q2 = uniform_ptr(tile2(R2, 0));
                    const char* q3 = uniform_ptr(tile2(v3 ? R3 : R2, 0));
                    const unsigned voff = (unsigned)fo * 16u;
                    struct Frag { v2d b0, b1, a2, a3; };
                    const int nk2 = 2 * nc, klast = nk2 - 1;
                    auto loadf = [&](Frag& f_, int k2) {
                        const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
                        f_.b0 = gload16(q0 + o, voff); f_.b1 = gload16(q1 + o, voff);
                        f_.a2 = gload16(q2 + o, voff); f_.a3 = gload16(q3 + o, voff);
                    };
                    // (R3 pure padding -- the last block of a matrix whose size is 1..16 past a multiple of 32, e.g. n = 514:
                    // four of the seven tiles are not needed, and in that block column this wavefront is the critical path)
#define HIPDRT_STEP7(B0, B1, A2, A3)                                                                    \
                    p20 = __builtin_amdgcn_mfma_f64_16x16x4f64(B0, A2, p20, 0, 0, 0);               \
                    p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(B1, A2, p21, 0, 0, 0);               \
                    e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, A2, e11, 0, 0, 0);               \
                    if (v3) {                                                                       \
                        p30 = __builtin_amdgcn_mfma_f64_16x16x4f64(B0, A3, p30, 0, 0, 0);           \
                        p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(B1, A3, p31, 0, 0, 0);           \
                        e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, A3, e21, 0, 0, 0);           \
                        e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, A3, e22, 0, 0, 0);           \
                    }
                    auto multf = [&](const Frag& f_) {
                        HIPDRT_STEP7(f_.b0.x, f_.b1.x, f_.a2.x, f_.a3.x)
                        HIPDRT_STEP7(f_.b0.y, f_.b1.y, f_.a2.y, f_.a3.y)
                        __builtin_amdgcn_sched_barrier(0);
                    };
#undef HIPDRT_STEP7
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile loads above: from here on the count is ours
                    __builtin_amdgcn_sched_barrier(0);
                    Frag f0, f1, f2, f3;
                    loadf(f0, 0); loadf(f1, 1); loadf(f2, 2);
                    for (int k2 = 0; k2 < nk2; k2 += 4) {           // nk2 = 4 jb: a multiple of 4
                        loadf(f3, k2 + 3); vm_wait<12>(); multf(f0);
                        loadf(f0, k2 + 4); vm_wait<12>(); multf(f1);
                        loadf(f1, k2 + 5); vm_wait<12>(); multf(f2);
                        loadf(f2, k2 + 6); vm_wait<12>(); multf(f3);
                    }
                    vm_wait<0>();
                    if (!v3) {
                        // R3 is pure padding (its operand was a stand-in): no panel tiles, identity diagonal
                        p30 = (v4d){0, 0, 0, 0}; p31 = (v4d){0, 0, 0, 0}; e21 = (v4d){0, 0, 0, 0};
                        e22 = init_tile(R3, R3, ntr, fo, li, kq);
                    }
                }
            }
            __syncthreads();                                    // (A)
            if (sm.flag[0]) return false;
            have_pre = false;
            if (v2) {
                double wn1[4], l21[4], wn2[4];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    wn1[s_] = -U[(size_t)(j0 + li) * PLD + 4 * s_ + kq];
                    l21[s_] = sm.t21[li * DLD + 4 * s_ + kq];
                    wn2[s_] = -U[(size_t)(j0 + 16 + li) * PLD + 16 + 4 * s_ + kq];
                }
                v4d x20 = (v4d){0, 0, 0, 0}, x30 = x20, x21_ = x20, x31 = x20;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    x20 = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], p20[s_], x20, 0, 0, 0);
                    x30 = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], p30[s_], x30, 0, 0, 0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x20[s_], p21, 0, 0, 0);
                    p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x30[s_], p31, 0, 0, 0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    x21_ = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], p21[s_], x21_, 0, 0, 0);
                    x31 = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], p31[s_], x31, 0, 0, 0);
                }
                {
                    double2* d0 = const_cast<double2*>(tile2(R2, 2 * jb)) + fo;
                    d0[0] = make_double2(x20[0], x20[1]);   d0[64] = make_double2(x20[2], x20[3]);
                    d0[128] = make_double2(x21_[0], x21_[1]); d0[192] = make_double2(x21_[2], x21_[3]);
                }
                if (v3) {
                    double2* d0 = const_cast<double2*>(tile2(R3, 2 * jb)) + fo;
                    d0[0] = make_double2(x30[0], x30[1]);   d0[64] = make_double2(x30[2], x30[3]);
                    d0[128] = make_double2(x31[0], x31[1]); d0[192] = make_double2(x31[2], x31[3]);
                }
                // next column's source tiles (rows tb+4, tb+5): in flight while the stores above drain
                if (R2 + 2 < ntr) {
                    const int N2 = R2 + 2, N3 = R3 + 2;
                    pre[0] = tile_src(N2, R2, ntr, fo); pre[1] = tile_src(N2, R3, ntr, fo);
                    pre[2] = tile_src(N3, R2, ntr, fo); pre[3] = tile_src(N3, R3, ntr, fo);
                    pre[4] = tile_src(N2, N2, ntr, fo); pre[5] = tile_src(N3, N2, ntr, fo);
                    pre[6] = tile_src(N3, N3, ntr, fo);
                    have_pre = true;
                }
                // the two chunks just produced complete the next diagonal block: a tile's register image is its own
                // operand fragment (register s <-> k-step s)
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x20[s_], x20[s_], e11, 0, 0, 0);
                    e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(x20[s_], x30[s_], e21, 0, 0, 0);
                    e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(x30[s_], x30[s_], e22, 0, 0, 0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21_[s_], x21_[s_], e11, 0, 0, 0);
                    e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21_[s_], x31[s_], e21, 0, 0, 0);
                    e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(x31[s_], x31[s_], e22, 0, 0, 0);
                }
                stage_dsc(e11);
                img21[lane] = e21;
                img22[lane] = e22;
                if (fwd) {
                    lds_barrier();                              // (A2)
                    fwd_update(x20, x21_, R2, j0, li, kq);
                    if (v3) fwd_update(x30, x31, R3, j0, li, kq);
                }
            } else if (fwd) {
                lds_barrier();                                  // (A2)
            }
            __syncthreads();                                    // (B) block column visible to everyone
        }
        return true;
    }

    // ======== wavefronts 2..7: the rows below ==========================================================================
    // rows of pass 0 of block column jb for this wavefront (balanced schedule: the first RMAXT set bits of its mask)
    __device__ __forceinline__ void first_rows(int jb, int wv, int lane, int ntr, int (&T)[RMAXT], bool (&act)[RMAXT]) const {
        constexpr int OW = RNW - 2;
        const int tb = 2 * jb;
        const int nsq = ntr - (tb + 4) > 0 ? ntr - (tb + 4) : 0;
        const int nothers = nsq + nex;
        if (balanced) {
            constexpr int SROW = Smem::SROW;
            const unsigned char* row = sm.sched + jb * SROW;
            unsigned long long m0 = __ballot(lane < nsq && row[lane] == wv), m1 = 0;
            if (SROW > 64) m1 = __ballot(lane + 64 < nsq && row[lane + 64] == wv);
#pragma unroll
            for (int u = 0; u < RMAXT; ++u) {
                int r = -1;
                if (m0) { r = __builtin_ctzll(m0); m0 &= m0 - 1; }
                else if (m1) { r = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
                T[u] = r >= 0 ? tb + 4 + r : nch;
                act[u] = r >= 0;
            }
        } else {
#pragma unroll
            for (int u = 0; u < RMAXT; ++u) {
                const int slot = (wv - 2) + u * OW;
                T[u] = slot < nsq ? tb + 4 + slot : nch + (slot - nsq);
                act[u] = slot < nothers;
            }
        }
    }

    __device__ __forceinline__ bool factor_rows(int wv) {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;              // this lane's double2 inside a 1 KB half tile
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* U = sm.U;
        constexpr int OW = RNW - 2;                       // wavefronts in this role
        // source tiles of pass 0 of the NEXT block column, requested before barrier (B) (tile_src)
        TileSrc pre[RMAXT][2];
        bool have_pre = false;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int tb = j0 >> 4;                 // first tile-row of the block
            const int nc = 2 * jb;                  // finished 16-column chunks
            const int nsq = ntr - (tb + 4) > 0 ? ntr - (tb + 4) : 0;   // rows of the square matrix below R3
            const int nothers = nsq + nex;                               // ... followed by the appended rows
            const bool two = (tb + 1) < ntr;                             // second tile column is not pure padding
            int npass = nothers > OW * RMAXT ? (nothers + OW * RMAXT - 1) / (OW * RMAXT) : 1;
            // balanced schedule (build_schedule): bit r of (m0, m1) = tile row tb + 4 + r is this wavefront's
            unsigned long long m0 = 0, m1 = 0;
            if (balanced) {
                constexpr int SROW = Smem::SROW;
                const unsigned char* row = sm.sched + jb * SROW;
                m0 = __ballot(lane < nsq && row[lane] == wv);
                if (SROW > 64) m1 = __ballot(lane + 64 < nsq && row[lane + 64] == wv);
                const int mine = __builtin_popcountll(m0) + __builtin_popcountll(m1);
                npass = mine > RMAXT ? (mine + RMAXT - 1) / RMAXT : 1;
            }
#pragma unroll 1
            for (int ps = 0; ps < npass; ++ps) {
                int T[RMAXT];
                bool act[RMAXT];
                if (balanced) {
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        int r = -1;
                        if (m0) { r = __builtin_ctzll(m0); m0 &= m0 - 1; }
                        else if (m1) { r = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
                        T[u] = r >= 0 ? tb + 4 + r : nch;
                        act[u] = r >= 0;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        const int slot = (wv - 2) + u * OW + ps * OW * RMAXT;
                        T[u] = slot < nsq ? tb + 4 + slot : nch + (slot - nsq);
                        act[u] = slot < nothers;
                    }
                }
                // ---- (1) accT = -(S' tile) + sum_c L(Cc, c) L(T, c)' --------------------------------------
                v4d acc[RMAXT][2];
                if (ps == 0 && have_pre) {
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[u][ct] = act[u] ? tile_image(pre[u][ct], T[u], tb + ct, li, kq) : (v4d){0, 0, 0, 0};
                } else {
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[u][ct] = act[u] ? init_tile(T[u], tb + ct, ntr, fo, li, kq) : (v4d){0, 0, 0, 0};
                }
                if (jb > 0 && act[0]) {
                    // Hand-pipelined operand ring (qp_common.hpp: gload16 / vm_wait), half-chunks of 8 columns: the A
                    // tiles (this wavefront's own rows, from HBM) are requested THREE half-chunks ahead, the B tiles
                    // (the block's two tile rows, shared by all wavefronts: L1 / L2) one ahead.  Per step 2 B + 4 A loads,
                    // B first, so "B of this step has arrived" is vmcnt(2 RMAXT + 2): A(k+2), B(k+1), A(k+3) may still be in
                    // flight.  Indices past the end are clamped (a redundant load) so that the counts stay uniform.
                    const char* rb0 = uniform_ptr(tile2(tb, 0));
                    const char* rb1 = uniform_ptr(tile2(two ? tb + 1 : tb, 0));      // stand-in when the row is padding
                    const char* ra[RMAXT];
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) ra[u] = uniform_ptr(tile2(act[u] ? T[u] : tb, 0));
                    const unsigned voff = (unsigned)fo * 16u;
                    struct SlA { v2d a[RMAXT]; };
                    struct SlB { v2d b0, b1; };
                    const int nk2 = 2 * nc, klast = nk2 - 1;
                    auto loadA = [&](SlA& s_, int k2) {
                        const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) s_.a[u] = gload16(ra[u] + o, voff);
                    };
                    auto loadB = [&](SlB& s_, int k2) {
                        const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
                        s_.b0 = gload16(rb0 + o, voff); s_.b1 = gload16(rb1 + o, voff);
                    };
                    auto mult = [&](const SlA& a_, const SlB& b_) {
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) {
                            if (act[u]) {
                                acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b0.x, a_.a[u].x, acc[u][0], 0, 0, 0);
                                if (two) acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b1.x, a_.a[u].x, acc[u][1], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) {
                            if (act[u]) {
                                acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b0.y, a_.a[u].y, acc[u][0], 0, 0, 0);
                                if (two) acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b1.y, a_.a[u].y, acc[u][1], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    SlA a0, a1, a2, a3;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile loads above: from here on the count is ours
                    __builtin_amdgcn_sched_barrier(0);
                    SlB b0, b1;
                    loadB(b0, 0); loadA(a0, 0); loadA(a1, 1); loadA(a2, 2);
                    for (int k2 = 0; k2 < nk2; k2 += 4) {       // nk2 = 4 jb: a multiple of 4
                        loadB(b1, k2 + 1); loadA(a3, k2 + 3); vm_wait<2 * RMAXT + 2>(); mult(a0, b0);
                        loadB(b0, k2 + 2); loadA(a0, k2 + 4); vm_wait<2 * RMAXT + 2>(); mult(a1, b1);
                        loadB(b1, k2 + 3); loadA(a1, k2 + 5); vm_wait<2 * RMAXT + 2>(); mult(a2, b0);
                        loadB(b0, k2 + 4); loadA(a2, k2 + 6); vm_wait<2 * RMAXT + 2>(); mult(a3, b1);
                    }
                    vm_wait<0>();
                }
                if (ps == 0) {
                    __syncthreads();                            // (A) W1, L21, W2 published by wavefront 0
                    if (sm.flag[0]) return false;
                    have_pre = false;
                }
                // ---- (3) X1' = W1 C1', C2' -= L21 X1', X2' = W2 C2' on the matrix pipe, from registers -----
                if (act[0]) {
                    double wn1[4], l21[4], wn2[4];
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        wn1[s_] = -U[(size_t)(j0 + li) * PLD + 4 * s_ + kq];
                        l21[s_] = sm.t21[li * DLD + 4 * s_ + kq];
                        wn2[s_] = -U[(size_t)(j0 + 16 + li) * PLD + 16 + 4 * s_ + kq];
                    }
                    v4d x1[RMAXT], x2[RMAXT];
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) { x1[u] = (v4d){0, 0, 0, 0}; x2[u] = (v4d){0, 0, 0, 0}; }
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u)
                            x1[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], acc[u][0][s_], x1[u], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u)
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x1[u][s_], acc[u][1], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u)
                            x2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], acc[u][1][s_], x2[u], 0, 0, 0);
                    // ---- (4) tiles straight from registers: two contiguous 1 KB stores per tile ------------
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u) {
                        if (act[u]) {
                            double2* d0 = const_cast<double2*>(tile2(T[u], 2 * jb)) + fo;
                            d0[0] = make_double2(x1[u][0], x1[u][1]);
                            d0[64] = make_double2(x1[u][2], x1[u][3]);
                            d0[128] = make_double2(x2[u][0], x2[u][1]);
                            d0[192] = make_double2(x2[u][2], x2[u][3]);
                        }
                    }
                    if (ps == npass - 1 && jb + 1 < nblk) {
                        // the source tiles of the next block column's first pass: in flight while the stores above drain
                        int Tn[RMAXT];
                        bool an[RMAXT];
                        first_rows(jb + 1, wv, lane, ntr, Tn, an);
                        if (an[0]) {
#pragma unroll
                            for (int u = 0; u < RMAXT; ++u)
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct)
                                    if (an[u]) pre[u][ct] = tile_src(Tn[u], tb + 2 + ct, ntr, fo);
                            have_pre = true;
                        }
                    }
                    if (fwd) {
                        if (ps == 0) lds_barrier();             // (A2) y_j published by wavefront 0
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u)
                            if (act[u] && T[u] < nch) fwd_update(x1[u], x2[u], T[u], j0, li, kq);
                    }
                } else if (fwd && ps == 0) {
                    lds_barrier();                              // (A2)
                }
            }
            __syncthreads();                                        // (B) block column visible to everyone
        }
        return true;
    }

    // =====================================================================================================================
    // factor64: the same factorisation with the finished part of L walked ONCE PER 64 COLUMNS (P64).
    //
    // The left-looking re-read of L is half of this kernel's HBM traffic (section 6 of DESIGN.md): a block column of 32 reads
    // every tile row below it over all finished columns.  Here a row wavefront accumulates FOUR tile columns (two 32-blocks a
    // and b = a "super column") per pass over the history, 3 tile rows x 4 tile columns of accumulators, so every A tile
    // is fetched once per 64 columns.  What stays 32 wide is everything sequential: the diagonal chain, the inverse blocks
    // in U, the triangular sweeps.  Per super column J (tile rows tA = 4J .. tA+3 hold its two diagonal blocks):
    //   wavefront 0   chain a (diagonal block a was completed by wavefront 1 in super column J-1), y_a of the fused forward
    //                 substitution, then itself the look-ahead of block b: tile rows tA+2, tA+3 -- four panel tiles in
    //                 columns a and the three tiles of diagonal block b -- over the history, their panel solve with its own
    //                 W_a, the rank-32 update of diagonal block b from registers, chain b, y_b.  All of it runs beside the
    //                 other wavefronts' history pass; nothing is handed over inside it.
    //   wavefront 1   tile rows tA+4, tA+5 (the diagonal rows of block a of super column J+1): eight panel tiles in columns
    //                 a and b plus the three tiles of the next diagonal block a', over the history.
    //   wavefronts 2..7  all tile rows from tA+6 on, dealt by build_schedule64.
    //   barrier (A):  history accumulated everywhere, W / L21 of BOTH blocks, y_a, y_b and the solved tiles of rows tA+2,
    //                 tA+3 published.  Then every wavefront solves its panel tiles against block a, applies block a's rank-32
    //                 update to its column-b tiles FROM REGISTERS (the solved tile is its own operand fragment; the other
    //                 operand, rows tA+2 / tA+3 in columns a, is read back from L: 8 KB through L1 / L2), solves against
    //                 block b, stores, and updates the right-hand side; wavefront 1 also completes diagonal block a' and
    //                 stages it for the chain.
    //   barrier (B):  the super column's tiles are visible to everyone.
    //   What of the two chains nobody needs before (A) -- the lower-left blocks W21 of the inverse diagonal blocks (used by the
    //   triangular sweeps only) and y_a, y_b of the fused forward substitution (used by the right-hand-side updates at the end
    //   of the panel solves) -- wavefront 0 computes behind (A), where it has nothing else to do (finish_later); the others'
    //   updates of the right-hand side wait for an LDS word that is set long before they get there.
    //   Behind that, still before (B), wavefront 0 already STARTS super column J+1's look-ahead (tile rows tA+6, tA+7): source
    //   tiles from P, then the part of their history that is final (columns < 4 J; this super column's own 64 columns are
    //   being solved by the others right now).  The other seven wavefronts count themselves into the LDS word flag[3] before
    //   they enter (B); wavefront 0 polls it between ring steps and takes (B) as soon as all seven stand there, finishing the
    //   old range afterwards -- so nobody ever waits at (B) for the early work, and the next super column's look-ahead has only
    //   the 64 new columns left (8.375 -> 8.34 ms per launch; profiles/r04o_*).
    // Other divisions of the sequential work were built and measured (all bit-identical, tools/experiments/
    // qp_factor64_roles_v3.hpp, profiles/r04e_*, r04g_*): chain b on wavefront 1 behind a second barrier with wavefront 0 taking the
    // heavier look-ahead rows (9.30 ms per launch), and wavefront 0 with the chains + the next diagonal block only, rows tA+4,
    // tA+5 as ordinary rows (9.07 ms) -- against 8.63 ms for this form: a third barrier per super column costs more than the
    // shorter chain path returns, and the two-row look-ahead wavefronts are what the busiest SIMD carries either way.
    // Two barriers per 64 columns instead of six.  Every tile receives exactly the MFMA sequence it receives in factor()
    // (history chunks ascending, x then y half of every half-chunk, the same operand order), so the factor, U and the forward-
    // substituted right-hand side are bit for bit those of the 32-column form (tools/dump_fit.py --cmp).
    // kFat: the four-wavefront form (one wavefront per SIMD, 512 registers each: 32 accumulator tiles in AccVGPRs).  Wavefront 0
    // is the chain + look-ahead of block b exactly as with eight wavefronts -- alone on its SIMD, so no row wavefront's MFMA
    // stream stretches the chains --, wavefront 1 carries tile rows tA+4, tA+5 (11 tiles) plus up to RM1 ordinary rows,
    // wavefronts 2 and 3 up to RM = 6 rows x 4 columns per pass: from super column 3 on every row is accumulated in ONE pass
    // (27 / 23 / 19 / 15 tile rows below the look-ahead rows in super columns 0 / 1 / 2 / 3 at n = 514).
    static constexpr bool kFat = P64 && RNW == 4;
    // (fat form: 6 rows x 4 columns = 24 of a wavefront's 32 accumulator tiles; the panel solves need two more tiles per row of
    // a group -- the solved tiles x1, x2 are MFMA results too -- and with 7 rows + groups of 4 = 32 tiles exactly hipcc kept ONE
    // tile for all of them and moved every intermediate result through VGPRs around every MFMA, 19 wait states each time:
    // panel solves 2.5 x slower than the matrix pipe, profiles/r06_fat_*)
    static constexpr int RM = kFat ? 6 : 3;      // tile rows per row wavefront and pass (x 4 tile columns of accumulator tiles)
    static constexpr int RM1 = 3;                // fat form: ordinary rows of wavefront 1, next to its 11 look-ahead tiles (one pass)
    static constexpr int PG = 3;                 // panel solves in groups of at most PG rows
    static constexpr int NARR = RNW - 1;         // wavefronts that count themselves into flag[3] per super column (all but wavefront 0)
    __device__ __forceinline__ void build_schedule64() {
        const int ntr = (n + 15) >> 4, nblk = (n + NB - 1) / NB, nsup = (nblk + 1) >> 1;
        constexpr int SROW = Smem::SROW;
        for (int J = threadIdx.x; J < nsup; J += RT) {
            const int tA = 4 * J, nk2 = 8 * J;
            const int nsq = ntr - (tA + 6) > 0 ? ntr - (tA + 6) : 0;
            const int c = 8 * nk2 + 40;                               // MFMAs of one row: history + two panel solves + block a's update
            if constexpr (kFat) {
                // one wavefront per SIMD: wavefront 1 starts with its look-ahead tiles' MFMAs and takes at most RM1 rows,
                // wavefronts 2 and 3 share the rest (as many passes of RM rows as that takes)
                const int rest = nsq > RM1 ? nsq - RM1 : 0;
                const int cap = RM * (rest > 2 * RM ? (rest + 2 * RM - 1) / (2 * RM) : 1);
                int l1 = (tA + 4 < ntr) ? kLa2Load * nk2 + 96 : 0, l2 = 0, l3 = 0;
                int c1 = 0, c2 = 0, c3 = 0;
                for (int r = 0; r < nsq; ++r) {
                    int best = -1, bl = 0x7fffffff;
                    if (J & 1) {                                      // (ties: the odd row of a super column goes to 2 and 3 in turn)
                        if (c3 < cap && l3 < bl) { best = 3; bl = l3; }
                        if (c2 < cap && l2 < bl) { best = 2; bl = l2; }
                    } else {
                        if (c2 < cap && l2 < bl) { best = 2; bl = l2; }
                        if (c3 < cap && l3 < bl) { best = 3; bl = l3; }
                    }
                    if (c1 < RM1 && l1 < bl) { best = 1; bl = l1; }
                    if (best < 0) best = c2 <= c3 ? 2 : 3;            // (cannot happen: the caps cover nsq)
                    sm.sched[J * SROW + r] = (unsigned char)best;
                    if (best == 1) { ++c1; l1 += c; }
                    else if (best == 2) { ++c2; l2 += c; }
                    else { ++c3; l3 += c; }
                }
                continue;
            }
            const int cap = RM * (nsq > 6 * RM ? (nsq + 6 * RM - 1) / (6 * RM) : 1);
            // SIMD 0 carries wavefront 0 (two chains + 14 MFMAs per half-chunk), SIMD 1 wavefront 1 (22 per half-chunk)
            int l0 = (tA + 2 < ntr) ? kLa1Load * nk2 + 48 : 0;
            int l1 = (tA + 4 < ntr) ? kLa2Load * nk2 + 96 : 0, l2 = 0, l3 = 0;
            int c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
            for (int r = 0; r < nsq; ++r) {
                const int w2 = c2 <= c6 ? 2 : 6, n2 = c2 <= c6 ? c2 : c6;
                const int w3 = c3 <= c7 ? 3 : 7, n3 = c3 <= c7 ? c3 : c7;
                int best = -1, bl = 0x7fffffff;
                if (c4 < cap && l0 < bl) { best = 4; bl = l0; }
                if (n2 < cap && l2 < bl) { best = w2; bl = l2; }
                if (n3 < cap && l3 < bl) { best = w3; bl = l3; }
                if (c5 < cap && l1 < bl) { best = 5; bl = l1; }
                sm.sched[J * SROW + r] = (unsigned char)best;
                if (best == 4) { ++c4; l0 += c; }
                else if (best == 5) { ++c5; l1 += c; }
                else if (best == 2) { ++c2; l2 += c; }
                else if (best == 6) { ++c6; l2 += c; }
                else if (best == 3) { ++c3; l3 += c; }
                else { ++c7; l3 += c; }
            }
        }
    }

    // operand fragments (k-steps 0..3) of L21 of block which (0 = a, 1 = b): register image of the chain's x21, lane for lane
    __device__ __forceinline__ void load_l21(double (&l21)[4], int which, int lane) const {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) l21[s_] = sm.t21[which * 256 + s_ * 64 + lane];
    }
    // -W1 and -W2 of the 32-block at row j0 as A operands
    __device__ __forceinline__ void load_wn(double (&wn1)[4], double (&wn2)[4], int j0, int li, int kq) const {
        const double* U = sm.U;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            wn1[s_] = -U[(size_t)(j0 + li) * PLD + 4 * s_ + kq];
            wn2[s_] = -U[(size_t)(j0 + 16 + li) * PLD + 16 + 4 * s_ + kq];
        }
    }
    // y of the fused forward substitution is in vec: flag[2] >= 2 J + 1 for block a, 2 J + 2 for block b (wavefront 0, behind (A))
    __device__ __forceinline__ void wait_y(int word) const {
        int spins = 0;
        while (lds_peek32(&sm.flag[2]) < word) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) __builtin_trap();           // (a protocol error must not hang the device)
        }
    }
    // Barrier (B) in two halves (HIPDRT_QP_SPLITB): a wavefront that has stored its tiles of super column J counts itself into
    // flag[3] (release) and goes straight on to super column J + 1 -- source tiles, then the part of the history that is final
    // already -- and waits for the other six (seven for wavefront 0, which stores nothing behind (A)) only in front of its
    // first load from the 64 columns just solved.  No s_barrier: who finishes its panel solves early is not held up by who
    // finishes late, and the phases of the wavefronts on one SIMD drift apart (one in its MFMA-bound history pass, the other
    // in its latency-bound panel solves) instead of coinciding.
    // (A) of the fat form is not a barrier either: the row wavefronts need wavefront 0's W / L21 (LDS) and solved look-ahead tiles
    // (global memory), nothing from each other, and wavefront 0 needs nothing from anybody -- it publishes super column J in
    // flag[1] (release, as arrive_b) and goes on to W21 / y and the next look-ahead while the rows are still in their history
    __device__ __forceinline__ void publish_a(int J_, int lane) {
        if (lane == 0) __hip_atomic_store(&sm.flag[1], J_ + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void wait_a(int J_) const {
        int spins = 0;
        while (lds_peek32(&sm.flag[1]) < J_ + 1) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) __builtin_trap();
        }
    }
    __device__ __forceinline__ void arrive_b(int lane) {
        if (lane == 0) __hip_atomic_fetch_add(&sm.flag[3], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void wait_b(int J_) const {        // everybody's tiles of super column J_ are stored
        int spins = 0;
        while (lds_peek32(&sm.flag[3]) < NARR * (J_ + 1)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) __builtin_trap();           // (a protocol error must not hang the device)
        }
    }
    // the four solved tiles of tile rows tA+2, tA+3 in columns a as B operands: bf[row][chunk][half]
    struct BFrag { v2d f[2][2][2]; };
    __device__ __forceinline__ void load_bfrag(BFrag& b_, int tA, bool v3, int fo) const {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double2* p = tile2((r == 1 && !v3) ? tA + 2 : tA + 2 + r, tA + k) + fo;       // (stand-in when row tA+3 is padding)
                const double2 d0 = p[0], d1 = p[64];
                b_.f[r][k][0] = (v2d){d0.x, d0.y};
                b_.f[r][k][1] = (v2d){d1.x, d1.y};
            }
    }
    // acc += (tile row tA+2+r, columns a) * xs'   with xs = (xa | xb) the solved tiles of one tile row in columns a: the eight
    // k-steps of that product one at a time (st = 0 .. 7: chunk tA steps 0..3, chunk tA+1 steps 0..3), so that a caller with
    // several accumulators rotates over them -- eight MFMAs in a row on ONE accumulator wait for each other (~100+ cycles each
    // instead of 64); per accumulator the order of the steps, hence the sum, is unchanged
    static __device__ __forceinline__ void upd_b_step(v4d& acc, const BFrag& b_, int r, const v4d& xa, const v4d& xb, int st) {
        const int k = st >> 2, h = (st >> 1) & 1;
        const double bo = (st & 1) ? b_.f[r][k][h].y : b_.f[r][k][h].x;
        const double xo = k ? xb[st & 3] : xa[st & 3];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bo, xo, acc, 0, 0, 0);
    }
    __device__ __forceinline__ void store_pair(int T, int c, const v4d& x1, const v4d& x2, int fo) const {
        double2* d0 = const_cast<double2*>(tile2(T, c)) + fo;
        d0[0] = make_double2(x1[0], x1[1]);   d0[64] = make_double2(x1[2], x1[3]);
        d0[128] = make_double2(x2[0], x2[1]); d0[192] = make_double2(x2[2], x2[3]);
    }
    // lower-left block of the inverse, W21 = -W2 (L21 W1), and y_j = M_j b_j of the fused forward substitution (wavefront 0;
    // x21 = this lane's registers of L21)
    __device__ __forceinline__ void finish_block(const v4d& x21, int j0, int lane, int li, int kq) {
        double* U = sm.U;
        v4d y = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(x21[s_], U[(size_t)(j0 + 4 * s_ + kq) * PLD + li], y, 0, 0, 0);
        v4d w21 = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
            w21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(size_t)(j0 + 16 + li) * PLD + 16 + 4 * s_ + kq], y[s_], w21, 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) U[(size_t)(j0 + 16 + kq + 4 * rg) * PLD + li] = w21[rg];
        __builtin_amdgcn_wave_barrier();
        const int r = lane & 31;
        const double* Mr = U + (size_t)(j0 + r) * PLD;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int c = 0; c < NB; c += 4) {
            s0 += Mr[c] * sm.vec[j0 + c];
            s1 += Mr[c + 1] * sm.vec[j0 + c + 1];
            s2 += Mr[c + 2] * sm.vec[j0 + c + 2];
            s3 += Mr[c + 3] * sm.vec[j0 + c + 3];
        }
        const double yv = (s0 + s1) + (s2 + s3);
        __builtin_amdgcn_wave_barrier();
        if (lane < NB) sm.vec[j0 + lane] = yv;
        __builtin_amdgcn_wave_barrier();
    }

    __device__ __forceinline__ bool factor64() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ntr = (n + 15) >> 4;
        if (wv == 1) {
            // prologue: diagonal block of column 0 straight from P
            const int li = lane & 15, kq = lane >> 4, fo = li * 4 + kq;
            v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
            const v4d d11 = init_tile(0, 0, ntr, fo, li, kq);
            const v4d d21 = init_tile(1, 0, ntr, fo, li, kq);
            const v4d d22 = init_tile(1, 1, ntr, fo, li, kq);
            stage_dsc(d11);
            img21[lane] = d21;
            img21[64 + lane] = d22;
            if (lane == 0) { sm.flag[1] = 0; sm.flag[2] = 0; sm.flag[3] = 0; TL_START(); }
        }
        __syncthreads();
        // (no barrier (B) behind the last super column either: everybody's tiles are stored before anybody starts a sweep)
        const bool ok = wv == 0 ? f64_chain() : (wv == 1 ? f64_look2() : f64_rows(wv));
        __syncthreads();
        return ok;
    }

    // ======== wavefront 0: both chains of a super column and, between them, the look-ahead of block b ====================
    __device__ __forceinline__ bool f64_chain() {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;
        const int nblk = (n + NB - 1) / NB, nsup = (nblk + 1) >> 1;
        const int ntr = (n + 15) >> 4;
        double* U = sm.U;
        v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
        v4d* const img22 = img21 + 64;
        // The look-ahead accumulators live across super columns: behind barrier (A) this wavefront has nothing to do but W21 / y,
        // so it starts the NEXT super column's look-ahead there -- its tiles from P and their history over everything that is
        // final already (columns < 4 J: the "old range") -- and only the 64 columns being solved right now are left for after
        // chain a'.  (This wavefront's chain a -> look-ahead history -> solve -> chain b is what the others wait for at (A).)
        v4d p20 = (v4d){0, 0, 0, 0}, p21 = p20, p30 = p20, p31 = p20, e11 = p20, e21 = p20, e22 = p20;
        bool la_ready = false;                   // the accumulators hold super column J's tiles already (+ la_done half-chunks of history)
        int la_done = 0;
        const unsigned voff = (unsigned)fo * 16u;
        struct Frag { v2d b0, b1, a2, a3; };
        // rank-k update of the seven tiles of super column Jc over half-chunks [k_lo, k_hi) (multiples of 4).  With `poll`, stops
        // in front of the first group of four whose start finds the others waiting at barrier (B) (flag[3] >= arrived);
        // returns the first half-chunk not accumulated.
        auto la1_ring = [&](int Jc, int k_lo, int k_hi, bool poll, int arrived) -> int {
            const int tAc = 4 * Jc;
            const bool v3c = tAc + 3 < ntr;
            const char* q0 = uniform_ptr(tile2(tAc, 0));
            const char* q1 = uniform_ptr(tile2(tAc + 1, 0));
            const char* q2 = uniform_ptr(tile2(tAc + 2, 0));
            const char* q3 = uniform_ptr(tile2(v3c ? tAc + 3 : tAc + 2, 0));
            const int klast = k_hi - 1;
            auto loadf = [&](Frag& f_, int k2) {
                const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
                f_.b0 = gload16(q0 + o, voff); f_.b1 = gload16(q1 + o, voff);
                f_.a2 = gload16(q2 + o, voff); f_.a3 = gload16(q3 + o, voff);
            };
#define HIPDRT_STEP7(B0, B1, A2, A3)                                                                    \
            p20 = __builtin_amdgcn_mfma_f64_16x16x4f64(B0, A2, p20, 0, 0, 0);                       \
            p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(B1, A2, p21, 0, 0, 0);                       \
            e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, A2, e11, 0, 0, 0);                       \
            if (v3c) {                                                                              \
                p30 = __builtin_amdgcn_mfma_f64_16x16x4f64(B0, A3, p30, 0, 0, 0);                   \
                p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(B1, A3, p31, 0, 0, 0);                   \
                e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, A3, e21, 0, 0, 0);                   \
                e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, A3, e22, 0, 0, 0);                   \
            }
            auto multf = [&](const Frag& f_) {
                HIPDRT_STEP7(f_.b0.x, f_.b1.x, f_.a2.x, f_.a3.x)
                HIPDRT_STEP7(f_.b0.y, f_.b1.y, f_.a2.y, f_.a3.y)
                __builtin_amdgcn_sched_barrier(0);
            };
#undef HIPDRT_STEP7
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // whatever was in flight: from here on the count is ours
            __builtin_amdgcn_sched_barrier(0);
            Frag f0, f1, f2, f3;
            loadf(f0, k_lo); loadf(f1, k_lo + 1); loadf(f2, k_lo + 2);
            int k2 = k_lo;
            for (; k2 < k_hi; k2 += 4) {
                if (poll && lds_peek32(&sm.flag[3]) >= arrived) break;
                loadf(f3, k2 + 3); vm_wait<12>(); multf(f0);
                loadf(f0, k2 + 4); vm_wait<12>(); multf(f1);
                loadf(f1, k2 + 5); vm_wait<12>(); multf(f2);
                loadf(f2, k2 + 6); vm_wait<12>(); multf(f3);
            }
            vm_wait<0>();
            return k2;
        };
        for (int J = 0; J < nsup; ++J) {
            const int j0a = J * 64, j0b = j0a + NB, tA = 4 * J;
            const bool hasb = (2 * J + 1) < nblk;
            const int R2 = tA + 2, R3 = tA + 3;
            const bool v3 = R3 < ntr;
            PROF_DECL
#ifdef HIPDRT_QP_PROFILE
            const unsigned long long _jt0 = __builtin_amdgcn_s_memtime();      // slots 26 + J: the whole super column, by J
#endif
            TL(0, J, 0);
            // ---- chain a ------------------------------------------------------------------------------------------
            bool ok = cholinv16_dsc(j0a, 0);
            PROF(12);
            {
                const v4d d21 = img21[lane];
                v4d d22 = img22[lane];
                v4d x21 = (v4d){0, 0, 0, 0};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    x21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(size_t)(j0a + li) * PLD + 4 * s_ + kq], d21[s_], x21, 0, 0, 0);
                // lane (li, kq) register rg = L21[li][kq + 4 rg]: as an A operand of k-step s it is its own fragment
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) sm.t21[rg * 64 + lane] = x21[rg];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    d22 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21[s_], x21[s_], d22, 0, 0, 0);
                PROF(14);
                ok = cholinv16(d22, j0a + 16, 16) && ok;
                PROF(15);
                TL(0, J, 4);
                // (W21 of the inverse block and y_a of the fused forward substitution are not needed before the panel solves:
                // they are computed behind barrier (A), while this wavefront has nothing else to do -- finish_later below)
            }
            v4d x20 = (v4d){0, 0, 0, 0}, x30 = x20, x21_ = x20, x31 = x20;
            if (hasb) {
                // ---- look-ahead of block b: rows R2, R3 in columns a, diagonal block b ------------------------------
                if (!la_ready) {                 // (super column 0, or no early start: tiles straight from P)
                    p20 = init_tile(R2, tA, ntr, fo, li, kq);      p21 = init_tile(R2, tA + 1, ntr, fo, li, kq);
                    p30 = init_tile(R3, tA, ntr, fo, li, kq);      p31 = init_tile(R3, tA + 1, ntr, fo, li, kq);
                    e11 = init_tile(R2, R2, ntr, fo, li, kq);      e21 = init_tile(R3, R2, ntr, fo, li, kq);
                    e22 = init_tile(R3, R3, ntr, fo, li, kq);
                    la_done = 0;
                }
                if (J > 0) {
                    if (la_done < 8 * J) la1_ring(J, la_done, 8 * J, false, 0);      // what is left: normally the last 64 columns
                    if (!v3) {
                        p30 = (v4d){0, 0, 0, 0}; p31 = (v4d){0, 0, 0, 0}; e21 = (v4d){0, 0, 0, 0};
                        e22 = init_tile(R3, R3, ntr, fo, li, kq);
                    }
                }
                PROF(40);
                TL(0, J, 5);
                // panel solve against block a (own W1, L21, W2)
                double wn1[4], l21[4], wn2[4];
                load_wn(wn1, wn2, j0a, li, kq);
                load_l21(l21, 0, lane);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    x20 = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], p20[s_], x20, 0, 0, 0);
                    x30 = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], p30[s_], x30, 0, 0, 0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x20[s_], p21, 0, 0, 0);
                    p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x30[s_], p31, 0, 0, 0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    x21_ = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], p21[s_], x21_, 0, 0, 0);
                    x31 = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], p31[s_], x31, 0, 0, 0);
                }
                store_pair(R2, tA, x20, x21_, fo);
                if (v3) store_pair(R3, tA, x30, x31, fo);
                // the two chunks just produced complete diagonal block b
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x20[s_], x20[s_], e11, 0, 0, 0);
                    e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(x20[s_], x30[s_], e21, 0, 0, 0);
                    e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(x30[s_], x30[s_], e22, 0, 0, 0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21_[s_], x21_[s_], e11, 0, 0, 0);
                    e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(x21_[s_], x31[s_], e21, 0, 0, 0);
                    e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(x31[s_], x31[s_], e22, 0, 0, 0);
                }
                PROF(41);
                TL(0, J, 6);
                // ---- chain b ---------------------------------------------------------------------------------------
                ok = cholinv16(e11, j0b, 0) && ok;
                v4d xb = (v4d){0, 0, 0, 0};
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    xb = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(size_t)(j0b + li) * PLD + 4 * s_ + kq], e21[s_], xb, 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) sm.t21[256 + rg * 64 + lane] = xb[rg];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
                    e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[s_], xb[s_], e22, 0, 0, 0);
                ok = cholinv16(e22, j0b + 16, 16) && ok;
                PROF(42);
            }
            if (lane == 0) sm.flag[0] = ok ? 0 : 1;
            TL(0, J, 1);
            if constexpr (kFat) {
                publish_a(J, lane);
                if (!ok) return false;
            } else {
                __syncthreads();                                // (A)
            }
            TL(0, J, 2);
            PROF(1);
#ifdef HIPDRT_QP_PROFILE
            const unsigned long long _ta = __builtin_amdgcn_s_memtime();       // slots 16 + J: (A) -> everybody's arrival at (B), by J
            unsigned long long _tarr = _ta;
#define HIPDRT_PROF_ARRIVED() do { _tarr = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HIPDRT_PROF_ARRIVED()
#endif
            if (sm.flag[0]) return false;
            // ---- finish_later: what of the two chains nobody needed before (A).  The others' updates of the right-hand side
            // wait for the word flag[2] (2 J + 1: y_a is in vec, 2 J + 2: y_b), which they reach long after it is set.
            {
                double l21[4];
                load_l21(l21, 0, lane);
                finish_block((v4d){l21[0], l21[1], l21[2], l21[3]}, j0a, lane, li, kq);
                if (lane == 0) sm.flag[2] = 2 * J + 1;
                if (hasb) {
                    // right-hand side of block b receives block a's update, then y_b
                    fwd_update(x20, x21_, R2, j0a, li, kq);
                    if (v3) fwd_update(x30, x31, R3, j0a, li, kq);
                    __builtin_amdgcn_wave_barrier();
                    load_l21(l21, 1, lane);
                    finish_block((v4d){l21[0], l21[1], l21[2], l21[3]}, j0b, lane, li, kq);
                    if (lane == 0) sm.flag[2] = 2 * J + 2;
                }
            }
            PROF(43);
            TL(0, J, 7);
            // ---- early start of the next super column's look-ahead (tile rows tA+6, tA+7): source tiles, then the old range of
            // their history, until it is done or the other wavefronts stand at barrier (B) -- then (B) first, the rest after
            bool at_b = false;
            la_ready = false;
            if (2 * J + 3 < nblk) {
                const int N2 = tA + 6, N3 = tA + 7, cA = tA + 4;
                p20 = init_tile(N2, cA, ntr, fo, li, kq);      p21 = init_tile(N2, cA + 1, ntr, fo, li, kq);
                p30 = init_tile(N3, cA, ntr, fo, li, kq);      p31 = init_tile(N3, cA + 1, ntr, fo, li, kq);
                e11 = init_tile(N2, N2, ntr, fo, li, kq);      e21 = init_tile(N3, N2, ntr, fo, li, kq);
                e22 = init_tile(N3, N3, ntr, fo, li, kq);
                la_ready = true;
                la_done = 0;
                const int kend = 8 * J;          // columns < 4 J are final; this super column's own 64 columns are being solved now
                while (la_done < kend) {
                    la_done = la1_ring(J + 1, la_done, kend, !at_b, NARR * (J + 1));
                    if (la_done < kend && !at_b) {
                        at_b = true;
                        HIPDRT_PROF_ARRIVED();
                    }
                }
            }
            if (!at_b) { wait_b(J); HIPDRT_PROF_ARRIVED(); }    // (B), this wavefront's half: the next chain overwrites t21 / img
            PROF(4);
            TL(0, J, 3);
#ifdef HIPDRT_QP_PROFILE
            if (threadIdx.x == 0 && blockIdx.x == 0 && J < 14) atomicAdd(&g_qp_prof[26 + J], __builtin_amdgcn_s_memtime() - _jt0);
            if (threadIdx.x == 0 && blockIdx.x == 0 && J < 10) atomicAdd(&g_qp_prof[16 + J], _tarr - _ta);
#endif
#undef HIPDRT_PROF_ARRIVED
        }
        return true;
    }

    // ======== wavefront 1: tile rows Q2 = tA+4, Q3 = tA+5 over all four columns + the next diagonal block a' =============
    __device__ __forceinline__ bool f64_look2() {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;
        const int nblk = (n + NB - 1) / NB, nsup = (nblk + 1) >> 1;
        const int ntr = (n + 15) >> 4;
        v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
        v4d* const img22 = img21 + 64;
        TileSrc pre[11];                         // source tiles of the next super column, requested before barrier (B)
        bool have_pre = false;
        for (int J = 0; J < nsup; ++J) {
            const int j0a = J * 64, j0b = j0a + NB, tA = 4 * J, tB = tA + 2;
            const int Q2 = tA + 4, Q3 = tA + 5;
            const bool v2 = Q2 < ntr, v3 = Q3 < ntr;
            TL(1, J, 0);
            if (v2) {
                v4d ca[2][2], cb[2][2], f11, f21, f22;
                if (have_pre) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            ca[r][c] = tile_image(pre[4 * r + c], Q2 + r, tA + c, li, kq);
                            cb[r][c] = tile_image(pre[4 * r + 2 + c], Q2 + r, tB + c, li, kq);
                        }
                    f11 = tile_image(pre[8], Q2, Q2, li, kq);
                    f21 = tile_image(pre[9], Q3, Q2, li, kq);
                    f22 = tile_image(pre[10], Q3, Q3, li, kq);
                } else {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            ca[r][c] = init_tile(Q2 + r, tA + c, ntr, fo, li, kq);
                            cb[r][c] = init_tile(Q2 + r, tB + c, ntr, fo, li, kq);
                        }
                    f11 = init_tile(Q2, Q2, ntr, fo, li, kq);
                    f21 = init_tile(Q3, Q2, ntr, fo, li, kq);
                    f22 = init_tile(Q3, Q3, ntr, fo, li, kq);
                }
                // fat form: this wavefront's ordinary rows (at most RM1, one pass), accumulated behind the look-ahead tiles
                RowSet<RM1> xr;
                v4d xacc[RM1][4];
                if constexpr (kFat) {
                    unsigned long long m0, m1;
                    my_rows(J, 1, lane, ntr, m0, m1);
                    rows_next(xr, m0, m1, tA);
                    rows_init(xacc, xr, tA, ntr, fo, li, kq);
                }
                if (J > 0) {
                    const char* qb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) qb[c] = uniform_ptr(tile2(tA + c, 0));
                    const char* q2 = uniform_ptr(tile2(Q2, 0));
                    const char* q3 = uniform_ptr(tile2(v3 ? Q3 : Q2, 0));
                    const unsigned voff = (unsigned)fo * 16u;
                    struct Frag { v2d b[4], a2, a3; };
                    const int nk2 = 8 * J, klast = nk2 - 1;
                    auto loadf = [&](Frag& f_, int k2) {
                        const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
#pragma unroll
                        for (int c = 0; c < 4; ++c) f_.b[c] = gload16(qb[c] + o, voff);
                        f_.a2 = gload16(q2 + o, voff); f_.a3 = gload16(q3 + o, voff);
                    };
#define HIPDRT_STEP11(H)                                                                                          \
                    ca[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[0].H, f_.a2.H, ca[0][0], 0, 0, 0);     \
                    ca[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[1].H, f_.a2.H, ca[0][1], 0, 0, 0);     \
                    cb[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[2].H, f_.a2.H, cb[0][0], 0, 0, 0);     \
                    cb[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[3].H, f_.a2.H, cb[0][1], 0, 0, 0);     \
                    f11 = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.a2.H, f_.a2.H, f11, 0, 0, 0);                 \
                    if (v3) {                                                                                   \
                        ca[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[0].H, f_.a3.H, ca[1][0], 0, 0, 0); \
                        ca[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[1].H, f_.a3.H, ca[1][1], 0, 0, 0); \
                        cb[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[2].H, f_.a3.H, cb[1][0], 0, 0, 0); \
                        cb[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.b[3].H, f_.a3.H, cb[1][1], 0, 0, 0); \
                        f21 = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.a2.H, f_.a3.H, f21, 0, 0, 0);             \
                        f22 = __builtin_amdgcn_mfma_f64_16x16x4f64(f_.a3.H, f_.a3.H, f22, 0, 0, 0);             \
                    }
                    auto multf = [&](const Frag& f_) {
                        HIPDRT_STEP11(x)
                        HIPDRT_STEP11(y)
                        __builtin_amdgcn_sched_barrier(0);
                    };
#undef HIPDRT_STEP11
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    Frag f0, f1, f2, f3;
                    const int knew = nk2 - 8;                   // first half-chunk of the 64 columns solved in super column J - 1
                    bool arrived = false;
                    if (knew <= 2) { wait_b(J - 1); arrived = true; }
                    loadf(f0, 0); loadf(f1, 1); loadf(f2, 2);
                    for (int k2 = 0; k2 < nk2; k2 += 4) {
                        if (!arrived && k2 + 6 >= knew) { wait_b(J - 1); arrived = true; }
                        loadf(f3, k2 + 3); vm_wait<18>(); multf(f0);
                        loadf(f0, k2 + 4); vm_wait<18>(); multf(f1);
                        loadf(f1, k2 + 5); vm_wait<18>(); multf(f2);
                        loadf(f2, k2 + 6); vm_wait<18>(); multf(f3);
                    }
                    vm_wait<0>();
                    if (!v3) {
                        ca[1][0] = (v4d){0, 0, 0, 0}; ca[1][1] = ca[1][0]; cb[1][0] = ca[1][0]; cb[1][1] = ca[1][0]; f21 = ca[1][0];
                        f22 = init_tile(Q3, Q3, ntr, fo, li, kq);
                    }
                    if constexpr (kFat) {
                        if (xr.act[0]) rows_ring(xacc, xr, J, tA, fo, true);     // (the ring above has waited for super column J - 1)
                    }
                }
                TL(1, J, 1);
                if constexpr (kFat) wait_a(J); else __syncthreads();      // (A)
                TL(1, J, 2);
                if (sm.flag[0]) return false;
                have_pre = false;
                BFrag bf;
                load_bfrag(bf, tA, true, fo);                   // (rows tA+2, tA+3 are valid whenever Q2 is)
                v4d xa[2][2], xb[2][2];
                {
                    double wn1[4], l21[4], wn2[4];
                    load_wn(wn1, wn2, j0a, li, kq);
                    load_l21(l21, 0, lane);
#pragma unroll
                    for (int r = 0; r < 2; ++r) { xa[r][0] = (v4d){0, 0, 0, 0}; xa[r][1] = (v4d){0, 0, 0, 0}; }
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int r = 0; r < 2; ++r) xa[r][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], ca[r][0][s_], xa[r][0], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int r = 0; r < 2; ++r) ca[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], xa[r][0][s_], ca[r][1], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int r = 0; r < 2; ++r) xa[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], ca[r][1][s_], xa[r][1], 0, 0, 0);
                }
                store_pair(Q2, tA, xa[0][0], xa[0][1], fo);
                if (v3) store_pair(Q3, tA, xa[1][0], xa[1][1], fo);
                // block a's rank-32 update of the column-b tiles and of the next diagonal block, from registers
#pragma unroll
                for (int st = 0; st < 8; ++st)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) upd_b_step(cb[r][c], bf, c, xa[r][0], xa[r][1], st);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        f11 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0][k][s_], xa[0][k][s_], f11, 0, 0, 0);
                        f21 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0][k][s_], xa[1][k][s_], f21, 0, 0, 0);
                        f22 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1][k][s_], xa[1][k][s_], f22, 0, 0, 0);
                    }
                wait_y(2 * J + 1);
                fwd_update(xa[0][0], xa[0][1], Q2, j0a, li, kq);
                if (v3) fwd_update(xa[1][0], xa[1][1], Q3, j0a, li, kq);
                {
                    double wn1[4], l21[4], wn2[4];
                    load_wn(wn1, wn2, j0b, li, kq);
                    load_l21(l21, 1, lane);
#pragma unroll
                    for (int r = 0; r < 2; ++r) { xb[r][0] = (v4d){0, 0, 0, 0}; xb[r][1] = (v4d){0, 0, 0, 0}; }
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int r = 0; r < 2; ++r) xb[r][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], cb[r][0][s_], xb[r][0], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int r = 0; r < 2; ++r) cb[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], xb[r][0][s_], cb[r][1], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int r = 0; r < 2; ++r) xb[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], cb[r][1][s_], xb[r][1], 0, 0, 0);
                }
                store_pair(Q2, tB, xb[0][0], xb[0][1], fo);
                if (v3) store_pair(Q3, tB, xb[1][0], xb[1][1], fo);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        f11 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[0][k][s_], xb[0][k][s_], f11, 0, 0, 0);
                        f21 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[0][k][s_], xb[1][k][s_], f21, 0, 0, 0);
                        f22 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[1][k][s_], xb[1][k][s_], f22, 0, 0, 0);
                    }
                stage_dsc(f11);
                img21[lane] = f21;
                img22[lane] = f22;
                wait_y(2 * J + 2);
                fwd_update(xb[0][0], xb[0][1], Q2, j0b, li, kq);
                if (v3) fwd_update(xb[1][0], xb[1][1], Q3, j0b, li, kq);
                if constexpr (kFat) rows_panel(xacc, xr, J, lane, li, kq, fo);
                {
                    // (unconditional, stand-in tile (0, 0) when there are no such rows: see wavefront 0)
                    have_pre = Q2 + 4 < ntr;
                    const int N2 = have_pre ? Q2 + 4 : 0, N3 = have_pre ? Q2 + 5 : 0, cA = have_pre ? tA + 4 : 0, st = have_pre ? 1 : 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { pre[c] = tile_src(N2, cA + st * c, ntr, fo); pre[4 + c] = tile_src(N3, cA + st * c, ntr, fo); }
                    pre[8] = tile_src(N2, N2, ntr, fo); pre[9] = tile_src(N3, N2, ntr, fo);
                    pre[10] = tile_src(N3, N3, ntr, fo);
                }
            } else {
                if constexpr (kFat) wait_a(J); else __syncthreads();      // (A)
                if (sm.flag[0]) return false;
            }
            TL(1, J, 3);
            arrive_b(lane);
        }
        return true;
    }

    // ======== ordinary rows: tile rows tA+6 .. over all four columns =====================================================
    // The pieces of a row wavefront's super column, over R rows x 4 columns of accumulator tiles (R = RM; fat form: also RM1,
    // wavefront 1's rows): who owns what, source tiles, the history ring, the panel solves in groups of at most PG rows.
    template <int R> struct RowSet { int T[R]; bool act[R]; };
    // the rows of super column J_ owned by wavefront wv: bit r of (m0, m1) = tile row 4 J_ + 6 + r
    __device__ __forceinline__ void my_rows(int J_, int wv, int lane, int ntr, unsigned long long& m0, unsigned long long& m1) const {
        constexpr int SROW = Smem::SROW;
        const int nsq_ = ntr - (4 * J_ + 6) > 0 ? ntr - (4 * J_ + 6) : 0;
        const unsigned char* row = sm.sched + J_ * SROW;
        m0 = __ballot(lane < nsq_ && row[lane] == wv);
        m1 = 0;
        if (SROW > 64) m1 = __ballot(lane + 64 < nsq_ && row[lane + 64] == wv);
    }
    template <int R>
    __device__ __forceinline__ void rows_next(RowSet<R>& rs, unsigned long long& m0, unsigned long long& m1, int tA) const {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            int r = -1;
            if (m0) { r = __builtin_ctzll(m0); m0 &= m0 - 1; }
            else if (m1) { r = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
            rs.T[u] = r >= 0 ? tA + 6 + r : nch;
            rs.act[u] = r >= 0;
        }
    }
    template <int R>
    __device__ __forceinline__ void rows_init(v4d (&acc)[R][4], const RowSet<R>& rs, int tA, int ntr, int fo, int li, int kq) const {
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[u][c] = rs.act[u] ? init_tile(rs.T[u], tA + c, ntr, fo, li, kq) : (v4d){0, 0, 0, 0};
    }
    // operand ring as in factor_rows(): A tiles (own rows, HBM) three half-chunks ahead, B tiles (tile rows tA .. tA+3, shared by
    // all wavefronts: L1 / L2) one ahead; per step 4 B + R A loads, B first.  `arrived`: everybody's tiles of super column J - 1
    // are known to be stored (a later pass, or a caller that has waited already); otherwise wait_b in front of the first load
    // from those 64 columns.
    template <int R>
    __device__ __forceinline__ void rows_ring(v4d (&acc)[R][4], const RowSet<R>& rs, int J, int tA, int fo, bool arrived) const {
        const char* rb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rb[c] = uniform_ptr(tile2(tA + c, 0));
        const char* ra[R];
#pragma unroll
        for (int u = 0; u < R; ++u) ra[u] = uniform_ptr(tile2(rs.act[u] ? rs.T[u] : tA, 0));
        const unsigned voff = (unsigned)fo * 16u;
        struct SlA { v2d a[R]; };
        struct SlB { v2d b[4]; };
        const int nk2 = 8 * J, klast = nk2 - 1;
        auto loadA = [&](SlA& s_, int k2) {
            const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
#pragma unroll
            for (int u = 0; u < R; ++u) s_.a[u] = gload16(ra[u] + o, voff);
        };
        auto loadB = [&](SlB& s_, int k2) {
            const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
#pragma unroll
            for (int c = 0; c < 4; ++c) s_.b[c] = gload16(rb[c] + o, voff);
        };
        auto mult = [&](const SlA& a_, const SlB& b_) {
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (rs.act[u]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[u][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b[c].x, a_.a[u].x, acc[u][c], 0, 0, 0);
                }
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (rs.act[u]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[u][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(b_.b[c].y, a_.a[u].y, acc[u][c], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile loads above: from here on the count is ours
        __builtin_amdgcn_sched_barrier(0);
        const int knew = nk2 - 8;                   // first half-chunk of the 64 columns solved in super column J - 1
        if (!arrived && knew <= 2) { wait_b(J - 1); arrived = true; }
        if constexpr (kFat) {
            // one wavefront per SIMD: nobody covers for a late operand, and the shared B rows come from as far away as the A
            // rows (32 problems' factors per XCD do not fit its L2) -- A and B both two half-chunks ahead (a half-chunk is
            // 8 R MFMAs = 3.6 k cycles at R = 7), three buffers each, three half-chunks per trip
            SlA a0, a1, a2;
            SlB b0, b1, b2;
            loadB(b0, 0); loadA(a0, 0); loadB(b1, 1); loadA(a1, 1);
            for (int k2 = 0; k2 < nk2; k2 += 3) {
                if (!arrived && k2 + 4 >= knew) { wait_b(J - 1); arrived = true; }
                loadB(b2, k2 + 2); loadA(a2, k2 + 2); vm_wait<2 * (R + 4)>(); mult(a0, b0);
                loadB(b0, k2 + 3); loadA(a0, k2 + 3); vm_wait<2 * (R + 4)>(); if (k2 + 1 < nk2) mult(a1, b1);
                loadB(b1, k2 + 4); loadA(a1, k2 + 4); vm_wait<2 * (R + 4)>(); if (k2 + 2 < nk2) mult(a2, b2);
            }
            vm_wait<0>();
            return;
        }
        SlA a0, a1, a2, a3;
        SlB b0, b1;
        loadB(b0, 0); loadA(a0, 0); loadA(a1, 1); loadA(a2, 2);
        for (int k2 = 0; k2 < nk2; k2 += 4) {       // nk2 = 8 J: a multiple of 4
            if (!arrived && k2 + 6 >= knew) { wait_b(J - 1); arrived = true; }
            loadB(b1, k2 + 1); loadA(a3, k2 + 3); vm_wait<2 * R + 4>(); mult(a0, b0);
            loadB(b0, k2 + 2); loadA(a0, k2 + 4); vm_wait<2 * R + 4>(); mult(a1, b1);
            loadB(b1, k2 + 3); loadA(a1, k2 + 5); vm_wait<2 * R + 4>(); mult(a2, b0);
            loadB(b0, k2 + 4); loadA(a2, k2 + 6); vm_wait<2 * R + 4>(); mult(a3, b1);
        }
        vm_wait<0>();
    }
    // behind (A): rows U0 .. U0 + G - 1 (those of them below na) solved against block a, block a's update of their column-b tiles
    // from registers, solved against block b, stored, right-hand side updated
#ifdef HIPDRT_QP_PROFILE
    // PROFILE builds: time line of wavefront 2's panel solves in the rows of the unused wavefronts 4..7 of g_qp_tl
    // (4 + 2 pass + group; stamps: 0 entry, 1 solved against block a, 2 stored + block a's update, 3 y_a there, 4 right-hand side
    // updated, 5 solved against block b, 6 stored + y_b there, 7 right-hand side updated)
    int dbg_tl = -1;
#define TLP(k) do { if (dbg_tl >= 0) TL(dbg_tl + (U0 ? 1 : 0), J, k); } while (0)
#else
#define TLP(k)
#endif
    template <int R, int U0, int G>
    __device__ __forceinline__ void rows_panel_group(v4d (&acc)[R][4], const RowSet<R>& rs, int na, const BFrag& bf, int J,
                                                     int lane, int li, int kq, int fo) {
        const int j0a = J * 64, j0b = j0a + NB, tA = 4 * J, tB = tA + 2;
        TLP(0);
        v4d x1[G], x2[G];
        {
            double wn1[4], l21[4], wn2[4];
            load_wn(wn1, wn2, j0a, li, kq);
            load_l21(l21, 0, lane);
#pragma unroll
            for (int g = 0; g < G; ++g) { x1[g] = (v4d){0, 0, 0, 0}; x2[g] = (v4d){0, 0, 0, 0}; }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int g = 0; g < G; ++g) if (U0 + g < na) x1[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], acc[U0 + g][0][s_], x1[g], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int g = 0; g < G; ++g) if (U0 + g < na) acc[U0 + g][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x1[g][s_], acc[U0 + g][1], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int g = 0; g < G; ++g) if (U0 + g < na) x2[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], acc[U0 + g][1][s_], x2[g], 0, 0, 0);
        }
        TLP(1);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (rs.act[U0 + g]) store_pair(rs.T[U0 + g], tA, x1[g], x2[g], fo);
#pragma unroll
        for (int st = 0; st < 8; ++st)
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (U0 + g < na) {
                    upd_b_step(acc[U0 + g][2], bf, 0, x1[g], x2[g], st);
                    upd_b_step(acc[U0 + g][3], bf, 1, x1[g], x2[g], st);
                }
        TLP(2);
        wait_y(2 * J + 1);
        TLP(3);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (rs.act[U0 + g] && rs.T[U0 + g] < nch) fwd_update(x1[g], x2[g], rs.T[U0 + g], j0a, li, kq);
        TLP(4);
        {
            double wn1[4], l21[4], wn2[4];
            load_wn(wn1, wn2, j0b, li, kq);
            load_l21(l21, 1, lane);
#pragma unroll
            for (int g = 0; g < G; ++g) { x1[g] = (v4d){0, 0, 0, 0}; x2[g] = (v4d){0, 0, 0, 0}; }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int g = 0; g < G; ++g) if (U0 + g < na) x1[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], acc[U0 + g][2][s_], x1[g], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int g = 0; g < G; ++g) if (U0 + g < na) acc[U0 + g][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x1[g][s_], acc[U0 + g][3], 0, 0, 0);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int g = 0; g < G; ++g) if (U0 + g < na) x2[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], acc[U0 + g][3][s_], x2[g], 0, 0, 0);
        }
        TLP(5);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (rs.act[U0 + g]) store_pair(rs.T[U0 + g], tB, x1[g], x2[g], fo);
        wait_y(2 * J + 2);
        TLP(6);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (rs.act[U0 + g] && rs.T[U0 + g] < nch) fwd_update(x1[g], x2[g], rs.T[U0 + g], j0b, li, kq);
        TLP(7);
    }
    template <int R>
    __device__ __forceinline__ void rows_panel(v4d (&acc)[R][4], const RowSet<R>& rs, int J, int lane, int li, int kq, int fo) {
        if (!rs.act[0]) return;
        // (rows are dealt in order: a wavefront with fewer than R rows in this pass skips the absent rows' MFMAs -- in the
        // last super columns most wavefronts hold one row)
        int na = 1;
#pragma unroll
        for (int u = 1; u < R; ++u) if (rs.act[u]) na = u + 1;
        BFrag bf;
        load_bfrag(bf, 4 * J, true, fo);               // (rows tA+2, tA+3 are valid whenever there are rows here)
        if constexpr (R <= PG) {
            rows_panel_group<R, 0, R>(acc, rs, na, bf, J, lane, li, kq, fo);
        } else {
            static_assert(R <= 2 * PG, "two groups");
            rows_panel_group<R, 0, PG>(acc, rs, na, bf, J, lane, li, kq, fo);
            if (na > PG) rows_panel_group<R, PG, R - PG>(acc, rs, na, bf, J, lane, li, kq, fo);
        }
    }

    // ======== the row wavefronts (2..7; fat form: 2, 3) ===================================================================
    __device__ __forceinline__ bool f64_rows(int wv) {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;
        const int nblk = (n + NB - 1) / NB, nsup = (nblk + 1) >> 1;
        const int ntr = (n + 15) >> 4;
        for (int J = 0; J < nsup; ++J) {
            const int tA = 4 * J;
            TL(wv, J, 0);
            unsigned long long m0, m1;
            my_rows(J, wv, lane, ntr, m0, m1);
            const int mine = __builtin_popcountll(m0) + __builtin_popcountll(m1);
            const int npass = mine > RM ? (mine + RM - 1) / RM : 1;
            RowSet<RM> rs;
            v4d acc[RM][4];
            rows_next(rs, m0, m1, tA);
            rows_init(acc, rs, tA, ntr, fo, li, kq);
#pragma unroll 1
            for (int ps = 0; ps < npass; ++ps) {
                if (ps > 0) {
                    rows_next(rs, m0, m1, tA);
                    rows_init(acc, rs, tA, ntr, fo, li, kq);
                }
                if (J > 0 && rs.act[0]) rows_ring(acc, rs, J, tA, fo, ps > 0);      // (a later pass starts behind (A))
                if (ps == 0) {
                    TL(wv, J, 1);
                    if constexpr (kFat) wait_a(J); else __syncthreads();      // (A)
                    TL(wv, J, 2);
                    if (sm.flag[0]) return false;
                }
#ifdef HIPDRT_QP_PROFILE
                dbg_tl = (wv == 2 && ps < 2) ? 4 + 2 * ps : -1;
#endif
                rows_panel(acc, rs, J, lane, li, kq, fo);
            }
            TL(wv, J, 3);
            arrive_b(lane);
        }
        return true;
    }

    // -----------------------------------------------------------------------------------------------------
    // vec := S^-1 vec.  Two roles.  Wavefront 0 multiplies by the inverse 32x32 diagonal blocks (LDS resident) AND applies the
    // urgent part of a block's rank-32 update itself -- the two tiles that touch the 32 entries its next diagonal step needs;
    // wavefronts 1.. apply the rest of the update while wavefront 0 is already on the next block.  The sweep's chain is then
    // diagonal step -> two tiles -> diagonal step on one wavefront, with ONE barrier per block: it publishes y_j to the others
    // and separates block j's trailing update from block j + 1's.  Every entry still receives the blocks' contributions in the
    // order 0, 1, 2, ..., each from one wavefront with the same arithmetic (fwd_tile / bwd_chunk): the sums are bit for bit
    // those of the strictly sequential sweep (diagonal step, barrier, update of all rows, barrier) this replaces.
    // The update operands (tiles of L in HBM) do not depend on the running solution: they are fetched TWO blocks ahead into
    // alternating register buffers by hand-issued loads and retired with a counted wait (left to hipcc, every use waits for
    // vmcnt(0..3), i.e. also for the block requested one step ago), behind the block's arithmetic, so that their issue cost
    // (~100 cycles per 1 KB load and wavefront) is off the chain.  Wavefront 0's loads take a per-lane 64-bit address
    // (gload16v): with a scalar base pair, as everywhere else, this role's prefetch faulted on the device -- at the kernel's
    // scalar-register limit the bases are reloaded from VGPR lanes right in front of the loads, inside the 5 wait states gfx9
    // wants between a VALU-written SGPR and a VMEM instruction reading it (qp_common.hpp: gload16; tools/sgpr_hazard.py).
    //   tile load map: instruction h (k-half) of a 2 KB tile covers double2 index h*64 + lane  ->  row i = lane/4,
    //   columns 8h + (lane%4) and 8h + (lane%4) + 4.
    static constexpr int SW_TW = RNW - 1;                // trailing-update wavefronts
    // (fat form: three updaters with up to ten tiles each per block -- seven in the two register buffers (2 x 7 x 16 registers:
    // they must stay in real VGPRs, a copy made before the hand-counted wait would copy stale registers), the rest of an
    // early block's tiles requested together behind the buffered ones: one more memory round trip in 4-5 of the 17 blocks)
    static constexpr int kCap = kFat ? 5 : kSweepCap;
    static constexpr int kOvr = kFat ? 1 : 1;
    static constexpr int SW_FT = (29 + SW_TW - 1) / SW_TW < kCap ? (29 + SW_TW - 1) / SW_TW : kCap;   // forward: buffered tiles each
    static constexpr int SW_BC = (30 + SW_TW - 1) / SW_TW < kCap ? (30 + SW_TW - 1) / SW_TW : kCap;   // backward: buffered chunks each
    static constexpr int SW_NBUF = SW_FT > SW_BC ? SW_FT : SW_BC;
    struct SweepBuf { v2d t[SW_NBUF][4]; };
    struct UrgentBuf { v2d t[2][4]; };

    // one tile row of the forward update: vec[rows of tile row T] -= L(T, 2jb..2jb+1) y_jb   (t: [chunk*2 + half])
    __device__ __forceinline__ void fwd_tile(const v2d (&t)[4], const double (&ya)[4], const double (&yb)[4], int T, int l4, int g4) {
        double pv = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) pv += t[q].x * ya[q] + t[q].y * yb[q];
        pv = quad_sum(pv);
        if (l4 == 0) {
            const int row = T * 16 + g4;
            if (row < n) sm.vec[row] -= pv;
        }
    }
    // one chunk of the backward update: vec[16c ..] -= L(tb..tb+1, c)' x_jb   (t: [tile*2 + half])
    __device__ __forceinline__ void bwd_chunk(const v2d (&t)[4], double x0, double x1, int c, int lane, int l4) {
        const double s0 = t[0].x * x0 + t[2].x * x1;     // column 16c + l4
        const double s1 = t[0].y * x0 + t[2].y * x1;     // column 16c + l4 + 4
        const double s2 = t[1].x * x0 + t[3].x * x1;     // column 16c + 8 + l4
        const double s3 = t[1].y * x0 + t[3].y * x1;     // column 16c + 12 + l4
        const double f = colsum4(s0, s1, s2, s3, lane);   // column 16c + l4 + 4*(lane>>4)
        if ((lane & 12) == 0) sm.vec[c * 16 + l4 + 4 * (lane >> 4)] -= f;
    }

    __device__ __forceinline__ void forward() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* vec = sm.vec;
        const double* U = sm.U;
        PROF_DECL
        // Two code paths with matching barrier sequences: one barrier per block (behind the diagonal step) and one at the end.
        // The barriers order LDS traffic only, so operand tiles stay in flight across them.
        if (wv == 0) {
            // ======== wavefront 0: multiply by the inverse 32x32 diagonal blocks ================================
            const int r = lane & 31;
            // U in global memory (GU): this lane's row of the NEXT inverse block is requested while the updaters work on the
            // current one (32 strided 8-byte loads per lane, an L2 round trip that used to sit on the sweep's critical path)
            double mr[NB];
            if (GU) {
#pragma unroll
                for (int c = 0; c < NB; ++c) mr[c] = U[(size_t)r * PLD + c];
            }
            const int l4 = lane & 3, g4 = lane >> 2;
            auto ucount = [&](int jb) {                 // urgent tile rows of block jb: tb + 2, tb + 3 where they exist
                const int k = jb < nblk ? ntr - (2 * jb + 2) : 0;
                return k < 0 ? 0 : (k < 2 ? k : 2);
            };
            auto upre = [&](UrgentBuf& B_, int jb) {
                const int nv = ucount(jb);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u < nv) {
                        const double2* p = tile2(2 * jb + 2 + u, 2 * jb) + lane;   // chunks 2jb, 2jb+1 are adjacent
                        B_.t[u][0] = gload16v<0>(p); B_.t[u][1] = gload16v<1024>(p);
                        B_.t[u][2] = gload16v<2048>(p); B_.t[u][3] = gload16v<3072>(p);
                    }
                }
            };
            auto urgent = [&](UrgentBuf& B_, int jb) {  // b_(jb+1) -= L(jb+1, jb) y_jb, then the tiles of two blocks on
                const int j0 = jb * NB, nv = ucount(jb);
                vm_wait_tiles(ucount(jb + 1));
                if (nv > 0) {
                    double ya[4], yb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ya[q] = vec[j0 + 8 * q + l4]; yb[q] = vec[j0 + 8 * q + l4 + 4]; }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (u < nv) fwd_tile(B_.t[u], ya, yb, 2 * jb + 2 + u, l4, g4);
                }
                upre(B_, jb + 2);
            };
            UrgentBuf ua, ub;
            upre(ua, 0);
            upre(ub, 1);
            auto dstep = [&](int jb) {                   // forward: y = M b (upper-right 16x16 of M is zero)
                const int j0 = jb * NB;
                const double* Mr = U + (size_t)(j0 + r) * PLD;
                PROF(6);
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int c = 0; c < NB; c += 4) {
                    s0 += (GU ? mr[c] : Mr[c]) * vec[j0 + c];
                    s1 += (GU ? mr[c + 1] : Mr[c + 1]) * vec[j0 + c + 1];
                    s2 += (GU ? mr[c + 2] : Mr[c + 2]) * vec[j0 + c + 2];
                    s3 += (GU ? mr[c + 3] : Mr[c + 3]) * vec[j0 + c + 3];
                }
                const double y = (s0 + s1) + (s2 + s3);
                __builtin_amdgcn_wave_barrier();
                if (lane < NB) vec[j0 + lane] = y;
                lds_barrier();
                PROF(5);
                if (GU && jb + 1 < nblk) {
#pragma unroll
                    for (int c = 0; c < NB; ++c) mr[c] = Mr[(size_t)NB * PLD + c];
                }
            };
            // (two blocks per trip, one buffer each: a buffer picked by the block's parity at run time would make hipcc copy
            // the freshly requested registers at the merge -- before the data has arrived)
            for (int jb = 0; jb < nblk; jb += 2) {
                dstep(jb);
                urgent(ua, jb);
                if (jb + 1 < nblk) { dstep(jb + 1); urgent(ub, jb + 1); }
            }
            vm_wait<0>();
            lds_barrier();
        } else {
            // ======== wavefronts 1..: the rest of the rank-32 updates ==========================================
            const int l4 = lane & 3, g4 = lane >> 2, tw_ = wv - 1;
            const unsigned voff = (unsigned)lane * 16u;
            auto fcount = [&](int jb) {                 // tiles this wavefront requests for block jb (tile rows tb + 4 ..)
                const int tbelow = jb < nblk ? ntr - (2 * jb + 4) : 0;
                const int k = tbelow - tw_ > 0 ? (tbelow - tw_ + SW_TW - 1) / SW_TW : 0;
                return k < SW_FT ? k : SW_FT;
            };
            auto fpre1 = [&](SweepBuf& B_, int jb, int nv, int u) {     // tile (tb+4+tt, 2jb..2jb+1) into slot u: [chunk*2 + half]
                if (u < nv) {
                    const char* p = uniform_ptr(tile2(2 * jb + 4 + tw_ + u * SW_TW, 2 * jb));   // chunks 2jb, 2jb+1 are adjacent
#pragma unroll
                    for (int q = 0; q < 4; ++q) B_.t[u][q] = gload16(p + q * 1024, voff);
                }
            };
            auto fpre = [&](SweepBuf& B_, int jb) {
                const int nv = fcount(jb);
#pragma unroll
                for (int u = 0; u < SW_FT; ++u) fpre1(B_, jb, nv, u);
            };
            auto fstep = [&](SweepBuf& B_, int jb) {
                const int j0 = jb * NB, tb = 2 * jb, tbelow = ntr - (tb + 4);
                PROFW(44);                                          // (PROFILE build, wavefronts 1..: arithmetic + requests | barrier | loads)
                lds_barrier();
                PROFW(45);
                vm_wait_tiles(fcount(jb + 1));                      // this block's tiles are in; the next block's may be in flight
                PROFW(46);
                if (tbelow > 0) {
                    // q = 2*chunk + half: lane holds columns 8q + l4 and 8q + l4 + 4 of tile row g4
                    double ya[4], yb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ya[q] = vec[j0 + 8 * q + l4]; yb[q] = vec[j0 + 8 * q + l4 + 4]; }
                    // (a slot's next tile -- block jb + 2 -- is requested right behind its arithmetic: the seven updaters' requests
                    // reach the CU's one address path spread over the step instead of as one burst behind it)
                    const int nv2 = fcount(jb + 2);
#pragma unroll
                    for (int u = 0; u < SW_FT; ++u) {
                        const int tt = tw_ + u * SW_TW;
                        if (tt < tbelow) fwd_tile(B_.t[u], ya, yb, tb + 4 + tt, l4, g4);
                        __builtin_amdgcn_sched_barrier(0);
                        fpre1(B_, jb + 2, nv2, u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {
                        // more tile rows below than the register buffers hold (n > 528, or fewer wavefronts): the rest straight
                        // from memory, four tile rows per round -- their sixteen loads are requested together, so a round costs
                        // one memory round trip instead of four
                        constexpr int OVR = kOvr;    // (eight wavefronts, n <= 528 never get here: no registers spent on it)
                        for (int tt0 = tw_ + SW_FT * SW_TW; tt0 < tbelow; tt0 += OVR * SW_TW) {
                            v2d t_[OVR][4];
#pragma unroll
                            for (int j = 0; j < OVR; ++j) {
                                const int tt = tt0 + j * SW_TW;
                                if (tt < tbelow) {
                                    const double2* p = tile2(tb + 4 + tt, 2 * jb) + lane;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) { const double2 d_ = p[q * 64]; t_[j][q] = (v2d){d_.x, d_.y}; }
                                }
                            }
#pragma unroll
                            for (int j = 0; j < OVR; ++j) {
                                const int tt = tt0 + j * SW_TW;
                                if (tt < tbelow) fwd_tile(t_[j], ya, yb, tb + 4 + tt, l4, g4);
                            }
                        }
                    }
                }
            };
            {
                SweepBuf fa, fb;
                fpre(fa, 0);
                fpre(fb, 1);
                for (int jb = 0; jb < nblk; jb += 2) {
                    fstep(fa, jb);
                    if (jb + 1 < nblk) fstep(fb, jb + 1);
                }
                vm_wait<0>();
                lds_barrier();
            }
        }
    }

    __device__ __forceinline__ void backward() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* vec = sm.vec;
        const double* U = sm.U;
        PROF_DECL
        // (roles and barriers as in forward(); the urgent part of block jb's update are chunks 2jb - 2 and 2jb - 1)
        if (wv == 0) {
            // ======== wavefront 0: multiply by the inverse 32x32 diagonal blocks ================================
            const int r = lane & 31;
            double mc[NB];                               // (GU: this lane's column of the next block, one block ahead: see forward())
            if (GU) {
#pragma unroll
                for (int q = 0; q < NB; ++q) mc[q] = U[((size_t)(nblk - 1) * NB + q) * PLD + r];
            }
            const int l4 = lane & 3, g4 = lane >> 2;
            auto ucount = [&](int jb) { return jb > 0 ? 2 : 0; };
            auto upre = [&](UrgentBuf& B_, int jb) {    // chunks 2jb - 2, 2jb - 1 of tile rows tb, tb + 1
                if (jb > 0) {
                    const int tb = 2 * jb;
                    const bool two = (tb + 1) < ntr;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = 2 * jb - 2 + u;
                        const double2* p0 = tile2(tb, c) + lane;
                        const double2* p1 = tile2(two ? tb + 1 : tb, c) + lane;
                        B_.t[u][0] = gload16v<0>(p0); B_.t[u][1] = gload16v<1024>(p0);
                        B_.t[u][2] = gload16v<0>(p1); B_.t[u][3] = gload16v<1024>(p1);
                    }
                }
            };
            auto urgent = [&](UrgentBuf& B_, int jb) {
                const int j0 = jb * NB, tb = 2 * jb;
                const bool two = (tb + 1) < ntr;                    // second tile-row of the block holds valid rows
                vm_wait_tiles(ucount(jb - 1));
                if (jb > 0) {
                    const double x0 = vec[j0 + g4];
                    const double x1 = two ? vec[j0 + 16 + g4] : 0.0;
#pragma unroll
                    for (int u = 0; u < 2; ++u) bwd_chunk(B_.t[u], x0, x1, 2 * jb - 2 + u, lane, l4);
                }
                upre(B_, jb - 2);
            };
            UrgentBuf ua, ub;
            upre(ua, nblk - 1);
            upre(ub, nblk - 2);
            auto dstep = [&](int jb) {                   // backward: x = M' y, lane = column of M
                const int j0 = jb * NB;
                const double* Mc = U + (size_t)j0 * PLD + r;
                PROF(8);
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int q = 0; q < NB; q += 4) {
                    s0 += (GU ? mc[q] : Mc[(size_t)q * PLD]) * vec[j0 + q];
                    s1 += (GU ? mc[q + 1] : Mc[(size_t)(q + 1) * PLD]) * vec[j0 + q + 1];
                    s2 += (GU ? mc[q + 2] : Mc[(size_t)(q + 2) * PLD]) * vec[j0 + q + 2];
                    s3 += (GU ? mc[q + 3] : Mc[(size_t)(q + 3) * PLD]) * vec[j0 + q + 3];
                }
                const double xv = (s0 + s1) + (s2 + s3);
                __builtin_amdgcn_wave_barrier();
                if (lane < NB) vec[j0 + lane] = xv;
                lds_barrier();
                PROF(7);
                if (GU && jb > 0) {
#pragma unroll
                    for (int q = 0; q < NB; ++q) mc[q] = U[((size_t)(j0 - NB) + q) * PLD + r];
                }
            };
            for (int jb = nblk - 1; jb >= 0; jb -= 2) {
                dstep(jb);
                urgent(ua, jb);
                if (jb - 1 >= 0) { dstep(jb - 1); urgent(ub, jb - 1); }
            }
            vm_wait<0>();
            lds_barrier();
        } else {
            // ======== wavefronts 1..: the chunks further left ===================================================
            const int l4 = lane & 3, g4 = lane >> 2, tw_ = wv - 1;
            const unsigned voff = (unsigned)lane * 16u;
            auto bcount = [&](int jb) {                 // chunks this wavefront requests for block jb (chunks 0 .. 2jb - 3)
                const int nc = jb > 0 ? 2 * jb - 2 : 0;
                const int k = nc - tw_ > 0 ? (nc - tw_ + SW_TW - 1) / SW_TW : 0;
                return k < SW_BC ? k : SW_BC;
            };
            auto bpre1 = [&](SweepBuf& B_, int jb, int nv, int u) {     // tiles (tb..tb+1, c) into slot u: [tile*2 + half]
                if (u < nv) {
                    const int tb = 2 * jb, c = tw_ + u * SW_TW;
                    const bool two = (tb + 1) < ntr;
                    const char* p0 = uniform_ptr(tile2(tb, c));
                    const char* p1 = uniform_ptr(tile2(two ? tb + 1 : tb, c));
                    B_.t[u][0] = gload16(p0, voff); B_.t[u][1] = gload16(p0 + 1024, voff);
                    B_.t[u][2] = gload16(p1, voff); B_.t[u][3] = gload16(p1 + 1024, voff);
                }
            };
            auto bpre = [&](SweepBuf& B_, int jb) {
                const int nv = bcount(jb);
#pragma unroll
                for (int u = 0; u < SW_BC; ++u) bpre1(B_, jb, nv, u);
            };
            auto bstep = [&](SweepBuf& B_, int jb) {
                const int j0 = jb * NB, tb = 2 * jb, nc = 2 * jb - 2;
                const bool two = (tb + 1) < ntr;                    // second tile-row of the block holds valid rows
                lds_barrier();
                vm_wait_tiles(bcount(jb - 1));
                if (nc > 0) {
                    // x of the block: row g4 of tile tb and of tile tb+1 (zero padding beyond n)
                    const double x0 = vec[j0 + g4];
                    const double x1 = two ? vec[j0 + 16 + g4] : 0.0;
                    const int nv2 = bcount(jb - 2);
#pragma unroll
                    for (int u = 0; u < SW_BC; ++u) {
                        const int c = tw_ + u * SW_TW;
                        if (c < nc) bwd_chunk(B_.t[u], x0, x1, c, lane, l4);
                        __builtin_amdgcn_sched_barrier(0);
                        bpre1(B_, jb - 2, nv2, u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {
                        // more finished chunks than the register buffers hold (n > 528, or fewer wavefronts): the rest straight
                        // from memory, four chunks per round (sixteen loads requested together: one round trip per round)
                        constexpr int OVR = kOvr;
                        for (int c0 = tw_ + SW_BC * SW_TW; c0 < nc; c0 += OVR * SW_TW) {
                            v2d t_[OVR][4];
#pragma unroll
                            for (int j = 0; j < OVR; ++j) {
                                const int c = c0 + j * SW_TW;
                                if (c < nc) {
                                    const double2* p0 = tile2(tb, c) + lane;
                                    const double2* p1 = tile2(two ? tb + 1 : tb, c) + lane;
                                    const double2 d0 = p0[0], d1 = p0[64], d2 = p1[0], d3 = p1[64];
                                    t_[j][0] = (v2d){d0.x, d0.y}; t_[j][1] = (v2d){d1.x, d1.y};
                                    t_[j][2] = (v2d){d2.x, d2.y}; t_[j][3] = (v2d){d3.x, d3.y};
                                }
                            }
#pragma unroll
                            for (int j = 0; j < OVR; ++j) {
                                const int c = c0 + j * SW_TW;
                                if (c < nc) bwd_chunk(t_[j], x0, x1, c, lane, l4);
                            }
                        }
                    }
                }
            };
            {
                SweepBuf ba, bb;
                bpre(ba, nblk - 1);
                bpre(bb, nblk - 2);
                for (int jb = nblk - 1; jb >= 0; jb -= 2) {
                    bstep(ba, jb);
                    if (jb - 1 >= 0) bstep(bb, jb - 1);
                }
                vm_wait<0>();
                lds_barrier();
            }
        }
    }

    // vec := S^-1 vec
    __device__ __forceinline__ void solve() {
        forward();
        backward();
    }

    // -----------------------------------------------------------------------------------------------------
    // dvec = P * vec from the packed lower tiles (1.2 MB instead of the 2.1 MB row-major matrix): every tile is read
    // once, as two contiguous 1 KB loads, and used for both  y_T += tile x_C  and  y_C += tile' x_T.  The tiles, in
    // column-major order, are dealt to the wavefronts in equal contiguous runs; a wavefront keeps the column sums of
    // its current tile column in registers and adds everything into its OWN partial result vector (the LDS array
    // of the inverse diagonal blocks is dead between two factorisations and serves as scratch), so the final sum
    // over wavefronts has a fixed order.
    __device__ __forceinline__ void matvec() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int l4 = lane & 3, g4 = lane >> 2;
        const int ntr = (n + 15) >> 4;
        const int NPd = nch * 16;
        const double* xin = sm.vec;
        // VT > 1: this wavefront also runs the tile runs of the virtual wavefronts wv + RNW, ... into their own partial vectors:
        // the sum over (virtual) wavefronts below is the one a workgroup of RNW VT wavefronts forms
        constexpr int RNWV = RNW * VT;
        const int ntile = ntr * (ntr + 1) / 2;
        const int per = (ntile + RNWV - 1) / RNWV;
#pragma unroll 1
        for (int vt = 0; vt < VT; ++vt) {
        const int vw = wv + RNW * vt;
        double* yw = sm.U + (size_t)vw * NPd;               // this (virtual) wavefront's partial result
        for (int i = lane; i < NPd; i += 64) yw[i] = 0.0;
        const int t0 = vw * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
        // locate (C, T) of tile t0: column C holds ntr - C tiles (T = C .. ntr-1)
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
        int Tp = T, Cp = C;                                   // prefetch cursor
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
                    // lane: row g4 of the tile, columns l4, l4+4 (d0) and l4+8, l4+12 (d1)
                    // (explicit fused multiply-adds: left to itself hipcc contracts `a b + c d` with EITHER product as the fused
                    // one, and which one changed with an unrelated edit of this function -- 7.5e-12 of the peak in x after a fit;
                    // this is the association every build up to round 5 happened to get)
                    double pr = __builtin_fma(cur.d1.y, xc3, __builtin_fma(cur.d1.x, xc2, __builtin_fma(cur.d0.y, xc1, cur.d0.x * xc0)));
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
        }
        __syncthreads();
        for (int i = tid; i < n; i += RT) {
            double s_ = 0.0;
#pragma unroll
            for (int w = 0; w < RNWV; ++w) s_ += sm.U[(size_t)w * NPd + i];
            sm.dvec[i] = s_;
        }
        __syncthreads();
        // the scratch goes back to the factorisation with its upper-right quarters zero
        for (int i = tid; i < RNWV * NPd; i += RT) sm.U[i] = 0.0;
    }
};

// ---------------------------------------------------------------------------------------------------------
// Posterior variance of the distribution on an evaluation grid, the diagonal of
// drt1d.estimate_distribution_cov (hybdrt/models/drt1d.py:3063-3151, 4116-4138):
//     var_i = b_i' P^-1 b_i = || L^-1 b_i ||^2 ,   P = L L' ,  b_i = row i of the basis-evaluation matrix
// (zero in the special-parameter slots).  The rows b_i ride through the factorisation as appended panel rows.
// ---------------------------------------------------------------------------------------------------------
struct CovArgs {
    int B, n;
    const double* Ppk; long long ppk_stride; int nchp;   // final P of every spectrum, packed tiles
    const double* Bex; int nex;                          // shared evaluation rows, packed tiles [nex][nchp][256]
    double* L; long long l_stride;                       // scratch (nch + nex) x nch tiles per spectrum
    double* out; long long out_stride;                   // [B][16 nex]
    int* status;                                         // [B]: 0 ok, -1 P not positive definite
};

// where U lives when it is not in LDS: behind the factor in the per-problem scratch (NP^2 doubles of tiles, then NP x 33)
template <bool GU>
__device__ __forceinline__ double* resident_u_ptr(double* Lb, int NP) { return Lb + (size_t)NP * NP; }

template <bool GU, int RTT = 512>
__global__ __launch_bounds__(RTT) void cov_kernel_resident(CovArgs a, int NP) {
    constexpr int RT = RTT, RNW = RTT / 64;
    const int b = blockIdx.x;
    extern __shared__ double smem[];
    OpsResidentT<GU, RTT> ops;
    ops.L = a.L + (size_t)b * a.l_stride; ops.nch = NP / 16; ops.n = a.n;
    ops.Ppk = a.Ppk + (size_t)b * a.ppk_stride; ops.nchp = a.nchp;
    ops.nex = a.nex; ops.Bex = a.Bex; ops.fwd = false;
    constexpr int VEC = ResSmemT<GU, RTT>::VEC;
    ops.sm.carve(smem);
    // with U outside LDS it sits behind the (nch + nex) x nch tiles of this spectrum's scratch
    if (GU) ops.sm.U = ops.L + (size_t)(NP / 16 + a.nex) * (NP / 16) * TSZ;
    for (int i = threadIdx.x; i < NP * PLD; i += RT) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < VEC; i += RT) { ops.sm.vec[i] = 0.0; ops.sm.dvec[i] = 0.0; }   // no diagonal shift
    __syncthreads();
    const bool ok = ops.factor();
    double* out = a.out + (size_t)b * a.out_stride;
    if (threadIdx.x == 0) a.status[b] = ok ? 0 : -1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l4 = lane & 3, g4 = lane >> 2;
    const int nblk = (a.n + NB - 1) / NB;
    for (int e = wv; e < a.nex; e += RNW) {
        double s_ = 0.0;
        if (ok) {
            for (int c = 0; c < 2 * nblk; ++c) {
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

// the QP kernel's factorisation: 64-column passes over the history (factor64) with eight wavefronts
template <int RTT> static constexpr bool kQpPanel64 = (HIPDRT_QP_PANEL64 != 0) && RTT == 512;

// (the second launch-bound argument is waves per SIMD: 2 in both forms, i.e. one 512-thread or two 256-thread workgroups per
// CU and at most 256 registers per lane; without it hipcc gives the 256-thread form 393 registers and one workgroup per CU)
// The FAT form (RTT = 256, WPS = 1, VT = 2): four wavefronts, one per SIMD, with the whole register file of their SIMD -- 256
// architectural registers for operand rings and 256 accumulation registers (AccVGPRs: hipcc puts every MFMA result there once
// a kernel may use more than 256 registers) = 32 accumulator tiles per wavefront.  The interior-point vectors run as two
// virtual threads per thread (qp_common.hpp), so its results are bit for bit the eight-wavefront kernel's.
template <bool GU, int RTT = 512, int WPS = 2, bool P64 = kQpPanel64<RTT>, int VT = 1>
__global__ __launch_bounds__(RTT, WPS) void qp_kernel_resident(QpArgs a, int NP) {
    constexpr int RT = RTT;
    const int b = a.order ? a.order[blockIdx.x] : blockIdx.x;
    if (a.active && !a.active[b]) return;
    extern __shared__ double smem[];
    OpsResidentT<GU, RTT, P64, VT> ops;
    ops.L = a.L + (size_t)b * a.l_stride; ops.nch = NP / 16; ops.n = a.n;
    ops.Ppk = a.Ppk ? a.Ppk + (size_t)b * a.ppk_stride : nullptr; ops.nchp = a.nchp;
    ops.sm.carve(smem);
    if (GU) ops.sm.U = resident_u_ptr<GU>(ops.L, NP);
    ops.build_schedule();
    // zero U (the upper-right quarter of every inverse block stays zero) and the padding of vec (read by the
    // updates of the last, partial block)
    for (int i = threadIdx.x; i < NP * PLD; i += RT) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < NP + 32; i += RT) ops.sm.vec[i] = 0.0;
    __syncthreads();
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    constexpr int VTH = RT * VT;             // virtual threads
    ipm_solve<RT, GU ? (2048 + VTH - 1) / VTH : (RNP_MAX + VTH - 1) / VTH, decltype(ops), VT>(a, b, ops, is);
}

// LDS bytes: everything for n <= 528, only the fixed part when U lives in global memory
// (qp = the QP kernel's layout, which differs from the posterior-variance kernel's when it runs factor64)
static size_t resident_lds_bytes(int NP, bool qp = false) {
    return ((size_t)NP * PLD + (qp ? ResSmemT<false, 512, kQpPanel64<512>>::FIXED : ResSmem::FIXED)) * sizeof(double);
}
// the fat four-wavefront form of the QP kernel
static constexpr bool kFatPanel64 = true;
using FatSmem = ResSmemT<false, 256, kFatPanel64, 2>;
static size_t resident_fat_lds_bytes(int NP) { return ((size_t)NP * PLD + FatSmem::FIXED) * sizeof(double); }
static_assert((FatSmem::FIXED + 544 * 33) * 8 <= 160 * 1024, "n = 514 must fit one CU's LDS (fat form)");
template <int RTT = 512>
static size_t resident_gu_lds_bytes(bool qp = false) {
    return (size_t)(qp ? ResSmemT<true, RTT, kQpPanel64<RTT>>::FIXED : ResSmemT<true, RTT>::FIXED) * sizeof(double);
}
// per-problem scratch doubles of the U-outside form: the tile-packed factor (NP^2) followed by U (NP x 33)
static size_t resident_gu_doubles(int n) {
    const size_t NP = (size_t)round_up(n, 32);
    return NP * NP + NP * PLD;
}

// scratch doubles per problem for the tile-packed factor: (NP/16)^2 tiles of 256 doubles
static size_t resident_l_doubles(int n) {
    const size_t nt = (size_t)round_up(n, 32) / 16;
    return nt * nt * TSZ;
}

}  // namespace hipdrt

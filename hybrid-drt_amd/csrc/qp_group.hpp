// coneqp for FEW, possibly LARGE problems: one problem on G co-resident workgroups ("members"), n <= 4096.
//
// The batch kernel (qp_resident.hpp) gives every problem one workgroup = one CU: the right shape for a thousand problems,
// the wrong one for a single spectrum (BASELINE configs[1]: 255 of 256 CUs idle), one joint fit (configs[4], n = 1078) or
// a coupled multi-observation QP (mapping/resolve.py, n = 7 x 514 = 3598 > the 2048 the one-CU kernel serves).  Here the
// factorisation of ONE problem -- the tile-packed left-looking blocked Cholesky of qp_resident.hpp, same tile layout, same
// per-tile arithmetic -- is shared by G workgroups:
//
//  * Tile rows belong to members in pairs (block row jb = tile rows 2 jb, 2 jb + 1 -> member jb % G) for the whole
//    factorisation, and inside the member to one row wavefront (static table, build_owner): a wavefront only re-reads tiles
//    it stored itself.
//  * The diagonal chain (wavefront 0: factor + invert the 32 x 32 diagonal block) runs REDUNDANTLY in every member, from
//    identical inputs, hence bit-identical: no member ever waits for another one's chain.
//  * The look-ahead tiles (the seven accumulators that become the next diagonal block and its panel rows: inner dimension
//    = every finished block column) are accumulated by ONE member per block column -- the owner of those two tile rows, all
//    seven working wavefronts on a slice of the inner dimension each -- and published: 14 KB in labuf[jb] + the laprog
//    word.  Every member's wavefront 1 fetches them while its chain still works on the current block, then does the last
//    rank-32 update, the panel solve and the staging of the next diagonal block itself (redundantly, identical bits).  This
//    publication is the only data that crosses between members inside a factorisation, and the only wait on another member.
//  * Inside a member there is no barrier at the end of a block column: the hand-offs between its wavefronts go through
//    progress bytes in LDS (rowdone[T]) and a "diagonal block staged" word, so a member's store drain overlaps its next
//    rank-k update.  Two workgroup barriers per block column remain: (A) inverse diagonal blocks published, (A2) the L21
//    scratch block may be rewritten.
//  * Everything else -- the O(n) interior-point vectors, the triangular sweeps, P x -- every member does redundantly on its
//    own copy (own state slot, own inverse diagonal blocks U), reading the shared factor: no communication, and the members
//    stay in lockstep because they compute the same bits.  Two group barriers per factorisation: before it (everybody has
//    finished sweeping the old factor; followed by an L1 invalidate) and after it (every tile is in memory; L1 invalidate).
//  * Visibility: the members of a group sit on ONE XCD (blocks b and b + 8 share an XCD under the round-robin dispatch;
//    checked at run time through HW_REG_XCC_ID -- a group that is spread over several XCDs reports HIPDRT_QP_ABORTED and the
//    launcher repeats that problem with G = 1), so the XCD's L2 is the point of coherence.  Stores are complete (s_waitcnt
//    vmcnt(0)) before a progress word is written.  Tiles of L are read with plain loads: a CU reads a tile for the first
//    time in a factorisation only after the tile is final (see "operand rings"), so its L1 cannot hold a stale copy; the
//    look-ahead accumulators are read with sc1 loads (L1 bypass).
//  * Every wait on another member is bounded (kSpinLimit): a protocol error or a member that never became resident traps the
//    launch instead of hanging the device.  Group launches of one device are chained through an event (qp.hip) so that two
//    of them cannot each occupy part of the CUs and wait for the rest.
//  * The result does not depend on G (who owns a row does not change its arithmetic; the look-ahead slices are cut by
//    wavefront, not by member).  Against the batch kernel it differs by rounding: there the predictor's forward substitution
//    is fused into the factorisation and the look-ahead tiles are summed in one pass.
//
// G = 1 is the same kernel without the global words: the single-workgroup form for n up to 4096.
#pragma once
#include "qp_resident.hpp"

namespace hipdrt {

#ifdef HIPDRT_GRP_TIMELINE
// diagnostic (-DHIPDRT_GRP_TIMELINE=<block column>, tools/probe_timeline.py): s_memtime stamps of one block column of the third
// factorisation, every member, every wavefront -> g_grp_tl[member][wavefront][10] (+ [2560 + member] = owner of the look-ahead rows)
__device__ unsigned long long g_grp_tl[32 * 8 * 10 + 32];
#define TS_DECL unsigned long long ts_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const bool ts_on_ = (fidx == 3 && jb == HIPDRT_GRP_TIMELINE);
#define TS(e) do { if (ts_on_) ts_[e] = __builtin_amdgcn_s_memtime(); } while (0)
#define TS_PRINT() do { if (ts_on_ && (threadIdx.x & 63) == 0 && g < 32) { const int w_ = threadIdx.x >> 6; for (int e_ = 0; e_ < 10; ++e_) g_grp_tl[(g * 8 + w_) * 10 + e_] = ts_[e_]; if (w_ == 1) g_grp_tl[2560 + g] = la_owner(jb); } } while (0)
#else
#define TS_DECL
#define TS(e)
#define TS_PRINT()
#endif

static constexpr int GRP_NMAX = 4096;                  // unknowns
static constexpr int GRP_OWN = 512;                    // tile rows incl. appended ones the tables cover
static constexpr int GRP_MAXG = 32;                    // members (CUs of one XCD)
#ifndef HIPDRT_GRP_LA_RING
#define HIPDRT_GRP_LA_RING 4
#endif
#ifndef HIPDRT_GRP_ROW_RING
#define HIPDRT_GRP_ROW_RING 4
#endif
static constexpr int GRP_LA_RING = HIPDRT_GRP_LA_RING;  // slots of the old-range operand ring (ring_la_slice): a ring issues
                                                        // ceil(len / slots) * slots + slots - 1 slots of loads whatever the length, a
                                                        // slice is 4 .. 22 half-chunks long, and a wavefront pays ~100 cycles per load
static constexpr int GRP_ROW_RING = HIPDRT_GRP_ROW_RING;  // slots of the rows' operand ring (ring_rows)
static constexpr int GRP_RMAXT = 2;                    // tile rows per row wavefront and pass: a member has few rows, and a short
                                                       // pass leaves registers for an eight-slot operand ring (below)
static constexpr int GRP_WORDS = 16 + GRP_OWN;         // ints of global sync state per problem
static constexpr int GRP_LA_SLOT = 7 * 256 + 32;       // doubles per block column in labuf: seven accumulator tiles + 32 right-hand-side entries
static constexpr int kNotResident = 1 << 30;           // poison bit in word [0]: a member gave up waiting for its partners
// words: [0] members arrived at the start | kNotResident (ONE word decides go / abort for every member), [1] OR of (1 << XCC id),
// [4] why a launch was aborted, [2] barrier counter, [3] laprog: factorisation count * 256
// + block columns whose look-ahead accumulators are in labuf (written by the owner of the column's look-ahead rows),
// [16 + T] rowprog: factorisation count * 256 + block columns of tile row T complete in memory (written by T's owner when T
// becomes a look-ahead row: the member that accumulates the next look-ahead block reads T as an operand)

// 16 bytes per lane that bypass the CU's vector L1 (written by another CU of the XCD in THIS factorisation after this CU may
// have read the same addresses: the look-ahead accumulators, whose slots are reused from one factorisation to the next)
static __device__ __forceinline__ v2d gload16_sc1(const char* sbase, unsigned voff) {
    v2d d;
    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
    return d;
}

struct OpsGroup : OpsResidentT<true, 512> {
    using Base = OpsResidentT<true, 512>;
    static constexpr int RT = 512, RNW = 8;
    // The predictor's forward substitution is fused into the factorisation as in the batch kernel (round 5; before, every member
    // ran a full forward sweep over the finished factor: one of four sweeps per interior-point iteration).  A member only solves
    // its own rows' tiles, so only ITS copy of those rows' right-hand-side entries receives the block columns' updates -- but
    // what every member needs is y_j = M_j b_j of every diagonal block, and b_j are the two look-ahead rows of block column
    // j - 1: their owner publishes the 32 entries (updated through block column j - 2 by its row wavefronts) behind the seven
    // look-ahead accumulators it publishes anyway, every member's wavefront 1 takes them over and applies block column j - 1's
    // update itself, as it does for the tiles.  Every member's chain then computes every y_j from identical bits, and at the end
    // of the factorisation every member's vector holds the whole substituted right-hand side.  Per row the block columns'
    // contributions arrive in ascending order whoever owns the row: the result does not depend on G.
    static constexpr bool kFusedForward = true;
    static constexpr int kRedSlots = 4;
    static constexpr int kSpinLimit = 1 << 24;
    static constexpr int kRendezvousLimit = 1 << 22;       // polls of the start rendezvous (~0.5 s) before the launch gives up cleanly
    int G = 1, g = 0;                      // members of the group, this member
    int* gs = nullptr;                     // global sync words of the problem (GRP_WORDS)
    int fidx = 0, gepoch = 0;              // factorisations started, group barriers passed (identical in every member)
    unsigned char* owner = nullptr;        // LDS [GRP_OWN]: row wavefront (2..7) owning tile row T in THIS member, 0 = not mine
    volatile unsigned char* rowdone = nullptr;   // LDS [GRP_OWN]: block columns of tile row T complete in memory (own member's rows
                                                 // and the look-ahead rows this member computes itself)
    double* latile = nullptr;              // LDS [6][256]: the look-ahead tiles computed by wavefronts 2..7 (register images)
    int* lacnt = nullptr;                  // LDS: look-ahead tiles delivered in this factorisation (6 per block column this member owns)
    double* labuf = nullptr;               // global [block columns][GRP_LA_SLOT]: look-ahead accumulators (+ the rows' right-hand side) published by their owner

    int* abort_status = nullptr;           // &status[problem]: a wait that expires reports HIPDRT_QP_ABORTED there
    // A bounded wait that expires INSIDE a running factorisation (it can only do so through a protocol error, not through the
    // environment -- residency is settled at the start rendezvous) used to trap, which kills the HIP context and the host process
    // with it.  Now the wavefront reports the problem as aborted (status + reason 3 in sync word [4]) and ends: a terminated
    // wavefront leaves its workgroup's barriers, the partner members run into their own time-outs at their next hand-over and end
    // the same way, the kernel returns, and the launcher repeats the problem on one workgroup from its untouched inputs (the
    // group kernel writes x, the iteration counts and the status only at the very end).
    __device__ __forceinline__ void trap_if(bool c) const {
        if (c) {
            __hip_atomic_store(abort_status, HIPDRT_QP_ABORTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gs[4], 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_endpgm();
        }
    }

    // ---- group barrier: every member's stores complete, then one arrival per member on a monotonic counter -------------
    __device__ __forceinline__ void group_sync() {
        PROF_DECL
        // every wavefront drains its own vector stores first: on gfx950 (no threadgroup-split mode) __syncthreads() is
        // `s_waitcnt lgkmcnt(0); s_barrier` -- a workgroup-scope release does not wait for vmcnt -- and the arrival below is a
        // relaxed atomic, so without this wait nothing would keep a member's tile or state stores in front of the counter
        // another member polls (tests/test_isa_hazards.py checks the assembly for it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (G > 1) {
            ++gepoch;
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(&gs[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = G * gepoch;
                int spins = 0;
                while (__hip_atomic_load(&gs[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(2);
                    trap_if(++spins > kSpinLimit);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // L1 invalidate: plain loads below see the other members' data
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        PROF(4);
    }

    // Tile rows belong to members in PAIRS (the two tile rows of a block row: member (T / 2) % G), so that the two look-ahead
    // rows of a block column have one owner.  Inside the member they are dealt to the row wavefronts from the bottom up,
    // round-robin in an order that fills the four SIMDs evenly (wavefront w runs on SIMD w % 4; SIMD 0 / 1 also carry the
    // chain / the look-ahead wavefront, which use the matrix pipe little, so wavefronts 4 and 5 come first): the rows still
    // active at any block column are a prefix of that deal.
    __device__ __forceinline__ bool mine(int T) const { return (T >> 1) % G == g; }
    __device__ __forceinline__ void build_owner() {
        if (threadIdx.x != 0) return;
        const int ntr = (n + 15) >> 4;
        const unsigned char order[6] = {2, 3, 5, 6, 7, 4};
        int k = 0;
        for (int T = ntr - 1; T >= 0; --T) {
            if (!mine(T)) { owner[T] = 0; continue; }
            owner[T] = order[k];
            k = k == 5 ? 0 : k + 1;
        }
    }
    // the member that accumulates the look-ahead tiles of block column jb (rows 2 jb + 2, 2 jb + 3: its own rows)
    __device__ __forceinline__ bool la_owner(int jb) const { return (jb + 1) % G == g; }
    // does this member own a tile row below the look-ahead rows of block column jb?  (first pair >= jb + 2 that is its own)
    __device__ __forceinline__ bool has_rows_below(int jb, int ntr) const {
        const int p0 = jb + 2, d = ((g - p0) % G + G) % G;
        return 2 * (p0 + d) < ntr;
    }

    // ---- progress words ----------------------------------------------------------------------------------------------------
    __device__ __forceinline__ int prog_value(int jb) const { return fidx * 256 + jb; }
    // tile row T of THIS member complete through block column jb - 1 (set by the row wavefront that stores it)
    __device__ __forceinline__ void wait_row(int T, int jb) const {
        int spins = 0;
        while (lds_peek8((const void*)(rowdone + T)) < jb) { __builtin_amdgcn_s_sleep(1); trap_if(++spins > kSpinLimit); }
        asm volatile("" ::: "memory");
    }
    // rows tb, tb + 1 of the current diagonal block: stored by this member's own look-ahead wavefront
    __device__ __forceinline__ void wait_diag_rows(int tb, bool two, int jb) const {
        int spins = 0;
        while (lds_peek8((const void*)(rowdone + tb)) < jb || (two && lds_peek8((const void*)(rowdone + tb + 1)) < jb)) {
            __builtin_amdgcn_s_sleep(1);
            trap_if(++spins > kSpinLimit);
        }
        asm volatile("" ::: "memory");
    }

    // ---- row masks: bit r = tile row tb + 4 + r (r < nsq) is this wavefront's ------------------------------------------------
    struct RowMask { unsigned long long m0, m1, m2, m3; };
    __device__ __forceinline__ RowMask my_rows(int jb, int wv, int lane, int ntr) const {
        const int tb = 2 * jb;
        const int nsq = ntr - (tb + 4) > 0 ? ntr - (tb + 4) : 0;
        auto word = [&](int w) -> unsigned long long {
            if (64 * w >= nsq) return 0ull;
            const int r = lane + 64 * w;
            return __ballot(r < nsq && owner[r < nsq ? tb + 4 + r : 0] == wv);
        };
        RowMask k;
        k.m0 = word(0); k.m1 = word(1); k.m2 = word(2); k.m3 = word(3);
        return k;
    }
    __device__ __forceinline__ int pop_row(RowMask& k, int tb) const {
        int r = -1;
        if (k.m0) { r = __builtin_ctzll(k.m0); k.m0 &= k.m0 - 1; }
        else if (k.m1) { r = 64 + __builtin_ctzll(k.m1); k.m1 &= k.m1 - 1; }
        else if (k.m2) { r = 128 + __builtin_ctzll(k.m2); k.m2 &= k.m2 - 1; }
        else if (k.m3) { r = 192 + __builtin_ctzll(k.m3); k.m3 &= k.m3 - 1; }
        return r < 0 ? -1 : tb + 4 + r;
    }
    static __device__ __forceinline__ int count_rows(const RowMask& k) {
        return __builtin_popcountll(k.m0) + __builtin_popcountll(k.m1) + __builtin_popcountll(k.m2) + __builtin_popcountll(k.m3);
    }

    // ---- the look-ahead tiles ------------------------------------------------------------------------------------------------
    // The seven tiles wavefront 1 of the batch kernel accumulates in one loop -- (R2|R3, tb|tb+1) and the next diagonal block
    // (R2,R2), (R3,R2), (R3,R3), over the whole inner dimension -- are the critical path of a block column here, and computed
    // by every member they would cost each member 14 MFMAs per half-chunk whatever the group size (measured: a group of 16
    // no faster than one of 8).  Only the member that OWNS rows R2, R3 accumulates them (la_owner), one tile per wavefront
    // 1..7 (same order of summation as the batch kernel: same bits), wavefronts 2..7 hand theirs to wavefront 1 through LDS,
    // and wavefront 1 publishes the seven raw accumulators in labuf[jb] + the laprog word.  Every member's wavefront 1
    // fetches them (sc1 loads: the XCD's L2) while its chain wavefront still factors the current diagonal block, and does the
    // last rank-32 update, the panel solve and the staging itself after barrier (A) -- redundantly, from identical bits.
    // ---- operand rings -------------------------------------------------------------------------------------------------------
    // Every rank-k loop of this kernel streams its operands through an eight-slot register ring, one half-chunk (1 KB per
    // operand) per slot, requested SEVEN slots ahead with hand-issued loads and counted waits (qp_resident.hpp, gload16 /
    // vm_wait).  The loads are plain, L1-cached ones although most of the B operand (rows tb, tb+1) was stored by another
    // member: this CU reads a tile of L for the first time in a factorisation only after the tile is final (its rows were
    // waited for by the member that published the look-ahead accumulators this member fetched before it passed barrier (A)
    // of the previous block column), and group_sync() invalidated the L1 after the previous factor's sweeps -- so no stale
    // line can be hit, and the six row wavefronts of a member, which walk the same B chunks, share them through the L1.
    // (An earlier form staged the shared rows through LDS with LDS-DMA loads and a software barrier per four half-chunks
    // among the seven wavefronts: 3.9k cycles per super-step of which 1.1k matrix work -- DESIGN.md section 7.)
    //
    // Registers written by a hand-issued load stay allocated until the load has landed only if something reads them after
    // the wait: slots past the end of the inner dimension are re-reads of the last chunk, and are "consumed" by an empty asm.

    // rows of this pass: acc[u][c] += chunk(T[u], k) chunk(tb + c, k)' over the finished block columns
    // (NR = rows actually present in this pass: a slot carries NR + 2 loads -- a wavefront pays ~100 cycles per 1 KB load, so the
    // single-row case, the usual one once a member has fewer rows than wavefronts, does not request a second row it does not have)
    template <int NR>
    __device__ __forceinline__ void ring_rows(int jb, int ntr, int lane, const int (&T)[GRP_RMAXT], const bool (&act)[GRP_RMAXT],
                                              v4d (&acc)[GRP_RMAXT][2]) const {
        const int li = lane & 15, kq = lane >> 4, fo = li * 4 + kq;
        const int tb = 2 * jb, nk2 = 4 * jb, klast = nk2 - 1;
        const bool two = tb + 1 < ntr;
        const char* rb0 = uniform_ptr(tile2(tb, 0));
        const char* rb1 = uniform_ptr(tile2(two ? tb + 1 : tb, 0));
        const char* ra[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) ra[u] = uniform_ptr(tile2(act[u] ? T[u] : tb, 0));
        const unsigned voff = (unsigned)fo * 16u;
        struct Sl { v2d b0, b1, a[NR]; };
        auto load = [&](Sl& s_, int k2) {
            const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
            s_.b0 = gload16(uniform_ptr(rb0 + o), voff); s_.b1 = gload16(uniform_ptr(rb1 + o), voff);
#pragma unroll
            for (int u = 0; u < NR; ++u) s_.a[u] = gload16(uniform_ptr(ra[u] + o), voff);
        };
        // Two accumulators per tile: the products of the first four columns of a half-chunk go onto the source tile, those of
        // the last four into a second accumulator from zero, added at the end -- with one row a wavefront has only two tiles, and a
        // v_mfma_f64_16x16x4 that has to wait for the one two instructions back costs ~100 cycles instead of 64 (400 cycles per
        // half-chunk measured for one row against 256 of issue).  The same for two rows, so that a tile's bits do not depend on
        // how many rows its wavefront happened to have.
        v4d accy[NR][2];
#pragma unroll
        for (int u = 0; u < NR; ++u) { accy[u][0] = (v4d){0, 0, 0, 0}; accy[u][1] = (v4d){0, 0, 0, 0}; }
        auto mult = [&](const Sl& s_, int k2) {
            if (k2 < nk2) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    if (act[u]) {
                        acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.b0.x, s_.a[u].x, acc[u][0], 0, 0, 0);
                        if (two) acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.b1.x, s_.a[u].x, acc[u][1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    if (act[u]) {
                        accy[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.b0.y, s_.a[u].y, accy[u][0], 0, 0, 0);
                        if (two) accy[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.b1.y, s_.a[u].y, accy[u][1], 0, 0, 0);
                    }
                }
            } else {
                asm volatile("" :: "v"(s_.b0), "v"(s_.b1));
#pragma unroll
                for (int u = 0; u < NR; ++u) asm volatile("" :: "v"(s_.a[u]));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        constexpr int LPS = 2 + NR;                  // loads per slot
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // the newest two chunks of rows tb, tb+1 were stored by this member's look-ahead wavefront in the previous block
        // column: wait for them just before the first request that reaches that far
        bool gated = false;
        if (GRP_ROW_RING - 2 >= nk2 - 4) { wait_diag_rows(tb, two, jb); gated = true; }
        constexpr int NS = GRP_ROW_RING;
        Sl r[NS];
#pragma unroll
        for (int i_ = 0; i_ < NS - 1; ++i_) load(r[i_], i_);
        for (int kb = 0; kb < nk2; kb += NS) {
            if (!gated && kb + 2 * NS - 2 >= nk2 - 4) { wait_diag_rows(tb, two, jb); gated = true; }
#pragma unroll
            for (int i_ = 0; i_ < NS; ++i_) {
                load(r[(i_ + NS - 1) % NS], kb + i_ + NS - 1);
                vm_wait<(NS - 1) * LPS>();
                mult(r[i_], kb + i_);
            }
        }
        vm_wait<0>();
#pragma unroll
        for (int i_ = 0; i_ < NS; ++i_) mult(r[i_], nk2);       // (pins)
#pragma unroll
        for (int u = 0; u < NR; ++u) { acc[u][0] += accy[u][0]; acc[u][1] += accy[u][1]; }
    }

    // The look-ahead accumulators of block column jbn (tile rows R2 = 2 jbn + 2, R3 = R2 + 1 against rows tbn = 2 jbn, tbn + 1
    // and themselves), in the member that owns R2, R3.  The seven tiles are products of chunks of FOUR tile rows, and a
    // single CU draws only ~20 bytes per cycle from beyond its L2, so (a) every operand chunk is requested once per member:
    // the inner dimension is cut into contiguous slices, one per row wavefront 2..7, each accumulating all seven tiles over
    // its slice, the partial sums added up in LDS in wavefront order (la_reduce: a fixed order, independent of the group
    // size); and (b) the bulk of it happens ONE BLOCK COLUMN EARLY: during block column jbn - 1 the owner accumulates the
    // "old range" -- block columns 0 .. jbn - 3, half-chunks [0, 4 (jbn - 2)) -- which needs nothing of the two newest
    // columns; when column jbn starts only its "new range" (the eight half-chunks of block columns jbn - 2, jbn - 1) is
    // left, one 32-load step of wavefront 1 (la_new_range), and every other member gets the accumulators while its chain
    // still works on the diagonal block.  (Accumulated within block column jbn itself, the seven-tile loop was what every
    // member waited for: 29 k of 46 k cycles per block column at n = 1078.  With the old range reaching up to column
    // jbn - 2 the owners serialise: each needs the previous owner's newest stores before it can start, 51 k per column.)
    struct LaAcc { v4d p20, p21, p30, p31, e11, e21, e22; };
    // old range of la(jbn), slice of row wavefront w = 2..7; called during block column jbn - 1 >= 1
    __device__ __forceinline__ void ring_la_slice(int jbn, int ntr, int lane, int w, LaAcc& A) const {
        const int li = lane & 15, kq = lane >> 4, fo = li * 4 + kq;
        const int jb = jbn;                                  // (names as in the formulas above)
        const int tb = 2 * jb, nk2 = 4 * (jb - 2);
        const bool v3 = tb + 3 < ntr;
        // rows R2, R3 (this member's own), complete through block column jbn - 3: their row wavefronts said so
        wait_row(tb + 2, jb - 2);
        if (v3) wait_row(tb + 3, jb - 2);
        // rows tbn, tbn + 1: the look-ahead rows of the column in progress, another member's in general -- through block
        // column jbn - 3 their owner published them a whole block column ago
        if (mine(tb)) {
            wait_row(tb, jb - 2);
            wait_row(tb + 1, jb - 2);
        } else {
            const int want = prog_value(jb - 2);
            for (int spins = 0;; ) {
                const int a_ = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&gs[16 + tb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const int b_ = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&gs[16 + tb + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (a_ >= want && b_ >= want) break;
                __builtin_amdgcn_s_sleep(1);
                trap_if(++spins > kSpinLimit);
            }
            asm volatile("" ::: "memory");
        }
        const int k0 = (w - 2) * nk2 / 6, k1 = (w - 1) * nk2 / 6, klast = k1 - 1;      // this wavefront's half-chunks
        if (k1 <= k0) return;
        const char* r0p = uniform_ptr(tile2(tb, 0));
        const char* r1p = uniform_ptr(tile2(tb + 1, 0));
        const char* r2p = uniform_ptr(tile2(tb + 2, 0));
        const char* r3p = uniform_ptr(tile2(v3 ? tb + 3 : tb + 2, 0));          // (padding row: its tiles are replaced by the caller)
        const unsigned voff = (unsigned)fo * 16u;
        struct Sl { v2d c0, c1, c2, c3; };
        auto load = [&](Sl& s_, int k2) {
            const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
            s_.c0 = gload16(uniform_ptr(r0p + o), voff); s_.c1 = gload16(uniform_ptr(r1p + o), voff);
            s_.c2 = gload16(uniform_ptr(r2p + o), voff); s_.c3 = gload16(uniform_ptr(r3p + o), voff);
        };
        auto mult = [&](const Sl& s_, int k2) {
            if (k2 < k1) {
                A.p20 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c0.x, s_.c2.x, A.p20, 0, 0, 0);
                A.p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c1.x, s_.c2.x, A.p21, 0, 0, 0);
                A.p30 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c0.x, s_.c3.x, A.p30, 0, 0, 0);
                A.p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c1.x, s_.c3.x, A.p31, 0, 0, 0);
                A.e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c2.x, s_.c2.x, A.e11, 0, 0, 0);
                A.e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c2.x, s_.c3.x, A.e21, 0, 0, 0);
                A.e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c3.x, s_.c3.x, A.e22, 0, 0, 0);
                A.p20 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c0.y, s_.c2.y, A.p20, 0, 0, 0);
                A.p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c1.y, s_.c2.y, A.p21, 0, 0, 0);
                A.p30 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c0.y, s_.c3.y, A.p30, 0, 0, 0);
                A.p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c1.y, s_.c3.y, A.p31, 0, 0, 0);
                A.e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c2.y, s_.c2.y, A.e11, 0, 0, 0);
                A.e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c2.y, s_.c3.y, A.e21, 0, 0, 0);
                A.e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(s_.c3.y, s_.c3.y, A.e22, 0, 0, 0);
            } else {
                asm volatile("" :: "v"(s_.c0), "v"(s_.c1), "v"(s_.c2), "v"(s_.c3));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NS = GRP_LA_RING;
        Sl r[NS];
#pragma unroll
        for (int i_ = 0; i_ < NS - 1; ++i_) load(r[i_], k0 + i_);
        for (int kb = k0; kb < k1; kb += NS) {
#pragma unroll
            for (int i_ = 0; i_ < NS; ++i_) {
                load(r[(i_ + NS - 1) % NS], kb + i_ + NS - 1);
                vm_wait<4 * (NS - 1)>();
                mult(r[i_], kb + i_);
            }
        }
        vm_wait<0>();
#pragma unroll
        for (int i_ = 0; i_ < NS; ++i_) mult(r[i_], k1);       // (pins)
    }
    // new range of la(jb) on top of the totals (wavefront 1 of the owner, at the start of block column jb): the half-chunks
    // of block columns jb - 2 and jb - 1.  The newest tiles this member stored itself a moment ago (rows tb, tb+1: this
    // wavefront's panel store; R2, R3: its row wavefronts'); those of rows tb, tb+1 in column jb - 2 their owner stored before
    // it published the look-ahead accumulators this member fetched in the previous block column.
    // four half-chunks of the four rows: sixteen loads in flight, then 56 MFMAs
    struct LaChunk4 { v2d c[4][4]; };
    __device__ __forceinline__ void la_load4(LaChunk4& q, const char* const (&rp)[4], int k0, unsigned voff) const {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) q.c[h][r] = gload16(uniform_ptr(rp[r] + (size_t)(k0 + h) * 1024), voff);
    }
    static __device__ __forceinline__ void la_mult4(const LaChunk4& q, LaAcc& A) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            A.p20 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][0].x, q.c[h][2].x, A.p20, 0, 0, 0);
            A.p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][1].x, q.c[h][2].x, A.p21, 0, 0, 0);
            A.p30 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][0].x, q.c[h][3].x, A.p30, 0, 0, 0);
            A.p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][1].x, q.c[h][3].x, A.p31, 0, 0, 0);
            A.e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][2].x, q.c[h][2].x, A.e11, 0, 0, 0);
            A.e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][2].x, q.c[h][3].x, A.e21, 0, 0, 0);
            A.e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][3].x, q.c[h][3].x, A.e22, 0, 0, 0);
            A.p20 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][0].y, q.c[h][2].y, A.p20, 0, 0, 0);
            A.p21 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][1].y, q.c[h][2].y, A.p21, 0, 0, 0);
            A.p30 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][0].y, q.c[h][3].y, A.p30, 0, 0, 0);
            A.p31 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][1].y, q.c[h][3].y, A.p31, 0, 0, 0);
            A.e11 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][2].y, q.c[h][2].y, A.e11, 0, 0, 0);
            A.e21 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][2].y, q.c[h][3].y, A.e21, 0, 0, 0);
            A.e22 = __builtin_amdgcn_mfma_f64_16x16x4f64(q.c[h][3].y, q.c[h][3].y, A.e22, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void la_new_range(int jb, int ntr, int lane, LaAcc& A) const {
        const int li = lane & 15, kq = lane >> 4, fo = li * 4 + kq;
        const int tb = 2 * jb;
        const bool v3 = tb + 3 < ntr;
        wait_row(tb + 2, jb);
        if (v3) wait_row(tb + 3, jb);
        wait_diag_rows(tb, true, jb);
        const char* const rp[4] = {uniform_ptr(tile2(tb, 0)), uniform_ptr(tile2(tb + 1, 0)), uniform_ptr(tile2(tb + 2, 0)),
                                   uniform_ptr(tile2(v3 ? tb + 3 : tb + 2, 0))};
        const unsigned voff = (unsigned)fo * 16u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // block column jb - 2 first (same order of summation as one pass over the inner dimension), then block column jb - 1
        // through the same sixteen registers (both in flight at once spills: this wavefront also carries the seven
        // accumulators through the panel solve)
        LaChunk4 qa;
#pragma unroll 1
        for (int c_ = jb >= 2 ? jb - 2 : 0; c_ < jb; ++c_) {
            la_load4(qa, rp, 4 * c_, voff);
            vm_wait<0>();
            __builtin_amdgcn_sched_barrier(0);
            la_mult4(qa, A);
        }
    }
    // Old-range partial sums -> LDS, in wavefront order: wavefront 1 writes the source tiles (init), wavefronts 2 .. 7 add
    // theirs.  `base` = 7 x the old ranges this member accumulated before this one: lacnt counts the whole factorisation's.
    __device__ __forceinline__ void la_reduce(int lane, int w, int base, const LaAcc& A, const LaAcc* init) const {
        for (int spins = 0; lds_peek32(lacnt) < base + (w - 1);) {
            __builtin_amdgcn_s_sleep(1);
            trap_if(++spins > kSpinLimit);
        }
        asm volatile("" ::: "memory");
        v4d* lt = reinterpret_cast<v4d*>(latile) + lane;
        if (init) {
            lt[0] = init->p20 + A.p20;   lt[64] = init->p21 + A.p21;   lt[128] = init->p30 + A.p30;  lt[192] = init->p31 + A.p31;
            lt[256] = init->e11 + A.e11; lt[320] = init->e21 + A.e21; lt[384] = init->e22 + A.e22;
        } else {
            lt[0] += A.p20;   lt[64] += A.p21;  lt[128] += A.p30; lt[192] += A.p31;
            lt[256] += A.e11; lt[320] += A.e21; lt[384] += A.e22;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(lacnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // =========================================================================================================================
    __device__ __forceinline__ bool factor() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ntr = (n + 15) >> 4;
        ++fidx;
        group_sync();                            // every member has finished sweeping the previous factor: its tiles may go
        for (int i = tid; i < GRP_OWN; i += RT) rowdone[i] = 0;
        if (tid == 0) { sm.flag[1] = 0; *lacnt = 0; }
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
        bool ok;
        if (wv == 0) ok = factor_chain();
        else if (wv == 1) ok = factor_lookahead();
        else ok = factor_rows(wv);
        if (ok) group_sync();                    // every tile of every member in memory (and this CU's L1 invalidated)
        return ok;                               // (a failed factorisation left through barrier (A) in every wavefront of every member)
    }

    // ======== wavefront 0: factor + invert the diagonal blocks (redundantly in every member) =====================================
    __device__ __forceinline__ bool factor_chain() {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;
        const int nblk = (n + NB - 1) / NB;
        double* U = sm.U;
        v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
        v4d* const img22 = img21 + 64;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            PROF_DECL
            TS_DECL
            TS(0);
            for (int spins = 0; lds_peek32(&sm.flag[1]) < jb;) {                        // diagonal block jb staged
                __builtin_amdgcn_s_sleep(1);
                trap_if(++spins > kSpinLimit);
            }
            asm volatile("" ::: "memory");
            PROF(2);
            TS(1);
            bool ok = cholinv16_dsc(j0, 0);
            TS(2);
            const v4d d21 = img21[lane];
            v4d d22 = img22[lane];
            v4d x21 = (v4d){0, 0, 0, 0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                x21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[(size_t)(j0 + li) * PLD + 4 * s_ + kq], d21[s_], x21, 0, 0, 0);
            // L21: into the LDS scratch block for this block column's first pass, and as tile (tb + 1, 2 jb) of L (a slot nothing
            // else uses) for later passes, which may run while this wavefront already factors the next diagonal block.  (Every
            // member stores the same bits there.)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) sm.t21[li * DLD + kq + 4 * rg] = x21[rg];
            {
                double2* d0 = const_cast<double2*>(tile2(2 * jb + 1, 2 * jb)) + fo;
                d0[0] = make_double2(x21[0], x21[1]);
                d0[64] = make_double2(x21[2], x21[3]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                d22 = __builtin_amdgcn_mfma_f64_16x16x4f64(sm.t21[li * DLD + 4 * s_ + kq], x21[s_], d22, 0, 0, 0);
            ok = cholinv16(d22, j0 + 16, 16) && ok;
            PROF(12);
            TS(5);
            if (lane == 0) sm.flag[0] = ok ? 0 : 1;
            __syncthreads();                                    // (A) W1, L21, W2 published
            PROF(1);
            TS(6);
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
                // fused forward substitution: y_j = M_j b_j (b_j: published by the rows' owner, last update by wavefront 1)
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
            }
            // (A2): separates this block column's readers of the L21 scratch block from its next writer, and publishes y_j
            TS(8);
            lds_barrier();                                      // (A2)
            PROF(3);
            TS(7);
            TS_PRINT();
        }
        return true;
    }

    // ======== wavefront 1: the two tile rows R2 = tb+2, R3 = tb+3 of the NEXT diagonal block (redundantly in every member) =======
    __device__ __forceinline__ bool factor_lookahead() {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* U = sm.U;
        v4d* const img21 = reinterpret_cast<v4d*>(sm.img);
        v4d* const img22 = img21 + 64;
        int laown = 0;                               // block columns whose look-ahead tiles this member accumulated
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int tb = j0 >> 4;
            const int R2 = tb + 2, R3 = tb + 3;
            const bool v2 = R2 < ntr, v3 = R3 < ntr;
            PROF_DECL
            if (jb > 0) {
                // the tiles this wavefront stored in the previous column (rows tb, tb+1 now) are in memory: tell the row wavefronts
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < 2) rowdone[tb + lane] = (unsigned char)jb;
            }
            PROFW(20);
            TS_DECL
            TS(0);
            v4d p20, p21, p30, p31, e11, e21, e22;
            if (v2) {
                if (jb == 0) {
                    p20 = init_tile(R2, tb, ntr, fo, li, kq);      p21 = init_tile(R2, tb + 1, ntr, fo, li, kq);
                    p30 = init_tile(R3, tb, ntr, fo, li, kq);      p31 = init_tile(R3, tb + 1, ntr, fo, li, kq);
                    e11 = init_tile(R2, R2, ntr, fo, li, kq);      e21 = init_tile(R3, R2, ntr, fo, li, kq);
                    e22 = init_tile(R3, R3, ntr, fo, li, kq);
                } else {
                    const bool la = la_owner(jb);
                    // (order: this member's finished old-range totals out of LDS first -- the buffer is about to be reused --,
                    // then the source tiles of the NEXT look-ahead block into it if this member accumulates that one, so that its
                    // row wavefronts never wait for what follows: the new range and the publication, or the owner's accumulators)
                    LaAcc T_;
                    if (la) {
                        // the old range was summed up in LDS during the previous block column (jb < 3: there is none, the
                        // totals are the source tiles); the new range on top of it
                        if (jb >= 3) {
                            for (int spins = 0; lds_peek32(lacnt) < 7 * laown;) {
                                __builtin_amdgcn_s_sleep(1);
                                trap_if(++spins > kSpinLimit);
                            }
                            asm volatile("" ::: "memory");
                            const v4d* lt = reinterpret_cast<const v4d*>(latile) + lane;
                            T_.p20 = lt[0]; T_.p21 = lt[64]; T_.p30 = lt[128]; T_.p31 = lt[192];
                            T_.e11 = lt[256]; T_.e21 = lt[320]; T_.e22 = lt[384];
                        } else {
                            T_.p20 = init_tile(R2, tb, ntr, fo, li, kq);      T_.p21 = init_tile(R2, tb + 1, ntr, fo, li, kq);
                            T_.p30 = init_tile(R3, tb, ntr, fo, li, kq);      T_.p31 = init_tile(R3, tb + 1, ntr, fo, li, kq);
                            T_.e11 = init_tile(R2, R2, ntr, fo, li, kq);      T_.e21 = init_tile(R3, R2, ntr, fo, li, kq);
                            T_.e22 = init_tile(R3, R3, ntr, fo, li, kq);
                        }
                    }
                    if (jb >= 2 && tb + 4 < ntr && la_owner(jb + 1)) {
                        LaAcc I_, Z_;
                        Z_.p20 = Z_.p21 = Z_.p30 = Z_.p31 = Z_.e11 = Z_.e21 = Z_.e22 = (v4d){0, 0, 0, 0};
                        I_.p20 = init_tile(R2 + 2, tb + 2, ntr, fo, li, kq);   I_.p21 = init_tile(R2 + 2, tb + 3, ntr, fo, li, kq);
                        I_.p30 = init_tile(R3 + 2, tb + 2, ntr, fo, li, kq);   I_.p31 = init_tile(R3 + 2, tb + 3, ntr, fo, li, kq);
                        I_.e11 = init_tile(R2 + 2, R2 + 2, ntr, fo, li, kq);   I_.e21 = init_tile(R3 + 2, R2 + 2, ntr, fo, li, kq);
                        I_.e22 = init_tile(R3 + 2, R3 + 2, ntr, fo, li, kq);
                        la_reduce(lane, 1, 7 * laown, Z_, &I_);
                        ++laown;
                    }
                    if (la) {
                        PROFW(21);
                        TS(1);
                        la_new_range(jb, ntr, lane, T_);
                        PROFW(22);
                        TS(3);
                        p20 = T_.p20; p21 = T_.p21; p30 = T_.p30; p31 = T_.p31; e11 = T_.e11; e21 = T_.e21; e22 = T_.e22;
                        if (G > 1) {
                            // publish the seven raw accumulators (register images) for the other members
                            v4d* dst = reinterpret_cast<v4d*>(labuf + (size_t)jb * GRP_LA_SLOT) + lane;
                            dst[0] = p20; dst[64] = p21; dst[128] = p30; dst[192] = p31; dst[256] = e11; dst[320] = e21; dst[384] = e22;
                            // ... and the rows' right-hand-side entries as this member's row wavefronts left them (block columns
                            // 0 .. jb - 1 applied: la_new_range waited for those rows)
                            if (fwd && lane < 32) labuf[(size_t)jb * GRP_LA_SLOT + 7 * 256 + lane] = sm.vec[R2 * 16 + lane];
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            if (lane == 0) __hip_atomic_store(&gs[3], prog_value(jb + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        TS(4);
                    } else {
                        // the owner's accumulators
                        PROFW(21);
                        TS(1);
                        const int want = prog_value(jb + 1);
                        for (int spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(&gs[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < want;) {
                            __builtin_amdgcn_s_sleep(1);
                            trap_if(++spins > kSpinLimit);
                        }
                        asm volatile("" ::: "memory");
                        PROFW(22);
                        TS(3);
                        const char* src = uniform_ptr(labuf + (size_t)jb * GRP_LA_SLOT);
                        const unsigned vo = (unsigned)lane * 32u;
                        v2d t_[14];
#pragma unroll
                        for (int q = 0; q < 7; ++q) {
                            t_[2 * q] = gload16_sc1(src + q * 2048, vo);
                            t_[2 * q + 1] = gload16_sc1(src + q * 2048 + 16, vo);
                        }
                        vm_wait<0>();
#pragma unroll
                        for (int q = 0; q < 14; ++q) asm volatile("" : "+v"(t_[q]));
                        auto cat = [&](int q) { return (v4d){t_[2 * q].x, t_[2 * q].y, t_[2 * q + 1].x, t_[2 * q + 1].y}; };
                        p20 = cat(0); p21 = cat(1); p30 = cat(2); p31 = cat(3); e11 = cat(4); e21 = cat(5); e22 = cat(6);
                        if (fwd && lane < 32) {
                            // the owner's right-hand-side entries of rows R2, R3 replace this member's stale ones (agent-scope load:
                            // past the L1, like the accumulators)
                            unsigned long long* rp_ = reinterpret_cast<unsigned long long*>(labuf + (size_t)jb * GRP_LA_SLOT + 7 * 256 + lane);
                            const unsigned long long bits = __hip_atomic_load(rp_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            sm.vec[R2 * 16 + lane] = __longlong_as_double((long long)bits);
                        }
                        TS(4);
                    }
                    if (!v3) {
                        // R3 is pure padding: no panel tiles, identity diagonal
                        p30 = (v4d){0, 0, 0, 0}; p31 = (v4d){0, 0, 0, 0}; e21 = (v4d){0, 0, 0, 0};
                        e22 = init_tile(R3, R3, ntr, fo, li, kq);
                    }
                }
            }
            TS(5);
            __syncthreads();                                    // (A)
            PROFW(23);
            TS(6);
            if (sm.flag[0]) return false;
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
                // (every member stores the same bits into the shared factor: its own row wavefronts and its own next look-ahead
                // pass read them back, and the sweeps read them after the group barrier)
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
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *(volatile int*)&sm.flag[1] = jb + 1;      // diagonal block jb + 1 staged: the chain may start
                lds_barrier();                                  // (A2)
                if (fwd) {
                    // block column jb's update of the right-hand side of rows R2, R3 (every member, identical bits)
                    fwd_update(x20, x21_, R2, j0, li, kq);
                    if (v3) fwd_update(x30, x31, R3, j0, li, kq);
                }
            } else {
                lds_barrier();                                  // (A2)
            }
            PROFW(24);
            TS(7);
            TS_PRINT();
        }
        return true;
    }

    // ======== wavefronts 2..7: this member's rows below =============================================================================
    __device__ __forceinline__ bool factor_rows(int wv) {
        const int lane = fresh_lane(), li = lane & 15, kq = lane >> 4;
        const int fo = li * 4 + kq;
        const int nblk = (n + NB - 1) / NB;
        const int ntr = (n + 15) >> 4;
        double* U = sm.U;
        int laown = 0;                               // block columns whose look-ahead tiles this member accumulated
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int tb = j0 >> 4;
            RowMask mask = my_rows(jb, wv, lane, ntr);
            PROF_DECL
            TS_DECL
            TS(0);
            if (jb > 0) {
                // everything this wavefront stored in the previous block column is in memory (the wait also covers the source
                // tiles requested after those stores).  The rows somebody else reads next: tb + 2, tb + 3 (the look-ahead rows
                // from now on: wavefront 1 of this member, and the member that accumulates the NEXT look-ahead block, through
                // the global word) and tb + 4, tb + 5 (this member's own old-range pass below)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < 4) {
                    const int T = tb + 2 + lane;
                    if (T < ntr && owner[T] == wv) {
                        rowdone[T] = (unsigned char)jb;
                        if (G > 1) __hip_atomic_store(&gs[16 + T], prog_value(jb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (wv == 2) PROFW(26);
            TS(1);
            // the old range of the NEXT block column's look-ahead accumulators, if this member owns those rows: this
            // wavefront's slice, added to the sums in LDS (ring_la_slice)
            if (jb >= 2 && tb + 4 < ntr && la_owner(jb + 1)) {
                LaAcc A_;
                A_.p20 = A_.p21 = A_.p30 = A_.p31 = A_.e11 = A_.e21 = A_.e22 = (v4d){0, 0, 0, 0};
                ring_la_slice(jb + 1, ntr, lane, wv, A_);
                la_reduce(lane, wv, 7 * laown, A_, nullptr);
                ++laown;
            }
            if (wv == 2) PROFW(27);
            TS(4);
            const int mine = count_rows(mask);
            const int npass = mine > GRP_RMAXT ? (mine + GRP_RMAXT - 1) / GRP_RMAXT : 1;
#pragma unroll 1
            for (int ps = 0; ps < npass; ++ps) {
                int T[GRP_RMAXT];
                bool act[GRP_RMAXT];
#pragma unroll
                for (int u = 0; u < GRP_RMAXT; ++u) {
                    const int t_ = pop_row(mask, tb);
                    T[u] = t_ >= 0 ? t_ : nch;
                    act[u] = t_ >= 0;
                }
                v4d acc[GRP_RMAXT][2];
#pragma unroll
                for (int u = 0; u < GRP_RMAXT; ++u)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[u][ct] = act[u] ? init_tile(T[u], tb + ct, ntr, fo, li, kq) : (v4d){0, 0, 0, 0};
                TS(9);
                if (jb > 0 && act[0]) {
                    if (act[GRP_RMAXT - 1]) ring_rows<GRP_RMAXT>(jb, ntr, lane, T, act, acc);
                    else ring_rows<1>(jb, ntr, lane, T, act, acc);
                }
                if (ps == 0) {
                    if (wv == 2) PROFW(28);
                    TS(5);
                    __syncthreads();                            // (A) W1, L21, W2 published by wavefront 0
                    if (wv == 2) PROFW(29);
                    TS(6);
                    if (sm.flag[0]) return false;
                }
                if (act[0]) {
                    double wn1[4], l21[4], wn2[4];
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) {
                        wn1[s_] = -U[(size_t)(j0 + li) * PLD + 4 * s_ + kq];
                        wn2[s_] = -U[(size_t)(j0 + 16 + li) * PLD + 16 + 4 * s_ + kq];
                    }
                    if (ps == 0) {
#pragma unroll
                        for (int s_ = 0; s_ < 4; ++s_) l21[s_] = sm.t21[li * DLD + 4 * s_ + kq];
                    } else {
                        // later passes: from the copy in L (the LDS block may already hold the next block's L21)
                        const double2* t_ = tile2(tb + 1, 2 * jb) + fo;
                        const double2 h0 = t_[0], h1 = t_[64];
                        l21[0] = h0.x; l21[1] = h0.y; l21[2] = h1.x; l21[3] = h1.y;
                    }
                    v4d x1[GRP_RMAXT], x2[GRP_RMAXT];
#pragma unroll
                    for (int u = 0; u < GRP_RMAXT; ++u) { x1[u] = (v4d){0, 0, 0, 0}; x2[u] = (v4d){0, 0, 0, 0}; }
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int u = 0; u < GRP_RMAXT; ++u)
                            x1[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn1[s_], acc[u][0][s_], x1[u], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int u = 0; u < GRP_RMAXT; ++u)
                            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s_], x1[u][s_], acc[u][1], 0, 0, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                        for (int u = 0; u < GRP_RMAXT; ++u)
                            x2[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(wn2[s_], acc[u][1][s_], x2[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < GRP_RMAXT; ++u) {
                        if (act[u]) {
                            double2* d0 = const_cast<double2*>(tile2(T[u], 2 * jb)) + fo;
                            d0[0] = make_double2(x1[u][0], x1[u][1]);
                            d0[64] = make_double2(x1[u][2], x1[u][3]);
                            d0[128] = make_double2(x2[u][0], x2[u][1]);
                            d0[192] = make_double2(x2[u][2], x2[u][3]);
                        }
                    }
                    if (ps == 0) lds_barrier();                 // (A2): y_j is in vec
                    if (fwd) {
#pragma unroll
                        for (int u = 0; u < GRP_RMAXT; ++u)
                            if (act[u] && T[u] < nch) fwd_update(x1[u], x2[u], T[u], j0, li, kq);
                    }
                } else if (ps == 0) {
                    lds_barrier();                              // (A2)
                }
            }
            if (wv == 2) PROFW(30);
            TS(7);
            TS_PRINT();
        }
        return true;
    }
};

// LDS of the group kernel (doubles): the fixed buffers of qp_resident.hpp, the two byte tables, the two n-vectors
static constexpr int GRP_FIXED = 4 * 8 * 4 + 2 * 16 * 17 + 8 + 512 + 2 * GRP_OWN / 8 + 7 * 256 + 8;
static size_t group_lds_bytes(int NP) { return (size_t)(GRP_FIXED + 2 * (NP + 64)) * sizeof(double); }

// per-problem scratch doubles: the tile-packed factor, one copy of U per member, the look-ahead accumulators of every block
// column (7 tiles each; a column's slot is written once per factorisation, so no member can overwrite what a slower one
// has not fetched yet)
static size_t group_scratch_doubles(int n, int G) {
    const size_t NP = (size_t)round_up(n, 32);
    return NP * NP + (size_t)G * NP * PLD + (NP / NB) * (size_t)GRP_LA_SLOT;
}

// grid: block 8 (r G + g) + s = member g of problem 8 r + s -- the members of a problem are 8 blocks apart, i.e. on one XCD
// under the round-robin dispatch (checked below)
__global__ __launch_bounds__(512, 2) void qp_kernel_group(QpArgs a, int NP, int G) {
    constexpr int RT = 512;
    const int s_ = blockIdx.x & 7, rg = blockIdx.x >> 3;
    const int b = 8 * (rg / G) + s_, g = rg % G;
    if (b >= a.B) return;
    if (a.active && !a.active[b]) return;
    if (a.redo_aborted && a.status[b] != HIPDRT_QP_ABORTED) return;
    extern __shared__ double smem[];
    OpsGroup ops;
    ops.G = G; ops.g = g;
    ops.gs = a.gsync + (size_t)b * GRP_WORDS;
    ops.abort_status = a.status + b;
    ops.L = a.L + (size_t)b * a.l_stride; ops.nch = NP / 16; ops.n = a.n;
    ops.Ppk = a.Ppk + (size_t)b * a.ppk_stride; ops.nchp = a.nchp;
    // LDS carve (the pointers of the base class's layout struct are set by hand: n-vectors sized by this launch)
    ops.sm.red = smem;
    ops.sm.t21 = ops.sm.red + 4 * 8 * 4;
    ops.sm.dsc = ops.sm.t21 + 16 * 17;
    ops.sm.flag = reinterpret_cast<int*>(ops.sm.dsc + 16 * 17);
    ops.sm.img = ops.sm.dsc + 16 * 17 + 8;
    ops.owner = reinterpret_cast<unsigned char*>(ops.sm.img + 512);
    ops.rowdone = ops.owner + GRP_OWN;
    ops.latile = ops.sm.img + 512 + 2 * GRP_OWN / 8;
    ops.lacnt = reinterpret_cast<int*>(ops.latile + 7 * 256);
    ops.sm.vec = ops.latile + 7 * 256 + 8;
    ops.sm.dvec = ops.sm.vec + NP + 64;
    ops.sm.U = ops.L + (size_t)NP * NP + (size_t)g * NP * PLD;          // this member's own inverse diagonal blocks
    ops.labuf = ops.L + (size_t)NP * NP + (size_t)G * NP * PLD;
    ops.build_owner();
    for (int i = threadIdx.x; i < NP * PLD; i += RT) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < NP + 64; i += RT) { ops.sm.vec[i] = 0.0; ops.sm.dvec[i] = 0.0; }
    // ---- rendezvous: all members resident, all on one XCD -------------------------------------------------------------------
    if (G > 1) {
        int& xcc_mask = ops.sm.flag[2];
        if (threadIdx.x == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;      // HW_REG_XCC_ID[3:0]
            __hip_atomic_fetch_or(&ops.gs[1], 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Go or abort is decided on ONE word, so that no two members can decide differently: word [0] = arrivals, plus the
            // poison bit of a member that did not see its partners in time (set by compare-and-swap on the very value it
            // saw: either the count was still short and EVERY later reader -- pollers, and arrivers through the value their
            // fetch_add returns -- finds the bit, or somebody arrived meanwhile and the member looks again).  "All arrived,
            // not poisoned" can therefore not be seen by one member and missed by another; the late ones, whenever they
            // get a CU, find the bit and leave before they have touched anything but their own scratch.
            // (release: the fetch_or of this member's XCC bit above is ordered before its count; the readers below acquire on
            // the count, so "all G arrived" implies all G bits are visible -- two relaxed atomics on different words would not)
            int seen = __hip_atomic_fetch_add(&ops.gs[0], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
            int spins = 0;
            while (!(seen & kNotResident) && seen < G) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > OpsGroup::kRendezvousLimit) {
                    int expected = seen;
                    if (__hip_atomic_compare_exchange_strong(&ops.gs[0], &expected, seen | kNotResident, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        seen |= kNotResident;
                        break;
                    }
                    seen = expected;                         // the count moved: judge the new value
                    continue;
                }
                seen = __hip_atomic_load(&ops.gs[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            }
            // the XCC mask is complete once every member has arrived (a member ORs its bit in before it counts itself)
            xcc_mask = (seen & kNotResident) ? kNotResident
                                             : __hip_atomic_load(&ops.gs[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (__builtin_popcount(xcc_mask) != 1 || (xcc_mask & kNotResident)) {
            // spread over several XCDs (the L2-coherence assumption does not hold) or not all resident: every member leaves,
            // the host repeats this problem on one workgroup; word [4] says which (1 spread, 2 not resident)
            if (g == 0 && threadIdx.x == 0) {
                a.status[b] = HIPDRT_QP_ABORTED;
                ops.gs[4] = (xcc_mask & kNotResident) ? 2 : 1;
            }
            return;
        }
    }
    __syncthreads();
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    ipm_solve<RT, (GRP_NMAX + RT - 1) / RT>(a, b, ops, is, b * G + g, g == 0);
}

}  // namespace hipdrt

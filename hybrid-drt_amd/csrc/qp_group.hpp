// coneqp for FEW, possibly LARGE problems: one problem on G co-resident workgroups ("members"), n <= 4096.
//
// The batch kernel (qp_resident.hpp) gives every problem one workgroup = one CU: the right shape for a thousand problems,
// the wrong one for a single spectrum (BASELINE configs[1]: 255 of 256 CUs idle), one joint fit (configs[4], n = 1078) or
// a coupled multi-observation QP (mapping/resolve.py, n = 7 x 514 = 3598 > the 2048 the one-CU kernel serves).  Here the
// factorisation of ONE problem -- the tile-packed left-looking blocked Cholesky of qp_resident.hpp, same tile layout, same
// per-tile arithmetic -- is shared by G workgroups:
//
//  * Tile row T belongs to member T % G for the whole factorisation, and inside the member to one row wavefront (static
//    table, build_owner): a wavefront only re-reads tiles it stored itself.  The diagonal chain (wavefront 0) and the
//    look-ahead rows (wavefront 1: the two tile rows of the next diagonal block) are computed REDUNDANTLY by every member
//    from identical inputs, hence bit-identical: the only data that crosses between members are the finished tiles of the two
//    rows that become look-ahead rows next, published by their owner through one progress word per tile row in global
//    memory (rowprog[T] = factorisation count * 256 + block columns complete).
//  * Inside a member there is no barrier at the end of a block column: the hand-offs between its wavefronts go through
//    progress bytes in LDS (rowdone[T], same meaning) and a "diagonal block staged" word, so a member's store drain overlaps
//    its next rank-k update.  Two workgroup barriers per block column remain: (A) inverse diagonal blocks published, (A2)
//    forward-substituted right-hand-side block published.
//  * Everything else -- the O(n) interior-point vectors, the triangular sweeps, P x -- every member does redundantly on its
//    own copy (own state slot, own inverse diagonal blocks U), reading the shared factor: no communication, and the members
//    stay in lockstep because they compute the same bits.  Two group barriers per factorisation: before it (everybody has
//    finished sweeping the old factor) and after it (every tile is in memory; followed by an agent-scope acquire, i.e. an L1
//    invalidate, so that the sweeps' plain loads see the other members' tiles).
//  * Visibility inside the factorisation without fences: the members of a group sit on ONE XCD (blocks b and b + 8 share an
//    XCD under the round-robin dispatch; checked at run time through HW_REG_XCC_ID -- a group that is spread over several
//    XCDs reports HIPDRT_QP_ABORTED and the launcher repeats that problem with G = 1), so the XCD's L2 is the point of
//    coherence: stores are complete (s_waitcnt vmcnt(0)) before the progress word is written, and every load of a tile
//    another member may have written is an `sc1` load, which bypasses the CU's L1 (MI355X_MICROARCH.md, inter-workgroup
//    visibility).  Tiles a wavefront wrote itself are read with plain loads.
//  * Every wait on another member is bounded (kSpinLimit): a protocol error or a member that never became resident traps the
//    launch instead of hanging the device.  Group launches of one device are chained through an event (qp.hip) so that two
//    of them cannot each occupy part of the CUs and wait for the rest.
//
// G = 1 is the same kernel without the global words: the single-workgroup form for n up to 4096.
#pragma once
#include "qp_resident.hpp"

namespace hipdrt {

static constexpr int GRP_NMAX = 4096;                  // unknowns
static constexpr int GRP_OWN = 512;                    // tile rows incl. appended ones the tables cover
static constexpr int GRP_MAXG = 32;                    // members (CUs of one XCD)
static constexpr int GRP_WORDS = 16 + GRP_OWN;         // ints of global sync state per problem
// words: [0] members arrived at the start, [1] OR of (1 << XCC id), [2] barrier counter, [16 + T] rowprog[T]

// 16 bytes per lane that bypass the CU's vector L1 (another CU of the XCD may have written them)
static __device__ __forceinline__ v2d gload16_sc1(const char* sbase, unsigned voff) {
    v2d d;
    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
    return d;
}

struct OpsGroup : OpsResidentT<true, 512, 4> {
    using Base = OpsResidentT<true, 512, 4>;
    static constexpr int RT = 512, RNW = 8;
    static constexpr int kSpinLimit = 1 << 24;
    int G = 1, g = 0;                      // members of the group, this member
    int* gs = nullptr;                     // global sync words of the problem (GRP_WORDS)
    int fidx = 0, gepoch = 0;              // factorisations started, group barriers passed (identical in every member)
    unsigned char* owner = nullptr;        // LDS [GRP_OWN]: row wavefront (2..7) owning tile row T in THIS member, 0 = not mine
    volatile unsigned char* rowdone = nullptr;   // LDS [GRP_OWN]: block columns of tile row T complete in memory (own member's rows
                                                 // and the look-ahead rows this member computes itself)

    __device__ __forceinline__ void trap_if(bool c) const { if (c) __builtin_trap(); }

    // ---- group barrier: every member's stores complete, then one arrival per member on a monotonic counter -------------
    __device__ __forceinline__ void group_sync() {
        __syncthreads();                                   // (s_waitcnt vmcnt(0) in every wavefront: this member's stores are in L2)
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
    }

    // rows of this member, dealt to its row wavefronts from the bottom up by weighted round-robin (qp_resident.hpp: SIMD 0 / 1
    // also run the chain / the look-ahead wavefront, so wavefronts 4 / 5 get smaller shares)
    __device__ __forceinline__ void build_owner() {
        if (threadIdx.x != 0) return;
        const int ntr = (n + 15) >> 4;
        int cnt[6] = {0, 0, 0, 0, 0, 0};
        const int wt[6] = {22, 22, 28, 10, 22, 22};
        for (int T = ntr - 1; T >= 0; --T) {
            if (T % G != g) { owner[T] = 0; continue; }
            int best = 0;
            for (int w = 1; w < 6; ++w)
                if ((cnt[w] + 1) * wt[best] < (cnt[best] + 1) * wt[w]) best = w;
            ++cnt[best];
            owner[T] = (unsigned char)(best + 2);
        }
    }

    // ---- progress words ----------------------------------------------------------------------------------------------------
    __device__ __forceinline__ int prog_value(int jb) const { return fidx * 256 + jb; }
    // tile row T complete through block column jb - 1 (published by its owner; local rows through LDS)
    __device__ __forceinline__ void wait_row(int T, int jb) const {
        int spins = 0;
        if (T % G == g || G == 1) {
            while (rowdone[T] < jb) { __builtin_amdgcn_s_sleep(1); trap_if(++spins > kSpinLimit); }
        } else {
            const int want = prog_value(jb);
            while (__hip_atomic_load(&gs[16 + T], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                trap_if(++spins > kSpinLimit);
            }
        }
        asm volatile("" ::: "memory");
    }
    // rows tb, tb + 1 of the current diagonal block: stored by this member's own look-ahead wavefront
    __device__ __forceinline__ void wait_diag_rows(int tb, bool two, int jb) const {
        int spins = 0;
        while (rowdone[tb] < jb || (two && rowdone[tb + 1] < jb)) { __builtin_amdgcn_s_sleep(1); trap_if(++spins > kSpinLimit); }
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

    // =========================================================================================================================
    __device__ __forceinline__ bool factor() {
        const int tid = opaque_u32(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int ntr = (n + 15) >> 4;
        ++fidx;
        group_sync();                            // every member has finished sweeping the previous factor: its tiles may go
        for (int i = tid; i < GRP_OWN; i += RT) rowdone[i] = 0;
        if (tid == 0) sm.flag[1] = 0;
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
            for (int spins = 0; *(volatile int*)&sm.flag[1] < jb;) {                    // diagonal block jb staged
                __builtin_amdgcn_s_sleep(1);
                trap_if(++spins > kSpinLimit);
            }
            asm volatile("" ::: "memory");
            bool ok = cholinv16_dsc(j0, 0);
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
            if (lane == 0) sm.flag[0] = ok ? 0 : 1;
            __syncthreads();                                    // (A) W1, L21, W2 published
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
            // fused forward substitution: y_j = M_j b_j (b_j has received every earlier column's update)
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (U lives in global memory: the row of W21 just stored is read back)
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
            lds_barrier();                                      // (A2) y_j published
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
        TileSrc pre[7];
        bool have_pre = false;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int tb = j0 >> 4;
            const int nc = 2 * jb;
            const int R2 = tb + 2, R3 = tb + 3;
            const bool v2 = R2 < ntr, v3 = R3 < ntr;
            if (jb > 0) {
                // the tiles this wavefront stored in the previous column (rows tb, tb+1 now) are in memory: tell the row wavefronts
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < 2) rowdone[tb + lane] = (unsigned char)jb;
            }
            v4d p20, p21, p30, p31, e11, e21, e22;
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
                    // operand ring as in qp_resident.hpp; every tile may have been written by another member: sc1 loads
                    const char* q0 = uniform_ptr(tile2(tb, 0));
                    const char* q1 = uniform_ptr(tile2(tb + 1, 0));
                    const char* q2 = uniform_ptr(tile2(R2, 0));
                    const char* q3 = uniform_ptr(tile2(v3 ? R3 : R2, 0));
                    const unsigned voff = (unsigned)fo * 16u;
                    struct Frag { v2d b0, b1, a2, a3; };
                    const int nk2 = 2 * nc, klast = nk2 - 1;
                    auto loadf = [&](Frag& f_, int k2) {
                        const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
                        f_.b0 = gload16_sc1(q0 + o, voff); f_.b1 = gload16_sc1(q1 + o, voff);
                        f_.a2 = gload16_sc1(q2 + o, voff); f_.a3 = gload16_sc1(q3 + o, voff);
                    };
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
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    // the newest two chunks (half-chunks nk2-4 ..) of rows R2, R3 were stored by their owners (possibly other
                    // members) in the previous block column: wait for them just before the first request that reaches that far
                    if (nk2 == 4) { wait_row(R2, jb); if (v3) wait_row(R3, jb); }
                    Frag f0, f1, f2, f3;
                    loadf(f0, 0); loadf(f1, 1); loadf(f2, 2);
                    for (int k2 = 0; k2 < nk2; k2 += 4) {
                        if (k2 == nk2 - 8) { wait_row(R2, jb); if (v3) wait_row(R3, jb); }
                        loadf(f3, k2 + 3); vm_wait<12>(); multf(f0);
                        loadf(f0, k2 + 4); vm_wait<12>(); multf(f1);
                        loadf(f1, k2 + 5); vm_wait<12>(); multf(f2);
                        loadf(f2, k2 + 6); vm_wait<12>(); multf(f3);
                    }
                    vm_wait<0>();
                    if (!v3) {
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
                // next column's source tiles (rows tb+4, tb+5), requested a column ahead
                if (R2 + 2 < ntr) {
                    const int N2 = R2 + 2, N3 = R3 + 2;
                    pre[0] = tile_src(N2, R2, ntr, fo); pre[1] = tile_src(N2, R3, ntr, fo);
                    pre[2] = tile_src(N3, R2, ntr, fo); pre[3] = tile_src(N3, R3, ntr, fo);
                    pre[4] = tile_src(N2, N2, ntr, fo); pre[5] = tile_src(N3, N2, ntr, fo);
                    pre[6] = tile_src(N3, N3, ntr, fo);
                    have_pre = true;
                }
                lds_barrier();                                  // (A2)
                fwd_update(x20, x21_, R2, j0, li, kq);
                if (v3) fwd_update(x30, x31, R3, j0, li, kq);
            } else {
                lds_barrier();                                  // (A2)
            }
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
        TileSrc pre[RMAXT][2];
        bool have_pre = false;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j0 = jb * NB;
            const int tb = j0 >> 4;
            const int nc = 2 * jb;
            const bool two = (tb + 1) < ntr;
            RowMask mask = my_rows(jb, wv, lane, ntr);
            if (jb > 0) {
                // everything this wavefront stored in the previous block column is in memory (the wait also covers the source
                // tiles requested after those stores).  The only rows anybody else reads next are tb + 2 and tb + 3 -- the
                // look-ahead wavefronts' rows from now on: their owner publishes them, to its own member through LDS and to
                // the other members through the global progress word
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < 2) {
                    const int T = tb + 2 + lane;
                    if (T < ntr && owner[T] == wv) {
                        rowdone[T] = (unsigned char)jb;
                        if (G > 1) __hip_atomic_store(&gs[16 + T], prog_value(jb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            const int mine = count_rows(mask);
            const int npass = mine > RMAXT ? (mine + RMAXT - 1) / RMAXT : 1;
#pragma unroll 1
            for (int ps = 0; ps < npass; ++ps) {
                int T[RMAXT];
                bool act[RMAXT];
#pragma unroll
                for (int u = 0; u < RMAXT; ++u) {
                    const int t_ = pop_row(mask, tb);
                    T[u] = t_ >= 0 ? t_ : nch;
                    act[u] = t_ >= 0;
                }
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
                    // operand ring as in qp_resident.hpp.  A tiles = this wavefront's own rows (plain loads: it stored them
                    // itself); B tiles = rows tb, tb+1, whose older chunks their owner -- possibly another member -- stored
                    // when they were ordinary rows: sc1 loads
                    const char* rb0 = uniform_ptr(tile2(tb, 0));
                    const char* rb1 = uniform_ptr(tile2(two ? tb + 1 : tb, 0));
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
                        for (int u = 0; u < RMAXT; ++u) s_.a[u] = act[u] ? gload16(ra[u] + o, voff) : gload16_sc1(ra[u] + o, voff);
                    };
                    auto loadB = [&](SlB& s_, int k2) {
                        const size_t o = (size_t)(k2 < klast ? k2 : klast) * 1024;
                        s_.b0 = gload16_sc1(rb0 + o, voff); s_.b1 = gload16_sc1(rb1 + o, voff);
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
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    // the newest two chunks of rows tb, tb+1 were stored by this member's look-ahead wavefront in the previous
                    // block column: wait for them just before the first request that reaches that far
                    if (nk2 == 4) wait_diag_rows(tb, two, jb);
                    SlB b0, b1;
                    loadB(b0, 0); loadA(a0, 0); loadA(a1, 1); loadA(a2, 2);
                    for (int k2 = 0; k2 < nk2; k2 += 4) {
                        if (k2 == nk2 - 8) wait_diag_rows(tb, two, jb);
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
                        // the source tiles of the next block column's first pass, requested a column ahead
                        RowMask nx = my_rows(jb + 1, wv, lane, ntr);
                        int Tn[RMAXT];
#pragma unroll
                        for (int u = 0; u < RMAXT; ++u) Tn[u] = pop_row(nx, tb + 2);
                        if (Tn[0] >= 0) {
#pragma unroll
                            for (int u = 0; u < RMAXT; ++u)
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct)
                                    if (Tn[u] >= 0) pre[u][ct] = tile_src(Tn[u], tb + 2 + ct, ntr, fo);
                            have_pre = true;
                        }
                    }
                    if (ps == 0) lds_barrier();                 // (A2) y_j published by wavefront 0
#pragma unroll
                    for (int u = 0; u < RMAXT; ++u)
                        if (act[u]) fwd_update(x1[u], x2[u], T[u], j0, li, kq);
                } else if (ps == 0) {
                    lds_barrier();                              // (A2)
                }
            }
        }
        return true;
    }
};

// LDS of the group kernel (doubles): the fixed buffers of qp_resident.hpp, the two byte tables, the two n-vectors
static constexpr int GRP_FIXED = 4 * 8 * 4 + 2 * 16 * 17 + 8 + 512 + 2 * GRP_OWN / 8;
static size_t group_lds_bytes(int NP) { return (size_t)(GRP_FIXED + 2 * (NP + 64)) * sizeof(double); }

// per-problem scratch doubles: the tile-packed factor followed by one copy of U per member
static size_t group_scratch_doubles(int n, int G) {
    const size_t NP = (size_t)round_up(n, 32);
    return NP * NP + (size_t)G * NP * PLD;
}

// grid: block 8 (r G + g) + s = member g of problem 8 r + s -- the members of a problem are 8 blocks apart, i.e. on one XCD
// under the round-robin dispatch (checked below)
__global__ __launch_bounds__(512, 2) void qp_kernel_group(QpArgs a, int NP, int G) {
    constexpr int RT = 512;
    const int s_ = blockIdx.x & 7, rg = blockIdx.x >> 3;
    const int b = 8 * (rg / G) + s_, g = rg % G;
    if (b >= a.B) return;
    if (a.active && !a.active[b]) return;
    extern __shared__ double smem[];
    OpsGroup ops;
    ops.G = G; ops.g = g;
    ops.gs = a.gsync + (size_t)b * GRP_WORDS;
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
    ops.sm.vec = ops.sm.img + 512 + 2 * GRP_OWN / 8;
    ops.sm.dvec = ops.sm.vec + NP + 64;
    ops.sm.U = ops.L + (size_t)NP * NP + (size_t)g * NP * PLD;          // this member's own inverse diagonal blocks
    ops.build_owner();
    for (int i = threadIdx.x; i < NP * PLD; i += RT) ops.sm.U[i] = 0.0;
    for (int i = threadIdx.x; i < NP + 64; i += RT) { ops.sm.vec[i] = 0.0; ops.sm.dvec[i] = 0.0; }
    // ---- rendezvous: all members resident, all on one XCD -------------------------------------------------------------------
    if (G > 1) {
        __shared__ int xcc_mask;
        if (threadIdx.x == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;      // HW_REG_XCC_ID[3:0]
            __hip_atomic_fetch_or(&ops.gs[1], 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&ops.gs[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&ops.gs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > OpsGroup::kSpinLimit) __builtin_trap();
            }
            xcc_mask = __hip_atomic_load(&ops.gs[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (__builtin_popcount(xcc_mask) != 1) {
            // spread over several XCDs: the L2-coherence assumption does not hold -- every member leaves, the host repeats
            // this problem on one workgroup
            if (g == 0 && threadIdx.x == 0) a.status[b] = HIPDRT_QP_ABORTED;
            return;
        }
    }
    __syncthreads();
    IpmSmem is{ops.sm.vec, ops.sm.dvec, ops.sm.red};
    ipm_solve<RT, (GRP_NMAX + RT - 1) / RT>(a, b, ops, is, b * G + g, g == 0);
}

}  // namespace hipdrt

// Shared pieces of the batched coneqp kernels: wave helpers, fixed-order block reductions and the interior-point
// driver itself (cvxopt coneprog.coneqp for one 'l' cone with G = -I; restated for tests in the repository's CPU
// checker).  The driver is templated on an Ops policy that provides the three linear-algebra
// services: factor S = P + diag(dvec), solve S vec = vec in place, and out = P * vec.
#pragma once
#include "common.hpp"

namespace hipdrt {

typedef double v4d __attribute__((ext_vector_type(4)));

#ifdef HIPDRT_QP_PROFILE
static constexpr int QP_PROF_SLOTS = 48;
__device__ unsigned long long g_qp_prof[QP_PROF_SLOTS];
// time line of workgroup 0's LAST factorisation (factor64): s_memtime of wavefront w in super column J at stamp k --
// 0 start of the super column, 1 at barrier (A), 2 behind (A), 3 tiles stored (arrival at (B); wavefront 0: everybody's arrival seen),
// wavefront 0 also 4 chain a done, 5 look-ahead history done, 6 look-ahead solve done, 7 W21 / y done
// g_qp_tl_sum: the same stamps as cycles since the start of the factorisation (TL_START, wavefront 1 in front of the first
// barrier), summed over all factorisations of workgroup 0 since the last reset; entry [8 * J * K] counts the factorisations
static constexpr int QP_TL_J = 16, QP_TL_K = 8;
__device__ unsigned long long g_qp_tl[8 * QP_TL_J * QP_TL_K];
__device__ unsigned long long g_qp_tl_sum[8 * QP_TL_J * QP_TL_K + 1];
__device__ unsigned long long g_qp_tl_ref;
#define TL_START() do { if (blockIdx.x == 0) { g_qp_tl_ref = __builtin_amdgcn_s_memtime(); atomicAdd(&g_qp_tl_sum[8 * QP_TL_J * QP_TL_K], 1ull); } } while (0)
#define TL(w, J, k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && (J) < QP_TL_J) { \
    const unsigned long long _tn = __builtin_amdgcn_s_memtime(); \
    g_qp_tl[((w) * QP_TL_J + (J)) * QP_TL_K + (k)] = _tn; \
    atomicAdd(&g_qp_tl_sum[((w) * QP_TL_J + (J)) * QP_TL_K + (k)], _tn - g_qp_tl_ref); } } while (0)
#define PROF_DECL unsigned long long _pt = __builtin_amdgcn_s_memtime();
#define PROF(slot) do { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long _n = __builtin_amdgcn_s_memtime(); \
    atomicAdd(&g_qp_prof[slot], _n - _pt); _pt = _n; } else { _pt = 0; } } while (0)
// the same interval into two slots (a total and a per-block-column breakdown)
#define PROF2(slot, slot2) do { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long _n = __builtin_amdgcn_s_memtime(); \
    atomicAdd(&g_qp_prof[slot], _n - _pt); atomicAdd(&g_qp_prof[slot2], _n - _pt); _pt = _n; } else { _pt = 0; } } while (0)
// the same for lane 0 of any wavefront of workgroup 0 (per-wavefront role timelines, slots 16..)
#define PROFW(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { unsigned long long _n = __builtin_amdgcn_s_memtime(); \
    atomicAdd(&g_qp_prof[slot], _n - _pt); _pt = _n; } else { _pt = 0; } } while (0)
#else
static constexpr int QP_PROF_SLOTS = 48;
#define PROF_DECL
#define PROF(slot)
#define PROF2(slot, slot2)
#define PROFW(slot)
#define TL(w, J, k)
#define TL_START()
#endif

static constexpr int NB = 32;     // Cholesky block
static constexpr int PLD = 33;    // LDS panel row stride (doubles): odd => conflict-free row-per-lane access

// the value unchanged, but opaque to the optimiser: address arithmetic derived from it is redone where it is used
// instead of being hoisted to the top of the kernel and kept (or spilled) across every phase
__device__ __forceinline__ unsigned opaque_u32(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for outstanding global
// loads (vmcnt), so operand tiles prefetched before the barrier stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// an int / a byte of LDS read NOW (polls of progress words: a volatile C++ load would be a FLAT instruction, which counts in
// vmcnt as well and returns out of order with the hand-counted loads).  The value goes through readfirstlane: the spin loops
// around these stay scalar branches -- a loop whose exit depends on a vector register runs under EXEC masking, and
// hand-issued loads after such a loop were observed to fault.
static __device__ __forceinline__ int lds_peek32(const void* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
static __device__ __forceinline__ int lds_peek8(const void* p) {
    int v;
    asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}

typedef double v2d __attribute__((ext_vector_type(2)));

// 16 bytes per lane from (wave-uniform base) + (per-lane byte offset), issued as ONE instruction the compiler neither
// moves nor waits for: the rank-k loops below keep several half-chunks in flight and count vmcnt by hand (hipcc's own
// schedule gathers all loads of an unrolled body at its top and drains them with vmcnt(0) at its bottom, so nothing
// stays in flight across iterations).  Every use of the result must come after an explicit vm_wait<N>().
// HAZARD: hipcc does not insert wait states for inline asm.  gfx9 needs 5 between a VALU instruction that writes an SGPR
// (v_readlane: every SGPR spill reload; v_readfirstlane) and a VMEM instruction that reads it -- a gload16 right behind a spill
// reload of its base loads from a stale pointer (observed: faults on address 0).  tools/sgpr_hazard.py scans the assembly for
// this, tests/test_isa_hazards.py runs it with every test run; where it cannot be avoided use gload16v (VGPR address).
static __device__ __forceinline__ v2d gload16(const char* sbase, unsigned voff) {
    v2d d;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
    return d;
}
// the same from a per-lane 64-bit address plus an immediate byte offset (0 .. 4095), no scalar base pair
template <int OFF>
static __device__ __forceinline__ v2d gload16v(const void* p) {
    v2d d;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
    return d;
}
template <int N>
static __device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);      // register-only consumers (MFMA) must not be hoisted above the wait
}
static __device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {     // quad_perm 0x00-0xFF, row_ror:n = 0x120 + n
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true));
}
// Column sums of a tile held as lane = 4*row + l4 with four values per lane (columns l4, l4+4, l4+8, l4+12):
// the sum over the 16 rows of value v ends up in the lanes of 16-lane row v (lane>>4 == v).  Rows 4-apart are
// combined with DPP row rotations, the four 16-lane rows with a halving exchange (3 swaps, not 16 moves).
// gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper 32-lane (odd 16-lane) rows of the first operand with
// the lower (even) rows of the second: (a', b') = swap(a, b), and a' + b' is at once "lower lanes: a summed over both halves,
// upper lanes: b summed over both halves" -- the halving exchange of a cross-lane reduction in two VALU instructions per
// double, no select and no LDS crossbar (ds_bpermute) round trip.  Same operands per addition as the shuffle form had.
__device__ __forceinline__ double swap_add32(double a, double b) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {
    const auto r0 = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double((int)r1[0], (int)r0[0]) + __hiloint2double((int)r1[1], (int)r0[1]);
}

__device__ __forceinline__ double colsum4(double s0, double s1, double s2, double s3, int lane) {
    s0 += dpp_mov_d<0x128>(s0); s1 += dpp_mov_d<0x128>(s1); s2 += dpp_mov_d<0x128>(s2); s3 += dpp_mov_d<0x128>(s3);
    s0 += dpp_mov_d<0x124>(s0); s1 += dpp_mov_d<0x124>(s1); s2 += dpp_mov_d<0x124>(s2); s3 += dpp_mov_d<0x124>(s3);
    const double k0 = swap_add32(s0, s2);        // lanes 0-31: value 0 over both halves, lanes 32-63: value 2
    const double k1 = swap_add32(s1, s3);
    return swap_add16(k0, k1);                   // 16-lane row v holds the sum of value v
}

// sum over the 4 lanes of a quad (lane ^ 1, lane ^ 2) with DPP quad permutes: VALU speed, no LDS crossbar
__device__ __forceinline__ double quad_sum(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    double o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true),
                                __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true));
    v += o;
    lo = __double2loint(v); hi = __double2hiint(v);
    o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true),
                         __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true));
    return v + o;
}

__device__ __forceinline__ double bcast_lane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// A value that is the same in every lane (the result of a block reduction) moved to scalar registers: the IPM's scalars
// (gap, mu, sigma, step, costs, norms) live across the factorisation and the solves, where vector registers are scarce
__device__ __forceinline__ double uni(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}

// wavefront sum / max, every lane gets the result: DPP quad permutes and row rotations inside the 16-lane rows,
// v_readlane across the four rows (no LDS crossbar)
__device__ __forceinline__ double wsum(double v) {
    v += dpp_mov_d<0xB1>(v);
    v += dpp_mov_d<0x4E>(v);
    v += dpp_mov_d<0x124>(v);
    v += dpp_mov_d<0x128>(v);
    return (bcast_lane(v, 0) + bcast_lane(v, 16)) + (bcast_lane(v, 32) + bcast_lane(v, 48));
}
__device__ __forceinline__ double wmax(double v) {
    v = fmax(v, dpp_mov_d<0xB1>(v));
    v = fmax(v, dpp_mov_d<0x4E>(v));
    v = fmax(v, dpp_mov_d<0x124>(v));
    v = fmax(v, dpp_mov_d<0x128>(v));
    return fmax(fmax(bcast_lane(v, 0), bcast_lane(v, 16)), fmax(bcast_lane(v, 32), bcast_lane(v, 48)));
}

// SLOTS >= 2 result slots used in rotation: a wavefront can be at most one reduction (= one barrier) ahead of the slowest
// reader of the previous one, so two suffice; the 64-column factorisation's LDS budget has room for exactly two.
// VT > 1: every physical thread stands for VT virtual ones (thread t, t + 64 NW, ...: the fat four-wavefront kernel runs
// the interior-point vectors as the eight-wavefront kernel does, two virtual threads each), a physical wavefront w for the
// virtual wavefronts w, w + NW, ...; partial sums are formed per VIRTUAL thread and combined per virtual wavefront in
// ascending order, so a reduction returns the same bits whatever the number of physical wavefronts.
template <int NW, int SLOTS = 4, int VT = 1>
struct Reducer {
    static_assert(SLOTS == 2 || SLOTS == 4, "");
    static constexpr int NWV = NW * VT;
    double* buf;   // LDS [SLOTS][NWV][4]
    int slot;
    __device__ Reducer(double* b) : buf(b), slot(0) {}
    // sums up to 4 values at once; every thread gets the totals (in v[0])
    template <int N>
    __device__ __forceinline__ void sum(double (&v)[VT][N]) {
        static_assert(N <= 4, "");
        double* s = buf + (slot & (SLOTS - 1)) * NWV * 4;
        ++slot;
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
            for (int i = 0; i < N; ++i) v[t][i] = wsum(v[t][i]);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int t = 0; t < VT; ++t)
#pragma unroll
                for (int i = 0; i < N; ++i) s[((threadIdx.x >> 6) + NW * t) * 4 + i] = v[t][i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NWV; ++w) t += s[w * 4 + i];
            v[0][i] = t;
        }
    }
    template <int N>
    __device__ __forceinline__ void max(double (&v)[VT][N]) {
        static_assert(N <= 4, "");
        double* s = buf + (slot & (SLOTS - 1)) * NWV * 4;
        ++slot;
#pragma unroll
        for (int t = 0; t < VT; ++t)
#pragma unroll
            for (int i = 0; i < N; ++i) v[t][i] = wmax(v[t][i]);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int t = 0; t < VT; ++t)
#pragma unroll
                for (int i = 0; i < N; ++i) s[((threadIdx.x >> 6) + NW * t) * 4 + i] = v[t][i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = s[i];
#pragma unroll
            for (int w = 1; w < NWV; ++w) t = fmax(t, s[w * 4 + i]);
            v[0][i] = t;
        }
    }
};


// LDS vectors every Ops policy provides
struct IpmSmem {
    double* vec;     // [n]  rhs / solution of the KKT solve; x for the mat-vec
    double* dvec;    // [n]  diagonal shift d^-2 for the factorisation; P x result
    double* red;     // [Ops::kRedSlots][NW][4]
};

// ---------------------------------------------------------------------------------------------------------
// The interior-point iteration.  O(n) vectors live in registers, element i owned by thread i % THREADS.
// ---------------------------------------------------------------------------------------------------------
// VT = virtual threads per physical thread (Reducer above): EPT is then the element count of a VIRTUAL thread, element i is
// owned by virtual thread i % (THREADS VT) = physical thread i % THREADS, and per-thread partial sums run over a virtual
// thread's elements -- the reductions, hence the whole trajectory, are bit for bit those of a THREADS VT-thread workgroup.
template <int THREADS, int EPT, class Ops, int VT = 1>
__device__ __forceinline__ void ipm_solve(const QpArgs& a, int b, Ops& ops, const IpmSmem& sm, int slot = -1,
                                          bool leader = true) {
    // slot / leader: several workgroups may run the same problem redundantly (qp.hip, workgroup groups); each then keeps
    // its own copy of the iterates and only the leader writes the results
    constexpr int NW = THREADS / 64;
    // element loops: fully unrolled up to 4 elements per thread (reads first, stores last: all loads of a statement in flight
    // together); beyond that (the group kernel, n <= 4096: 8 per thread) in pairs -- eight elements' worth of operands at once
    // would not fit the register file next to the factorisation's rings
    constexpr int EUNR = EPT <= 4 ? EPT : 2;
    const int n = a.n;
    const unsigned tid = threadIdx.x;      // unsigned indices: SGPR base + 32-bit VGPR offset addressing, no per-vector
                                           // 64-bit address registers kept alive across the whole kernel
    const double* qg = a.q + (size_t)b * n;
    const double* hg = a.h + (size_t)b * a.h_stride;
    Reducer<NW, Ops::kRedSlots, VT> red(sm.red);
    // P x is normally not formed by a pass over P: every step direction solves (P + D) dx = r to the backward error of
    // the Cholesky solve, so P dx = r - D dx and P x follows the iterate by an O(n) recurrence (start point:
    // (P + I) x0 = -q - h).  The recurrence carries rounding errors of size eps |P| |step dx| along, a direct product
    // only eps |P| |x|: `drift` sums max|step dx| since the last direct product, and once it exceeds kDriftTol max|x|
    // (iterates that come down from a far-away start point: h = 1e5 when nonneg is off) the next residual is taken
    // from a direct product again.  The rule depends on reduced values only, so it is the same in every thread and for
    // every batch size.  -DHIPDRT_QP_MATVEC builds the direct product in every iteration (diagnostic).
#ifdef HIPDRT_QP_MATVEC
    constexpr bool kRecurPx = false;
#else
    constexpr bool kRecurPx = true;
#endif
    constexpr double kDriftTol = 8.0;
    constexpr double kCancUlps = 64.0;
    double drift = 0.0, canc = 0.0;
    bool refresh = false;

    // The O(n) iterates live in a per-problem global scratch (L1/L2 resident, 16 vectors), element i touched
    // only by its owner thread, so no synchronisation is needed for them; keeping them out of registers leaves
    // the 128-VGPR budget of the 1024-thread kernel to the factorisation / solve phases (fewer spills there).
    double* const S_ = a.state + (size_t)(slot >= 0 ? slot : b) * a.state_stride;
    const int sld = a.state_ld;
#define SV(k) (S_ + (size_t)(k) * sld)
    double* const x = SV(0); double* const z = SV(1); double* const s = SV(2); double* const d = SV(3);
    double* const di = SV(4); double* const lm = SV(5); double* const qv = SV(6); double* const hv = SV(7);
    double* const rx = SV(8); double* const rz = SV(9); double* const dx = SV(10); double* const ds = SV(11);
    double* const dz = SV(12); double* const ws3 = SV(13); double* const zz = SV(14); double* const sv = SV(15);
    double* const px = SV(16);      // P x, carried along by the recurrence below
#define FOR_E _Pragma("unroll") for (unsigned v_ = 0; v_ < (unsigned)VT; ++v_) \
        _Pragma("unroll EUNR") for (unsigned e_ = 0, i = opaque_u32(tid) + v_ * THREADS; e_ < (unsigned)EPT; ++e_, i += THREADS * VT)
#define VALID (i < (unsigned)n)
    FOR_E if (VALID) { qv[i] = qg[i]; hv[i] = hg[i]; x[i] = z[i] = 0.0; s[i] = lm[i] = 1.0; d[i] = di[i] = 1.0; }

    double nq[VT][2] = {};
    FOR_E if (VALID) { nq[v_][0] += qv[i] * qv[i]; nq[v_][1] += hv[i] * hv[i]; }
    red.sum(nq);
    const double resx0 = uni(fmax(1.0, sqrt(nq[0][0])));
    const double resz0 = uni(fmax(1.0, sqrt(nq[0][1])));

    PROF_DECL
#ifdef HIPDRT_QP_PROFILE
    const unsigned long long _rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz: slot 13 / slot 10 gives the shader clock
#endif
    int status = HIPDRT_QP_MAXITER, iters = 0;
    double pcost = 0.0, gap = 0.0;

    // One loop body serves the start point (W = I, one KKT solve) and every Mehrotra iteration (two KKT solves)
    // so that factor(), solve() and matvec() are each inlined exactly once.
    bool start = true;
    for (;;) {
        if (!start) {
            // ---- residuals, costs, stopping test ----------------------------------------------------------
            const bool direct = !kRecurPx || refresh;       // this pass's P x is a direct product
            if (direct) {
                __syncthreads();
                FOR_E if (VALID) sm.vec[i] = x[i];
                __syncthreads();
                { PROF_DECL
                ops.matvec();
                __syncthreads();
                PROF(9); }
                if (kRecurPx) {
                    FOR_E if (VALID) px[i] = sm.dvec[i];
                    refresh = false;
                    canc = 0.0;
                }
            }
            // (the element loops read everything they need first and store last: the state vectors are slices of one
            // buffer, so a store in between would order every later load behind it -- one L2 round trip per statement)
            double t4[VT][4] = {}, zr[VT][1] = {};
            FOR_E {
                if (VALID) {
                    const double px_ = kRecurPx ? px[i] : sm.dvec[i], q_ = qv[i], x_ = x[i], z_ = z[i], s_ = s[i], h_ = hv[i];
                    double r = px_ + q_;                    // P x + q
                    t4[v_][0] += x_ * r;                      // x'(Px+q)
                    t4[v_][1] += x_ * q_;                      // x'q
                    r -= z_;                                // + G'z
                    t4[v_][2] += r * r;
                    const double rzz = s_ - h_ - x_;        // s + Gx - h
                    t4[v_][3] += rzz * rzz;
                    zr[v_][0] += z_ * rzz;
                    rx[i] = r;
                    rz[i] = rzz;
                }
            }
            red.sum(t4);
            red.sum(zr);
            const double f0 = 0.5 * (t4[0][0] + t4[0][1]);
            const double resx = sqrt(t4[0][2]), resz = sqrt(t4[0][3]);
            pcost = uni(f0);
            const double dcost = f0 + zr[0][0] - gap;
            bool has_rel = false;
            double relgap = 0.0;
            if (pcost < 0.0) { relgap = gap / -pcost; has_rel = true; }
            else if (dcost > 0.0) { relgap = gap / dcost; has_rel = true; }
            const double pres = resz / resz0, dres = resx / resx0;
            const bool conv = pres <= a.opts.feastol && dres <= a.opts.feastol &&
                              (gap <= a.opts.abstol || (has_rel && relgap <= a.opts.reltol));
            // The drift rule does not see cancellation in r - D dx when di^2 is huge (active constraints late in the iteration),
            // nor the residual of the Cholesky solve on such an S.  `canc` sums, since the last direct product, the largest
            // magnitude that went into an element of the recurrence (step (|r| + di^2 |dx|)): each update loses at most a few
            // ulps of that, so kCancUlps eps sqrt(n) canc bounds what the carried residual norm can be off by.  A verdict --
            // "optimal", or the last test at maxiters -- taken on a carried P x whose bound exceeds 5 % of the dual
            // feasibility threshold is repeated once on P x itself, as cvxopt forms it in every iteration; if the direct
            // residual does not pass, the iteration simply goes on from the corrected P x.
            if ((conv || iters == a.opts.maxiters) && !direct &&
                kCancUlps * 1.1102230246251565e-16 * sqrt((double)n) * canc > 0.05 * a.opts.feastol * resx0) {
                refresh = true;
                continue;
            }
            if (conv) { status = HIPDRT_QP_OPTIMAL; break; }
            if (iters == a.opts.maxiters) { status = HIPDRT_QP_MAXITER; break; }
            if (iters == 0) {
                FOR_E if (VALID) { const double s_ = s[i], z_ = z[i], d_ = sqrt(s_ / z_); d[i] = d_; di[i] = 1.0 / d_; lm[i] = sqrt(s_ * z_); }
            }
        }
        // ---- factor S = P + diag(di^2)  (di = 1 at the start point) --------------------------------------
        const double mu = uni(gap / (double)n);
        double sigma = 0.0, step = 1.0;
        const int nsolve = start ? 1 : 2;
        // right-hand side of KKT solve pc (sigma = 0 for the predictor, so its rhs is known before the factorisation)
        auto set_rhs = [&](int pc) {
            FOR_E {
                if (VALID) {
                    if (start) {
                        sm.vec[i] = -qv[i] - hv[i];           // bx + Gs' bz with bx = -q, bz = h
                    } else {
                        const double lm_ = lm[i], rz_ = rz[i], d_ = d[i], di_ = di[i], rx_ = rx[i];
                        double t = (pc == 1) ? (-ws3[i] - lm_ * lm_) : (-(lm_ * lm_));
                        t += sigma * mu;
                        const double sv_ = t / lm_;
                        const double bz = -rz_ - d_ * sv_;
                        const double zz_ = bz * di_;
                        sv[i] = sv_;
                        zz[i] = zz_;
                        sm.vec[i] = -rx_ - di_ * zz_;
                    }
                }
            }
        };
        __syncthreads();
        FOR_E if (VALID) sm.dvec[i] = di[i] * di[i];
        if (Ops::kFusedForward) set_rhs(0);     // the factorisation also forward-substitutes the first rhs
        __syncthreads();
        if (!ops.factor()) {
            status = (start || iters == 0) ? HIPDRT_QP_SINGULAR : HIPDRT_QP_SINGULAR_LATE;
            break;
        }

#pragma nounroll
        for (int pc = 0; pc < nsolve; ++pc) {
            if (Ops::kFusedForward && pc == 0) {
                ops.backward();
            } else {
                set_rhs(pc);
                __syncthreads();
                ops.solve();
            }
            if (start) {
                double st[VT][2] = {}, mx[VT][2];
                for (int t_ = 0; t_ < VT; ++t_) mx[t_][0] = mx[t_][1] = -INFINITY;
                FOR_E {
                    if (VALID) {
                        const double x_ = sm.vec[i], q_ = qv[i], h_ = hv[i];
                        const double z_ = -x_ - h_, s_ = -z_;
                        st[v_][0] += s_ * s_; st[v_][1] += z_ * z_;
                        mx[v_][0] = fmax(mx[v_][0], -s_); mx[v_][1] = fmax(mx[v_][1], -z_);
                        x[i] = x_;
                        if (kRecurPx) px[i] = (-q_ - h_) - x_;           // (P + I) x = -q - h
                        z[i] = z_;
                        s[i] = s_;
                    }
                }
                red.sum(st);
                red.max(mx);
                if (kRecurPx) {
                    double mm[VT][2] = {};
                    FOR_E if (VALID) { const double ax = fabs(x[i]); mm[v_][0] = fmax(mm[v_][0], ax); mm[v_][1] = fmax(mm[v_][1], fabs(qv[i] + hv[i]) + ax); }
                    red.max(mm);
                    drift = uni(mm[0][0]);
                    canc = uni(mm[0][1]);
                }
                const double nrms = sqrt(st[0][0]), nrmz = sqrt(st[0][1]);
                if (mx[0][0] >= -1e-8 * fmax(nrms, 1.0)) {
                    FOR_E if (VALID) s[i] += 1.0 + mx[0][0];
                }
                if (mx[0][1] >= -1e-8 * fmax(nrmz, 1.0)) {
                    FOR_E if (VALID) z[i] += 1.0 + mx[0][1];
                }
                double gp[VT][1] = {};
                FOR_E if (VALID) gp[v_][0] += s[i] * z[i];
                red.sum(gp);
                gap = uni(gp[0][0]);
            } else {
                double dd[VT][1] = {}, mx[VT][2];
                for (int t_ = 0; t_ < VT; ++t_) mx[t_][0] = mx[t_][1] = -INFINITY;
                FOR_E {
                    if (VALID) {
                        const double dx_ = sm.vec[i], di_ = di[i], zz_ = zz[i], sv_ = sv[i], lm_ = lm[i];
                        double dz_ = -di_ * dx_ - zz_;
                        double ds_ = sv_ - dz_;
                        const double w3 = ds_ * dz_;
                        dd[v_][0] += w3;
                        ds_ /= lm_;
                        dz_ /= lm_;
                        mx[v_][0] = fmax(mx[v_][0], -ds_);
                        mx[v_][1] = fmax(mx[v_][1], -dz_);
                        dx[i] = dx_;
                        if (pc == 0) ws3[i] = w3;
                        ds[i] = ds_;
                        dz[i] = dz_;
                    }
                }
                red.sum(dd);
                red.max(mx);
                const double t = fmax(0.0, fmax(mx[0][0], mx[0][1]));
                if (t == 0.0) step = 1.0;
                else if (pc == 0) step = uni(fmin(1.0, 1.0 / t));
                else step = uni(fmin(1.0, 0.99 / t));
                if (pc == 0) {
                    const double sg = fmin(1.0, fmax(0.0, 1.0 - step + dd[0][0] / gap * (step * step)));
                    sigma = uni(sg * sg * sg);
                }
            }
        }
        if (start) { start = false; continue; }
        // ---- update ---------------------------------------------------------------------------------------
        double g2[VT][1] = {}, mm[VT][3] = {};
        FOR_E {
            if (VALID) {
                const double dx_ = dx[i], ds_ = ds[i], dz_ = dz[i], lm_ = lm[i], d_ = d[i], x_ = x[i];
                double px_ = 0.0;
                if (kRecurPx) {
                    const double rx_ = rx[i], di_ = di[i], zz_ = zz[i];
                    const double r_ = -rx_ - di_ * zz_, t_ = (di_ * di_) * dx_;
                    px_ = px[i] + step * (r_ - t_);
                    mm[v_][0] = fmax(mm[v_][0], fabs(step * dx_));
                    mm[v_][2] = fmax(mm[v_][2], step * (fabs(r_) + fabs(t_)));
                }
                const double xn = x_ + step * dx_;
                mm[v_][1] = fmax(mm[v_][1], fabs(xn));
                const double dss = (1.0 + step * ds_) * lm_;
                const double dzz = (1.0 + step * dz_) * lm_;
                const double sqs = sqrt(dss), sqz = sqrt(dzz);
                const double dn = d_ * sqs / sqz;
                const double din = 1.0 / dn;
                const double lmn = sqs * sqz;
                g2[v_][0] += lmn * lmn;
                if (kRecurPx) px[i] = px_;
                x[i] = xn;
                d[i] = dn;
                di[i] = din;
                lm[i] = lmn;
                s[i] = lmn * dn;
                z[i] = lmn * din;
            }
        }
        red.sum(g2);
        gap = uni(g2[0][0]);
        if (kRecurPx) {
            red.max(mm);
            drift = uni(drift + mm[0][0]);
            canc = uni(canc + mm[0][2]);
            if (drift > kDriftTol * mm[0][1]) { refresh = true; drift = uni(mm[0][1]); }
        }
        ++iters;
    }

    PROF(10);
#ifdef HIPDRT_QP_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_qp_prof[13], __builtin_amdgcn_s_memrealtime() - _rt0);
#endif
    if (leader) {
        FOR_E if (VALID) a.x[(size_t)b * n + i] = x[i];
    }
    if (tid == 0 && leader) {
#ifdef HIPDRT_QP_PROFILE
        if (blockIdx.x == 0) atomicAdd(&g_qp_prof[11], (unsigned long long)(iters + 1));      // the workgroup the tick counters follow
#endif
        if (a.iters) a.iters[b] = iters;
        if (a.pcost) a.pcost[b] = pcost;
        a.status[b] = status;
        if (a.iters_accum) a.iters_accum[b] += iters;
    }
#undef FOR_E
#undef VALID
#undef SV
}

}  // namespace hipdrt

// Batched coneqp for gfx950: cvxopt.solvers.qp(P, q, G=-I, h) as called at hybdrt/models/qphb.py:512-519,
// one workgroup per problem, the whole interior-point trajectory inside one launch (no host round trips).
//
// Algorithm = cvxopt coneprog.coneqp restricted to one 'l' cone with G = -I (restated in oracle/coneqp.py,
// SURVEY.md Appendix A): default start point, Nesterov-Todd scaling W = diag(d), Mehrotra predictor-corrector
// (STEP 0.99, EXPON 3), KKT solves through the Cholesky factor of S = P + diag(d^-2) ('chol2' solver),
// cvxopt's stopping test.  Everything FP64.  The interior-point driver is qp_common.hpp: ipm_solve; the linear
// algebra (tile-packed left-looking Cholesky, triangular sweeps) is qp_resident.hpp, one kernel for every
// n <= 2048: inverse diagonal blocks in LDS up to n = 528, in global memory beyond.  This file holds the launchers.
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <cstdio>

#include "qp_common.hpp"
#include "qp_resident.hpp"
#include "qp_group.hpp"

namespace hipdrt {

size_t qp_scratch_ld(int n) { return (size_t)round_up(n, 16); }

// doubles of factor scratch per problem: the tile-packed factor, beyond n = 528 followed by the inverse diagonal blocks
// (group kernel: one copy of those per member)
size_t qp_scratch_doubles(int n, int G) {
    if (G >= 1) return group_scratch_doubles(n, G);
    return n <= RNP_MAX ? resident_l_doubles(n) : resident_gu_doubles(n);
}
size_t qp_gsync_ints() { return GRP_WORDS; }

int device_cus() {
    static const int cus = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return cus;
}

// One problem per workgroup keeps the chip full from a few hundred problems on.  Below that -- a single spectrum, one joint
// fit, the handful of coupled QPs of resolve_group -- most CUs would idle, and beyond n = 2048 the one-workgroup kernel does
// not reach: such launches go to the group kernel.  Members per problem: one per two block rows (a member's fixed cost is
// its redundant chain wavefront; the look-ahead tiles are accumulated by one member per block column), all members of a
// problem on one XCD (32 CUs) with up to eight problems side by side.
int qp_group_size(int B, int n, int force) {
    if (n > GRP_NMAX) return -1;
    if (force == 0 && n <= 2048) return 0;
    if (force >= 1) {
        int G = force;
        const int ntr = (n + 15) / 16, rounds = (B + 7) / 8;
        if (G > 32 / rounds) G = 32 / rounds;
        if (G > ntr / 2) G = ntr / 2;        // (at least one block row per member)
        return G >= 1 ? G : 1;
    }
    const bool few = B * 16 <= device_cus();              // <= 16 problems on a 256-CU part
    if (n <= 2048 && !(few && n > 256)) return 0;       // (small n: a block column is all chain, nothing to share)
    const int ntr = (n + 15) / 16, rounds = (B + 7) / 8;
    int G = ntr / 4;                                       // two block rows (pairs of tile rows) per member
    if (G > 32 / rounds) G = 32 / rounds;
    if (G > GRP_MAXG) G = GRP_MAXG;
    return G >= 2 ? G : 1;
}

#ifdef HIPDRT_GRP_TIMELINE
}  // namespace hipdrt
extern "C" int hipdrt_debug_group_timeline(unsigned long long* out) {      // [32 members][8 wavefronts][10 stamps] + [32] owner flags
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hipdrt::g_grp_tl), sizeof(unsigned long long) * (32 * 8 * 10 + 32)) == hipSuccess ? 0 : -1;
}
namespace hipdrt {
#endif

int qp_profile_read(unsigned long long* out, int n, int reset) {
#ifdef HIPDRT_QP_PROFILE
    unsigned long long h[QP_PROF_SLOTS];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qp_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n; ++i) out[i] = i < QP_PROF_SLOTS ? h[i] : 0;
    if (reset) { unsigned long long z[QP_PROF_SLOTS] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qp_prof), z, sizeof(z)); }
    if (n > QP_PROF_SLOTS && hyper_profile_read(out + QP_PROF_SLOTS, (n < 64 ? n : 64) - QP_PROF_SLOTS, reset) < 0) return -1;
    if (n > 64) {            // behind the two kernels' phase counters: the factorisation time line (qp_common.hpp, g_qp_tl), then its sums
        constexpr int NT = 8 * QP_TL_J * QP_TL_K;
        static unsigned long long tl[NT], ts[NT + 1];
        if (hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_qp_tl), sizeof(tl)) != hipSuccess) return -1;
        if (hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_qp_tl_sum), sizeof(ts)) != hipSuccess) return -1;
        for (int i = 64; i < n; ++i) out[i] = i - 64 < NT ? tl[i - 64] : (i - 64 - NT <= NT ? ts[i - 64 - NT] : 0);
        if (reset) { static unsigned long long z[NT + 1]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qp_tl_sum), z, sizeof(z)); }
    }
    return 1;
#else
    for (int i = 0; i < n; ++i) out[i] = 0;
    return 0;
#endif
}

// Longest-processing-time-first dispatch: the iteration count of a spectrum's previous QP predicts the next one's,
// and workgroups are dispatched in blockIdx order, so starting the long problems first trims the tail of the launch
// (4 problems per CU with 2..9 iterations each otherwise leave a third of the CUs idle at the end).  Rank by brute
// force (B^2 comparisons, B ~ 1e3); ties keep index order, so the permutation is deterministic.
__global__ __launch_bounds__(256) void lpt_order_kernel(int B, const int* __restrict__ iters,
                                                        const int* __restrict__ active, int* __restrict__ order) {
    // sixteen threads per problem, each ranking it against a sixteenth of the keys (one thread against all of them took
    // 30 us per outer iteration on the critical path in front of every coneqp launch)
    extern __shared__ int keys[];
    for (int i = threadIdx.x; i < B; i += blockDim.x) keys[i] = (active && !active[i]) ? -1 : iters[i];
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x, b = t >> 4, part = t & 15;
    int rank = 0;
    if (b < B) {
        const int kb = keys[b];
        for (int j = part; j < B; j += 16) {
            const int kj = keys[j];
            rank += (kj > kb) || (kj == kb && j < b);
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) rank += __shfl_xor(rank, o, 16);
    if (b < B && part == 0) order[rank] = b;
}

void launch_lpt_order(hipStream_t st, int B, const int* iters, const int* active, int* order) {
    const int blocks = (B * 16 + 255) / 256;
    hipLaunchKernelGGL(lpt_order_kernel, dim3(blocks), dim3(256), (size_t)B * sizeof(int), st, B, iters, active, order);
}

static int launch_qp_resident(hipStream_t st, const QpArgs& a) {
    const int NP = round_up(a.n, 32);
    if (!a.Ppk) { set_error("qp resident: packed copy of P missing"); return HIPDRT_E_INVALID; }
    const bool gu = a.n > RNP_MAX;                      // inverse diagonal blocks in global memory: any n <= 2048
    if (!gu && a.waves == 4) {
        // the fat form: four wavefronts with 512 registers each (qp_resident.hpp), n <= 528
        const size_t ldsf = resident_fat_lds_bytes(NP);
        const void* ff = reinterpret_cast<const void*>(qp_kernel_resident<false, 256, 1, kFatPanel64, 2>);
        hipError_t ef = hipFuncSetAttribute(ff, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
        if (ef != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp fat): ") + hipGetErrorString(ef)); return HIPDRT_E_HIP; }
        hipLaunchKernelGGL((qp_kernel_resident<false, 256, 1, kFatPanel64, 2>), dim3(a.B), dim3(256), ldsf, st, a, NP);
        ef = hipGetLastError();
        if (ef != hipSuccess) { set_error(std::string("qp fat launch: ") + hipGetErrorString(ef)); return HIPDRT_E_HIP; }
        return HIPDRT_OK;
    }
    const size_t lds = gu ? resident_gu_lds_bytes(true) : resident_lds_bytes(NP, true);
    const void* fn = gu ? reinterpret_cast<const void*>(qp_kernel_resident<true>) : reinterpret_cast<const void*>(qp_kernel_resident<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp resident): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    if (gu) hipLaunchKernelGGL(qp_kernel_resident<true>, dim3(a.B), dim3(RT), lds, st, a, NP);
    else hipLaunchKernelGGL(qp_kernel_resident<false>, dim3(a.B), dim3(RT), lds, st, a, NP);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("qp launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

int launch_dist_var(hipStream_t st, int B, int n, const double* Ppk, long long ppk_stride, const double* Bex, int nex,
                    double* L, long long l_stride, double* out, long long out_stride, int* status) {
    if (n > GRP_NMAX) { set_error("posterior variance: n > 4096 not supported"); return HIPDRT_E_INVALID; }
    if (n > RNP_MAX) {
        const int NP = round_up(n, 32);
        const size_t lds = resident_gu_lds_bytes();
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_kernel_resident<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(cov): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
        CovArgs a;
        a.B = B; a.n = n; a.Ppk = Ppk; a.ppk_stride = ppk_stride; a.nchp = qp_nchp(n); a.Bex = Bex; a.nex = nex;
        a.L = L; a.l_stride = l_stride; a.out = out; a.out_stride = out_stride; a.status = status;
        hipLaunchKernelGGL(cov_kernel_resident<true>, dim3(B), dim3(RT), lds, st, a, NP);
        e = hipGetLastError();
        if (e != hipSuccess) { set_error(std::string("cov launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
        return HIPDRT_OK;
    }
    const int NP = round_up(n, 32);
    const size_t lds = resident_lds_bytes(NP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cov_kernel_resident<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(cov): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    CovArgs a;
    a.B = B; a.n = n; a.Ppk = Ppk; a.ppk_stride = ppk_stride; a.nchp = qp_nchp(n); a.Bex = Bex; a.nex = nex;
    a.L = L; a.l_stride = l_stride; a.out = out; a.out_stride = out_stride; a.status = status;
    hipLaunchKernelGGL(cov_kernel_resident<false>, dim3(B), dim3(RT), lds, st, a, NP);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("cov launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    return HIPDRT_OK;
}

// doubles of factor scratch per spectrum for the posterior-variance kernel: (nch + nex) x nch tiles
size_t dist_var_scratch_doubles(int n, int nex) {
    const size_t nch = (size_t)round_up(n, 32) / 16;
    return (nch + (size_t)nex) * nch * TSZ + (n > RNP_MAX ? (size_t)round_up(n, 32) * PLD : 0);   // + U when it lives outside LDS
}

// diagnostic (tools/): resident workgroups per CU the runtime reports for the coneqp kernel of n unknowns
int qp_occupancy(int threads, int n) {
    const int NP = round_up(n, 32);
    int nb = -1;
    hipError_t e;
    if (threads != 512) return -1;
    if (n > RNP_MAX) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qp_kernel_resident<true, 512>, 512, resident_gu_lds_bytes(true));
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qp_kernel_resident<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)resident_lds_bytes(NP, true));
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qp_kernel_resident<false, 512>, 512, resident_lds_bytes(NP, true));
    }
    return e == hipSuccess ? nb : -1;
}

// The group kernel's members wait for each other, so a group launch must become fully resident.  Two group launches from
// different streams could each occupy part of the CUs and wait for the rest: they are chained through one event per device
// (stream-ordered; other kernels still overlap freely).  Every wait inside the kernel is bounded on top of that.
static int launch_qp_group(hipStream_t st, const QpArgs& a, int G) {
    const int NP = round_up(a.n, 32);
    if (!a.Ppk || !a.gsync) { set_error("qp group: packed copy of P or sync words missing"); return HIPDRT_E_INVALID; }
    const size_t lds = group_lds_bytes(NP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(qp_kernel_group), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error(std::string("hipFuncSetAttribute(qp group): ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    e = hipMemsetAsync(a.gsync, 0, (size_t)a.B * GRP_WORDS * sizeof(int), st);
    if (e != hipSuccess) { set_error(std::string("qp group sync reset: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    const int blocks = 8 * G * ((a.B + 7) / 8);
    static std::mutex mtx;
    static hipEvent_t last[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mtx);
    hipEvent_t& ev = last[dev & 63];
    if (G > 1) {
        if (ev) {
            e = hipStreamWaitEvent(st, ev, 0);
            if (e != hipSuccess) { set_error(std::string("qp group chain: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
        } else {
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) { ev = nullptr; set_error(std::string("qp group event: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
        }
    }
    hipLaunchKernelGGL(qp_kernel_group, dim3(blocks), dim3(512), lds, st, a, NP, G);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error(std::string("qp group launch: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    if (G > 1) {
        e = hipEventRecord(ev, st);
        if (e != hipSuccess) { set_error(std::string("qp group record: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    }
    return HIPDRT_OK;
}

// per device: set once a group launch has found its members spread over several XCDs (the placement the fence-free
// hand-offs rely on does not hold on this device / driver / partition mode): from then on every group on that device runs
// with one member.  A launch whose members did not all become resident in time (another process's long kernel, a CU mask)
// is repeated with one member as well, but that is not remembered: it says nothing about the next launch.
static std::atomic<bool> g_group_spread[64];

int launch_qp(hipStream_t st, const QpArgs& a) {
    if (a.n > GRP_NMAX) { set_error("qp: n > 4096 not supported"); return HIPDRT_E_INVALID; }
    if (a.G < 1) {
        if (a.n > 2048) { set_error("qp: n > 2048 needs the group kernel's buffers (QpArgs.G >= 1)"); return HIPDRT_E_INVALID; }
        return launch_qp_resident(st, a);
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<bool>& spread_here = g_group_spread[dev & 63];
    int G = spread_here.load(std::memory_order_relaxed) ? 1 : a.G;
    if (G > 1) {
        // co-residency is a precondition of the members' waits: never launch more workgroups than the device can hold at once
        // by its own account (occupancy query x CUs); a launch that does not fit runs with fewer members per problem
        const int NP = round_up(a.n, 32);
        int per_cu = 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qp_kernel_group), hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds_bytes(NP));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qp_kernel_group, 512, group_lds_bytes(NP)) != hipSuccess || per_cu < 1) per_cu = 1;
        const int rounds = (a.B + 7) / 8;
        while (G > 1 && 8 * G * rounds > per_cu * device_cus()) --G;
    }
    int rc = launch_qp_group(st, a, G);
    if (rc != HIPDRT_OK || G == 1) return rc;
    // did every group find its members resident and on one XCD?  (one short synchronisation per group launch: these are the
    // launches of single fits, tens of microseconds against milliseconds of kernel)
    std::vector<int> stt((size_t)a.B);
    hipError_t e = hipMemcpyAsync(stt.data(), a.status, (size_t)a.B * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error(std::string("qp group status: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    bool aborted = false;
    for (int b = 0; b < a.B; ++b) aborted = aborted || stt[(size_t)b] == HIPDRT_QP_ABORTED;
    if (!aborted) return HIPDRT_OK;
    // why: gsync word [4] of an aborted problem is 1 (spread over XCDs), 2 (a member did not arrive in time) or 3 (a bounded wait
    // inside the factorisation expired: qp_group.hpp, trap_if)
    std::vector<int> gs((size_t)a.B * GRP_WORDS);
    e = hipMemcpyAsync(gs.data(), a.gsync, gs.size() * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error(std::string("qp group sync words: ") + hipGetErrorString(e)); return HIPDRT_E_HIP; }
    for (int b = 0; b < a.B; ++b)
        if (stt[(size_t)b] == HIPDRT_QP_ABORTED && gs[(size_t)b * GRP_WORDS + 4] == 1) spread_here.store(true, std::memory_order_relaxed);
    QpArgs again = a;
    again.redo_aborted = 1;                  // the aborted groups once more, one member each (they had changed nothing)
    return launch_qp_group(st, again, 1);
}

}  // namespace hipdrt

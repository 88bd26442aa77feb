// Batched coneqp for gfx950: cvxopt.solvers.qp(P, q, G=-I, h) as called at hybdrt/models/qphb.py:512-519,
// one workgroup per problem, the whole interior-point trajectory inside one launch (no host round trips).
//
// Algorithm = cvxopt coneprog.coneqp restricted to one 'l' cone with G = -I (restated in oracle/coneqp.py,
// SURVEY.md Appendix A): default start point, Nesterov-Todd scaling W = diag(d), Mehrotra predictor-corrector
// (STEP 0.99, EXPON 3), KKT solves through the Cholesky factor of S = P + diag(d^-2) ('chol2' solver),
// cvxopt's stopping test.  Everything FP64.  The interior-point driver is qp_common.hpp: ipm_solve; the linear
// algebra (tile-packed left-looking Cholesky, triangular sweeps) is qp_resident.hpp, one kernel for every
// n <= 2048: inverse diagonal blocks in LDS up to n = 528, in global memory beyond.  This file holds the launchers.
#include <mutex>
#include <cstdlib>
#include <cstdio>

#include "qp_common.hpp"
#include "qp_resident.hpp"

namespace hipdrt {

size_t qp_scratch_ld(int n) { return (size_t)round_up(n, 16); }

// doubles of factor scratch per problem: the tile-packed factor, beyond n = 528 followed by the inverse diagonal blocks
size_t qp_scratch_doubles(int n) { return n <= RNP_MAX ? resident_l_doubles(n) : resident_gu_doubles(n); }

int qp_profile_read(unsigned long long* out, int n, int reset) {
#ifdef HIPDRT_QP_PROFILE
    unsigned long long h[QP_PROF_SLOTS];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_qp_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < n; ++i) out[i] = i < QP_PROF_SLOTS ? h[i] : 0;
    if (reset) { unsigned long long z[QP_PROF_SLOTS] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qp_prof), z, sizeof(z)); }
    if (n > QP_PROF_SLOTS && hyper_profile_read(out + QP_PROF_SLOTS, n - QP_PROF_SLOTS, reset) < 0) return -1;
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

static int launch_qp_resident(hipStream_t st, const QpArgs& a) {
    const int NP = round_up(a.n, 32);
    if (!a.Ppk) { set_error("qp resident: packed copy of P missing"); return HIPDRT_E_INVALID; }
    const bool gu = a.n > RNP_MAX;                      // inverse diagonal blocks in global memory: any n <= 2048
    const size_t lds = gu ? resident_gu_lds_bytes() : resident_lds_bytes(NP);
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
    if (n > 2048) { set_error("posterior variance: n > 2048 not supported"); return HIPDRT_E_INVALID; }
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
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qp_kernel_resident<true, 512>, 512, resident_gu_lds_bytes());
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qp_kernel_resident<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)resident_lds_bytes(NP));
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, qp_kernel_resident<false, 512>, 512, resident_lds_bytes(NP));
    }
    return e == hipSuccess ? nb : -1;
}

int launch_qp(hipStream_t st, const QpArgs& a) {
    if (a.n > 2048) { set_error("qp: n > 2048 not supported"); return HIPDRT_E_INVALID; }
    return launch_qp_resident(st, a);
}

}  // namespace hipdrt

/* hipdrt_debug.h -- diagnostic and test hooks of libhipdrt.so.
 *
 * NOT part of the drop-in boundary (include/hipdrt.h): nothing here replaces a reference interface, and the Python host
 * layer's product path never calls these.  They exist for tests/ (kernel-choice independence of the results) and tools/
 * (in-kernel phase counters of a PROFILE build, occupancy queries).  Every hook takes a context: there is no process-wide
 * switch.
 */
#ifndef HIPDRT_DEBUG_H
#define HIPDRT_DEBUG_H

#include "hipdrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* diagnostic: in-kernel phase cycle counters of qp_kernel (workgroup 0 only), non-zero only in a build with
 * -DHIPDRT_QP_PROFILE (make PROFILE=1); slots documented in csrc/qp.hip.  Never used in timed runs.        */
int hipdrt_qp_profile(hipdrt_ctx* ctx, unsigned long long* cycles, int n, int reset);
/* diagnostic: workgroups per CU the runtime reports for the coneqp kernel of n unknowns (threads = 512)          */
int hipdrt_debug_qp_occupancy(hipdrt_ctx* ctx, int threads, int n);
/* diagnostic (tests): workgroups per problem of THIS CONTEXT's coneqp launches sized from now on (plans created on it,
 * hipdrt_qp_batch calls through it; other contexts are not affected) -- members >= 1 forces the group kernel
 * with (at most) that many members for every problem size, 0 forces the one-workgroup batch kernel (n <= 2048), -1 gives the
 * choice back to the library (few problems of n > 528, or n > 2048 -> group kernel).  The group kernel's results do not
 * depend on the group size (bit for bit); batch and group kernel differ by rounding (the batch kernel fuses the forward
 * substitution into the factorisation: another summation order), same iteration counts.                                */
int hipdrt_debug_qp_group(hipdrt_ctx* ctx, int members);
/* diagnostic (tests, tools): wavefronts per workgroup of THIS CONTEXT's batch coneqp launches at n <= 528 -- 4 = the fat form
 * (four wavefronts, one per SIMD, 512 registers each), 8 = eight wavefronts with 256 registers, -1 = the library's choice.
 * A context starts from the environment variable HIPDRT_QP_WAVES (4 | 8) when it is set.  Both kernels run the same arithmetic
 * in the same order: the results are the same bits (tests/test_gpu_qp.py).                                             */
int hipdrt_debug_qp_waves(hipdrt_ctx* ctx, int waves);
/* diagnostic (tests): on = 0 makes THIS CONTEXT's fits visit the exact zeros of the penalty matrices as well -- the Gram
 * epilogue adds the L2 part to every tile and the hyper kernel's Toeplitz convolutions run over all columns instead of the
 * penalties' reach (csrc/gram.hip, csrc/hyper.hip).  The results are the same bits either way; tests/test_gpu_fit.py checks it. */
int hipdrt_debug_exact_zero_shortcuts(hipdrt_ctx* ctx, int on);
/* diagnostic (tests): the library's own streams on the context's device (hipdrt.h: hipdrt_stream) -- how many there are
 * (return value through *size), and for the first min(*size, cap) of them the hipStream_t, the number of contexts holding it and
 * the number of device loops running on it right now.  Any of the three arrays may be NULL.                               */
int hipdrt_debug_stream_pool(hipdrt_ctx* ctx, int cap, void** streams, int* holders, int* running, int* size);

#ifdef __cplusplus
}
#endif

#endif /* HIPDRT_DEBUG_H */

// Internal declarations shared by the HIP translation units of libhipdrt.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hipdrt.h"
#include "../../include/hipdrt_debug.h"

namespace hipdrt {

void set_error(const std::string& msg);

#define HIPDRT_CHECK(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            hipdrt::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +       \
                              __FILE__ + ":" + std::to_string(__LINE__) + ")");                \
            return HIPDRT_E_HIP;                                                               \
        }                                                                                      \
    } while (0)

#define HIPDRT_REQUIRE(cond, msg)                                                              \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            hipdrt::set_error(std::string("invalid argument: ") + msg);                        \
            return HIPDRT_E_INVALID;                                                           \
        }                                                                                      \
    } while (0)

// RAII device buffer (freed on destruction; the ctx/plan own these)
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool own = true;          // false: a window into another buffer (sub-batch views of a plan), never freed here
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && own) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        own = true;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 8;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    // window [offset, offset + n) of `src` (nothing when src is empty)
    void alias(const DevBuf& src, size_t offset, size_t n) {
        release();
        if (!src.p) return;
        p = static_cast<char*>(src.p) + offset;
        bytes = n;
        own = false;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
    double* d() const { return static_cast<double*>(p); }
    int* i() const { return static_cast<int*>(p); }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace hipdrt

struct hipdrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // one of the library's own streams (api.hip: StreamPool), held for the context's lifetime
    int pool_idx = -1;
    int num_cu = 0;
    size_t hbm_bytes = 0;
    std::string arch;
    // lifetime: plans refer to their context, and a garbage collector may release the two in either order -- a released
    // context lives on until its last plan is destroyed
    int plans = 0;
    bool released = false;
    // hipdrt_debug_qp_group (include/hipdrt_debug.h; tests): workgroups per problem of this context's coneqp launches,
    // -1 = the library chooses
    int qp_force_group = -1;
    // hipdrt_debug_exact_zero_shortcuts (tests): 0 = the Gram epilogue and the hyper kernel visit the penalty matrices' exact
    // zeros as well (no reach restriction, no tile skip)
    int zero_shortcuts = 1;
    // hipdrt_debug_qp_waves (tests, tools): wavefronts per workgroup of this context's batch coneqp launches at n <= 528 --
    // 4 = the fat form, 8 = eight wavefronts, -1 = the library chooses.  Initialised from HIPDRT_QP_WAVES when the context is made.
    int qp_waves = -1;
};

// ---- launchers implemented in the .hip files (all asynchronous on `st`) ---------------------------------
namespace hipdrt {

// api.hip: the library's own streams (one per hardware queue) exist from here on
void ensure_stream_pool(int device);

// matrices.hip
void launch_lookup(hipStream_t st, double eps, int ngrid, int ny, const double* wt_re, const double* wt_im,
                   double* z_re, double* z_im);
void launch_lookup_slopes(hipStream_t st, int ngrid, const double* xp, const double* fp, double* slopes);
// lut = {log_wt_re, z_re, slope_re, log_wt_im, z_im, slope_im} each [ngrid], device
void launch_impedance_matrix(hipStream_t st, int B, int freq_batched, const double* freq, int nf, const double* tau,
                             int ntau, int mode, int toeplitz, double eps, int ngrid, const double* lut6, int ny,
                             double* a_re, double* a_im, double* cr_scratch);
void launch_phasor_z(hipStream_t st, const double* freq, int nf, const double* nu, int nnu, double eps, double* zre,
                     double* zim);
void launch_phasor_v(hipStream_t st, const double* times, int nt, const double* nu, int nnu, double eps,
                     const double* step_times, const double* step_sizes, int nsteps, double* rm, double* layered);
void launch_chrono_vmm(hipStream_t st, const double* tt, int nt, const int* seg, int nseg, double eps, int uniform,
                       double* vmm);
void launch_response_lookup(hipStream_t st, double eps, int ngrid, int ny, const double* td, double* v);
void launch_response_matrix(hipStream_t st, const double* times, int nt, const double* tau, int ntau,
                            const double* step_times, const double* step_sizes, int nsteps, int mode, double eps,
                            int ngrid, const double* lut3, int ny, double* a, double* layered);
void launch_response_variant(hipStream_t st, const double* times, int nt, const double* tau, int ntau,
                             const double* step_times, const double* step_sizes, const double* tau_rise, int nsteps,
                             int variant, double eps, int ny, double* a, double* layered);
void launch_nonuniform_gauss(hipStream_t st, const double* y, int n, const double* sigma, const int* seg_of, const int* seg,
                             const double* nodes, int K, const double* node_delta, const double* weights, const int* woff,
                             const int* radius, double* out);
void launch_penalty(hipStream_t st, const double* ln_tau, int n, double eps, int toeplitz, double* m0, double* m1,
                    double* m2, int ld, int pad);
void launch_eis_vmm(hipStream_t st, const double* freq, int nf, double vmm_eps, double reim_cor, int uniform,
                    double* vmm);

// gram.hip
struct GramL2 {
    // explicit L2 (stand-alone API) ...
    const double* l2;
    long long l2_stride;
    int ldl2;
    // ... or hyper-parameter form (fit loop)
    const double* mk[3];   // padded penalty matrices [n][ldm]
    int ldm;
    const double* s;       // [B][3][n]
    const double* rho;     // [B][3]
    double dfac[3];        // l2_lambda_0 * derivative_weight[k]   (0 => order skipped)
    int ns;
    int use_rho;
    int sym;               // penalty matrices are bitwise symmetric (Toeplitz build): read them along rows
    int toep;              // the DRT block (indices >= ns) of every mk is symmetric Toeplitz: mk[i][j] = mk[ns][ns + |i - j|]
    int toep_maxd;         // largest |i - j| at which any order's first row is non-zero (-1: not known): the Gaussian penalties
                           // underflow to exactly 0 a few dozen grid points off the diagonal, so a tile further out adds exact zeros
    int spec_zero;         // every penalty entry that couples a special parameter with a DRT coefficient is exactly zero
    // x_dop block (qphb.py:92-100): entries with both indices in [dop_start, dop_start + dop_size) are scaled by
    // dop_dfac[k] * dop_rho[b][k] instead
    int dop_start, dop_size;
    const double* dop_rho; // [B][3]
    double dop_dfac[3];    // dop_l2_lambda_0 * dop_derivative_weight[k]
};
// a_stride: doubles between the response matrices of consecutive problems (0 = one shared matrix)
void launch_gram_l2(hipStream_t st, int B, int m, int n, const double* A, int lda, const double* w, const GramL2& g,
                    double* P, int ldp, long long p_stride, const int* active, double* Ppk = nullptr,
                    long long ppk_stride = 0, int nchp = 0, long long a_stride = 0);
// row-major symmetric P -> accumulator-native lower tiles
void launch_pack_p(hipStream_t st, int B, int n, const double* P, int ldp, long long p_stride, double* Ppk,
                   long long ppk_stride, int nchp);
inline int qp_nchp(int n) { return round_up(n, 32) / 16; }
inline size_t qp_ppk_doubles(int n) { return (size_t)qp_nchp(n) * qp_nchp(n) * 256; }
void launch_qvec_batched(hipStream_t s, int B, int m, int n, const double* rm, int ldrm, const double* w, const double* y,
                         const double* l1, double l1_scalar, double* q, const int* active);
void launch_qvec(hipStream_t st, int B, int m, int n, const double* A, int lda, const double* w, const double* y,
                 const double* l1, double l1_scalar, double* q, const int* active, long long a_stride = 0);
void launch_weighted_gram(hipStream_t st, int B, int m, int n, const double* A, int lda, const double* w,
                          const double* b, const double* l2, long long l2_stride, int ldl2, const double* l1,
                          double* P, int ldp, long long p_stride, double* q, const int* active);

// hyper.hip: device-resident state of a batched fit (passed by value to the kernels)
struct FitState {
    int nf, m, n, ns, ldrm, ldm;
    int toeplitz_m;        // penalty blocks are symmetric Toeplitz (uniform ln-tau grid): first column suffices
    int toep_reach;        // ... and exactly zero beyond this distance from the diagonal in every order (-1: not known)
    int continue_mode;     // 1: warm restart (_continue_from_init): xmx norms stay frozen, no rescaling;
                           // 2: a bare iterate_qphb (hipdrt_plan_iterate): as 1 and no vz_offset column rewrite
    int min_iter;          // a spectrum may only stop once it has done this many outer iterations (fit: 1)
    double basis_area;     // area of one tau basis function (sqrt(pi) / epsilon): predict_r_p of update_scale
    hipdrt_fit_opts opts;
    // prepared-matrix plans (hipdrt_plan_create_prepared): any data type, optional x_dop block and vz_offset column
    int prepared;
    hipdrt_prepared_desc desc;
    long long rm_stride;   // doubles between the response matrices of consecutive spectra (0 = shared)
    double* rm_rw;         // == rm when the matrices are per spectrum (the vz_offset column is rewritten every iteration)
    const double* vz_strength;   // [m]
    const double* vz_entry;      // [B][m] or null: warm restarts (continue_mode 1) predict the vz_offset column from a matrix whose
                                 // offset column is frozen as it stood when the restart began (drt1d.py:1295-1298), not zero
    double *dop_rho, *dop_xmx;   // [B][3]
    double* outlier_t;           // [B][m] or null: 1 - posterior outlier probability of the last estimate_weights (outlier_p set)
    double* hist_dop_rho;
    // shared (plan) matrices
    const double* rm;      // [m][ldrm]  stacked [Re; Im] response matrix incl. special columns
    const double* vmm;     // [m][m]
    const double* vmm_iw;  // [m][m]  variance matrix of initialize_weights (self-excluded rows when outlier_p is set)
    const double* mk[3];   // [n][ldm]   padded penalty matrices
    // per-spectrum
    const double *z_re, *z_im;   // [B][nf]
    double *rv, *w, *est_w;      // [B][m]
    double *x, *x_in;            // [B][n]
    double* s;                   // [B][3][n]
    double *rho, *xmx;           // [B][3]
    double *coef_scale, *var_floor;   // [B]
    int *active, *outer_iters, *fit_status, *qp_iters_total, *qp_status, *qp_iters;   // [B]
    int* n_active;               // [1]
    // few, large fits: the three matrix-vector products of an outer iteration (rm @ x, vmm @ resid^2, rm @ x for the
    // vz_offset column) computed by a many-workgroup kernel before hyper_kernel, which then only reads them ([3][B][m], or null)
    double* premv;
    int premv_batched;     // premv is filled by batch_products_kernel (many fits sharing rm and vmm: two products on the matrix pipe)
    // optional history of one spectrum
    int hist_b, hist_cap;
    double *hist_x, *hist_w, *hist_rho;
    int *hist_qp, *hist_rows;
};
size_t hyper_lds_bytes(int n, int m, int ns);
int launch_prep(hipStream_t s, const FitState& st, int B);
// stage 0: weights from the first overfit only (outlier branch); stage 1: final est_weights -> init weights, x reset
// stage 2: est_weights of rows [r0, r1) only (init_weights_separately); stage 3: final step only
int launch_init_weights(hipStream_t s, const FitState& st, int B, int stage, int r0 = 0, int r1 = 0);
void launch_row_mask(hipStream_t s, int B, int m, int r0, int r1, double* w);
void launch_weight_method(hipStream_t s, const FitState& st, int B, double fixed_chrono, double fixed_eis, double* wrow,
                          double* wfac);
void launch_vmm_exclude_self(hipStream_t s, const double* vmm, int m, double* out);
int launch_hyper(hipStream_t s, const FitState& st, int B, int it);
int device_cus();      // compute units of the current device (qp.hip)
void launch_scale_weights(hipStream_t s, const FitState& st, int B, double factor);
void launch_scale_rows(hipStream_t s, int B, int m, const double* w, const double* rows, int batched, double factor,
                       const int* active, double* out);
void launch_copy_column(hipStream_t s, int B, int m, const double* rm, long long rm_stride, int ldrm, int col, double* out);
int launch_llh(hipStream_t s, const FitState& st, int B, double* rss, double* slw, int stored = 0, double scalar_w = 1.0);
void launch_assemble_rm(hipStream_t s, const FitState& st, const double* a_re, const double* a_im, const double* freq,
                        double* rm, int idx_rinf, int idx_induc);
void launch_special_penalty(hipStream_t s, double* m0, double* m1, double* m2, int ld, int idx_rinf, int idx_induc,
                            double pen_r, double pen_l);
void launch_make_h(hipStream_t s, double* h, int n, int ns, int nonneg);

// qp.hip
struct QpArgs {
    int B, n;
    const double* P;      // [B or 1][n][ldp]
    long long p_stride;   // 0 if shared
    int ldp;
    const double* Ppk;    // optional [B or 1][nchp][nchp][256]: P in accumulator-native 16x16 tiles (lower tiles)
    long long ppk_stride; // 0 if shared
    int nchp;
    const double* q;      // [B][n]
    const double* h;      // [B or 1][n]
    long long h_stride;
    double* L;            // scratch [B][n_pad][ldl]
    int ldl;
    long long l_stride;
    double* x;            // [B][n] out
    int* iters;           // [B] out (may be null)
    double* pcost;        // [B] out (may be null)
    int* status;          // [B] out
    const int* active;    // [B] or null: skip problems with active[b]==0
    const int* order = nullptr;   // [B] or null: workgroup i solves problem order[i] (longest-first dispatch)
    int* iters_accum;     // [B] or null: += iterations
    double* state;        // scratch [B * max(G, 1)][17][state_ld] for the IPM iterates (one copy per member of a group)
    int state_ld;
    long long state_stride;
    hipdrt_qp_opts opts;
    // G >= 1: the group kernel (qp_group.hpp), every problem on G co-resident workgroups (qp_group_size picks G when the
    // buffers are sized); 0: the batch kernel, one workgroup per problem
    int G = 0;
    int* gsync = nullptr; // [B][qp_gsync_ints()] global sync words of the groups (zeroed by the launcher)
    int redo_aborted = 0; // group kernel: only the problems whose status is HIPDRT_QP_ABORTED (second pass of launch_qp)
    int waves = 0;        // batch kernel, n <= 528: 4 = the fat form (four wavefronts x 512 registers), anything else = eight wavefronts
};
int launch_qp(hipStream_t st, const QpArgs& a);
// workgroups per problem for a launch of B problems of n unknowns: 0 = batch kernel (one workgroup per problem, n <= 2048),
// >= 1 = group kernel with that many members (few problems, or n > 2048); -1 = n not supported
// (`force`: the context's hipdrt_debug_qp_group setting, -1 = automatic)
int qp_group_size(int B, int n, int force = -1);
size_t qp_gsync_ints();
// posterior variance on an evaluation grid (qp_resident.hpp: cov_kernel_resident); Bex = evaluation rows as packed tiles
int launch_dist_var(hipStream_t st, int B, int n, const double* Ppk, long long ppk_stride, const double* Bex, int nex,
                    double* L, long long l_stride, double* out, long long out_stride, int* status);
size_t dist_var_scratch_doubles(int n, int nex);
void launch_rows_outer(hipStream_t st, const double* Y, int nch, int ncol, int nex, int nrow, double scale, double* out, int ld);
void launch_pack_rows(hipStream_t st, int nrow, int ncol, int col_offset, const double* M, int ldm, int ntile_rows,
                      double* tiles, int nchp);
// order[] = problem indices sorted by descending iteration count of the previous solve (inactive ones last)
void launch_lpt_order(hipStream_t st, int B, const int* iters, const int* active, int* order);
size_t qp_scratch_ld(int n);
size_t qp_scratch_doubles(int n, int G = 0);
inline int qp_state_ld(int n) { return round_up(n, 32) + 32; }
inline size_t qp_state_doubles(int n) { return (size_t)17 * qp_state_ld(n); }
int qp_profile_read(unsigned long long* out, int n, int reset);
int hyper_profile_read(unsigned long long* out, int n, int reset);   // slots 48.. of hipdrt_qp_profile (hyper.hip)
int qp_occupancy(int threads, int n);

}  // namespace hipdrt

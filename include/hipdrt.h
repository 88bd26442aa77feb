/* hipdrt.h -- C-ABI of libhipdrt.so, the MI355X (gfx950) implementation of hybrid-drt's hot path.
 *
 * The reference (jdhuang-csm/hybrid-drt) is pure Python; it has no FFI of its own.  The entry points below
 * are what a ctypes binding for the hot path replaces, each citing the reference interface it stands for
 * (paths relative to the reference root).  Conventions:
 *   - plain C: opaque handles, pointers and sizes only; no C++/torch types;
 *   - every array is row-major contiguous float64 (int32 for counters/status) unless stated otherwise;
 *   - "host" pointers are caller-owned host memory, the library does the H2D/D2H copies;
 *     "_dev" entry points take device pointers and neither copy nor synchronise more than documented;
 *   - every function returns 0 on success, <0 on error (HIPDRT_E_*); hipdrt_last_error() gives the text;
 *     nothing throws, nothing falls back to the CPU: without a gfx950 device hipdrt_create fails;
 *   - one hipdrt_ctx per (process, device); a ctx is not re-entrant, distinct ctxs are independent.
 */
#ifndef HIPDRT_H
#define HIPDRT_H

#ifdef __cplusplus
extern "C" {
#endif

#define HIPDRT_OK 0
#define HIPDRT_E_INVALID (-1)   /* bad argument                                         */
#define HIPDRT_E_HIP (-2)       /* HIP runtime error (text in hipdrt_last_error)         */
#define HIPDRT_E_NODEVICE (-3)  /* no usable gfx950 device                              */
#define HIPDRT_E_NUMERIC (-4)   /* numerical breakdown for the whole call               */

/* per-problem status codes written to status[] arrays */
#define HIPDRT_QP_OPTIMAL 0     /* coneqp stopping test met                                        */
#define HIPDRT_QP_MAXITER 1     /* maxiters reached (cvxopt status 'unknown')                      */
#define HIPDRT_QP_SINGULAR_LATE 2 /* Cholesky breakdown after iteration 0: cvxopt returns current x */
#define HIPDRT_QP_SINGULAR (-1) /* breakdown at the start point: cvxopt raises ValueError          */
#define HIPDRT_QP_ABORTED (-2)  /* internal to the several-workgroups-per-problem kernel: the workgroups of this problem were
                                   not placed on one XCD; the launcher repeats such a problem on one workgroup before it
                                   returns, so a caller sees this only if that repeat could not run     */

#define HIPDRT_MODE_INTERP 0    /* integrate_method='interp' (drtbase.py:155)  */
#define HIPDRT_MODE_TRAPZ 1     /* integrate_method='trapz'  (drtbase.py:159)  */

typedef struct hipdrt_ctx hipdrt_ctx;
typedef struct hipdrt_plan hipdrt_plan;
typedef struct hipdrt_comm hipdrt_comm;

/* ---- context -------------------------------------------------------------------------------------- */
int hipdrt_create(int device, hipdrt_ctx** out);
int hipdrt_destroy(hipdrt_ctx* ctx);   /* with plans still alive on it the context is freed by the last hipdrt_plan_destroy */
const char* hipdrt_last_error(void);
/* HIP stream the ctx launches on (so callers can record events / order other work): returns hipStream_t.
 * The stream is one of the library's own (created once per device, one per hardware queue beside the null stream's:
 * GPU_MAX_HW_QUEUES - 1, i.e. 3 by default and 7 under the host layer's default of 8; HIPDRT_STREAM_POOL=<n> overrides) and is
 * held by the context for its lifetime; contexts beyond that count share streams (their work is then ordered with each other's,
 * never wrong).  The ranges of a sub-batched hipdrt_plan_fit borrow the least busy of these streams for the duration of the call. */
void* hipdrt_stream(hipdrt_ctx* ctx);
int hipdrt_synchronize(hipdrt_ctx* ctx);
/* name of the device architecture, e.g. "gfx950" */
int hipdrt_device_info(hipdrt_ctx* ctx, char* arch, int arch_len, int* num_cu, long long* hbm_bytes);

/* ---- L1 kernel matrices --------------------------------------------------------------------------- */

/* basis.generate_impedance_lookup (hybdrt/matrices/basis.py:648-669), Gaussian basis.
 * in : wt_re[ngrid], wt_im[ngrid]  the omega*tau abscissae (np.logspace(-2.7,2.7,ngrid) / (-5.4,5.4))
 * out: z_re[ngrid], z_im[ngrid]    ny-point trapezoid integrals over y = linspace(-20,20,ny)          */
int hipdrt_impedance_lookup(hipdrt_ctx* ctx, double epsilon, int ngrid, int ny,
                            const double* wt_re, const double* wt_im, double* z_re, double* z_im);

/* mat1d.construct_impedance_matrix (hybdrt/matrices/mat1d.py:212-374), both parts in one call,
 * batched over B frequency grids.
 * freq[B or 1][nf] (freq_batched selects), tau[ntau]; mode INTERP uses the lookups (log_wt_*, z_* of
 * length ngrid, np.interp semantics incl. end clamping); mode TRAPZ integrates ny points.
 * toeplitz != 0 reproduces the reference's Toeplitz shortcut (mat1d.py:353-360): first column/row are
 * evaluated and scattered (the caller decides with the reference's rule, it requires B-independent grids).
 * out: a_re[B][nf][ntau], a_im[B][nf][ntau]                                                          */
int hipdrt_impedance_matrix(hipdrt_ctx* ctx, int B, int freq_batched, const double* freq, int nf,
                            const double* tau, int ntau, int mode, int toeplitz, double epsilon,
                            int ngrid, const double* log_wt_re, const double* z_re,
                            const double* log_wt_im, const double* z_im, int ny,
                            double* a_re, double* a_im);
/* same, outputs stay on the device (a_re_dev/a_im_dev are device pointers); used by bench.py to time the
 * build with HIP events without the D2H copy.  elapsed_ms (may be NULL) = kernel time over `repeat` launches */
int hipdrt_impedance_matrix_dev(hipdrt_ctx* ctx, int B, int freq_batched, const double* freq, int nf,
                                const double* tau, int ntau, int mode, int toeplitz, double epsilon,
                                int ngrid, const double* log_wt_re, const double* z_re,
                                const double* log_wt_im, const double* z_im, int ny,
                                void* a_re_dev, void* a_im_dev, int repeat, float* elapsed_ms);

/* phasance.construct_phasor_z_matrix (hybdrt/matrices/phasance.py:108-118), nu_basis_type='gaussian', normalize=False:
 * the distribution-of-phasances impedance columns.  out: zm_re, zm_im [nf][n_nu] (real and imaginary part).           */
int hipdrt_phasor_z_matrix(hipdrt_ctx* ctx, const double* freq, int nf, const double* basis_nu, int n_nu,
                           double nu_epsilon, double* zm_re, double* zm_im);
/* phasance.construct_phasor_v_matrix (phasance.py:121-144), gaussian nu basis, galvanostatic ideal steps.
 * out: rm[nt][n_nu] = sum over steps; layered[nsteps][nt][n_nu] (may be NULL)                                        */
int hipdrt_phasor_v_matrix(hipdrt_ctx* ctx, const double* times, int nt, const double* basis_nu, int n_nu,
                           double nu_epsilon, const double* step_times, const double* step_sizes, int nsteps,
                           double* rm, double* layered);

/* mat1d.construct_chrono_var_matrix (hybdrt/matrices/mat1d.py:457-490).  tt[nt] = the samples' transformed times
 * (utils/chrono.py:5-44 fwd_transform), seg[nseg+1] = sample-index bounds of the step segments
 * ([0, step indices..., nt], preprocessing.py:161-178); uniform != 0 -> error_structure='uniform' (ones / nt).
 * out: vmm[nt][nt], Gaussian in transformed time inside a segment, zero across segments, rows normalised.   */
int hipdrt_chrono_var_matrix(hipdrt_ctx* ctx, const double* tt, int nt, const int* seg, int nseg,
                             double vmm_epsilon, int uniform, double* vmm);

/* basis.generate_response_lookup (hybdrt/matrices/basis.py:672-689; integrand basis.py:616-618), Gaussian basis,
 * galvanostatic ideal step.
 * in : td[ngrid]   the (t - t_step)/tau abscissae (np.logspace(-6, 2, ngrid))
 * out: v[ngrid]    ny-point trapezoid integrals over y = linspace(-20,20,ny)                            */
int hipdrt_response_lookup(hipdrt_ctx* ctx, double epsilon, int ngrid, int ny, const double* td, double* v);

/* mat1d.construct_response_matrix (hybdrt/matrices/mat1d.py:16-122) for basis_type='gaussian', op_mode='galv',
 * step_model='ideal'.  times[nt], tau[ntau], step_times/step_sizes[nsteps]; mode INTERP uses the lookup
 * (log_td, v of length ngrid, np.interp semantics incl. end clamping), mode TRAPZ integrates ny points.
 * out: a[nt][ntau] = sum over steps; layered[nsteps][nt][ntau] per step (may be NULL)                   */
int hipdrt_response_matrix(hipdrt_ctx* ctx, const double* times, int nt, const double* tau, int ntau,
                           const double* step_times, const double* step_sizes, int nsteps, int mode,
                           double epsilon, int ngrid, const double* log_td, const double* v, int ny,
                           double* a, double* layered);

/* The non-default forms of mat1d.construct_response_matrix (hybdrt/matrices/mat1d.py:96-118), Gaussian basis:
 *   HIPDRT_RESPONSE_POT       op_mode='pot' (mat1d.py:114-118): exp(-(t - t_k) / tau) * unit_step(t, t_k) * size_k, rows before a
 *                             step 0 (tau_rise, epsilon, ny unused)
 *   HIPDRT_RESPONSE_EXPDECAY  op_mode='galv', step_model='expdecay', integrate_method='trapz' (integrand basis.py:619-637):
 *                             tau_rise[nsteps] = rise time of every step, ny-point trapezoid over y = linspace(-20, 20, ny)
 * same outputs as hipdrt_response_matrix                                                                    */
#define HIPDRT_RESPONSE_POT 0
#define HIPDRT_RESPONSE_EXPDECAY 1
int hipdrt_response_matrix_variant(hipdrt_ctx* ctx, const double* times, int nt, const double* tau, int ntau,
                                   const double* step_times, const double* step_sizes, const double* tau_rise, int nsteps,
                                   int variant, double epsilon, int ny, double* a, double* layered);

/* filters.nonuniform_gaussian_filter1d (hybdrt/filters/_filters.py:261-343; order 0, mode 'reflect', empty=False) applied
 * segment by segment: the anti-aliasing filter of the chrono down-sampling (preprocessing.filter_chrono_signal, 507-572,
 * called by downsample_data, 423-432).  y[n], sigma[n] (per-sample filter widths in samples, already capped);
 * seg[nseg+1] sample-index bounds of the step segments; per segment s its log-spaced sigma nodes nodes[s*K .. s*K+K)
 * (0 = unused slot), node_delta[nseg] = log spacing of the nodes, radius[s*K+k] = int(4 sigma + 0.5) or -1 for a node
 * below min_sigma (output = input), weights[woff[s*K+k] + j], j = 0..radius, the normalised Gaussian kernel halves
 * (scipy.ndimage._filters._gaussian_kernel1d); filtered[s] = 0 skips a segment (all widths zero).  out[n].
 * The caller (hipdrt.filters / hipdrt.preprocessing) derives nodes and kernels exactly as the reference does. */
int hipdrt_nonuniform_gaussian_filter1d(hipdrt_ctx* ctx, const double* y, int n, const double* sigma, const int* seg,
                                        int nseg, const int* filtered, const double* nodes, int K,
                                        const double* node_delta, const double* weights, long long nweights,
                                        const int* woff, const int* radius, double* out);

/* mat1d.construct_integrated_derivative_matrix (hybdrt/matrices/mat1d.py:125-209), orders 0,1,2 of the
 * Gaussian basis (closed forms basis.py:382-395).  toeplitz != 0: first column scattered (mat1d.py:158-168).
 * out: m0, m1, m2 each [n][n]                                                                        */
int hipdrt_penalty_matrices(hipdrt_ctx* ctx, const double* ln_tau, int n, double epsilon, int toeplitz,
                            double* m0, double* m1, double* m2);

/* mat1d.construct_eis_var_matrix (hybdrt/matrices/mat1d.py:493-515): out vmm[2nf][2nf], rows normalised.
 * uniform != 0 is error_structure='uniform'.                                                         */
int hipdrt_eis_var_matrix(hipdrt_ctx* ctx, const double* freq, int nf, double vmm_epsilon, double reim_cor,
                          int uniform, double* vmm);

/* ---- L2 QP ------------------------------------------------------------------------------------------ */

typedef struct {
    double abstol, reltol, feastol; /* cvxopt.solvers.options defaults 1e-7, 1e-6, 1e-7 */
    int maxiters;                   /* 100 */
} hipdrt_qp_opts;

/* cvxopt.solvers.qp(P, q, G=-I, h) as called from qphb.solve_convex_opt (hybdrt/models/qphb.py:512-519):
 * B independent problems min 1/2 x'Px + q'x s.t. -x <= h, solved with coneqp's trajectory.
 * P[B or 1][n][n] (p_batched selects; only the lower triangle is read), q[B][n], h[B or 1][n].
 * out: x[B][n], iters[B], pcost[B] ('primal objective'), status[B]
 * n <= 4096.  Many problems: one workgroup each (n <= 2048).  Few problems (B <= #CUs / 16, n > 256) or n > 2048: every
 * problem on up to 32 co-resident workgroups of one XCD; same iteration counts, x equal to rounding.   */
int hipdrt_qp_batch(hipdrt_ctx* ctx, int B, int n, int p_batched, const double* P, const double* q,
                    int h_batched, const double* h, const hipdrt_qp_opts* opts,
                    double* x, int* iters, double* pcost, int* status);

/* (diagnostic entry points -- kernel phase counters, occupancy, forcing a kernel choice -- are declared in hipdrt_debug.h:
 * they are not part of the drop-in boundary and nothing in the host layer's product path calls them)                          */

/* P = (W A)'(W A) + L2, q = -(W A)'(W b) + l1 of qphb.solve_convex_opt (qphb.py:465-466) for B weight
 * vectors over one shared A[m][n]:  w[B][m], b[B][m], l2[B or 1][n][n], l1[n] -> P[B][n][n], q[B][n]   */
int hipdrt_weighted_gram(hipdrt_ctx* ctx, int B, int m, int n, const double* A, const double* w,
                         const double* b, int l2_batched, const double* l2, const double* l1,
                         double* P, double* q);

/* ---- L3/L4 batched fit ------------------------------------------------------------------------------ */

typedef struct {
    /* qphb.get_default_hypers (hybdrt/models/qphb.py:208-255), eff_hp=True */
    double rp_scale;                 /* 14 */
    double derivative_weights[3];    /* 1.5, 1.0, 0.5 */
    double sigma_ds[3];              /* 1, 1000, 1000 */
    double l1_lambda_0;              /* 0 */
    double l2_lambda_0;              /* 142 */
    double s_alpha[3];               /* 5, 10, 25 */
    double s_0[3];                   /* 1, 1, 1 */
    double rho_alpha[3];             /* 0.15, 0.2, 0.25 */
    double rho_0[3];                 /* 1, 1, 1 */
    /* DRT._qphb_fit_core keyword defaults (hybdrt/models/drt1d.py:102-137) */
    double iw_l1_lambda_0, iw_l2_lambda_0;   /* 1e-4, 1e-4 */
    double ohmic_penalty, inductance_penalty; /* 1e-6, 1e-6 */
    double inductance_scale;         /* 1e-5 */
    double eis_vmm_epsilon, eis_reim_cor;    /* 0.25, 0.25 */
    double xtol;                     /* 1e-2 */
    int max_iter;                    /* 50 */
    int nonneg;                      /* 1 */
    int scale_data;                  /* 1 */
    int fit_ohmic, fit_inductance;   /* 1, 1 */
    int eis_error_uniform;           /* 0 (eis_error_structure=None) */
    int update_scale;                /* 0; 1: re-scale the data every iteration from the second on (drt1d.py:903-927) */
    int eff_hp;                      /* 1; 0: solve_s sees rho_k instead of 1 (qphb.py:747-750; other default alphas) */
    /* optional branches of the weight estimation; <= 0 means None (the reference defaults) */
    double outlier_p;                /* prior outlier probability, qphb.py:1497-1553, 1629-1656 */
    double iw_alpha, iw_beta;        /* prior on the initial weights, qphb.py:1471-1479, 1679 */
    hipdrt_qp_opts qp;
} hipdrt_fit_opts;

void hipdrt_default_fit_opts(hipdrt_fit_opts* o);

/* A plan = everything that does not depend on the measured impedances: lookups (DRTBase.__init__,
 * hybdrt/models/drtbase.py:138-156), Z'/Z'' (DRT._prep_impedance_fit_matrix, drt1d.py:5625-5657), penalty
 * matrices (_prep_penalty_matrices, 5673-5734), the stacked [Re;Im] response matrix and padded M_k
 * (_format_qp_matrices, 5736-5963), the variance-estimation matrix (drt1d.py:622-636), all built on the
 * device, plus work space for `capacity` spectra.  All spectra of a plan share freq[nf] and tau[ntau]
 * (the DRTMD case: mapping/drtmd.py:245-319 with one DRT instance and its recalc cache).
 * toeplitz_a / toeplitz_m carry the reference's Toeplitz decisions (host logic); wt_* are the lookup
 * abscissae and log_wt_* = np.log(wt_*) (both computed by the caller exactly as basis.py:655-669 does). */
int hipdrt_plan_create(hipdrt_ctx* ctx, const double* freq, int nf, const double* tau, int ntau,
                       double epsilon, int mode, int toeplitz_a, int toeplitz_m, int ngrid, int ny,
                       const double* wt_re, const double* wt_im, const double* log_wt_re,
                       const double* log_wt_im, const hipdrt_fit_opts* opts, int capacity,
                       hipdrt_plan** out);
int hipdrt_plan_destroy(hipdrt_plan* plan);
/* dimensions: n = ns + ntau unknowns, m = 2 nf rows, ns special parameters */
int hipdrt_plan_dims(hipdrt_plan* plan, int* n, int* m, int* ns);
/* copy a shared matrix of the plan back to the host: which = "lut_z_re","lut_z_im" [ngrid],
 * "a_re","a_im" [nf][ntau], "rm" [m][n], "m0","m1","m2" [n][n] (padded), "vmm" [m][m]                  */
int hipdrt_plan_get(hipdrt_plan* plan, const char* which, double* out, long long count);
/* replace the plan's lookup tables with externally supplied ones (multi-GPU: rank 0 builds them, RCCL
 * broadcasts, the other ranks install them) and rebuild the dependent matrices            */
int hipdrt_plan_set_lookup(hipdrt_plan* plan, const double* z_re, const double* z_im);

/* stage B <= capacity spectra (z_re[B][nf], z_im[B][nf], host) into the plan's device buffers */
int hipdrt_plan_upload(hipdrt_plan* plan, int B, const double* z_re, const double* z_im);
/* DRT._qphb_fit_core (drt1d.py:102-1104) for the staged spectra, entirely on the device: scale_data,
 * initialize_weights (qphb.py:1609-1681), the iterate_qphb loop (qphb.py:606-972) with per-spectrum
 * convergence masks, calculate_pq's q (qphb.py:1154-1183).  Asynchronous on the ctx stream except for one
 * 4-byte "all converged" read-back per outer iteration.                                                */
int hipdrt_plan_fit(hipdrt_plan* plan);
/* How many contiguous ranges ("sub-batches") hipdrt_plan_fit cuts the staged spectra into: every range runs the same device
 * loop on a stream of its own, side by side, inside the one call and the plan's own buffers -- what DRTMD's serial loop over
 * observations (hybdrt/mapping/drtmd.py:303-319) becomes when the tail of one range's launches is filled by the others'.
 * k = 0 (default): chosen from the batch size -- 1 below 600 spectra, 2 from 600 on, 4 from 1000 on (capped at the number of the
 * library's own streams, see hipdrt_stream: 3 under the runtime's default of 4 hardware queues; profiles/r06_subbatch_sweep.txt); the
 * ranges run on those streams, picked per fit by activity and compute pipe, so the choice no longer depends on what else the process
 * has created (profiles/r06_trace_queue_placement.txt).  k >= 1: that many (capped so that a range keeps >= 64 spectra).  A caller
 * that keeps four or more plans fitting at the same time (own contexts, own threads) should set k = 1 on each: the command processor
 * has four compute pipes, and launch sequences beyond four take turns (4 plans x 2 ranges: 1900 fits/s against 2372 for 4 x 1 at
 * 1250 spectra, profiles/r06_inflight_ranges.txt).  Plans with prepared matrices, a recorded history, weight factors or outlier_p
 * fit in one range.
 * Per-spectrum results do not depend on k (every kernel of the loop works per spectrum).                                    */
int hipdrt_plan_set_subbatches(hipdrt_plan* plan, int k);
/* device bytes ONE staged spectrum costs an EIS plan of this shape (nf frequencies, ntau basis points, ns special parameters):
 * 4.9 MB at 256 x 512 -- the factor and P in tile order are the bulk.  A map driver sizes its batches with it
 * (mapping.fit_observations cuts a map that would not fit 80 % of the device's memory into consecutive batches).             */
int hipdrt_plan_bytes_per_spectrum(int nf, int ntau, int ns, long long* bytes);
/* results for the B staged spectra (any pointer may be NULL):
 * x[B][n] QP solution in scaled units, fit_x[B][ntau] / r_inf[B] / induc[B] rescaled like
 * extract_qphb_parameters (drt1d.py:6228-6289), weights[B][m] (1/sigma, scaled units),
 * coef_scale[B], rho[B][3], s_vectors[B][3][n], q_vector[B][n], outer_iters[B], qp_iters_total[B],
 * status[B] (0 converged, 1 max_iter reached, <0 failed)                                              */
int hipdrt_plan_download(hipdrt_plan* plan, double* x, double* fit_x, double* r_inf, double* induc,
                         double* weights, double* coef_scale, double* rho, double* s_vectors,
                         double* q_vector, int* outer_iters, int* qp_iters_total, int* status);
/* final P (calculate_pq) of spectrum b: p[n][n].  After hipdrt_plan_fit: the matrix calculate_pq builds from the fit's final
 * state (true_weights x chrono / eis row factors, drt1d.py:990-1006).  After hipdrt_plan_continue (upstream's
 * _continue_from_init returns its history only and leaves fit_parameters alone): the same construction on the state the
 * restart ended with -- its last s / rho and the last re-estimated weights times the factors the restart's QPs saw (row
 * factors x weight_factor), q_vector of hipdrt_plan_download to match.  The posterior entry points below use this P.     */
int hipdrt_plan_get_p_matrix(hipdrt_plan* plan, int b, double* p);
/* Posterior variance of the fitted distribution on an evaluation grid: the diagonal of
 * DRT.estimate_distribution_cov (hybdrt/models/drt1d.py:3063-3151 with estimate_param_cov, 4116-4138; order 0, no
 * normalisation), i.e. what DRTMD.fit_observation stores as obs_drt_var (hybdrt/mapping/drtmd.py:278-279) before its
 * extend_var post-processing.  basis_eval[neval][ntau] = basis.construct_func_eval_matrix(ln basis_tau, ln tau_eval).
 * For every fitted spectrum: final P (calculate_pq state), P = L L' on the device, out[b][i] = |L^-1 b_i|^2 * cs_b^2.
 * status[b] (may be NULL): 0 ok, -1 P not positive definite (the reference's LinAlgError -> None).  n <= 4096. */
int hipdrt_plan_distribution_var(hipdrt_plan* plan, const double* basis_eval, int neval, double* out, int* status);
/* same machinery with the identity as evaluation rows: out[b][i] = diag(inv(P_b))_i * cs_b^2, the parameter variances
 * np.diag(DRT.estimate_param_cov()) (hybdrt/models/drt1d.py:4116-4138) of every fitted spectrum.  n <= 4096.          */
int hipdrt_plan_param_var(hipdrt_plan* plan, double* out, int* status);
/* The FULL matrices, for one fitted spectrum b: out[n][n] = inv(P_b) * cs_b^2 = DRT.estimate_param_cov() (drt1d.py:4116-4138;
 * the DOP rescaling of 4126-4130 is the caller's: it needs dop_scale_vector) -- what DRTMD's PFRT post-processing takes per
 * step (hybdrt/mapping/drtmd.py:934-941) -- and out[neval][neval] = basis_eval inv(P_b)[DRT block] basis_eval' * cs_b^2 =
 * DRT.estimate_distribution_cov (drt1d.py:3063-3151; order 0, no normalisation, before extend_var).  Same factorisation as
 * the variances: the evaluation rows ride along as extra panel rows and come out as rows L^-T, one more product gives
 * rows L^-T L^-1 rows'.  *status (may be NULL): 0 ok, -1 P not positive definite (out is then NaN).  n <= 4096.            */
int hipdrt_plan_param_cov(hipdrt_plan* plan, int b, double* out, int* status);
int hipdrt_plan_distribution_cov(hipdrt_plan* plan, int b, const double* basis_eval, int neval, double* out, int* status);

/* per-outer-iteration history of spectrum b recorded when record_history was enabled before the fit:
 * hist_x[iters][n], hist_rho[iters][3], hist_w[iters][m], qp_iters[iters+1]                            */
int hipdrt_plan_record_history(hipdrt_plan* plan, int b_or_minus1);
int hipdrt_plan_get_history(hipdrt_plan* plan, double* hist_x, double* hist_rho, double* hist_w,
                            int* qp_iters, int max_rows, int* rows);
/* Ingredients of DRT.evaluate_llh(weights=qphb.estimate_weights(x, rv, vmm, rm), x=x) (hybdrt/models/drt1d.py:4457-4496,
 * qphb.py:1347-1377) for the current x of every spectrum, as the PFRT driver evaluates it after each step
 * (drt1d.py:2618-2622): out rss[B] = weighted residual sum of squares, sum_log_w[B] = sum(log(weights)).
 * llh = a0 ln b0 - an ln(b0 + rss/2) + lgamma(an) - lgamma(a0) + sum_log_w with an = a0 - 1 + m/2.                 */
int hipdrt_plan_llh_terms(hipdrt_plan* plan, double* rss, double* sum_log_w);
/* The same two sums with the fit's own est_weights (qphb_params['est_weights']) as the weights: DRT.evaluate_rss() and
 * DRT.evaluate_llh() with their default arguments (hybdrt/models/drt1d.py:4433-4496), which DRTMD.fit_observation
 * records for every observation as obs_rss / obs_llh (hybdrt/mapping/drtmd.py:259-260).                              */
int hipdrt_plan_obs_llh_terms(hipdrt_plan* plan, double* rss, double* sum_log_w);
/* The same two sums for the other forms of the `weights` argument of DRT.evaluate_rss / evaluate_llh
 * (hybdrt/models/drt1d.py:4434-4443, 4459-4472): HIPDRT_LLH_W_EST = None (the fit's est_weights, as above),
 * HIPDRT_LLH_W_UNIFORM = "uniform" (within each domain -- chrono rows, impedance rows -- every weight is the mean of that
 * domain's est_weights: DRTMD's default llh_kw / rss_kw, hybdrt/mapping/drtmd.py:129-131), HIPDRT_LLH_W_SCALAR = one scalar
 * weight for every row.  `normalize=True` (division by the number of rows) is left to the caller.                     */
#define HIPDRT_LLH_W_EST 1
#define HIPDRT_LLH_W_UNIFORM 2
#define HIPDRT_LLH_W_SCALAR 3
int hipdrt_plan_obs_llh_terms_w(hipdrt_plan* plan, int weights_mode, double scalar_weight, double* rss, double* sum_log_w);

/* Overwrite parts of the fitted batch's state on the device (NULL = keep): x[B][n] (also becomes the previous iterate),
 * rho[B][3], s[B][3][n], weights[B][m].  The inputs of drt1d._continue_from_init (x_init, rho_vector, s_vectors, weights). */
int hipdrt_plan_set_state(hipdrt_plan* plan, const double* x, const double* rho, const double* s, const double* weights);
/* ... and dop_rho[B][3] of a prepared plan with a distribution of phasances (dop_rho_vector of _continue_from_init) */
int hipdrt_plan_set_state_dop(hipdrt_plan* plan, const double* dop_rho);

/* drt1d._continue_from_init (hybdrt/models/drt1d.py:1270-1365), any data type: re-enter the outer loop from the state on the
 * device with updated hyper-parameters (opts: s_0, l2_lambda_0, ..., xtol, max_iter), at least min_iter iterations;
 * every iteration first multiplies the weights by weight_factor -- and, on prepared plans, by the plan's row factors (the
 * chrono / eis weight factors, 1314-1316: hipdrt_plan_set_weight_factors(plan, 1, rows, batched) beforehand).  est_weights,
 * xmx / dop_xmx norms and the data scale are kept.  Joint fits: the vz_offset column of every measurement's matrix is
 * rewritten after each iteration as in the fit, but from a copy whose offset column is frozen as this call found it
 * (1295-1298, 1353-1357) -- the reference's behaviour, so a chain of warm restarts reproduces pfrt_fit_hybrid (2558-2715).
 * Results through hipdrt_plan_download / _get_history as after hipdrt_plan_fit (outer_iters = iterations of this call).
 * Used by the candidate generators (drt1d.py:1497-1632) and PFRT (2558-2715, DRTMD fit_type='pfrt': drtmd.py:98-100, 1338).
 * opts->outlier_p > 0 (any plan; it may differ from the fit's): estimate_weights forms outlier_t and T V T anew from every
 * iterate (qphb.py:1545-1594) -- the outlier_t / outlier_tvt a caller hands _continue_from_init are never read (1300-1304).  */
int hipdrt_plan_continue(hipdrt_plan* plan, const hipdrt_fit_opts* opts, double weight_factor, int min_iter);

/* ---- prepared-matrix plans: the same device loop for any data type (config-5 family) -----------------------------
 * DRT._qphb_fit_core (hybdrt/models/drt1d.py:551-1006) is data-type agnostic once _prep_for_fit / _format_qp_matrices
 * (5439-5558, 5736-5963) have produced the stacked matrix rzm, the data vector rzv, the padded penalty matrices and the
 * stacked variance-estimation matrix.  A prepared plan takes exactly those (built by the caller from the stand-alone
 * builders above) and runs initialize_weights + the iterate_qphb loop on the device, including
 *   - the x_dop hyper-parameter pass (hybdrt/models/qphb.py:822-933) and the DOP block of the L2 matrix (qphb.py:92-100),
 *   - the per-iteration rewrite of the vz_offset column of a hybrid fit (drt1d.py:973-979).
 * Weight factors are 1 (hybrid_weight_factor_method=None, drt1d.py:785-787).                                          */
typedef struct {
    int m, n, ns;              /* rows of rzm, unknowns, special (non-DRT) unknowns ahead of the DRT block            */
    int dop_start, dop_size;   /* the x_dop block inside the specials (dop_size 0: none)                              */
    int vz_index;              /* column of vz_offset (-1: none)                                                      */
    int vb_start, vb_size;     /* v_baseline columns: excluded from the vz prediction (drt1d.py:507-511)              */
    int num_chrono;            /* rows [0, num_chrono) are chrono samples, the rest [Re; Im] impedance rows           */
    int toeplitz_m;            /* DRT block of the penalty matrices is symmetric Toeplitz (uniform ln tau)            */
    int chrono_vmm_uniform;    /* chrono block of vmm is the 'uniform' error structure (all rows equal): one row is read */
    double basis_area;         /* area of one tau basis function, sqrt(pi)/epsilon for the Gaussian basis (update_scale) */
    int init_weights_separately; /* 1: initialize_weights once per data block (chrono rows, impedance rows), each QP seeing
                                  only its block and each block with its own variance floor (drt1d.py:648-672)         */
    int weight_method;         /* 1: hybrid_weight_factor_method='weight' -- row factors from the blocks' weight scales
                                  after initialize_weights (drt1d.py:748-760); fixed_*_factor > 0 overrides one of them;
                                  the factors used are returned by hipdrt_plan_get("weight_factors") [B][2] = chrono, eis */
    double fixed_chrono_factor, fixed_eis_factor;
    double dop_l2_lambda_0;                                       /* qphb.py:243-253 */
    double dop_derivative_weights[3], dop_s_alpha[3], dop_rho_alpha[3], dop_s_0[3], dop_rho_0[3];
} hipdrt_prepared_desc;
/* m0,m1,m2 [n][n] padded penalty matrices; vmm [m][m]; h [n] (make_h_constraint, qphb.py:521-557); l1 [n]
 * (l1_lambda_vector, drt1d.py:552-556); vz_strength [m] (drt1d.py:514-519; NULL when vz_index < 0).
 * opts: the fit options (scale_data / rp_scale / fit_ohmic / fit_inductance / eis_* are ignored: the caller prepared
 * the data).                                                                                                         */
int hipdrt_plan_create_prepared(hipdrt_ctx* ctx, const hipdrt_prepared_desc* desc, const double* m0, const double* m1,
                                const double* m2, const double* vmm, const double* h, const double* l1,
                                const double* vz_strength, const hipdrt_fit_opts* opts, int capacity,
                                hipdrt_plan** out);
/* stage B measurements: rzm [B][m][n] (rm_batched != 0) or one shared [m][n]; rzv [B][m].  A vz_offset column needs
 * per-measurement matrices.  Then hipdrt_plan_fit; results through hipdrt_plan_download (x, weights, rho, s_vectors,
 * q_vector, iteration counts, status; pass NULL for fit_x / r_inf / induc), hipdrt_plan_get_p_matrix and
 * hipdrt_plan_get ("dop_rho" [B][3], "xmx", "dop_xmx" [B][3], "est_weights", "rzm" [B or 1][m][n] = final matrix;
 * with outlier_p set also "outlier_t" [B][m], qphb.py:1497-1520, of the last weight estimation -- with max_iter = 0 that
 * is initialize_weights', which is what remove_outliers thresholds, drt1d.py:817-833).                                  */
int hipdrt_plan_upload_prepared(hipdrt_plan* plan, int B, int rm_batched, const double* rzm, const double* rzv);

/* One qphb.iterate_qphb (hybdrt/models/qphb.py:606-972) on every staged measurement of a prepared plan -- the body of
 * _qphb_fit_core's loop as the reference exposes it: solve the QP built from (weights, s_vectors, rho_vector, dop_rho_vector),
 * update s_vectors / rho_vector (733-817) and the DOP pair (822-933) from the new x with the given xmx norms, re-estimate the
 * weights against est_weights (936), and test is_converged(x_in, x) (967-970).  A NULL member keeps the value on the device
 * (defaults after hipdrt_plan_upload_prepared: x 1e-6, s = s_0, rho = rho_0, weights = est_weights... = 1, norms 1; after
 * hipdrt_plan_fit: the fit's), so repeated calls with in = NULL continue from their own results.  Not included: what
 * _qphb_fit_core does around the call (xmx norms of the first iteration, update_scale, the vz_offset column, weight factors).
 * Results through hipdrt_plan_download (x, weights, rho, s_vectors) and hipdrt_plan_get("dop_rho"); the optional
 * outputs [B]: is_converged, and of the QP its status (HIPDRT_QP_*), iteration count and 'primal objective'. */
typedef struct {
    const double* x_in;           /* [B][n]    */
    const double* s_vectors;      /* [B][3][n] */
    const double* rho;            /* [B][3]    */
    const double* dop_rho;        /* [B][3]    */
    const double* weights;        /* [B][m]    */
    const double* est_weights;    /* [B][m]    */
    const double* xmx_norms;      /* [B][3]    */
    const double* dop_xmx_norms;  /* [B][3]    */
} hipdrt_iterate_state;
int hipdrt_plan_iterate(hipdrt_plan* plan, const hipdrt_iterate_state* in, int* converged, int* qp_status, int* qp_iters,
                        double* primal_objective);

/* Weight factors of _qphb_fit_core (hybdrt/models/drt1d.py:887-901, 990-1000): every outer iteration solves its QP with
 * weights * row_factors (chrono_weight_factor on the chrono rows, eis_weight_factor on the impedance rows; NULL = 1),
 * from the second iteration on also * weight_factor; estimate_weights keeps working on the unscaled weights.  After the
 * fit hipdrt_plan_download returns weights * weight_factor ("true_weights"); q_vector / p_matrix / the posterior
 * variances use those times the row factors ("scaled weights").  row_factors: [m], or [capacity][m] when batched != 0.
 * Applies to every later hipdrt_plan_fit of the plan (any plan kind); (1.0, NULL) switches it off.  batched bit 1 (value
 * 2): the row factors are a vector-valued weight_factor (kk_fit, drt1d.py:1393-1411) -- applied from the second iteration
 * on only and folded into the returned weights.                                                                        */
int hipdrt_plan_set_weight_factors(hipdrt_plan* plan, double weight_factor, const double* row_factors, int batched);
/* Constraint vector of the initialize_weights QP when it differs from the loop's (neg_allowed_tau_range: the loop allows
 * negative coefficients only inside a tau window, initialize_weights everywhere; hybdrt/models/drt1d.py:467, 657-660 vs
 * 944, hybdrt/models/qphb.py:521-557).  h_init[n]; NULL restores the plan's single h.                                   */
int hipdrt_plan_set_init_h(hipdrt_plan* plan, const double* h_init);

/* kernel-time breakdown of the last hipdrt_plan_fit in ms (HIP events on the ctx stream):
 * t[0]=total, t[1]=gram, t[2]=qp, t[3]=hyper, t[4]=setup/other; launches[5] same order                 */
int hipdrt_plan_timings(hipdrt_plan* plan, float* t, int* launches);

/* one-shot convenience: create plan, upload, fit, download, destroy */
int hipdrt_fit_eis_batch(hipdrt_ctx* ctx, int B, const double* freq, int nf, const double* z_re,
                         const double* z_im, const double* tau, int ntau, double epsilon, int mode,
                         int toeplitz_a, int toeplitz_m, int ngrid, int ny, const double* wt_re,
                         const double* wt_im, const double* log_wt_re, const double* log_wt_im,
                         const hipdrt_fit_opts* opts, double* x, double* fit_x,
                         double* r_inf, double* induc, double* weights, double* coef_scale, double* rho,
                         double* q_vector, int* outer_iters, int* status);

/* ---- one map over the GPUs of a node: RCCL behind the C ABI -------------------------------------------------------------
 * The reference fits the observations of a map one after the other in ONE process (DRTMD.fit_observations,
 * hybdrt/mapping/drtmd.py:303-319); they share only the lookup tables, the tau supergrid and cached matrices (245-301).  Here a
 * map is sharded over one process per GPU; the two exchanges that needs are a broadcast of rank 0's lookup tables and ONE gather
 * of the per-observation result rows.  librccl.so is loaded on first use.  Rendezvous: rank 0 calls hipdrt_comm_unique_id and
 * gets its 128 bytes to the other ranks by any out-of-band channel (hybrid-drt_amd/mapping/dist.py: a file next to
 * MASTER_PORT); every rank then calls hipdrt_comm_create with the same bytes (collective: returns when all `world` ranks did). */
#define HIPDRT_COMM_ID_BYTES 128
int hipdrt_comm_unique_id(char* id128);
int hipdrt_comm_create(int device, int rank, int world, const char* id128, hipdrt_comm** out);
int hipdrt_comm_destroy(hipdrt_comm* comm);
int hipdrt_comm_info(hipdrt_comm* comm, int* rank, int* world, int* device);
/* device buffers in, device buffers out.  broadcast: `count` doubles of `root` to everyone, in place.  gather: every rank
 * sends `count` doubles, `root` receives world x count (rank r's block at r * count; dev_recv may be NULL elsewhere) -- a
 * true gather, one ncclSend per rank and `world` ncclRecv on the root inside one group.  Both return when the data is there. */
int hipdrt_comm_broadcast_dev(hipdrt_comm* comm, double* dev_buf, long long count, int root);
int hipdrt_comm_gather_dev(hipdrt_comm* comm, const double* dev_send, long long count, double* dev_recv, int root);
/* the same for host arrays (numpy in, numpy out), staged through the communicator's own device buffers */
int hipdrt_comm_broadcast(hipdrt_comm* comm, double* host_buf, long long count, int root);
int hipdrt_comm_gather(hipdrt_comm* comm, const double* host_send, long long count, double* host_recv, int root);
/* max over the ranks of *value, in every rank (a benchmark's "slowest rank" time); value == NULL: a plain barrier */
int hipdrt_comm_allreduce_max(hipdrt_comm* comm, double* value);
int hipdrt_comm_barrier(hipdrt_comm* comm);
/* device memory for callers that keep matrices resident (hipdrt_impedance_matrix_dev and the *_dev collectives take it) */
int hipdrt_device_alloc(hipdrt_ctx* ctx, long long bytes, void** out);
int hipdrt_device_free(hipdrt_ctx* ctx, void* ptr);
/* hipDeviceSynchronize on the context's device: every stream drained (a benchmark brackets its timed region with it)       */
int hipdrt_device_synchronize(hipdrt_ctx* ctx);
/* 0 when device `device` exists and is a gfx950 part (HIPDRT_E_NODEVICE otherwise); creates nothing on the device                */
int hipdrt_device_probe(int device);

#ifdef __cplusplus
}
#endif
#endif /* HIPDRT_H */

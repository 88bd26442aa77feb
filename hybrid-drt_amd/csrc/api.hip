// C-ABI of libhipdrt.so (include/hipdrt.h): context, stand-alone operators, and the plan that runs
// DRT._qphb_fit_core (hybdrt/models/drt1d.py:102-1104, EIS branch) for a batch of spectra on the device.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>

#include "common.hpp"
#include <thread>
#include <vector>

namespace hipdrt {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace hipdrt

using namespace hipdrt;

// ---------------------------------------------------------------------------------------------------------
struct hipdrt_plan {
    hipdrt_ctx* ctx = nullptr;
    int nf = 0, ntau = 0, n = 0, m = 0, ns = 0, ngrid = 0, ny = 0, mode = 0, toeplitz_a = 0, toeplitz_m = 0;
    int idx_rinf = -1, idx_induc = -1;
    int ldrm = 0, ldm = 0, ldp = 0, ldl = 0;
    int capacity = 0, B = 0;
    double eps = 0;
    hipdrt_fit_opts opts{};
    // prepared-matrix plans (hipdrt_plan_create_prepared)
    int prepared = 0;
    int prepped = 0;           // launch_prep has run on the staged batch (hipdrt_plan_iterate runs it once)
    hipdrt_prepared_desc desc{};
    long long rm_stride = 0;
    DevBuf vz_strength, dop_rho, dop_xmx, hist_dop_rho, outlier_t, vz_entry;
    // weight factors (hipdrt_plan_set_weight_factors): w_eff = w * row factor * weight_factor is what the QP sees
    double weight_factor = 1.0;
    int wrow_batched = 0;
    int wrow_late = 0;      // row factors are a vector-valued weight_factor: applied from the second iteration on only
    DevBuf wrow, w_eff, h_init, wfac;
    bool has_weight_factors() const { return weight_factor != 1.0 || wrow.p != nullptr; }
    // shared
    DevBuf freq, tau, ln_tau, wt_re, wt_im, lut6, a_re, a_im, cr, rm, mk[3], vmm, h, l1;
    // per spectrum
    DevBuf z_re, z_im, rv, w, est_w, x, x_in, q, s, rho, xmx, coef_scale, var_floor;
    DevBuf active, outer_iters, fit_status, qp_iters_total, qp_status, qp_iters, n_active, pcost;
    DevBuf premv;          // [3][capacity][m]: hyper-parameter step of few, large fits (hyper.hip, premv_kernel)
    DevBuf L, Ptmp, qpstate, Ppk, order, vmm_base, gsync;
    int toep_maxd = -1;     // reach of the Toeplitz penalty blocks in grid points (plan_toep_reach), -1 = not determined
    int spec_zero = 0;      // the special-parameter rows / columns of the penalty matrices are zero outside the special block
    int qp_G = 0;           // workgroups per QP when the plan is full (qp_group_size at its capacity): 0 = the batch kernel
    // The kernel is chosen per fit from the number of spectra actually staged: a plan sized for a thousand spectra that is
    // handed one or a handful runs them on several workgroups each, inside the scratch it already has.
    void qp_layout(int B, QpArgs& qa) const {
        int G = qp_group_size(B, n, ctx ? ctx->qp_force_group : -1);
        const size_t have_l = L.bytes / sizeof(double), have_s = qpstate.bytes / sizeof(double);
        if (G >= 1 && G != qp_G) {
            const bool fits = gsync.p && (size_t)B * qp_scratch_doubles(n, G) <= have_l &&
                              (size_t)B * G * qp_state_doubles(n) <= have_s && (size_t)B * qp_gsync_ints() * sizeof(int) <= gsync.bytes;
            if (!fits) G = qp_G;
        } else if (G < 1) {
            G = qp_G;            // (a plan created for few spectra keeps its group layout when it is full)
        }
        qa.G = G; qa.gsync = gsync.i(); qa.l_stride = (long long)qp_scratch_doubles(n, G);
        qa.waves = ctx ? ctx->qp_waves : -1;
    }
    // history
    int hist_b = -1, hist_cap = 0;
    DevBuf hist_x, hist_w, hist_rho, hist_qp, hist_rows;
    // timings of the last fit
    float t_ms[5] = {0, 0, 0, 0, 0};
    int launches[5] = {0, 0, 0, 0, 0};
    // sub-batches of one fit (hipdrt_plan_set_subbatches): the staged spectra split into `k` contiguous ranges, every range
    // fitted by the same device loop on its own stream, all inside ONE hipdrt_plan_fit call and the plan's own buffers
    int subbatches = 0;                                   // 0 = automatic (subbatch_count), >= 1 fixed
    std::vector<std::unique_ptr<struct hipdrt_subfit>> subs;
    DevBuf n_active_sub;                                  // one "still active" counter per sub-batch
    hipdrt_plan() = default;
    ~hipdrt_plan();

    FitState state() const {
        FitState st{};
        st.nf = nf; st.m = m; st.n = n; st.ns = ns; st.ldrm = ldrm; st.ldm = ldm; st.toeplitz_m = toeplitz_m;
        st.toep_reach = (toeplitz_m && !(ctx && !ctx->zero_shortcuts)) ? toep_maxd : -1;
        st.opts = opts; st.continue_mode = 0; st.min_iter = 1;
        st.basis_area = prepared ? desc.basis_area : (eps > 0 ? 1.7724538509055159 / eps : 0.0);   // sqrt(pi) / epsilon
        st.prepared = prepared; st.desc = desc; st.rm_stride = rm_stride; st.rm_rw = rm.d();
        st.vz_strength = vz_strength.d(); st.vz_entry = nullptr; st.dop_rho = dop_rho.d(); st.dop_xmx = dop_xmx.d();
        st.hist_dop_rho = hist_dop_rho.d(); st.outlier_t = outlier_t.d();
        st.rm = rm.d(); st.vmm = vmm.d(); st.vmm_iw = vmm_base.p ? vmm_base.d() : vmm.d();
        for (int k = 0; k < 3; ++k) st.mk[k] = mk[k].d();
        st.z_re = z_re.d(); st.z_im = z_im.d();
        st.rv = rv.d(); st.w = w.d(); st.est_w = est_w.d();
        st.x = x.d(); st.x_in = x_in.d(); st.s = s.d(); st.rho = rho.d(); st.xmx = xmx.d();
        st.coef_scale = coef_scale.d(); st.var_floor = var_floor.d();
        st.active = active.i(); st.outer_iters = outer_iters.i(); st.fit_status = fit_status.i();
        st.qp_iters_total = qp_iters_total.i(); st.qp_status = qp_status.i(); st.qp_iters = qp_iters.i();
        st.n_active = n_active.i();
        st.hist_b = hist_b; st.hist_cap = hist_cap;
        st.hist_x = hist_x.d(); st.hist_w = hist_w.d(); st.hist_rho = hist_rho.d();
        st.hist_qp = hist_qp.i(); st.hist_rows = hist_rows.i();
        st.premv = nullptr; st.premv_batched = 0;
        return st;
    }
};

// one sub-batch of a plan: a plan object whose buffers are windows into the parent's, with a stream of its own
struct hipdrt_subfit {
    hipdrt_ctx ctx;
    hipdrt_plan view;
    int rc = 0;
    std::string err;
    // (ctx.stream is borrowed from the library's pool for the duration of one fit: hipdrt_plan_fit)
};
hipdrt_plan::~hipdrt_plan() = default;

static int upload(DevBuf& buf, const void* src, size_t bytes, hipStream_t st) {
    HIPDRT_CHECK(buf.alloc(bytes));
    if (src) HIPDRT_CHECK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, st));
    return 0;
}
#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define LAUNCH_OK() HIPDRT_CHECK(hipGetLastError())

// No exception may cross the C ABI: every entry point below is a function-try-block.  (Host-side std::vector buffers -- an
// n x n identity of 134 MB at n = 4096, download staging, the Toeplitz reach scan of plan creation -- can throw std::bad_alloc.)
#define HIPDRT_CATCH                                                                                              \
    catch (const std::bad_alloc&) { hipdrt::set_error("out of host memory"); return HIPDRT_E_HIP; }                \
    catch (const std::exception& e) { hipdrt::set_error(std::string("internal error: ") + e.what()); return HIPDRT_E_HIP; } \
    catch (...) { hipdrt::set_error("internal error"); return HIPDRT_E_HIP; }

extern "C" {

static int plan_hist_reserve(hipdrt_plan* p, int rows);
static int plan_toep_reach(hipdrt_plan* p);

const char* hipdrt_last_error(void) { return g_err.c_str(); }

// what the environment promises about the runtime's hardware-queue count (read when the runtime started; 4 when unset)
static int hw_queues_hint() {
    static const int q = [] { const char* e = std::getenv("GPU_MAX_HW_QUEUES"); const int v = e ? std::atoi(e) : 4; return v > 0 ? std::min(v, 32) : 4; }();
    return q;
}

// ---- the library's own streams ------------------------------------------------------------------------------------------
// The HIP runtime maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (4 unless the environment says otherwise), and two
// launch sequences on ONE queue run one behind the other.  Its rule, read off rocprofv3's queue ids (tools/queue_map_probe.sh,
// profiles/r06_queue_map.txt): queue 1 belongs to the null stream; a new stream gets a NEW queue while fewer than the maximum
// exist, afterwards the LAST queue among those with the fewest streams -- counting idle streams like busy ones.  So the stream that
// fills the pool and the one created right after it share a queue, and any two streams created one pool's length apart do:
//   * four ranges of one plan behind three or five other (idle!) streams land on three queues, two of them back to back:
//     1778 fits/s instead of 2304 (tools/trace_queue_placement.sh, profiles/r06_trace_queue_placement.txt);
//   * two plans in flight whose contexts were created seven streams apart share queue 8: 1730 instead of 2266
//     (profiles/r06_trace_plans_placement.txt);
//   * the "wrapped" placements of profiles/r06_hw_queue_placement.txt (-4 %) and round 5's "three and four ranges flip between
//     fast and slow".
// The library therefore creates its streams ONCE per device -- as many as queues are left beside the null stream's, each on a
// queue of its own when nothing else has created streams before -- and hands them out itself, by ACTIVITY: a context holds one
// for its lifetime (the one with the fewest holders), the ranges of a fit borrow the ones with the fewest device loops running
// for the duration of that fit.  Idle contexts and idle plans no longer push active ones onto shared queues.
// HIPDRT_STREAM_POOL=<n> sets the count (1 ... 32).
extern "C++" {
namespace {
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> st;
    std::vector<int> holders, running;
};

StreamPool* stream_pool(int device) {
    static std::mutex mu;
    static std::vector<std::pair<int, StreamPool*>> pools;         // (never freed: the streams live as long as the process)
    std::lock_guard<std::mutex> lock(mu);
    for (auto& pr : pools) if (pr.first == device) return pr.second;
    int n = std::max(2, hw_queues_hint() - 1);
    if (const char* e = std::getenv("HIPDRT_STREAM_POOL")) { const int v = std::atoi(e); if (v >= 1 && v <= 32) n = v; }
    auto* pl = new StreamPool();
    int before = 0;
    (void)hipGetDevice(&before);
    (void)hipSetDevice(device);
    for (int i = 0; i < n; ++i) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        pl->st.push_back(st);
    }
    (void)hipSetDevice(before);
    pl->holders.assign(pl->st.size(), 0);
    pl->running.assign(pl->st.size(), 0);
    pools.emplace_back(device, pl);
    return pl;
}

// Streams whose queues sit on the same compute pipe of the command processor take turns at dispatching: queues are dealt to
// the four pipes in creation order, so pool streams i and i + 4 are such a pair, and four ranges on queues of pipes 1, 2, 3, 1
// measure 2205 fits/s where pipes 0 ... 3 measure 2300 (profiles/r06_trace_queue_placement.txt, second half: every placement the
// trace shows as "four queues" but slow has two ranges one pipe apart).
constexpr int kPipes = 4;

// the stream for one more device loop: none running on it, the fewest loops running on its pipe, the fewest holders, the lowest index
int pool_pick(const StreamPool& pl) {
    int on_pipe[kPipes] = {0, 0, 0, 0};
    for (int i = 0; i < (int)pl.st.size(); ++i) on_pipe[i % kPipes] += pl.running[i];
    auto better = [&](int a, int b) {
        if (pl.running[a] != pl.running[b]) return pl.running[a] < pl.running[b];
        if (on_pipe[a % kPipes] != on_pipe[b % kPipes]) return on_pipe[a % kPipes] < on_pipe[b % kPipes];
        return pl.holders[a] < pl.holders[b];
    };
    int best = 0;
    for (int i = 1; i < (int)pl.st.size(); ++i) if (better(i, best)) best = i;
    return best;
}

// a context's stream for its lifetime
bool pool_hold(hipdrt_ctx* c) {
    StreamPool* pl = stream_pool(c->device);
    if (pl->st.empty()) return false;
    std::lock_guard<std::mutex> lock(pl->mu);
    int best = 0;
    for (int i = 1; i < (int)pl->st.size(); ++i) if (pl->holders[i] < pl->holders[best]) best = i;
    ++pl->holders[best];
    c->stream = pl->st[best];
    c->pool_idx = best;
    return true;
}
void pool_drop(hipdrt_ctx* c) {
    if (c->pool_idx < 0) return;
    StreamPool* pl = stream_pool(c->device);
    std::lock_guard<std::mutex> lock(pl->mu);
    --pl->holders[c->pool_idx];
    c->pool_idx = -1; c->stream = nullptr;
}
// a device loop starts / ends on stream `idx` (a fit on the context's own stream)
void pool_running(int device, int idx, int delta) {
    if (idx < 0) return;
    StreamPool* pl = stream_pool(device);
    std::lock_guard<std::mutex> lock(pl->mu);
    pl->running[idx] += delta;
}
// k streams for the ranges of one fit, the least busy first (more ranges than streams: they repeat)
void pool_borrow(int device, int k, int* idx, hipStream_t* st) {
    StreamPool* pl = stream_pool(device);
    std::lock_guard<std::mutex> lock(pl->mu);
    for (int i = 0; i < k; ++i) {
        idx[i] = pool_pick(*pl);
        ++pl->running[idx[i]];
        st[i] = pl->st[idx[i]];
    }
}
int pool_size(int device) { return (int)stream_pool(device)->st.size(); }
void pool_return(int device, int k, const int* idx) {
    StreamPool* pl = stream_pool(device);
    std::lock_guard<std::mutex> lock(pl->mu);
    for (int i = 0; i < k; ++i) --pl->running[idx[i]];
}
struct LoopOnContextStream {       // RAII: "a device loop runs on this context's stream" for the ranges of other fits to avoid
    hipdrt_ctx* c;
    explicit LoopOnContextStream(hipdrt_ctx* c_) : c(c_) { pool_running(c->device, c->pool_idx, +1); }
    ~LoopOnContextStream() { pool_running(c->device, c->pool_idx, -1); }
};
}  // namespace
}

extern "C++" {
namespace hipdrt {
// (comm.hip: a communicator made before the first context must not take one of the queues the pool would get)
void ensure_stream_pool(int device) { (void)stream_pool(device); }
}
}

int hipdrt_create(int device, hipdrt_ctx** out) try {
    HIPDRT_REQUIRE(out != nullptr, "out is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device visible (libhipdrt has no CPU fallback)");
        return HIPDRT_E_NODEVICE;
    }
    HIPDRT_REQUIRE(device >= 0 && device < count, "device index out of range");
    HIPDRT_CHECK(hipSetDevice(device)); (void)hipGetLastError();
    hipDeviceProp_t prop;
    HIPDRT_CHECK(hipGetDeviceProperties(&prop, device));
    std::string arch = prop.gcnArchName;
    if (arch.rfind("gfx950", 0) != 0) {
        set_error("device " + std::to_string(device) + " is " + arch + "; libhipdrt is built for gfx950 only");
        return HIPDRT_E_NODEVICE;
    }
    auto* c = new hipdrt_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->hbm_bytes = prop.totalGlobalMem;
    c->arch = arch.substr(0, arch.find(':'));
    if (const char* wv = std::getenv("HIPDRT_QP_WAVES")) {        // (tools/: A/B of the two batch coneqp kernels without a code change)
        const int w = std::atoi(wv);
        if (w == 4 || w == 8) c->qp_waves = w;
    }
    if (!pool_hold(c)) { delete c; set_error("no HIP stream could be created on device " + std::to_string(device)); return HIPDRT_E_HIP; }
    *out = c;
    return HIPDRT_OK;
} HIPDRT_CATCH

static std::mutex g_life;          // context / plan creation and destruction (any thread, e.g. a garbage collector's)

static void free_ctx(hipdrt_ctx* ctx) {
    pool_drop(ctx);                    // (the stream itself belongs to the library's pool and lives on)
    delete ctx;
}

int hipdrt_destroy(hipdrt_ctx* ctx) try {
    if (!ctx) return HIPDRT_OK;
    std::lock_guard<std::mutex> lk(g_life);
    if (ctx->plans > 0) { ctx->released = true; return HIPDRT_OK; }     // the last plan's destruction frees it
    free_ctx(ctx);
    return HIPDRT_OK;
} HIPDRT_CATCH

void* hipdrt_stream(hipdrt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int hipdrt_synchronize(hipdrt_ctx* ctx) try {
    HIPDRT_REQUIRE(ctx, "ctx is NULL");
    HIPDRT_CHECK(hipStreamSynchronize(ctx->stream));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_device_info(hipdrt_ctx* ctx, char* arch, int arch_len, int* num_cu, long long* hbm_bytes) try {
    HIPDRT_REQUIRE(ctx, "ctx is NULL");
    if (arch && arch_len > 0) { std::strncpy(arch, ctx->arch.c_str(), arch_len - 1); arch[arch_len - 1] = 0; }
    if (num_cu) *num_cu = ctx->num_cu;
    if (hbm_bytes) *hbm_bytes = (long long)ctx->hbm_bytes;
    return HIPDRT_OK;
} HIPDRT_CATCH

// ---- stand-alone operators --------------------------------------------------------------------------------

int hipdrt_impedance_lookup(hipdrt_ctx* ctx, double epsilon, int ngrid, int ny, const double* wt_re,
                            const double* wt_im, double* z_re, double* z_im) try {
    HIPDRT_REQUIRE(ctx && wt_re && wt_im && z_re && z_im, "NULL pointer");
    HIPDRT_REQUIRE(ngrid >= 2 && ny >= 2 && ny <= 6000, "ngrid >= 2, 2 <= ny <= 6000");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dwr, dwi, dzr, dzi;
    const size_t gb = (size_t)ngrid * sizeof(double);
    TRY(upload(dwr, wt_re, gb, st)); TRY(upload(dwi, wt_im, gb, st));
    HIPDRT_CHECK(dzr.alloc(gb)); HIPDRT_CHECK(dzi.alloc(gb));
    launch_lookup(st, epsilon, ngrid, ny, dwr.d(), dwi.d(), dzr.d(), dzi.d());
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(z_re, dzr.p, gb, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(z_im, dzi.p, gb, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_phasor_z_matrix(hipdrt_ctx* ctx, const double* freq, int nf, const double* basis_nu, int n_nu, double nu_epsilon,
                           double* zm_re, double* zm_im) try {
    HIPDRT_REQUIRE(ctx && freq && basis_nu && zm_re && zm_im, "NULL pointer");
    HIPDRT_REQUIRE(nf >= 1 && n_nu >= 1 && nu_epsilon > 0.0, "nf, n_nu >= 1, nu_epsilon > 0");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf df, dn, dr, di;
    TRY(upload(df, freq, (size_t)nf * sizeof(double), st));
    TRY(upload(dn, basis_nu, (size_t)n_nu * sizeof(double), st));
    const size_t ob = (size_t)nf * n_nu * sizeof(double);
    HIPDRT_CHECK(dr.alloc(ob)); HIPDRT_CHECK(di.alloc(ob));
    launch_phasor_z(st, df.d(), nf, dn.d(), n_nu, nu_epsilon, dr.d(), di.d());
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(zm_re, dr.p, ob, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(zm_im, di.p, ob, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_phasor_v_matrix(hipdrt_ctx* ctx, const double* times, int nt, const double* basis_nu, int n_nu, double nu_epsilon,
                           const double* step_times, const double* step_sizes, int nsteps, double* rm, double* layered) try {
    HIPDRT_REQUIRE(ctx && times && basis_nu && step_times && step_sizes && rm, "NULL pointer");
    HIPDRT_REQUIRE(nt >= 1 && n_nu >= 1 && nsteps >= 1 && nu_epsilon > 0.0, "nt, n_nu, nsteps >= 1, nu_epsilon > 0");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dt, dn, ds, da, dr, dl;
    TRY(upload(dt, times, (size_t)nt * sizeof(double), st));
    TRY(upload(dn, basis_nu, (size_t)n_nu * sizeof(double), st));
    TRY(upload(ds, step_times, (size_t)nsteps * sizeof(double), st));
    TRY(upload(da, step_sizes, (size_t)nsteps * sizeof(double), st));
    const size_t ob = (size_t)nt * n_nu * sizeof(double);
    HIPDRT_CHECK(dr.alloc(ob));
    if (layered) HIPDRT_CHECK(dl.alloc(ob * nsteps));
    launch_phasor_v(st, dt.d(), nt, dn.d(), n_nu, nu_epsilon, ds.d(), da.d(), nsteps, dr.d(), layered ? dl.d() : nullptr);
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(rm, dr.p, ob, hipMemcpyDeviceToHost, st));
    if (layered) HIPDRT_CHECK(hipMemcpyAsync(layered, dl.p, ob * nsteps, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_chrono_var_matrix(hipdrt_ctx* ctx, const double* tt, int nt, const int* seg, int nseg, double vmm_epsilon,
                             int uniform, double* vmm) try {
    HIPDRT_REQUIRE(ctx && tt && seg && vmm, "NULL pointer");
    HIPDRT_REQUIRE(nt >= 1 && nseg >= 1, "nt >= 1, nseg >= 1");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dtt, dseg, dv;
    TRY(upload(dtt, tt, (size_t)nt * sizeof(double), st));
    TRY(upload(dseg, seg, (size_t)(nseg + 1) * sizeof(int), st));
    HIPDRT_CHECK(dv.alloc((size_t)nt * nt * sizeof(double)));
    launch_chrono_vmm(st, dtt.d(), nt, dseg.i(), nseg, vmm_epsilon, uniform, dv.d());
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(vmm, dv.p, (size_t)nt * nt * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_response_lookup(hipdrt_ctx* ctx, double epsilon, int ngrid, int ny, const double* td, double* v) try {
    HIPDRT_REQUIRE(ctx && td && v, "NULL pointer");
    HIPDRT_REQUIRE(ngrid >= 2 && ny >= 2 && ny <= 6000, "ngrid >= 2, 2 <= ny <= 6000");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dtd, dv;
    const size_t gb = (size_t)ngrid * sizeof(double);
    TRY(upload(dtd, td, gb, st));
    HIPDRT_CHECK(dv.alloc(gb));
    launch_response_lookup(st, epsilon, ngrid, ny, dtd.d(), dv.d());
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(v, dv.p, gb, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_response_matrix(hipdrt_ctx* ctx, const double* times, int nt, const double* tau, int ntau,
                           const double* step_times, const double* step_sizes, int nsteps, int mode, double epsilon,
                           int ngrid, const double* log_td, const double* v, int ny, double* a, double* layered) try {
    HIPDRT_REQUIRE(ctx && times && tau && step_times && step_sizes && a, "NULL pointer");
    HIPDRT_REQUIRE(nt >= 1 && ntau >= 1 && nsteps >= 1, "nt, ntau, nsteps >= 1");
    HIPDRT_REQUIRE(mode == HIPDRT_MODE_INTERP || mode == HIPDRT_MODE_TRAPZ, "mode must be INTERP or TRAPZ");
    if (mode == HIPDRT_MODE_INTERP) {
        HIPDRT_REQUIRE(log_td && v && ngrid >= 2, "interpolate_grids must be provided for integrate_method 'interp'");
        HIPDRT_REQUIRE(3 * (size_t)ngrid * sizeof(double) <= 150 * 1024, "lookup too long for LDS staging");
    } else {
        HIPDRT_REQUIRE(ny >= 2 && ny <= 6000, "2 <= ny <= 6000");
    }
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dt, dtau, dst, dsa, lut3, da, dl;
    TRY(upload(dt, times, (size_t)nt * sizeof(double), st));
    TRY(upload(dtau, tau, (size_t)ntau * sizeof(double), st));
    TRY(upload(dst, step_times, (size_t)nsteps * sizeof(double), st));
    TRY(upload(dsa, step_sizes, (size_t)nsteps * sizeof(double), st));
    if (mode == HIPDRT_MODE_INTERP) {
        const size_t gb = (size_t)ngrid * sizeof(double);
        HIPDRT_CHECK(lut3.alloc(3 * gb));
        HIPDRT_CHECK(hipMemcpyAsync(lut3.d(), log_td, gb, hipMemcpyHostToDevice, st));
        HIPDRT_CHECK(hipMemcpyAsync(lut3.d() + ngrid, v, gb, hipMemcpyHostToDevice, st));
        launch_lookup_slopes(st, ngrid, lut3.d(), lut3.d() + ngrid, lut3.d() + 2 * (size_t)ngrid);
    }
    const size_t ab = (size_t)nt * ntau * sizeof(double);
    HIPDRT_CHECK(da.alloc(ab));
    if (layered) HIPDRT_CHECK(dl.alloc(ab * nsteps));
    launch_response_matrix(st, dt.d(), nt, dtau.d(), ntau, dst.d(), dsa.d(), nsteps, mode, epsilon, ngrid, lut3.d(), ny,
                           da.d(), layered ? dl.d() : nullptr);
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(a, da.p, ab, hipMemcpyDeviceToHost, st));
    if (layered) HIPDRT_CHECK(hipMemcpyAsync(layered, dl.p, ab * nsteps, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_response_matrix_variant(hipdrt_ctx* ctx, const double* times, int nt, const double* tau, int ntau,
                                   const double* step_times, const double* step_sizes, const double* tau_rise, int nsteps,
                                   int variant, double epsilon, int ny, double* a, double* layered) try {
    HIPDRT_REQUIRE(ctx && times && tau && step_times && step_sizes && a, "NULL pointer");
    HIPDRT_REQUIRE(nt >= 1 && ntau >= 1 && nsteps >= 1, "nt, ntau, nsteps >= 1");
    HIPDRT_REQUIRE(variant == HIPDRT_RESPONSE_POT || variant == HIPDRT_RESPONSE_EXPDECAY, "variant must be POT or EXPDECAY");
    if (variant == HIPDRT_RESPONSE_EXPDECAY) {
        HIPDRT_REQUIRE(tau_rise, "the expdecay step model needs tau_rise");
        HIPDRT_REQUIRE(ny >= 2 && ny <= 6000, "2 <= ny <= 6000");
    }
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dt, dtau, dst, dsa, dtr, da, dl;
    TRY(upload(dt, times, (size_t)nt * sizeof(double), st));
    TRY(upload(dtau, tau, (size_t)ntau * sizeof(double), st));
    TRY(upload(dst, step_times, (size_t)nsteps * sizeof(double), st));
    TRY(upload(dsa, step_sizes, (size_t)nsteps * sizeof(double), st));
    if (variant == HIPDRT_RESPONSE_EXPDECAY) TRY(upload(dtr, tau_rise, (size_t)nsteps * sizeof(double), st));
    const size_t ab = (size_t)nt * ntau * sizeof(double);
    HIPDRT_CHECK(da.alloc(ab));
    if (layered) HIPDRT_CHECK(dl.alloc(ab * nsteps));
    launch_response_variant(st, dt.d(), nt, dtau.d(), ntau, dst.d(), dsa.d(), dtr.d(), nsteps, variant, epsilon, ny, da.d(),
                            layered ? dl.d() : nullptr);
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(a, da.p, ab, hipMemcpyDeviceToHost, st));
    if (layered) HIPDRT_CHECK(hipMemcpyAsync(layered, dl.p, ab * nsteps, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

// lut6 = {log_wt_re, z_re, slope_re, log_wt_im, z_im, slope_im}
static int build_lut6(hipStream_t st, DevBuf& lut6, int ngrid, const double* log_wt_re, const double* z_re,
                      const double* log_wt_im, const double* z_im, bool z_on_device) {
    const size_t gb = (size_t)ngrid * sizeof(double);
    if (!lut6.p) HIPDRT_CHECK(lut6.alloc(6 * gb));
    double* base = lut6.d();
    const hipMemcpyKind zk = z_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIPDRT_CHECK(hipMemcpyAsync(base, log_wt_re, gb, hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipMemcpyAsync(base + 3 * (size_t)ngrid, log_wt_im, gb, hipMemcpyHostToDevice, st));
    if (z_re) HIPDRT_CHECK(hipMemcpyAsync(base + ngrid, z_re, gb, zk, st));
    if (z_im) HIPDRT_CHECK(hipMemcpyAsync(base + 4 * (size_t)ngrid, z_im, gb, zk, st));
    launch_lookup_slopes(st, ngrid, base, base + ngrid, base + 2 * (size_t)ngrid);
    launch_lookup_slopes(st, ngrid, base + 3 * (size_t)ngrid, base + 4 * (size_t)ngrid, base + 5 * (size_t)ngrid);
    LAUNCH_OK();
    return 0;
}

static int impedance_matrix_common(hipdrt_ctx* ctx, int B, int freq_batched, const double* freq, int nf,
                                   const double* tau, int ntau, int mode, int toeplitz, double epsilon, int ngrid,
                                   const double* log_wt_re, const double* z_re, const double* log_wt_im,
                                   const double* z_im, int ny, double* a_re_dev, double* a_im_dev, int repeat,
                                   float* elapsed_ms) {
    HIPDRT_REQUIRE(ctx && freq && tau && a_re_dev && a_im_dev, "NULL pointer");
    HIPDRT_REQUIRE(B >= 1 && nf >= 1 && ntau >= 1, "B, nf, ntau >= 1");
    HIPDRT_REQUIRE(mode == HIPDRT_MODE_INTERP || mode == HIPDRT_MODE_TRAPZ, "mode");
    HIPDRT_REQUIRE(!(toeplitz && freq_batched), "Toeplitz shortcut needs one shared frequency grid");
    if (mode == HIPDRT_MODE_INTERP)
        HIPDRT_REQUIRE(log_wt_re && z_re && log_wt_im && z_im && ngrid >= 2 && ngrid <= 3400,
                       "interp needs lookups with 2 <= ngrid <= 3400");
    else HIPDRT_REQUIRE(ny >= 2 && ny <= 6000, "2 <= ny <= 6000");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dfreq, dtau, lut6, cr;
    TRY(upload(dfreq, freq, (size_t)(freq_batched ? B : 1) * nf * sizeof(double), st));
    TRY(upload(dtau, tau, (size_t)ntau * sizeof(double), st));
    if (mode == HIPDRT_MODE_INTERP) TRY(build_lut6(st, lut6, ngrid, log_wt_re, z_re, log_wt_im, z_im, false));
    HIPDRT_CHECK(cr.alloc(((size_t)(freq_batched ? B : 1) * nf + 2 * (size_t)(nf + ntau)) * sizeof(double)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (elapsed_ms) { HIPDRT_CHECK(hipEventCreate(&e0)); HIPDRT_CHECK(hipEventCreate(&e1)); HIPDRT_CHECK(hipEventRecord(e0, st)); }
    for (int r = 0; r < (repeat < 1 ? 1 : repeat); ++r)
        launch_impedance_matrix(st, B, freq_batched, dfreq.d(), nf, dtau.d(), ntau, mode, toeplitz, epsilon, ngrid,
                                lut6.d(), ny, a_re_dev, a_im_dev, cr.d());
    LAUNCH_OK();
    if (elapsed_ms) { HIPDRT_CHECK(hipEventRecord(e1, st)); }
    HIPDRT_CHECK(hipStreamSynchronize(st));
    if (elapsed_ms) {
        HIPDRT_CHECK(hipEventElapsedTime(elapsed_ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    return HIPDRT_OK;
}

int hipdrt_impedance_matrix_dev(hipdrt_ctx* ctx, int B, int freq_batched, const double* freq, int nf,
                                const double* tau, int ntau, int mode, int toeplitz, double epsilon, int ngrid,
                                const double* log_wt_re, const double* z_re, const double* log_wt_im,
                                const double* z_im, int ny, void* a_re_dev, void* a_im_dev, int repeat,
                                float* elapsed_ms) try {
    return impedance_matrix_common(ctx, B, freq_batched, freq, nf, tau, ntau, mode, toeplitz, epsilon, ngrid,
                                   log_wt_re, z_re, log_wt_im, z_im, ny, (double*)a_re_dev, (double*)a_im_dev, repeat,
                                   elapsed_ms);
} HIPDRT_CATCH

int hipdrt_impedance_matrix(hipdrt_ctx* ctx, int B, int freq_batched, const double* freq, int nf, const double* tau,
                            int ntau, int mode, int toeplitz, double epsilon, int ngrid, const double* log_wt_re,
                            const double* z_re, const double* log_wt_im, const double* z_im, int ny, double* a_re,
                            double* a_im) try {
    HIPDRT_REQUIRE(ctx && a_re && a_im, "NULL pointer");
    HIPDRT_REQUIRE(B >= 1 && nf >= 1 && ntau >= 1, "B, nf, ntau >= 1");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    DevBuf dre, dim;
    const size_t bytes = (size_t)B * nf * ntau * sizeof(double);
    HIPDRT_CHECK(dre.alloc(bytes)); HIPDRT_CHECK(dim.alloc(bytes));
    TRY(impedance_matrix_common(ctx, B, freq_batched, freq, nf, tau, ntau, mode, toeplitz, epsilon, ngrid, log_wt_re,
                                z_re, log_wt_im, z_im, ny, dre.d(), dim.d(), 1, nullptr));
    HIPDRT_CHECK(hipMemcpy(a_re, dre.p, bytes, hipMemcpyDeviceToHost));
    HIPDRT_CHECK(hipMemcpy(a_im, dim.p, bytes, hipMemcpyDeviceToHost));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_nonuniform_gaussian_filter1d(hipdrt_ctx* ctx, const double* y, int n, const double* sigma, const int* seg, int nseg,
                                        const int* filtered, const double* nodes, int K, const double* node_delta,
                                        const double* weights, long long nweights, const int* woff, const int* radius,
                                        double* out) try {
    HIPDRT_REQUIRE(ctx && y && sigma && seg && filtered && nodes && node_delta && weights && woff && radius && out, "NULL pointer");
    HIPDRT_REQUIRE(n >= 1 && nseg >= 1 && K >= 1 && nweights >= 1, "n, nseg, K, nweights >= 1");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    // sample -> segment map (or -1 for an unfiltered segment)
    std::vector<int> seg_of(n, -1);
    for (int s_ = 0; s_ < nseg; ++s_) {
        HIPDRT_REQUIRE(seg[s_] >= 0 && seg[s_] <= seg[s_ + 1] && seg[s_ + 1] <= n, "segment bounds");
        if (filtered[s_]) for (int i = seg[s_]; i < seg[s_ + 1]; ++i) seg_of[i] = s_;
    }
    DevBuf dy, dsg, dso, dseg, dnodes, dnd, dw, dwo, drad, dout;
    TRY(upload(dy, y, (size_t)n * sizeof(double), st));
    TRY(upload(dsg, sigma, (size_t)n * sizeof(double), st));
    TRY(upload(dso, seg_of.data(), (size_t)n * sizeof(int), st));
    TRY(upload(dseg, seg, (size_t)(nseg + 1) * sizeof(int), st));
    TRY(upload(dnodes, nodes, (size_t)nseg * K * sizeof(double), st));
    TRY(upload(dnd, node_delta, (size_t)nseg * sizeof(double), st));
    TRY(upload(dw, weights, (size_t)nweights * sizeof(double), st));
    TRY(upload(dwo, woff, (size_t)nseg * K * sizeof(int), st));
    TRY(upload(drad, radius, (size_t)nseg * K * sizeof(int), st));
    HIPDRT_CHECK(dout.alloc((size_t)n * sizeof(double)));
    launch_nonuniform_gauss(st, dy.d(), n, dsg.d(), dso.i(), dseg.i(), dnodes.d(), K, dnd.d(), dw.d(), dwo.i(), drad.i(),
                            dout.d());
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_penalty_matrices(hipdrt_ctx* ctx, const double* ln_tau, int n, double epsilon, int toeplitz, double* m0,
                            double* m1, double* m2) try {
    HIPDRT_REQUIRE(ctx && ln_tau && m0 && m1 && m2, "NULL pointer");
    HIPDRT_REQUIRE(n >= 1, "n >= 1");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dl, d0, d1, d2;
    const size_t bytes = (size_t)n * n * sizeof(double);
    TRY(upload(dl, ln_tau, (size_t)n * sizeof(double), st));
    HIPDRT_CHECK(d0.alloc(bytes)); HIPDRT_CHECK(d1.alloc(bytes)); HIPDRT_CHECK(d2.alloc(bytes));
    launch_penalty(st, dl.d(), n, epsilon, toeplitz, d0.d(), d1.d(), d2.d(), n, 0);
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(m0, d0.p, bytes, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(m1, d1.p, bytes, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(m2, d2.p, bytes, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_eis_var_matrix(hipdrt_ctx* ctx, const double* freq, int nf, double vmm_epsilon, double reim_cor,
                          int uniform, double* vmm) try {
    HIPDRT_REQUIRE(ctx && freq && vmm, "NULL pointer");
    HIPDRT_REQUIRE(nf >= 1, "nf >= 1");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf df, dv;
    const size_t bytes = (size_t)4 * nf * nf * sizeof(double);
    TRY(upload(df, freq, (size_t)nf * sizeof(double), st));
    HIPDRT_CHECK(dv.alloc(bytes));
    launch_eis_vmm(st, df.d(), nf, vmm_epsilon, reim_cor, uniform, dv.d());
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(vmm, dv.p, bytes, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

static hipdrt_qp_opts default_qp_opts() { return hipdrt_qp_opts{1e-7, 1e-6, 1e-7, 100}; }

int hipdrt_qp_batch(hipdrt_ctx* ctx, int B, int n, int p_batched, const double* P, const double* q, int h_batched,
                    const double* h, const hipdrt_qp_opts* opts, double* x, int* iters, double* pcost, int* status) try {
    HIPDRT_REQUIRE(ctx && P && q && h && x && status, "NULL pointer");
    HIPDRT_REQUIRE(B >= 1 && n >= 1 && n <= 4096, "B >= 1, 1 <= n <= 4096");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dP, dq, dh, dL, dx, dit, dpc, dst, dstate, dPpk, dgs;
    const int ldl = (int)qp_scratch_ld(n);
    const int G = qp_group_size(B, n, ctx->qp_force_group);           // 0: one workgroup per problem; >= 1: that many workgroups per problem
    // device copy of P with an even leading dimension (16-byte row-pair loads in the kernels), pad column zeroed
    const int ldp = round_up(n, 2);
    const size_t nmat = (size_t)(p_batched ? B : 1);
    HIPDRT_CHECK(dP.alloc(nmat * n * ldp * sizeof(double)));
    if (ldp != n) HIPDRT_CHECK(hipMemsetAsync(dP.p, 0, dP.bytes, st));
    HIPDRT_CHECK(hipMemcpy2DAsync(dP.p, (size_t)ldp * sizeof(double), P, (size_t)n * sizeof(double),
                                  (size_t)n * sizeof(double), nmat * n, hipMemcpyHostToDevice, st));
    TRY(upload(dq, q, (size_t)B * n * sizeof(double), st));
    TRY(upload(dh, h, (size_t)(h_batched ? B : 1) * n * sizeof(double), st));
    HIPDRT_CHECK(dL.alloc((size_t)B * qp_scratch_doubles(n, G) * sizeof(double)));
    HIPDRT_CHECK(dx.alloc((size_t)B * n * sizeof(double)));
    HIPDRT_CHECK(dit.alloc((size_t)B * sizeof(int)));
    HIPDRT_CHECK(dpc.alloc((size_t)B * sizeof(double)));
    HIPDRT_CHECK(dst.alloc((size_t)B * sizeof(int)));
    QpArgs a{};
    a.B = B; a.n = n; a.P = dP.d(); a.p_stride = p_batched ? (long long)n * ldp : 0; a.ldp = ldp;
    a.q = dq.d(); a.h = dh.d(); a.h_stride = h_batched ? n : 0;
    a.L = dL.d(); a.ldl = ldl; a.l_stride = (long long)qp_scratch_doubles(n, G);
    a.x = dx.d(); a.iters = dit.i(); a.pcost = dpc.d(); a.status = dst.i();
    a.active = nullptr; a.iters_accum = nullptr;
    a.G = G;
    a.waves = ctx->qp_waves;
    if (G >= 1) {
        HIPDRT_CHECK(dgs.alloc((size_t)B * qp_gsync_ints() * sizeof(int)));
        a.gsync = dgs.i();
    }
    HIPDRT_CHECK(dPpk.alloc(nmat * qp_ppk_doubles(n) * sizeof(double)));
    launch_pack_p(st, (int)nmat, n, dP.d(), ldp, (long long)n * ldp, dPpk.d(), (long long)qp_ppk_doubles(n), qp_nchp(n));
    a.Ppk = dPpk.d(); a.ppk_stride = p_batched ? (long long)qp_ppk_doubles(n) : 0; a.nchp = qp_nchp(n);
    HIPDRT_CHECK(dstate.alloc((size_t)B * (G > 1 ? G : 1) * qp_state_doubles(n) * sizeof(double)));
    a.state = dstate.d(); a.state_ld = qp_state_ld(n); a.state_stride = (long long)qp_state_doubles(n);
    a.opts = opts ? *opts : default_qp_opts();
    TRY(launch_qp(st, a));
    HIPDRT_CHECK(hipMemcpyAsync(x, dx.p, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (iters) HIPDRT_CHECK(hipMemcpyAsync(iters, dit.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (pcost) HIPDRT_CHECK(hipMemcpyAsync(pcost, dpc.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(status, dst.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_debug_qp_group(hipdrt_ctx* ctx, int members) try {
    HIPDRT_REQUIRE(ctx, "NULL pointer");
    ctx->qp_force_group = members;
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_debug_stream_pool(hipdrt_ctx* ctx, int cap, void** streams, int* holders, int* running, int* size) try {
    HIPDRT_REQUIRE(ctx && size && cap >= 0, "NULL pointer");
    StreamPool* pl = stream_pool(ctx->device);
    std::lock_guard<std::mutex> lock(pl->mu);
    *size = (int)pl->st.size();
    for (int i = 0; i < std::min(cap, *size); ++i) {
        if (streams) streams[i] = pl->st[i];
        if (holders) holders[i] = pl->holders[i];
        if (running) running[i] = pl->running[i];
    }
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_debug_qp_waves(hipdrt_ctx* ctx, int waves) try {
    HIPDRT_REQUIRE(ctx, "NULL pointer");
    HIPDRT_REQUIRE(waves == -1 || waves == 4 || waves == 8, "waves: 4, 8 or -1");
    ctx->qp_waves = waves;
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_debug_exact_zero_shortcuts(hipdrt_ctx* ctx, int on) try {
    HIPDRT_REQUIRE(ctx, "NULL pointer");
    ctx->zero_shortcuts = on ? 1 : 0;
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_debug_qp_occupancy(hipdrt_ctx* ctx, int threads, int n) try {
    if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return -1;
    return qp_occupancy(threads, n);
} HIPDRT_CATCH

int hipdrt_qp_profile(hipdrt_ctx* ctx, unsigned long long* cycles, int n, int reset) try {
    HIPDRT_REQUIRE(ctx && cycles, "NULL pointer");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    HIPDRT_CHECK(hipStreamSynchronize(ctx->stream));
    return qp_profile_read(cycles, n, reset) < 0 ? HIPDRT_E_HIP : HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_weighted_gram(hipdrt_ctx* ctx, int B, int m, int n, const double* A, const double* w, const double* b,
                         int l2_batched, const double* l2, const double* l1, double* P, double* q) try {
    HIPDRT_REQUIRE(ctx && A && w && b && P && q, "NULL pointer");
    HIPDRT_REQUIRE(B >= 1 && m >= 1 && n >= 1, "B, m, n >= 1");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    DevBuf dA, dw, db, dl2, dl1, dP, dq;
    TRY(upload(dA, A, (size_t)m * n * sizeof(double), st));
    TRY(upload(dw, w, (size_t)B * m * sizeof(double), st));
    TRY(upload(db, b, (size_t)B * m * sizeof(double), st));
    if (l2) TRY(upload(dl2, l2, (size_t)(l2_batched ? B : 1) * n * n * sizeof(double), st));
    if (l1) TRY(upload(dl1, l1, (size_t)n * sizeof(double), st));
    HIPDRT_CHECK(dP.alloc((size_t)B * n * n * sizeof(double)));
    HIPDRT_CHECK(dq.alloc((size_t)B * n * sizeof(double)));
    launch_weighted_gram(st, B, m, n, dA.d(), n, dw.d(), db.d(), l2 ? dl2.d() : nullptr,
                         l2_batched ? (long long)n * n : 0, n, l1 ? dl1.d() : nullptr, dP.d(), n, (long long)n * n,
                         dq.d(), nullptr);
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(P, dP.p, (size_t)B * n * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(q, dq.p, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

// ---- plan ---------------------------------------------------------------------------------------------------

void hipdrt_default_fit_opts(hipdrt_fit_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->rp_scale = 14;
    const double dw[3] = {1.5, 1.0, 0.5}, sd[3] = {1, 1000, 1000}, sa[3] = {5, 10, 25}, ra[3] = {0.15, 0.2, 0.25};
    for (int k = 0; k < 3; ++k) {
        o->derivative_weights[k] = dw[k]; o->sigma_ds[k] = sd[k]; o->s_alpha[k] = sa[k]; o->s_0[k] = 1.0;
        o->rho_alpha[k] = ra[k]; o->rho_0[k] = 1.0;
    }
    o->l1_lambda_0 = 0; o->l2_lambda_0 = 142;
    o->iw_l1_lambda_0 = 1e-4; o->iw_l2_lambda_0 = 1e-4;
    o->ohmic_penalty = 1e-6; o->inductance_penalty = 1e-6; o->inductance_scale = 1e-5;
    o->eis_vmm_epsilon = 0.25; o->eis_reim_cor = 0.25;
    o->xtol = 1e-2; o->max_iter = 50; o->nonneg = 1; o->scale_data = 1; o->fit_ohmic = 1; o->fit_inductance = 1;
    o->update_scale = 0; o->eff_hp = 1;
    o->eis_error_uniform = 0;
    o->outlier_p = -1.0; o->iw_alpha = -1.0; o->iw_beta = -1.0;
    o->qp = default_qp_opts();
}

static int plan_build_matrices(hipdrt_plan* p, bool build_lookup) {
    hipStream_t st = p->ctx->stream;
    const size_t gb = (size_t)p->ngrid * sizeof(double);
    if (p->mode == HIPDRT_MODE_INTERP && build_lookup) {
        double* base = p->lut6.d();
        launch_lookup(st, p->eps, p->ngrid, p->ny, p->wt_re.d(), p->wt_im.d(), base + p->ngrid, base + 4 * (size_t)p->ngrid);
        LAUNCH_OK();
    }
    if (p->mode == HIPDRT_MODE_INTERP) {
        double* base = p->lut6.d();
        launch_lookup_slopes(st, p->ngrid, base, base + p->ngrid, base + 2 * (size_t)p->ngrid);
        launch_lookup_slopes(st, p->ngrid, base + 3 * (size_t)p->ngrid, base + 4 * (size_t)p->ngrid, base + 5 * (size_t)p->ngrid);
    }
    (void)gb;
    launch_impedance_matrix(st, 1, 0, p->freq.d(), p->nf, p->tau.d(), p->ntau, p->mode, p->toeplitz_a, p->eps, p->ngrid,
                            p->lut6.d(), p->ny, p->a_re.d(), p->a_im.d(), p->cr.d());
    LAUNCH_OK();
    FitState fs = p->state();
    launch_assemble_rm(st, fs, p->a_re.d(), p->a_im.d(), p->freq.d(), p->rm.d(), p->idx_rinf, p->idx_induc);
    LAUNCH_OK();
    return 0;
}

// work space for `capacity` spectra
static int plan_alloc_batch(hipdrt_plan* p) {
    const size_t cap = (size_t)p->capacity;
    const int n = p->n, m = p->m, nf = p->nf > 0 ? p->nf : 1;
    HIPDRT_CHECK(p->z_re.alloc(cap * nf * sizeof(double))); HIPDRT_CHECK(p->z_im.alloc(cap * nf * sizeof(double)));
    HIPDRT_CHECK(p->rv.alloc(cap * m * sizeof(double))); HIPDRT_CHECK(p->w.alloc(cap * m * sizeof(double)));
    HIPDRT_CHECK(p->est_w.alloc(cap * m * sizeof(double)));
    HIPDRT_CHECK(p->x.alloc(cap * n * sizeof(double))); HIPDRT_CHECK(p->x_in.alloc(cap * n * sizeof(double)));
    HIPDRT_CHECK(p->q.alloc(cap * n * sizeof(double)));
    HIPDRT_CHECK(p->s.alloc(cap * 3 * n * sizeof(double)));
    HIPDRT_CHECK(p->rho.alloc(cap * 3 * sizeof(double))); HIPDRT_CHECK(p->xmx.alloc(cap * 3 * sizeof(double)));
    HIPDRT_CHECK(p->coef_scale.alloc(cap * sizeof(double))); HIPDRT_CHECK(p->var_floor.alloc(cap * sizeof(double)));
    HIPDRT_CHECK(p->pcost.alloc(cap * sizeof(double)));
    for (DevBuf* ib : {&p->active, &p->outer_iters, &p->fit_status, &p->qp_iters_total, &p->qp_status, &p->qp_iters})
        HIPDRT_CHECK(ib->alloc(cap * sizeof(int)));
    HIPDRT_CHECK(p->n_active.alloc(sizeof(int)));
    p->qp_G = qp_group_size(p->capacity, n, p->ctx->qp_force_group);
    HIPDRT_REQUIRE(p->qp_G >= 0, "n too large for the QP kernels");
    HIPDRT_CHECK(p->L.alloc(cap * qp_scratch_doubles(n, p->qp_G) * sizeof(double)));
    HIPDRT_CHECK(p->Ptmp.alloc((size_t)n * p->ldp * sizeof(double)));
    HIPDRT_CHECK(p->qpstate.alloc(cap * (p->qp_G > 1 ? p->qp_G : 1) * qp_state_doubles(n) * sizeof(double)));
    HIPDRT_CHECK(p->gsync.alloc(cap * qp_gsync_ints() * sizeof(int)));
    HIPDRT_CHECK(p->Ppk.alloc(cap * qp_ppk_doubles(n) * sizeof(double)));
    HIPDRT_CHECK(p->order.alloc(cap * sizeof(int)));
    if (p->opts.outlier_p > 0.0) {
        HIPDRT_CHECK(p->vmm_base.alloc((size_t)m * m * sizeof(double)));
        HIPDRT_CHECK(p->outlier_t.alloc(cap * m * sizeof(double)));
    }
    HIPDRT_CHECK(p->hist_rows.alloc(sizeof(int)));
    return 0;
}

int hipdrt_plan_create(hipdrt_ctx* ctx, const double* freq, int nf, const double* tau, int ntau, double epsilon,
                       int mode, int toeplitz_a, int toeplitz_m, int ngrid, int ny, const double* wt_re,
                       const double* wt_im, const double* log_wt_re, const double* log_wt_im,
                       const hipdrt_fit_opts* opts, int capacity, hipdrt_plan** out) try {
    HIPDRT_REQUIRE(ctx && freq && tau && out, "NULL pointer");
    HIPDRT_REQUIRE(nf >= 2 && ntau >= 2 && capacity >= 1, "nf, ntau >= 2, capacity >= 1");
    HIPDRT_REQUIRE(mode == HIPDRT_MODE_INTERP || mode == HIPDRT_MODE_TRAPZ, "mode");
    if (mode == HIPDRT_MODE_INTERP)
        HIPDRT_REQUIRE(wt_re && wt_im && log_wt_re && log_wt_im && ngrid >= 2 && ngrid <= 3400, "interp lookups");
    HIPDRT_REQUIRE(ny >= 2 && ny <= 6000, "2 <= ny <= 6000");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    std::unique_ptr<hipdrt_plan> p(new hipdrt_plan());
    p->ctx = ctx;
    if (opts) p->opts = *opts; else hipdrt_default_fit_opts(&p->opts);
    p->nf = nf; p->ntau = ntau; p->eps = epsilon; p->mode = mode; p->ngrid = ngrid; p->ny = ny;
    p->toeplitz_a = toeplitz_a; p->toeplitz_m = toeplitz_m; p->capacity = capacity;
    // special parameters in registration order (drt1d.py:383-388): R_inf, then inductance
    int ns = 0;
    if (p->opts.fit_ohmic) p->idx_rinf = ns++;
    if (p->opts.fit_inductance) p->idx_induc = ns++;
    p->ns = ns; p->n = ns + ntau; p->m = 2 * nf;
    const int n = p->n, m = p->m;
    HIPDRT_REQUIRE(n <= 4096, "ns + ntau <= 4096");
    p->ldrm = round_up(n, 2); p->ldm = round_up(n, 2); p->ldp = round_up(n, 2); p->ldl = (int)qp_scratch_ld(n);

    std::vector<double> ln_tau(ntau);
    for (int i = 0; i < ntau; ++i) ln_tau[i] = std::log(tau[i]);
    TRY(upload(p->freq, freq, (size_t)nf * sizeof(double), st));
    TRY(upload(p->tau, tau, (size_t)ntau * sizeof(double), st));
    const size_t gb = (size_t)(ngrid > 0 ? ngrid : 1) * sizeof(double);
    if (mode == HIPDRT_MODE_INTERP) {
        TRY(upload(p->wt_re, wt_re, gb, st)); TRY(upload(p->wt_im, wt_im, gb, st));
        HIPDRT_CHECK(p->lut6.alloc(6 * gb));
        HIPDRT_CHECK(hipMemcpyAsync(p->lut6.d(), log_wt_re, gb, hipMemcpyHostToDevice, st));
        HIPDRT_CHECK(hipMemcpyAsync(p->lut6.d() + 3 * (size_t)ngrid, log_wt_im, gb, hipMemcpyHostToDevice, st));
    }
    HIPDRT_CHECK(p->a_re.alloc((size_t)nf * ntau * sizeof(double)));
    HIPDRT_CHECK(p->a_im.alloc((size_t)nf * ntau * sizeof(double)));
    HIPDRT_CHECK(p->cr.alloc(2 * (size_t)(nf + ntau) * sizeof(double)));
    HIPDRT_CHECK(p->rm.alloc((size_t)m * p->ldrm * sizeof(double)));
    HIPDRT_CHECK(hipMemsetAsync(p->rm.p, 0, p->rm.bytes, st));
    for (int k = 0; k < 3; ++k) {
        HIPDRT_CHECK(p->mk[k].alloc((size_t)n * p->ldm * sizeof(double)));
        HIPDRT_CHECK(hipMemsetAsync(p->mk[k].p, 0, p->mk[k].bytes, st));
    }
    HIPDRT_CHECK(p->vmm.alloc((size_t)m * m * sizeof(double)));
    HIPDRT_CHECK(p->h.alloc((size_t)n * sizeof(double)));
    // l1_lambda_vector: 0 on specials, l1_lambda_0 on DRT coefficients (drt1d.py:552-553)
    std::vector<double> l1(n, 0.0);
    for (int i = ns; i < n; ++i) l1[i] = p->opts.l1_lambda_0;
    TRY(upload(p->l1, l1.data(), (size_t)n * sizeof(double), st));
    // ln(tau) on the host: np.log(self.basis_tau) (drt1d.py:5694)
    TRY(upload(p->ln_tau, ln_tau.data(), (size_t)ntau * sizeof(double), st));
    HIPDRT_CHECK(hipStreamSynchronize(st));   // host vectors above go out of scope

    TRY(plan_alloc_batch(p.get()));

    // shared matrices on the device
    TRY(plan_build_matrices(p.get(), true));
    launch_penalty(st, p->ln_tau.d(), ntau, epsilon, toeplitz_m, p->mk[0].d(), p->mk[1].d(), p->mk[2].d(), p->ldm, ns);
    launch_special_penalty(st, p->mk[0].d(), p->mk[1].d(), p->mk[2].d(), p->ldm, p->idx_rinf, p->idx_induc,
                           p->opts.ohmic_penalty, p->opts.inductance_penalty);
    launch_eis_vmm(st, p->freq.d(), nf, p->opts.eis_vmm_epsilon, p->opts.eis_reim_cor, p->opts.eis_error_uniform,
                   p->vmm.d());
    if (p->vmm_base.p) launch_vmm_exclude_self(st, p->vmm.d(), m, p->vmm_base.d());
    launch_make_h(st, p->h.d(), n, ns, p->opts.nonneg);
    LAUNCH_OK();
    HIPDRT_CHECK(hipStreamSynchronize(st));
    TRY(plan_toep_reach(p.get()));
    { std::lock_guard<std::mutex> lk(g_life); ++ctx->plans; }
    *out = p.release();
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_create_prepared(hipdrt_ctx* ctx, const hipdrt_prepared_desc* d, const double* m0, const double* m1,
                                const double* m2, const double* vmm, const double* h, const double* l1,
                                const double* vz_strength, const hipdrt_fit_opts* opts, int capacity, hipdrt_plan** out) try {
    HIPDRT_REQUIRE(ctx && d && m0 && m1 && m2 && vmm && h && l1 && out, "NULL pointer");
    HIPDRT_REQUIRE(d->m >= 2 && d->n >= 2 && d->ns >= 0 && d->ns < d->n && capacity >= 1, "m, n >= 2, 0 <= ns < n, capacity >= 1");
    HIPDRT_REQUIRE(d->n <= 4096, "n <= 4096");
    HIPDRT_REQUIRE(d->dop_size >= 0 && (d->dop_size == 0 || (d->dop_start >= 0 && d->dop_start + d->dop_size <= d->ns)),
                   "the x_dop block must lie inside the special parameters");
    HIPDRT_REQUIRE(d->dop_size <= d->n - d->ns, "x_dop block larger than the DRT block");
    HIPDRT_REQUIRE(d->vz_index < d->ns && (d->vz_index < 0 || vz_strength), "vz_offset column / strength vector");
    HIPDRT_REQUIRE(d->vb_size >= 0 && d->vb_start >= 0 && d->vb_start + d->vb_size <= d->ns, "v_baseline columns");
    HIPDRT_REQUIRE(d->num_chrono >= 0 && d->num_chrono <= d->m, "num_chrono");
    HIPDRT_REQUIRE(!(opts && opts->update_scale) || d->basis_area > 0.0, "update_scale needs desc.basis_area");
    HIPDRT_CHECK(hipSetDevice(ctx->device)); (void)hipGetLastError();
    hipStream_t st = ctx->stream;
    std::unique_ptr<hipdrt_plan> p(new hipdrt_plan());
    p->ctx = ctx;
    if (opts) p->opts = *opts; else hipdrt_default_fit_opts(&p->opts);
    p->prepared = 1; p->desc = *d;
    p->n = d->n; p->m = d->m; p->ns = d->ns; p->ntau = d->n - d->ns; p->nf = 0; p->toeplitz_m = d->toeplitz_m;
    p->capacity = capacity;
    const int n = p->n, m = p->m;
    p->ldrm = round_up(n, 2); p->ldm = round_up(n, 2); p->ldp = round_up(n, 2); p->ldl = (int)qp_scratch_ld(n);
    const double* mk[3] = {m0, m1, m2};
    for (int k = 0; k < 3; ++k) {
        HIPDRT_CHECK(p->mk[k].alloc((size_t)n * p->ldm * sizeof(double)));
        HIPDRT_CHECK(hipMemsetAsync(p->mk[k].p, 0, p->mk[k].bytes, st));
        HIPDRT_CHECK(hipMemcpy2DAsync(p->mk[k].p, (size_t)p->ldm * sizeof(double), mk[k], (size_t)n * sizeof(double),
                                      (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, st));
    }
    TRY(upload(p->vmm, vmm, (size_t)m * m * sizeof(double), st));
    TRY(upload(p->h, h, (size_t)n * sizeof(double), st));
    TRY(upload(p->l1, l1, (size_t)n * sizeof(double), st));
    if (vz_strength) TRY(upload(p->vz_strength, vz_strength, (size_t)m * sizeof(double), st));
    TRY(plan_alloc_batch(p.get()));
    if (p->vmm_base.p) launch_vmm_exclude_self(st, p->vmm.d(), m, p->vmm_base.d());     // outlier_p: qphb.py:1644-1648
    HIPDRT_CHECK(p->dop_rho.alloc((size_t)capacity * 3 * sizeof(double)));
    HIPDRT_CHECK(p->dop_xmx.alloc((size_t)capacity * 3 * sizeof(double)));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    TRY(plan_toep_reach(p.get()));
    { std::lock_guard<std::mutex> lk(g_life); ++ctx->plans; }
    *out = p.release();
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_upload_prepared(hipdrt_plan* p, int B, int rm_batched, const double* rzm, const double* rzv) try {
    HIPDRT_REQUIRE(p && rzm && rzv, "NULL pointer");
    HIPDRT_REQUIRE(p->prepared, "not a prepared plan");
    HIPDRT_REQUIRE(B >= 1 && B <= p->capacity, "1 <= B <= capacity");
    HIPDRT_REQUIRE(rm_batched || p->desc.vz_index < 0, "a vz_offset column needs one response matrix per measurement");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int n = p->n, m = p->m;
    const size_t nmat = rm_batched ? (size_t)B : 1;
    const size_t need = nmat * m * p->ldrm * sizeof(double);
    if (p->rm.bytes < need) HIPDRT_CHECK(p->rm.alloc(need));
    HIPDRT_CHECK(hipMemsetAsync(p->rm.p, 0, need, st));
    HIPDRT_CHECK(hipMemcpy2DAsync(p->rm.p, (size_t)p->ldrm * sizeof(double), rzm, (size_t)n * sizeof(double),
                                  (size_t)n * sizeof(double), nmat * m, hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipMemcpyAsync(p->rv.p, rzv, (size_t)B * m * sizeof(double), hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    p->rm_stride = rm_batched ? (long long)m * p->ldrm : 0;
    p->B = B;
    p->prepped = 0;
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_set_weight_factors(hipdrt_plan* p, double weight_factor, const double* row_factors, int batched) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(weight_factor > 0.0, "weight_factor > 0");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    p->weight_factor = weight_factor;
    p->wrow_batched = (batched & 1) ? 1 : 0;
    p->wrow_late = (batched & 2) ? 1 : 0;
    if (row_factors) {
        const size_t cnt = ((batched & 1) ? (size_t)p->capacity : 1) * p->m;   // bit 1 (late) does not make it per spectrum
        TRY(upload(p->wrow, row_factors, cnt * sizeof(double), st));
    } else {
        p->wrow.release();
    }
    if (p->has_weight_factors() && !p->w_eff.p) HIPDRT_CHECK(p->w_eff.alloc((size_t)p->capacity * p->m * sizeof(double)));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_set_init_h(hipdrt_plan* p, const double* h_init) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    if (!h_init) { p->h_init.release(); return HIPDRT_OK; }
    TRY(upload(p->h_init, h_init, (size_t)p->n * sizeof(double), p->ctx->stream));
    HIPDRT_CHECK(hipStreamSynchronize(p->ctx->stream));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_destroy(hipdrt_plan* plan) try {
    if (!plan) return HIPDRT_OK;
    std::lock_guard<std::mutex> lk(g_life);
    hipdrt_ctx* ctx = plan->ctx;
    (void)hipSetDevice(ctx->device);
    delete plan;
    if (--ctx->plans == 0 && ctx->released) free_ctx(ctx);
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_dims(hipdrt_plan* plan, int* n, int* m, int* ns) try {
    HIPDRT_REQUIRE(plan, "plan is NULL");
    if (n) *n = plan->n;
    if (m) *m = plan->m;
    if (ns) *ns = plan->ns;
    return HIPDRT_OK;
} HIPDRT_CATCH

static int copy_strided(double* out, const double* dev, int rows, int cols, int ld, hipStream_t st) {
    HIPDRT_CHECK(hipMemcpy2DAsync(out, (size_t)cols * sizeof(double), dev, (size_t)ld * sizeof(double),
                                  (size_t)cols * sizeof(double), rows, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return 0;
}

int hipdrt_plan_get(hipdrt_plan* p, const char* which, double* out, long long count) try {
    HIPDRT_REQUIRE(p && which && out, "NULL pointer");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const std::string w = which;
    const double* src = nullptr; int rows = 0, cols = 0, ld = 0;
    if (w == "lut_z_re") { src = p->lut6.d() + p->ngrid; rows = 1; cols = ld = p->ngrid; }
    else if (w == "lut_z_im") { src = p->lut6.d() + 4 * (size_t)p->ngrid; rows = 1; cols = ld = p->ngrid; }
    else if (w == "a_re") { src = p->a_re.d(); rows = p->nf; cols = ld = p->ntau; }
    else if (w == "a_im") { src = p->a_im.d(); rows = p->nf; cols = ld = p->ntau; }
    else if (w == "rm") { src = p->rm.d(); rows = p->m; cols = p->n; ld = p->ldrm; }
    else if (w == "m0" || w == "m1" || w == "m2") { src = p->mk[w[1] - '0'].d(); rows = cols = p->n; ld = p->ldm; }
    else if (w == "vmm") { src = p->vmm.d(); rows = cols = ld = p->m; }
    else if (w == "h") { src = p->h.d(); rows = 1; cols = ld = p->n; }
    else if (w == "est_weights") { src = p->est_w.d(); rows = p->B; cols = ld = p->m; }   // per spectrum of the last batch
    else if (w == "rv") { src = p->rv.d(); rows = p->B; cols = ld = p->m; }
    else if (w == "xmx") { src = p->xmx.d(); rows = p->B; cols = ld = 3; }
    else if (w == "outlier_t" && p->outlier_t.p) { src = p->outlier_t.d(); rows = p->B; cols = ld = p->m; }
    else if (w == "weight_factors" && p->wfac.p) { src = p->wfac.d(); rows = p->B; cols = ld = 2; }
    else if (w == "dop_rho" && p->prepared) { src = p->dop_rho.d(); rows = p->B; cols = ld = 3; }
    else if (w == "dop_xmx" && p->prepared) { src = p->dop_xmx.d(); rows = p->B; cols = ld = 3; }
    else if (w == "rzm") { src = p->rm.d(); rows = (p->rm_stride ? p->B : 1) * p->m; cols = p->n; ld = p->ldrm; }
    else if (w == "hist_dop_rho" && p->prepared && p->hist_b >= 0) { src = p->hist_dop_rho.d(); rows = p->hist_cap; cols = ld = 3; }
    else { set_error("unknown matrix name: " + w); return HIPDRT_E_INVALID; }
    HIPDRT_REQUIRE(src != nullptr, "matrix not available in this mode");
    HIPDRT_REQUIRE(count == (long long)rows * cols, "count does not match the matrix size");
    return copy_strided(out, src, rows, cols, ld, st);
} HIPDRT_CATCH

int hipdrt_plan_set_lookup(hipdrt_plan* p, const double* z_re, const double* z_im) try {
    HIPDRT_REQUIRE(p && z_re && z_im, "NULL pointer");
    HIPDRT_REQUIRE(p->mode == HIPDRT_MODE_INTERP, "plan is not in interp mode");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const size_t gb = (size_t)p->ngrid * sizeof(double);
    HIPDRT_CHECK(hipMemcpyAsync(p->lut6.d() + p->ngrid, z_re, gb, hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipMemcpyAsync(p->lut6.d() + 4 * (size_t)p->ngrid, z_im, gb, hipMemcpyHostToDevice, st));
    TRY(plan_build_matrices(p, false));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_upload(hipdrt_plan* p, int B, const double* z_re, const double* z_im) try {
    HIPDRT_REQUIRE(p && z_re && z_im, "NULL pointer");
    HIPDRT_REQUIRE(!p->prepared, "prepared plans take hipdrt_plan_upload_prepared");
    HIPDRT_REQUIRE(B >= 1 && B <= p->capacity, "1 <= B <= capacity");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const size_t bytes = (size_t)B * p->nf * sizeof(double);
    HIPDRT_CHECK(hipMemcpyAsync(p->z_re.p, z_re, bytes, hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipMemcpyAsync(p->z_im.p, z_im, bytes, hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    p->B = B;
    return HIPDRT_OK;
} HIPDRT_CATCH

// L2 part of P in hyper-parameter form (calculate_qp_l2_matrix, qphb.py:53-120) for the plan's current state
static GramL2 plan_l2(const hipdrt_plan* p, double l2_lambda_0, const double* derivative_weights, double dop_l2_lambda_0) {
    GramL2 g{};
    g.l2 = nullptr; g.ldm = p->ldm; g.ns = p->ns; g.use_rho = 1;
    g.sym = p->prepared ? 0 : p->toeplitz_m;      // caller-supplied matrices are not assumed bitwise symmetric
    g.toep = p->toeplitz_m;                       // log-uniform tau grid (the hyper kernel relies on the same structure)
    g.toep_maxd = (p->toeplitz_m && !(p->ctx && !p->ctx->zero_shortcuts)) ? p->toep_maxd : -1;
    g.spec_zero = p->spec_zero;
    for (int k = 0; k < 3; ++k) { g.mk[k] = p->mk[k].d(); g.dfac[k] = l2_lambda_0 * derivative_weights[k]; }
    g.s = p->s.d(); g.rho = p->rho.d();
    if (p->prepared && p->desc.dop_size > 0) {
        g.dop_start = p->desc.dop_start; g.dop_size = p->desc.dop_size; g.dop_rho = p->dop_rho.d();
        for (int k = 0; k < 3; ++k) g.dop_dfac[k] = dop_l2_lambda_0 * p->desc.dop_derivative_weights[k];
    }
    return g;
}

// Reach of the penalty matrices on a log-uniform grid: the largest distance from the diagonal at which the first row of the DRT
// block of any order is not exactly zero (Gaussian basis: e^(-a^2 / 2) underflows ~39 grid points out at 10 points per decade,
// whatever the matrix size).  The Gram kernel's L2 epilogue skips tiles that lie wholly beyond it.  Once per plan.
static int plan_toep_reach(hipdrt_plan* p) {
    p->toep_maxd = -1;
    if (!p->toeplitz_m) return HIPDRT_OK;
    const int nd = p->n - p->ns;
    std::vector<double> row(nd);
    int reach = 0;
    for (int k = 0; k < 3; ++k) {
        HIPDRT_CHECK(hipMemcpy(row.data(), p->mk[k].d() + (size_t)p->ns * p->ldm + p->ns, (size_t)nd * sizeof(double), hipMemcpyDeviceToHost));
        for (int d = nd - 1; d > reach; --d)
            if (row[d] != 0.0) { reach = d; break; }
    }
    p->toep_maxd = reach;
    // the columns of the special parameters below the special block, and their rows to the right of it
    p->spec_zero = 1;
    if (p->ns > 0) {
        std::vector<double> cols((size_t)nd * p->ns), rows((size_t)p->ns * nd);
        for (int k = 0; k < 3 && p->spec_zero; ++k) {
            HIPDRT_CHECK(hipMemcpy2D(cols.data(), (size_t)p->ns * sizeof(double), p->mk[k].d() + (size_t)p->ns * p->ldm,
                                     (size_t)p->ldm * sizeof(double), (size_t)p->ns * sizeof(double), nd, hipMemcpyDeviceToHost));
            HIPDRT_CHECK(hipMemcpy2D(rows.data(), (size_t)nd * sizeof(double), p->mk[k].d() + p->ns, (size_t)p->ldm * sizeof(double),
                                     (size_t)nd * sizeof(double), p->ns, hipMemcpyDeviceToHost));
            for (double v : cols) if (v != 0.0) { p->spec_zero = 0; break; }
            for (double v : rows) if (v != 0.0) { p->spec_zero = 0; break; }
        }
    }
    return HIPDRT_OK;
}

// hyper-parameter step of one outer iteration.  Few fits with large matrices: their matrix-vector products are spread over
// many workgroups first (premv_kernel), else the one workgroup per fit of hyper_kernel would stream them through one CU each.
static int plan_hyper(hipdrt_plan* p, hipStream_t st, const FitState& fs_in, int B, int it) {
    FitState fs = fs_in;
    const size_t premv_need = 3 * (size_t)(p->capacity > B ? p->capacity : B) * p->m * sizeof(double);
    // (the options of THIS loop decide -- a warm restart may switch outlier_p on or off against the plan's fit)
    const bool outl = fs_in.opts.outlier_p > 0.0;
    if (B * 8 <= device_cus() && (size_t)p->m * p->n >= ((size_t)1 << 20) && !outl) {
        if (p->premv.bytes < premv_need) HIPDRT_CHECK(p->premv.alloc(premv_need));
        fs.premv = p->premv.d();
    } else if (p->rm_stride == 0 && !outl && !(p->prepared && p->desc.vz_index >= 0)) {
        // one response matrix and one variance matrix for the whole batch (every EIS plan, prepared plans without a vz_offset
        // column): rm @ x and vmm @ resid^2 of all spectra as two batched products (hyper.hip: batch_products_kernel) -- for any
        // batch size, so that a spectrum's bits do not depend on whether it is fitted alone or among a thousand
        if (p->premv.bytes < premv_need) HIPDRT_CHECK(p->premv.alloc(premv_need));     // (a sub-batch view: a window of the parent's)
        fs.premv = p->premv.d();
        fs.premv_batched = 1;
    }
    return launch_hyper(st, fs, B, it);
}

namespace {
struct PhaseTimer {
    hipStream_t st;
    std::vector<hipEvent_t> ev;
    std::vector<int> cat;
    explicit PhaseTimer(hipStream_t s) : st(s) {}
    ~PhaseTimer() { for (auto e : ev) (void)hipEventDestroy(e); }
    void mark(int category) {   // closes the previous phase, opens `category` (-1 = end)
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, st);
        ev.push_back(e);
        cat.push_back(category);
    }
    void collect(float* t_ms, int* launches) {
        for (int i = 0; i < 5; ++i) { t_ms[i] = 0; launches[i] = 0; }
        for (size_t i = 0; i + 1 < ev.size(); ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess && cat[i] >= 1 && cat[i] <= 4) {
                t_ms[cat[i]] += ms; launches[cat[i]] += 1;
            }
        }
        if (ev.size() >= 2) { float ms = 0; (void)hipEventElapsedTime(&ms, ev.front(), ev.back()); t_ms[0] = ms; launches[0] = 1; }
    }
};
}  // namespace

// the whole fit of the plan's staged spectra on the plan's stream (hipdrt_plan_fit; also run per sub-batch view)
static int plan_fit_one(hipdrt_plan* p) {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(p->B >= 1, "no spectra staged (call hipdrt_plan_upload)");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int B = p->B, n = p->n, m = p->m;
    FitState fs = p->state();
    PhaseTimer tm(st);
    tm.mark(4);
    if (p->hist_b >= 0) HIPDRT_CHECK(hipMemsetAsync(p->hist_rows.p, 0, sizeof(int), st));
    launch_prep(st, fs, B);
    LAUNCH_OK();
    p->prepped = 1;

    // initialize_weights runs with iw_l2_lambda_0 and the DOP / DRT ratio kept (drt1d.py:640-646)
    const double dop_l2 = p->prepared ? p->desc.dop_l2_lambda_0 : 0.0;
    GramL2 g = plan_l2(p, p->opts.iw_l2_lambda_0, p->opts.derivative_weights,
                       dop_l2 / p->opts.l2_lambda_0 * p->opts.iw_l2_lambda_0);
    const long long astr = p->rm_stride;
    const bool shared_rm = astr == 0;

    QpArgs qa{};
    qa.B = B; qa.n = n; qa.ldp = p->ldp; qa.q = p->q.d(); qa.h = p->h.d(); qa.h_stride = 0;
    qa.L = p->L.d(); qa.ldl = p->ldl;
    p->qp_layout(B, qa);
    qa.x = p->x.d(); qa.iters = p->qp_iters.i(); qa.pcost = p->pcost.d(); qa.status = p->qp_status.i();
    qa.iters_accum = p->qp_iters_total.i(); qa.opts = p->opts.qp;
    qa.state = p->qpstate.d(); qa.state_ld = qp_state_ld(n); qa.state_stride = (long long)qp_state_doubles(n);

    // ---- initialize_weights (qphb.py:1609-1681): one un-weighted, weakly penalised QP; P is the same for
    //      every spectrum (weights = 1, s = s_0, rho = rho_0), only q differs -------------------------------
    tm.mark(1);
    double* const Prow = nullptr;             // the QP reads P through its packed tile copy only (Ppk)
    const long long pstr = (long long)n * p->ldp, pkstr = (long long)qp_ppk_doubles(n);
    const int nc = p->prepared ? p->desc.num_chrono : 0;
    const bool separately = p->prepared && p->desc.init_weights_separately && nc > 0 && nc < m;
    HIPDRT_REQUIRE(!(separately && p->opts.outlier_p > 0.0), "init_weights_separately with outlier_p is not built");
    if (separately) {
        // drt1d.py:648-672: initialize_weights once for the chrono rows and once for the impedance rows.  A QP that sees
        // only one block = unit weights on its rows and zero on the others (the zero rows add exact zeros to P and q)
        qa.P = Prow; qa.p_stride = shared_rm ? 0 : pstr; qa.active = nullptr;
        qa.Ppk = p->Ppk.d(); qa.ppk_stride = shared_rm ? 0 : pkstr; qa.nchp = qp_nchp(n);
        if (p->h_init.p) qa.h = p->h_init.d();
        const int bounds[3] = {0, nc, m};
        for (int blk = 0; blk < 2; ++blk) {
            tm.mark(1);
            launch_row_mask(st, B, m, bounds[blk], bounds[blk + 1], p->w.d());
            launch_gram_l2(st, shared_rm ? 1 : B, m, n, p->rm.d(), p->ldrm, p->w.d(), g, Prow, p->ldp, shared_rm ? 0 : pstr,
                           nullptr, p->Ppk.d(), shared_rm ? 0 : pkstr, qp_nchp(n), astr);
            launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, p->w.d(), p->rv.d(), nullptr, p->opts.iw_l1_lambda_0, p->q.d(),
                        nullptr, astr);
            LAUNCH_OK();
            tm.mark(2);
            TRY(launch_qp(st, qa));
            tm.mark(3);
            TRY(launch_init_weights(st, fs, B, 2, bounds[blk], bounds[blk + 1]));
            LAUNCH_OK();
        }
        TRY(launch_init_weights(st, fs, B, 3));
        LAUNCH_OK();
    } else {
    // one P for the whole batch when the response matrix is shared, else one per measurement
    launch_gram_l2(st, shared_rm ? 1 : B, m, n, p->rm.d(), p->ldrm, p->w.d(), g, Prow, p->ldp, shared_rm ? 0 : pstr, nullptr,
                   p->Ppk.d(), shared_rm ? 0 : pkstr, qp_nchp(n), astr);
    launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, p->w.d(), p->rv.d(), nullptr, p->opts.iw_l1_lambda_0, p->q.d(), nullptr,
                astr);
    LAUNCH_OK();
    tm.mark(2);
    qa.P = Prow; qa.p_stride = shared_rm ? 0 : pstr; qa.active = nullptr;
    qa.Ppk = p->Ppk.d(); qa.ppk_stride = shared_rm ? 0 : pkstr; qa.nchp = qp_nchp(n);
    if (p->h_init.p) qa.h = p->h_init.d();          // initialize_weights' own constraint vector
    TRY(launch_qp(st, qa));
    tm.mark(3);
    if (p->opts.outlier_p > 0.0) {
        // qphb.py:1629-1656: weights from the first overfit with outlier down-weighting (variance matrix without each
        // point's own residual), a second ridge QP weighted by them (per-spectrum P now), weights again
        TRY(launch_init_weights(st, fs, B, 0));
        LAUNCH_OK();
        tm.mark(1);
        launch_gram_l2(st, B, m, n, p->rm.d(), p->ldrm, p->est_w.d(), g, Prow, p->ldp, (long long)n * p->ldp, nullptr,
                       p->Ppk.d(), (long long)qp_ppk_doubles(n), qp_nchp(n), astr);
        launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, p->est_w.d(), p->rv.d(), nullptr, p->opts.iw_l1_lambda_0, p->q.d(),
                    nullptr, astr);
        LAUNCH_OK();
        tm.mark(2);
        qa.p_stride = (long long)n * p->ldp; qa.ppk_stride = (long long)qp_ppk_doubles(n);
        TRY(launch_qp(st, qa));
        tm.mark(3);
    }
    TRY(launch_init_weights(st, fs, B, 1));
    LAUNCH_OK();
    }
    if (p->prepared && p->desc.weight_method == 1 && nc > 0 && nc < m) {
        // hybrid_weight_factor_method='weight' (drt1d.py:748-760): per-measurement row factors from the initial weights
        if (!p->w_eff.p) HIPDRT_CHECK(p->w_eff.alloc((size_t)p->capacity * m * sizeof(double)));
        if (p->wrow.bytes < (size_t)p->capacity * m * sizeof(double)) HIPDRT_CHECK(p->wrow.alloc((size_t)p->capacity * m * sizeof(double)));
        if (!p->wfac.p) HIPDRT_CHECK(p->wfac.alloc((size_t)p->capacity * 2 * sizeof(double)));
        p->wrow_batched = 1;
        launch_weight_method(st, fs, B, p->desc.fixed_chrono_factor, p->desc.fixed_eis_factor, p->wrow.d(), p->wfac.d());
        LAUNCH_OK();
    }

    // ---- outer loop (drt1d.py:877-988) ----------------------------------------------------------------------
    g = plan_l2(p, p->opts.l2_lambda_0, p->opts.derivative_weights, dop_l2);
    qa.h = p->h.d();
    qa.p_stride = (long long)n * p->ldp; qa.active = p->active.i();
    qa.ppk_stride = (long long)qp_ppk_doubles(n);
    int it = 0;
    for (; it < p->opts.max_iter; ++it) {
        tm.mark(1);
        HIPDRT_CHECK(hipMemsetAsync(p->n_active.p, 0, sizeof(int), st));
        const double* wq = p->w.d();
        if (p->has_weight_factors()) {           // drt1d.py:889-901: row factors every iteration, weight_factor from the second
            launch_scale_rows(st, B, m, p->w.d(), (p->wrow_late && it == 0) ? nullptr : p->wrow.d(), p->wrow_batched,
                              it > 0 ? p->weight_factor : 1.0, p->active.i(), p->w_eff.d());
            wq = p->w_eff.d();
        }
        launch_gram_l2(st, B, m, n, p->rm.d(), p->ldrm, wq, g, Prow, p->ldp, (long long)n * p->ldp, p->active.i(),
                       p->Ppk.d(), (long long)qp_ppk_doubles(n), qp_nchp(n), astr);
        launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, wq, p->rv.d(), p->l1.d(), 0.0, p->q.d(), p->active.i(), astr);
        LAUNCH_OK();
        tm.mark(2);
        if (B * sizeof(int) <= 48 * 1024) {     // dispatch order from the previous QP's iteration counts
            launch_lpt_order(st, B, p->qp_iters.i(), p->active.i(), p->order.i());
            qa.order = p->order.i();
        }
        TRY(launch_qp(st, qa));
        tm.mark(3);
        TRY(plan_hyper(p, st, fs, B, it));
        LAUNCH_OK();
        int n_active = 0;
        HIPDRT_CHECK(hipMemcpyAsync(&n_active, p->n_active.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPDRT_CHECK(hipStreamSynchronize(st));
        if (n_active == 0) break;
    }
    // ---- calculate_pq's q with the final weights (qphb.py:1154-1183) ---------------------------------------
    tm.mark(4);
    const double* wfin = p->w.d();
    if (p->has_weight_factors()) {
        // drt1d.py:990-1000: weights *= weight_factor (these are `true_weights`); calculate_pq sees them times the row factors
        if (p->wrow_late) {      // vector weight_factor: part of the weights themselves, no separate "scaled" weights
            launch_scale_rows(st, B, m, p->w.d(), p->wrow.d(), p->wrow_batched, p->weight_factor, nullptr, p->w.d());
            launch_scale_rows(st, B, m, p->w.d(), nullptr, 0, 1.0, nullptr, p->w_eff.d());
        } else {
            launch_scale_rows(st, B, m, p->w.d(), nullptr, 0, p->weight_factor, nullptr, p->w.d());
            launch_scale_rows(st, B, m, p->w.d(), p->wrow.d(), p->wrow_batched, 1.0, nullptr, p->w_eff.d());
        }
        wfin = p->w_eff.d();
    }
    launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, wfin, p->rv.d(), p->l1.d(), 0.0, p->q.d(), nullptr, astr);
    LAUNCH_OK();
    tm.mark(-1);
    HIPDRT_CHECK(hipStreamSynchronize(st));
    tm.collect(p->t_ms, p->launches);
    return HIPDRT_OK;
}

// ---- sub-batches ---------------------------------------------------------------------------------------------------------
// Spectra finish after 4 ... 50 outer iterations, so the tail of ONE batch's launch sequence leaves most CUs idle, and between
// two kernels of a sequence the device waits for the host's "anyone still active?" read-back.  Several sequences side by side
// fill both gaps.  bench.py / mapping.fit_observations(inflight=k) do that with k plans (k x the memory, k host threads of the
// caller); here the SAME effect comes from inside one plan: its staged batch is cut into contiguous ranges, every range is
// fitted by plan_fit_one on a view whose buffers are windows into the plan's own (nothing is allocated per range but a
// stream and a 4-byte counter), each on its own stream and worker thread, and the call returns when all are done.  Every
// kernel of the loop works per spectrum (reductions included), so a spectrum's result does not depend on which range it is in:
// bit-identical to the un-split fit as long as both use the same coneqp kernel (ranges of more than #CUs / 16 spectra).
static int subbatch_count(const hipdrt_plan* p) {
    if (p->prepared || p->hist_b >= 0 || p->has_weight_factors() || p->opts.outlier_p > 0.0 || p->qp_G != 0) return 1;
    // measured on one MI355X (profiles/r04_subbatch_sweep.txt): ranges below ~300 spectra lose to launch-wave quantisation
    // (fits/s with k = 1 / 2 / 3 / 4 ranges: 1024 spectra 1902 / 2110 / 2106 / 1660, 1250: 2001 / 2205 / 2219 / 1796, 2500: 2229 / 2375 / 2408 / 2104).
    // Round 5: TWO ranges from 600 spectra on, never three.  The kernel trace says why k = 3 and k = 4 lose (tools/trace_ranges.sh,
    // profiles/r05w_trace_ranges_1250.txt): the runtime maps streams onto 4 hardware queues by default, the ranges' streams landed
    // on TWO of them -- with k = 3 one queue carries two ranges' launch sequences one behind the other (102 coneqp launches
    // against 51 on the other queue), with k = 4 two each, and never more than two coneqp launches run at a time.  With
    // GPU_MAX_HW_QUEUES=8 in the process environment every range has its own queue and k = 2 / 3 / 4 measure 2285 / 2281 / 2330
    // at 1250 spectra (profiles/r05x_ab_hw_queues.txt) -- the library cannot set that for its host (it is read when the HIP
    // runtime starts), so it keeps the choice that is right with either setting.
    // Round 6: the ranges run on the library's own streams, picked per fit by activity and compute pipe (StreamPool above), so every
    // range has a queue and a pipe to itself whatever else the process has created (profiles/r06_trace_queue_placement.txt: 2254 ...
    // 2324 fits/s in all placements tried, against 1778 with two ranges on one queue and 2205 with two on one pipe), and FOUR ranges
    // from 1000 spectra on are the best cut (profiles/r06_subbatch_sweep.txt, k = 1 / 2 / 3 / 4 / 6: 1024 spectra 2048 / 2262 / 2261 /
    // 2320 / 2214, 1250: 2150 / 2368 / 2351 / 2422 / 2317, 2500: 2392 / 2556 / 2587 / 2600 / 2524; six lose: four pipes) -- as many
    // as the pool has streams: three under the runtime's default of 4 hardware queues, four with GPU_MAX_HW_QUEUES >= 5 (the host
    // layer's loader exports 8 unless its caller has set the variable).
    const int k_auto = p->B >= 1000 ? std::min(4, pool_size(p->ctx->device)) : (p->B >= 600 ? 2 : 1);
    int k = p->subbatches >= 1 ? std::min(p->subbatches, std::max(1, p->B / 64)) : k_auto;
    // the promise is "the bits of the un-split fit": the whole batch AND the smallest range must choose the batch coneqp kernel as
    // the views will see it (qp_layout runs qp_group_size on the view's own count with the context's current override, which may
    // have been set after the plan was allocated) -- otherwise fewer ranges, down to one
    const int force = p->ctx ? p->ctx->qp_force_group : -1;
    if (qp_group_size(p->B, p->n, force) != 0) return 1;
    while (k > 1 && qp_group_size(p->B / k, p->n, force) != 0) --k;
    return k;
}

static int make_view(hipdrt_plan* p, hipdrt_subfit& sf, int idx, int b0, int nb) {
    hipdrt_plan& v = sf.view;
    sf.ctx.device = p->ctx->device; sf.ctx.num_cu = p->ctx->num_cu; sf.ctx.hbm_bytes = p->ctx->hbm_bytes; sf.ctx.arch = p->ctx->arch;
    sf.ctx.qp_force_group = p->ctx->qp_force_group;
    sf.ctx.zero_shortcuts = p->ctx->zero_shortcuts;
    sf.ctx.qp_waves = p->ctx->qp_waves;
    v.ctx = &sf.ctx;
    v.nf = p->nf; v.ntau = p->ntau; v.n = p->n; v.m = p->m; v.ns = p->ns; v.ngrid = p->ngrid; v.ny = p->ny; v.mode = p->mode;
    v.toeplitz_a = p->toeplitz_a; v.toeplitz_m = p->toeplitz_m; v.toep_maxd = p->toep_maxd; v.spec_zero = p->spec_zero; v.idx_rinf = p->idx_rinf; v.idx_induc = p->idx_induc;
    v.ldrm = p->ldrm; v.ldm = p->ldm; v.ldp = p->ldp; v.ldl = p->ldl; v.eps = p->eps; v.opts = p->opts;
    v.capacity = nb; v.B = nb; v.qp_G = p->qp_G; v.subbatches = 1;
    // shared, read-only in the loop
    auto whole = [](DevBuf& d, const DevBuf& s_) { d.alias(s_, 0, s_.bytes); };
    whole(v.freq, p->freq); whole(v.tau, p->tau); whole(v.ln_tau, p->ln_tau); whole(v.wt_re, p->wt_re); whole(v.wt_im, p->wt_im);
    whole(v.lut6, p->lut6); whole(v.a_re, p->a_re); whole(v.a_im, p->a_im); whole(v.cr, p->cr); whole(v.rm, p->rm);
    for (int k = 0; k < 3; ++k) whole(v.mk[k], p->mk[k]);
    whole(v.vmm, p->vmm); whole(v.h, p->h); whole(v.l1, p->l1); whole(v.h_init, p->h_init); whole(v.vmm_base, p->vmm_base);
    whole(v.Ptmp, p->Ptmp);
    // per spectrum: `per` bytes each
    auto rows = [&](DevBuf& d, const DevBuf& s_, size_t per) { d.alias(s_, (size_t)b0 * per, (size_t)nb * per); };
    const size_t D = sizeof(double), I = sizeof(int), nn = (size_t)p->n, mm = (size_t)p->m;
    rows(v.z_re, p->z_re, p->nf * D); rows(v.z_im, p->z_im, p->nf * D);
    rows(v.rv, p->rv, mm * D); rows(v.w, p->w, mm * D); rows(v.est_w, p->est_w, mm * D);
    rows(v.x, p->x, nn * D); rows(v.x_in, p->x_in, nn * D); rows(v.q, p->q, nn * D); rows(v.s, p->s, 3 * nn * D);
    rows(v.rho, p->rho, 3 * D); rows(v.xmx, p->xmx, 3 * D); rows(v.coef_scale, p->coef_scale, D); rows(v.var_floor, p->var_floor, D);
    rows(v.pcost, p->pcost, D);
    rows(v.active, p->active, I); rows(v.outer_iters, p->outer_iters, I); rows(v.fit_status, p->fit_status, I);
    rows(v.qp_iters_total, p->qp_iters_total, I); rows(v.qp_status, p->qp_status, I); rows(v.qp_iters, p->qp_iters, I);
    rows(v.order, p->order, I);
    rows(v.L, p->L, qp_scratch_doubles(p->n, p->qp_G) * D);
    rows(v.qpstate, p->qpstate, (size_t)(p->qp_G > 1 ? p->qp_G : 1) * qp_state_doubles(p->n) * D);
    rows(v.gsync, p->gsync, qp_gsync_ints() * I);
    rows(v.Ppk, p->Ppk, qp_ppk_doubles(p->n) * D);
    rows(v.premv, p->premv, 3 * mm * D);                   // [3][nb][m] of this range (plan_hyper: batched products)
    v.n_active.alias(p->n_active_sub, (size_t)idx * I, I);
    return HIPDRT_OK;
}

// device bytes one more staged spectrum costs an EIS plan (the per-spectrum buffers plan_alloc and hipdrt_plan_fit size by the
// capacity): what a map driver divides the device's memory by before it forms its batches
int hipdrt_plan_bytes_per_spectrum(int nf, int ntau, int ns, long long* bytes) try {
    HIPDRT_REQUIRE(bytes && nf >= 1 && ntau >= 1 && ns >= 0, "arguments");
    const size_t n = (size_t)ntau + ns, m = 2 * (size_t)nf, D = sizeof(double), I = sizeof(int);
    size_t b = 0;
    b += 2 * (size_t)nf * D;                           // z_re, z_im
    b += 3 * m * D + 3 * m * D;                        // rv, w, est_w; the three batched products of the hyper phase
    b += 3 * n * D + 3 * n * D;                        // x, x_in, q; s
    b += (3 + 3 + 1 + 1 + 1) * D + 8 * I;              // rho, xmx, scales, cost; flags and counters
    b += (qp_scratch_doubles((int)n, 0) + qp_state_doubles((int)n) + qp_ppk_doubles((int)n)) * D + qp_gsync_ints() * I;
    *bytes = (long long)b;
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_set_subbatches(hipdrt_plan* p, int k) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(k >= 0 && k <= 16, "0 (automatic) <= k <= 16");
    p->subbatches = k;
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_fit(hipdrt_plan* p) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(p->B >= 1, "no spectra staged (call hipdrt_plan_upload)");
    const int k = subbatch_count(p);
    if (k <= 1) {
        LoopOnContextStream busy(p->ctx);
        return plan_fit_one(p);
    }
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    HIPDRT_CHECK(hipStreamSynchronize(p->ctx->stream));       // whatever staged the batch is done
    if (p->n_active_sub.bytes < (size_t)k * sizeof(int)) HIPDRT_CHECK(p->n_active_sub.alloc(16 * sizeof(int)));
    if (p->premv.bytes < 3 * (size_t)p->capacity * p->m * sizeof(double))      // the ranges' products buffers are windows of this one
        HIPDRT_CHECK(p->premv.alloc(3 * (size_t)p->capacity * p->m * sizeof(double)));
    while ((int)p->subs.size() < k) p->subs.emplace_back(new hipdrt_subfit());
    const int B = p->B;
    for (int i = 0; i < k; ++i) {
        const int b0 = (int)((long long)B * i / k), b1 = (int)((long long)B * (i + 1) / k);
        TRY(make_view(p, *p->subs[i], i, b0, b1 - b0));
    }
    // the ranges' streams: borrowed from the library's pool for this fit, the least busy ones (the context's own may be among
    // them: it is idle until the ranges are done)
    struct Borrowed {
        int device, k; int idx[16]; hipStream_t st[16];
        Borrowed(int device_, int k_) : device(device_), k(k_) { pool_borrow(device, k, idx, st); }
        ~Borrowed() { pool_return(device, k, idx); }
    } streams(p->ctx->device, k);
    for (int i = 0; i < k; ++i) p->subs[i]->ctx.stream = streams.st[i];
    const auto t0 = std::chrono::steady_clock::now();
    // no exception may cross the C ABI, and a joinable std::thread must not be destroyed: ranges whose worker thread cannot
    // be created (std::system_error) are fitted right here, on the caller's thread, after the started ones were joined
    std::vector<std::thread> workers;
    workers.reserve(k);
    int started = 0;
    for (int i = 0; i < k; ++i) {
        hipdrt_subfit* sf = p->subs[i].get();
        sf->rc = HIPDRT_OK;
        try {
            workers.emplace_back([sf] {
                sf->rc = plan_fit_one(&sf->view);
                if (sf->rc) sf->err = hipdrt_last_error();
            });
            ++started;
        } catch (...) {
            break;
        }
    }
    for (auto& w : workers) w.join();
    for (int i = started; i < k; ++i) {
        hipdrt_subfit* sf = p->subs[i].get();
        sf->rc = plan_fit_one(&sf->view);
        if (sf->rc) sf->err = hipdrt_last_error();
    }
    const float wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < 5; ++i) { p->t_ms[i] = 0; p->launches[i] = 0; }
    for (int i = 0; i < k; ++i) {
        hipdrt_subfit* sf = p->subs[i].get();
        if (sf->rc) { set_error("sub-batch " + std::to_string(i) + ": " + sf->err); return sf->rc; }
        // phase times are HIP-event intervals on streams that share the GPU: summed over the ranges they exceed the wall time
        for (int c = 1; c < 5; ++c) { p->t_ms[c] += sf->view.t_ms[c]; p->launches[c] += sf->view.launches[c]; }
    }
    p->t_ms[0] = wall_ms; p->launches[0] = 1;
    p->prepped = 1;
    return HIPDRT_OK;
} HIPDRT_CATCH

static int plan_llh_terms(hipdrt_plan* p, double* rss, double* sum_log_w, int stored, double scalar_w = 1.0);

int hipdrt_plan_llh_terms(hipdrt_plan* p, double* rss, double* sum_log_w) { return plan_llh_terms(p, rss, sum_log_w, 0); }

int hipdrt_plan_obs_llh_terms(hipdrt_plan* p, double* rss, double* sum_log_w) { return plan_llh_terms(p, rss, sum_log_w, 1); }

int hipdrt_plan_obs_llh_terms_w(hipdrt_plan* p, int weights_mode, double scalar_weight, double* rss, double* sum_log_w) try {
    HIPDRT_REQUIRE(weights_mode == HIPDRT_LLH_W_EST || weights_mode == HIPDRT_LLH_W_UNIFORM || weights_mode == HIPDRT_LLH_W_SCALAR,
                   "weights_mode");
    HIPDRT_REQUIRE(weights_mode != HIPDRT_LLH_W_SCALAR || scalar_weight > 0.0, "scalar weight must be positive");
    return plan_llh_terms(p, rss, sum_log_w, weights_mode, scalar_weight);
} HIPDRT_CATCH

static int plan_llh_terms(hipdrt_plan* p, double* rss, double* sum_log_w, int stored, double scalar_w) {
    HIPDRT_REQUIRE(p && rss && sum_log_w, "NULL pointer");
    HIPDRT_REQUIRE(p->B >= 1, "no fitted batch in the plan");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const size_t bb = (size_t)p->B * sizeof(double);
    DevBuf d1, d2;
    HIPDRT_CHECK(d1.alloc(bb)); HIPDRT_CHECK(d2.alloc(bb));
    TRY(launch_llh(st, p->state(), p->B, d1.d(), d2.d(), stored, scalar_w));
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(rss, d1.p, bb, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(sum_log_w, d2.p, bb, hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
}

int hipdrt_plan_set_state(hipdrt_plan* p, const double* x, const double* rho, const double* s, const double* weights) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(p->B >= 1, "no fitted batch in the plan");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const size_t B = p->B, n = p->n, m = p->m;
    if (x) {
        HIPDRT_CHECK(hipMemcpyAsync(p->x.p, x, B * n * sizeof(double), hipMemcpyHostToDevice, st));
        HIPDRT_CHECK(hipMemcpyAsync(p->x_in.p, x, B * n * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (rho) HIPDRT_CHECK(hipMemcpyAsync(p->rho.p, rho, B * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    if (s) HIPDRT_CHECK(hipMemcpyAsync(p->s.p, s, B * 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
    if (weights) HIPDRT_CHECK(hipMemcpyAsync(p->w.p, weights, B * m * sizeof(double), hipMemcpyHostToDevice, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_set_state_dop(hipdrt_plan* p, const double* dop_rho) try {
    HIPDRT_REQUIRE(p && dop_rho, "NULL pointer");
    HIPDRT_REQUIRE(p->prepared && p->desc.dop_size > 0, "the plan has no distribution of phasances");
    HIPDRT_REQUIRE(p->B >= 1, "no fitted batch in the plan");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    HIPDRT_CHECK(hipMemcpyAsync(p->dop_rho.p, dop_rho, (size_t)p->B * 3 * sizeof(double), hipMemcpyHostToDevice, p->ctx->stream));
    HIPDRT_CHECK(hipStreamSynchronize(p->ctx->stream));
    return HIPDRT_OK;
} HIPDRT_CATCH

// QP arguments of the outer loop: one P per spectrum in the packed tile layout, the loop's constraint vector
static QpArgs loop_qp_args(hipdrt_plan* p, const hipdrt_qp_opts& qpo) {
    const int n = p->n;
    QpArgs qa{};
    qa.B = p->B; qa.n = n; qa.ldp = p->ldp; qa.q = p->q.d(); qa.h = p->h.d(); qa.h_stride = 0;
    qa.L = p->L.d(); qa.ldl = p->ldl;
    p->qp_layout(p->B, qa);
    qa.x = p->x.d(); qa.iters = p->qp_iters.i(); qa.pcost = p->pcost.d(); qa.status = p->qp_status.i();
    qa.iters_accum = p->qp_iters_total.i(); qa.opts = qpo;
    qa.state = p->qpstate.d(); qa.state_ld = qp_state_ld(n); qa.state_stride = (long long)qp_state_doubles(n);
    qa.P = nullptr; qa.p_stride = (long long)n * p->ldp; qa.active = p->active.i();
    qa.Ppk = p->Ppk.d(); qa.ppk_stride = (long long)qp_ppk_doubles(n); qa.nchp = qp_nchp(n);
    return qa;
}

// drt1d._continue_from_init (hybdrt/models/drt1d.py:1270-1365) for the fitted batch: the same outer loop re-entered from
// the state on the device (x, s, rho [, dop_rho], weights; est_weights, xmx / dop_xmx norms and data scale stay) with updated
// hyper-parameters.  Any data type: on prepared plans (chrono / joint fits, DOP) the plan's row factors -- chrono / eis weight
// factors, hipdrt_plan_set_weight_factors -- multiply the weights at the top of every iteration together with `weight_factor`
// (1314-1318), the DOP pass runs as in the fit, and the vz_offset column is rewritten after every iteration from a matrix
// whose offset column is FROZEN as this call found it (1295-1298, 1353-1357: the reference copies rm at entry, zeroing the
// baseline columns only; the fit itself copied while the column was still zero).  The plan's scalar weight_factor is not used.
int hipdrt_plan_continue(hipdrt_plan* p, const hipdrt_fit_opts* opts, double weight_factor, int min_iter) try {
    HIPDRT_REQUIRE(p && opts, "NULL pointer");
    HIPDRT_REQUIRE(p->B >= 1, "no fitted batch in the plan");
    HIPDRT_REQUIRE(opts->max_iter >= 1 && min_iter >= 1, "max_iter, min_iter >= 1");
    LoopOnContextStream busy(p->ctx);
    // rejected calls must leave the finished fit as it is: every check comes before the first write
    HIPDRT_REQUIRE(p->prepared || !p->has_weight_factors(),
                   "warm restarts take their weight_factor argument; clear the plan's weight factors");
    HIPDRT_REQUIRE(!(p->wrow.p && p->wrow_late), "a vector-valued weight_factor belongs to the fit, not to its warm restarts");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int B = p->B, n = p->n, m = p->m;
    TRY(plan_hist_reserve(p, opts->max_iter));
    // the row factors may be new with this call (hipdrt_plan_set_weight_factors after the fit): the buffer of the scaled final
    // weights is filled behind the loop, below
    const bool rowfac = p->prepared && p->wrow.p;
    if (rowfac && !p->w_eff.p) HIPDRT_CHECK(p->w_eff.alloc((size_t)p->capacity * m * sizeof(double)));
    // outlier_p (qphb.py:1545-1594: estimate_weights forms outlier_t and the T V T matrix anew from every iterate, what
    // _continue_from_init is handed is never read, drt1d.py:1300-1304): only the record of 1 - outlier probability needs room
    if (opts->outlier_p > 0.0 && !p->outlier_t.p) HIPDRT_CHECK(p->outlier_t.alloc((size_t)p->capacity * m * sizeof(double)));
    FitState fs = p->state();
    fs.opts = *opts; fs.continue_mode = 1; fs.min_iter = min_iter;
    const long long astr = p->rm_stride;
    if (p->prepared && p->desc.vz_index >= 0) {
        if (!p->vz_entry.p) HIPDRT_CHECK(p->vz_entry.alloc((size_t)p->capacity * m * sizeof(double)));
        launch_copy_column(st, B, m, p->rm.d(), astr, p->ldrm, p->desc.vz_index, p->vz_entry.d());
        LAUNCH_OK();
        fs.vz_entry = p->vz_entry.d();
    }
    PhaseTimer tm(st);
    tm.mark(4);
    if (p->hist_b >= 0) HIPDRT_CHECK(hipMemsetAsync(p->hist_rows.p, 0, sizeof(int), st));
    {   // every spectrum takes part again; QP iteration totals restart
        std::vector<int> ones(B, 1);
        HIPDRT_CHECK(hipMemcpyAsync(p->active.p, ones.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
        HIPDRT_CHECK(hipMemsetAsync(p->qp_iters_total.p, 0, (size_t)B * sizeof(int), st));
        HIPDRT_CHECK(hipStreamSynchronize(st));
    }
    GramL2 g = plan_l2(p, opts->l2_lambda_0, opts->derivative_weights, p->prepared ? p->desc.dop_l2_lambda_0 : 0.0);
    QpArgs qa = loop_qp_args(p, opts->qp);
    double* const Prow = nullptr;
    for (int it = 0; it < opts->max_iter; ++it) {
        tm.mark(1);
        HIPDRT_CHECK(hipMemsetAsync(p->n_active.p, 0, sizeof(int), st));
        // in place, like the reference's `weights[:num_chrono] *= ...; weights = weights * weight_factor`: the hyper step
        // replaces the weights with a fresh estimate afterwards
        if (p->prepared && p->wrow.p)
            launch_scale_rows(st, B, m, p->w.d(), p->wrow.d(), p->wrow_batched, weight_factor, p->active.i(), p->w.d());
        else if (weight_factor != 1.0) launch_scale_weights(st, fs, B, weight_factor);
        launch_gram_l2(st, B, m, n, p->rm.d(), p->ldrm, p->w.d(), g, Prow, p->ldp, (long long)n * p->ldp, p->active.i(),
                       p->Ppk.d(), (long long)qp_ppk_doubles(n), qp_nchp(n), astr);
        launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, p->w.d(), p->rv.d(), p->l1.d(), 0.0, p->q.d(), p->active.i(), astr);
        LAUNCH_OK();
        tm.mark(2);
        if (B * sizeof(int) <= 48 * 1024) {
            launch_lpt_order(st, B, p->qp_iters.i(), p->active.i(), p->order.i());
            qa.order = p->order.i();
        }
        TRY(launch_qp(st, qa));
        tm.mark(3);
        TRY(plan_hyper(p, st, fs, B, it));
        LAUNCH_OK();
        int n_active = 0;
        HIPDRT_CHECK(hipMemcpyAsync(&n_active, p->n_active.p, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPDRT_CHECK(hipStreamSynchronize(st));
        if (n_active == 0) break;
    }
    tm.mark(4);
    // What the posterior entry points call "the final P" (hipdrt_plan_p_matrix, _param_cov, _distribution_cov, _param_var read
    // w_eff whenever the plan has weight factors): the weights this restart ended with -- the last iteration's fresh estimate --
    // times the factors its QPs saw, i.e. the matrix the NEXT iteration would have solved with, and q to match.  (The fit
    // leaves true_weights x row factors there, drt1d.py:990-1006; left alone, w_eff would still hold the FIRST fit's scaled
    // weights, or nothing at all when the factors came with this call.)
    const double* wfin = p->w.d();
    if (rowfac) {
        launch_scale_rows(st, B, m, p->w.d(), p->wrow.d(), p->wrow_batched, weight_factor, nullptr, p->w_eff.d());
        wfin = p->w_eff.d();
    }
    launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, wfin, p->rv.d(), p->l1.d(), 0.0, p->q.d(), nullptr, astr);
    LAUNCH_OK();
    tm.mark(-1);
    HIPDRT_CHECK(hipStreamSynchronize(st));
    tm.collect(p->t_ms, p->launches);
    return HIPDRT_OK;
} HIPDRT_CATCH

// qphb.iterate_qphb (hybdrt/models/qphb.py:606-972) for every staged measurement of a prepared plan: the QP on
// (weights, s_vectors, rho) as given, then the s / rho / DOP hyper-parameter pass, estimate_weights and is_converged
// against x_in.  What _qphb_fit_core does around the call (xmx norms of the first iteration, data rescaling, the
// vz_offset column; drt1d.py:903-979) is not part of it.
int hipdrt_plan_iterate(hipdrt_plan* p, const hipdrt_iterate_state* in, int* converged, int* qp_status, int* qp_iters,
                        double* primal_objective) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(p->prepared, "hipdrt_plan_iterate works on prepared plans (the caller's rm, rv as iterate_qphb takes them)");
    HIPDRT_REQUIRE(p->B >= 1, "no measurements staged (call hipdrt_plan_upload_prepared)");
    HIPDRT_REQUIRE(!p->has_weight_factors(), "weight factors belong to _qphb_fit_core, not to iterate_qphb");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int B = p->B, n = p->n, m = p->m;
    FitState fs = p->state();
    fs.continue_mode = 2; fs.min_iter = 1;
    fs.opts.max_iter = 2;                      // never "stopped at max_iter": fit_status 0 <=> converged
    if (!p->prepped) {                         // variance floor of estimate_weights + default state (qphb.py:1569)
        launch_prep(st, fs, B);
        LAUNCH_OK();
        p->prepped = 1;
    }
    if (in) {
        const size_t b = (size_t)B;
        struct { const double* src; void* dst; size_t cnt; } cp[] = {
            {in->x_in, p->x_in.p, b * n}, {in->x_in, p->x.p, b * n}, {in->s_vectors, p->s.p, b * 3 * n},
            {in->rho, p->rho.p, b * 3}, {in->dop_rho, p->dop_rho.p, b * 3}, {in->weights, p->w.p, b * m},
            {in->est_weights, p->est_w.p, b * m}, {in->xmx_norms, p->xmx.p, b * 3},
            {in->dop_xmx_norms, p->dop_xmx.p, b * 3}};
        for (auto& c : cp)
            if (c.src) HIPDRT_CHECK(hipMemcpyAsync(c.dst, c.src, c.cnt * sizeof(double), hipMemcpyHostToDevice, st));
    }
    PhaseTimer tm(st);
    tm.mark(4);
    {
        std::vector<int> ones(B, 1);
        HIPDRT_CHECK(hipMemcpyAsync(p->active.p, ones.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
        HIPDRT_CHECK(hipMemsetAsync(p->n_active.p, 0, sizeof(int), st));
        HIPDRT_CHECK(hipStreamSynchronize(st));         // `ones` and the caller's arrays may go once this returns
    }
    const GramL2 g = plan_l2(p, p->opts.l2_lambda_0, p->opts.derivative_weights, p->desc.dop_l2_lambda_0);
    QpArgs qa = loop_qp_args(p, p->opts.qp);
    tm.mark(1);
    launch_gram_l2(st, B, m, n, p->rm.d(), p->ldrm, p->w.d(), g, nullptr, p->ldp, (long long)n * p->ldp, p->active.i(),
                   p->Ppk.d(), (long long)qp_ppk_doubles(n), qp_nchp(n), p->rm_stride);
    launch_qvec(st, B, m, n, p->rm.d(), p->ldrm, p->w.d(), p->rv.d(), p->l1.d(), 0.0, p->q.d(), p->active.i(), p->rm_stride);
    LAUNCH_OK();
    tm.mark(2);
    TRY(launch_qp(st, qa));
    tm.mark(3);
    TRY(plan_hyper(p, st, fs, B, 0));
    LAUNCH_OK();
    tm.mark(-1);
    std::vector<int> act(B);
    HIPDRT_CHECK(hipMemcpyAsync(act.data(), p->active.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (qp_status) HIPDRT_CHECK(hipMemcpyAsync(qp_status, p->qp_status.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (qp_iters) HIPDRT_CHECK(hipMemcpyAsync(qp_iters, p->qp_iters.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (primal_objective)
        HIPDRT_CHECK(hipMemcpyAsync(primal_objective, p->pcost.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    if (converged) for (int b = 0; b < B; ++b) converged[b] = act[b] == 0;
    tm.collect(p->t_ms, p->launches);
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_timings(hipdrt_plan* p, float* t, int* launches) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    if (t) std::memcpy(t, p->t_ms, sizeof(p->t_ms));
    if (launches) std::memcpy(launches, p->launches, sizeof(p->launches));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_download(hipdrt_plan* p, double* x, double* fit_x, double* r_inf, double* induc, double* weights,
                         double* coef_scale, double* rho, double* s_vectors, double* q_vector, int* outer_iters,
                         int* qp_iters_total, int* status) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_REQUIRE(p->B >= 1, "nothing fitted");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    const int B = p->B, n = p->n, m = p->m, ns = p->ns, ntau = p->ntau;
    std::vector<double> hx((size_t)B * n), hcs(B);
    HIPDRT_CHECK(hipMemcpy(hx.data(), p->x.p, hx.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPDRT_CHECK(hipMemcpy(hcs.data(), p->coef_scale.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost));
    if (x) std::memcpy(x, hx.data(), hx.size() * sizeof(double));
    if (coef_scale) std::memcpy(coef_scale, hcs.data(), (size_t)B * sizeof(double));
    // extract_qphb_parameters (drt1d.py:6228-6289)
    for (int b = 0; b < B; ++b) {
        const double cs = hcs[b];
        const double* xb = hx.data() + (size_t)b * n;
        if (fit_x) for (int i = 0; i < ntau; ++i) fit_x[(size_t)b * ntau + i] = xb[ns + i] * cs;
        if (r_inf) r_inf[b] = p->idx_rinf >= 0 ? xb[p->idx_rinf] * cs : 0.0;
        if (induc) induc[b] = p->idx_induc >= 0 ? xb[p->idx_induc] * (cs * p->opts.inductance_scale) : 0.0;
    }
    if (weights) HIPDRT_CHECK(hipMemcpy(weights, p->w.p, (size_t)B * m * sizeof(double), hipMemcpyDeviceToHost));
    if (rho) HIPDRT_CHECK(hipMemcpy(rho, p->rho.p, (size_t)B * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (s_vectors) HIPDRT_CHECK(hipMemcpy(s_vectors, p->s.p, (size_t)B * 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    if (q_vector) HIPDRT_CHECK(hipMemcpy(q_vector, p->q.p, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost));
    if (outer_iters) HIPDRT_CHECK(hipMemcpy(outer_iters, p->outer_iters.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
    if (qp_iters_total) HIPDRT_CHECK(hipMemcpy(qp_iters_total, p->qp_iters_total.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
    if (status) HIPDRT_CHECK(hipMemcpy(status, p->fit_status.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_plan_get_p_matrix(hipdrt_plan* p, int b, double* out) try {
    HIPDRT_REQUIRE(p && out, "NULL pointer");
    HIPDRT_REQUIRE(b >= 0 && b < p->B, "spectrum index out of range");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int n = p->n, m = p->m;
    GramL2 g = plan_l2(p, p->opts.l2_lambda_0, p->opts.derivative_weights, p->prepared ? p->desc.dop_l2_lambda_0 : 0.0);
    g.s = p->s.d() + (size_t)b * 3 * n; g.rho = p->rho.d() + (size_t)b * 3;
    if (g.dop_size > 0) g.dop_rho = p->dop_rho.d() + (size_t)b * 3;
    const double* wfin = p->has_weight_factors() ? p->w_eff.d() : p->w.d();     // scaled_weights of calculate_pq
    launch_gram_l2(st, 1, m, n, p->rm.d() + (size_t)b * p->rm_stride, p->ldrm, wfin + (size_t)b * m, g, p->Ptmp.d(),
                   p->ldp, 0, nullptr);
    LAUNCH_OK();
    return copy_strided(out, p->Ptmp.d(), n, n, p->ldp, st);
} HIPDRT_CATCH

// out[b][i] = rows_i' P_b^-1 rows_i * cs_b^2 for the fitted batch; rows[nrow][ncol] sits at columns col_offset.. of the
// unknown vector (zero elsewhere)
static int plan_quadratic_forms(hipdrt_plan* p, const double* basis_eval, int neval, int ncol, int col_offset, double* out,
                                int* status) {
    HIPDRT_REQUIRE(p->B > 0, "no fitted batch in the plan");
    HIPDRT_REQUIRE(neval >= 1, "neval >= 1");
    HIPDRT_REQUIRE(p->n <= 4096, "posterior variance: n <= 4096");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int n = p->n, m = p->m, B = p->B;
    const int nex = (neval + 15) / 16, nchp = qp_nchp(n);
    // final P of every spectrum (calculate_pq with the final weights / s / rho, drt1d.py:1006), packed tiles only
    GramL2 g = plan_l2(p, p->opts.l2_lambda_0, p->opts.derivative_weights, p->prepared ? p->desc.dop_l2_lambda_0 : 0.0);
    launch_gram_l2(st, B, m, n, p->rm.d(), p->ldrm, p->has_weight_factors() ? p->w_eff.d() : p->w.d(), g, nullptr, p->ldp, 0,
                   nullptr, p->Ppk.d(), (long long)qp_ppk_doubles(n), nchp, p->rm_stride);
    LAUNCH_OK();
    // evaluation rows -> packed tiles, shifted past the special-parameter slots
    DevBuf dbe, bex, scratch, dout, dstat;
    TRY(upload(dbe, basis_eval, (size_t)neval * ncol * sizeof(double), st));
    HIPDRT_CHECK(bex.alloc((size_t)nex * nchp * 256 * sizeof(double)));
    launch_pack_rows(st, neval, ncol, col_offset, dbe.d(), ncol, nex, bex.d(), nchp);
    LAUNCH_OK();
    const int chunk = B < 256 ? B : 256;
    const size_t lsz = dist_var_scratch_doubles(n, nex);
    HIPDRT_CHECK(scratch.alloc((size_t)chunk * lsz * sizeof(double)));
    HIPDRT_CHECK(dout.alloc((size_t)B * nex * 16 * sizeof(double)));
    HIPDRT_CHECK(dstat.alloc((size_t)B * sizeof(int)));
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0) < chunk ? (B - b0) : chunk;
        TRY(launch_dist_var(st, nb, n, p->Ppk.d() + (size_t)b0 * qp_ppk_doubles(n), (long long)qp_ppk_doubles(n), bex.d(),
                            nex, scratch.d(), (long long)lsz, dout.d() + (size_t)b0 * nex * 16, (long long)nex * 16,
                            dstat.i() + b0));
    }
    std::vector<double> hv((size_t)B * nex * 16), cs(B);
    std::vector<int> hs(B);
    HIPDRT_CHECK(hipMemcpyAsync(hv.data(), dout.p, hv.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(cs.data(), p->coef_scale.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(hs.data(), dstat.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    // estimate_param_cov scales the inverse by coefficient_scale^2 (drt1d.py:4133)
    for (int b = 0; b < B; ++b) {
        const double c2 = cs[b] * cs[b];
        for (int i = 0; i < neval; ++i) out[(size_t)b * neval + i] = hv[((size_t)b * nex) * 16 + i] * c2;
        if (status) status[b] = hs[b];
    }
    return HIPDRT_OK;
}

// out[neval][neval] = rows P_b^-1 rows' * cs_b^2 for ONE fitted spectrum: the variance kernel leaves Y = rows L^-T behind the
// factor (one more panel of the same factorisation), rows_outer_kernel forms Y Y'
static int plan_full_cov(hipdrt_plan* p, int b, const double* rows, int neval, int ncol, int col_offset, double* out, int* status) {
    HIPDRT_REQUIRE(p->B > 0, "no fitted batch in the plan");
    HIPDRT_REQUIRE(b >= 0 && b < p->B, "spectrum index out of range");
    HIPDRT_REQUIRE(neval >= 1, "neval >= 1");
    HIPDRT_REQUIRE(p->n <= 4096, "posterior covariance: n <= 4096");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    hipStream_t st = p->ctx->stream;
    const int n = p->n, m = p->m;
    const int nex = (neval + 15) / 16, nchp = qp_nchp(n), nch = round_up(n, 32) / 16;
    // final P of this spectrum (calculate_pq with the final weights / s / rho), packed tiles, into its own slot of Ppk
    GramL2 g = plan_l2(p, p->opts.l2_lambda_0, p->opts.derivative_weights, p->prepared ? p->desc.dop_l2_lambda_0 : 0.0);
    g.s = p->s.d() + (size_t)b * 3 * n; g.rho = p->rho.d() + (size_t)b * 3;
    if (g.dop_size > 0) g.dop_rho = p->dop_rho.d() + (size_t)b * 3;
    const double* wfin = p->has_weight_factors() ? p->w_eff.d() : p->w.d();
    double* ppk = p->Ppk.d() + (size_t)b * qp_ppk_doubles(n);
    launch_gram_l2(st, 1, m, n, p->rm.d() + (size_t)b * p->rm_stride, p->ldrm, wfin + (size_t)b * m, g, nullptr, p->ldp, 0,
                   nullptr, ppk, 0, nchp, 0);
    LAUNCH_OK();
    DevBuf dbe, bex, scratch, dvar, dstat, dcov;
    TRY(upload(dbe, rows, (size_t)neval * ncol * sizeof(double), st));
    HIPDRT_CHECK(bex.alloc((size_t)nex * nchp * 256 * sizeof(double)));
    launch_pack_rows(st, neval, ncol, col_offset, dbe.d(), ncol, nex, bex.d(), nchp);
    LAUNCH_OK();
    HIPDRT_CHECK(scratch.alloc(dist_var_scratch_doubles(n, nex) * sizeof(double)));
    HIPDRT_CHECK(dvar.alloc((size_t)nex * 16 * sizeof(double)));
    HIPDRT_CHECK(dstat.alloc(sizeof(int)));
    HIPDRT_CHECK(dcov.alloc((size_t)neval * neval * sizeof(double)));
    TRY(launch_dist_var(st, 1, n, ppk, 0, bex.d(), nex, scratch.d(), 0, dvar.d(), 0, dstat.i()));
    double cs = 1.0;
    int hs = 0;
    HIPDRT_CHECK(hipMemcpyAsync(&cs, p->coef_scale.d() + b, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipMemcpyAsync(&hs, dstat.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    if (status) *status = hs;
    if (hs != 0) {                               // P not positive definite (np.linalg.inv would still return something; the
        for (size_t i = 0; i < (size_t)neval * neval; ++i) out[i] = __builtin_nan("");      // reference warns and returns None)
        return HIPDRT_OK;
    }
    // Y = rows L^-T sits in tile rows nch .. nch + nex - 1 of the scratch; only the first ceil(n / 16) tile columns are non-zero
    launch_rows_outer(st, scratch.d() + (size_t)nch * nch * 256, nch, nch, nex, neval, cs * cs, dcov.d(), neval);
    LAUNCH_OK();
    HIPDRT_CHECK(hipMemcpyAsync(out, dcov.p, (size_t)neval * neval * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPDRT_CHECK(hipStreamSynchronize(st));
    return HIPDRT_OK;
}

int hipdrt_plan_distribution_cov(hipdrt_plan* p, int b, const double* basis_eval, int neval, double* out, int* status) try {
    HIPDRT_REQUIRE(p && basis_eval && out, "NULL pointer");
    return plan_full_cov(p, b, basis_eval, neval, p->ntau, p->ns, out, status);
} HIPDRT_CATCH

int hipdrt_plan_param_cov(hipdrt_plan* p, int b, double* out, int* status) try {
    HIPDRT_REQUIRE(p && out, "NULL pointer");
    const int n = p->n;
    std::vector<double> eye((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) eye[(size_t)i * n + i] = 1.0;
    return plan_full_cov(p, b, eye.data(), n, n, 0, out, status);
} HIPDRT_CATCH

int hipdrt_plan_distribution_var(hipdrt_plan* p, const double* basis_eval, int neval, double* out, int* status) try {
    HIPDRT_REQUIRE(p && basis_eval && out, "NULL pointer");
    return plan_quadratic_forms(p, basis_eval, neval, p->ntau, p->ns, out, status);
} HIPDRT_CATCH

int hipdrt_plan_param_var(hipdrt_plan* p, double* out, int* status) try {
    HIPDRT_REQUIRE(p && out, "NULL pointer");
    const int n = p->n;
    std::vector<double> eye((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) eye[(size_t)i * n + i] = 1.0;
    return plan_quadratic_forms(p, eye.data(), n, n, 0, out, status);
} HIPDRT_CATCH

// history buffers for `rows` outer iterations (grown when a later call asks for more than the first one did)
static int plan_hist_reserve(hipdrt_plan* p, int rows) {
    if (rows < 1) rows = 1;
    if (p->hist_b >= 0 && p->hist_cap < rows) {
        p->hist_cap = rows;
        HIPDRT_CHECK(p->hist_x.alloc((size_t)p->hist_cap * p->n * sizeof(double)));
        HIPDRT_CHECK(p->hist_w.alloc((size_t)p->hist_cap * p->m * sizeof(double)));
        HIPDRT_CHECK(p->hist_rho.alloc((size_t)p->hist_cap * 3 * sizeof(double)));
        HIPDRT_CHECK(p->hist_qp.alloc((size_t)(p->hist_cap + 1) * sizeof(int)));
        HIPDRT_CHECK(p->hist_dop_rho.alloc((size_t)p->hist_cap * 3 * sizeof(double)));
    }
    return HIPDRT_OK;
}

int hipdrt_plan_record_history(hipdrt_plan* p, int b) try {
    HIPDRT_REQUIRE(p, "plan is NULL");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    p->hist_b = b;
    return plan_hist_reserve(p, p->opts.max_iter);
} HIPDRT_CATCH

int hipdrt_plan_get_history(hipdrt_plan* p, double* hist_x, double* hist_rho, double* hist_w, int* qp_iters,
                            int max_rows, int* rows) try {
    HIPDRT_REQUIRE(p && rows, "NULL pointer");
    HIPDRT_REQUIRE(p->hist_b >= 0 && p->hist_cap > 0, "history recording was not enabled");
    HIPDRT_CHECK(hipSetDevice(p->ctx->device)); (void)hipGetLastError();
    int r = 0;
    HIPDRT_CHECK(hipMemcpy(&r, p->hist_rows.p, sizeof(int), hipMemcpyDeviceToHost));
    if (r > max_rows) r = max_rows;
    *rows = r;
    if (hist_x) HIPDRT_CHECK(hipMemcpy(hist_x, p->hist_x.p, (size_t)r * p->n * sizeof(double), hipMemcpyDeviceToHost));
    if (hist_w) HIPDRT_CHECK(hipMemcpy(hist_w, p->hist_w.p, (size_t)r * p->m * sizeof(double), hipMemcpyDeviceToHost));
    if (hist_rho) HIPDRT_CHECK(hipMemcpy(hist_rho, p->hist_rho.p, (size_t)r * 3 * sizeof(double), hipMemcpyDeviceToHost));
    if (qp_iters) HIPDRT_CHECK(hipMemcpy(qp_iters, p->hist_qp.p, (size_t)(r + 1) * sizeof(int), hipMemcpyDeviceToHost));
    return HIPDRT_OK;
} HIPDRT_CATCH

int hipdrt_fit_eis_batch(hipdrt_ctx* ctx, int B, const double* freq, int nf, const double* z_re, const double* z_im,
                         const double* tau, int ntau, double epsilon, int mode, int toeplitz_a, int toeplitz_m,
                         int ngrid, int ny, const double* wt_re, const double* wt_im, const double* log_wt_re,
                         const double* log_wt_im, const hipdrt_fit_opts* opts, double* x, double* fit_x, double* r_inf,
                         double* induc, double* weights, double* coef_scale, double* rho, double* q_vector,
                         int* outer_iters, int* status) try {
    hipdrt_plan* p = nullptr;
    TRY(hipdrt_plan_create(ctx, freq, nf, tau, ntau, epsilon, mode, toeplitz_a, toeplitz_m, ngrid, ny, wt_re, wt_im,
                           log_wt_re, log_wt_im, opts, B, &p));
    int rc = hipdrt_plan_upload(p, B, z_re, z_im);
    if (!rc) rc = hipdrt_plan_fit(p);
    if (!rc) rc = hipdrt_plan_download(p, x, fit_x, r_inf, induc, weights, coef_scale, rho, nullptr, q_vector,
                                       outer_iters, nullptr, status);
    hipdrt_plan_destroy(p);
    return rc;
} HIPDRT_CATCH

}  // extern "C"

// libigx C ABI: host logic (knot-vector bookkeeping, patch set-up, dispatch).  See include/igx.h.
#include "igx_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace igx {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Gauss-Legendre rule on [-1,1] by Newton iteration on P_n (used when the caller gives no rule;
// the Python host passes numpy.polynomial.legendre.leggauss like pyiga/quadrature.py:8)
static void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w)
{
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
        double pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 0; j < n; ++j) {
                double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-16) break;
        }
        x[i] = -z;
        x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
    if (n % 2 == 1) x[n / 2] = 0.0;
}

template <class T>
static int dev_alloc_copy(T **dst, const T *src, size_t n, hipStream_t st)
{
    IGX_HIP(hipMalloc((void **)dst, std::max<size_t>(1, n) * sizeof(T)));
    if (n) IGX_HIP(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, st));
    return IGX_OK;
}

// Knot-vector bookkeeping: restates KnotVector.mesh / mesh_support_idx_all / mesh_span_indices
// (pyiga/bspline.py:110-144), compute_sparsity_ij (pyiga/mlmatrix.py:420-440) and the iterated
// Gauss rule (pyiga/quadrature.py:3-16).
static int setup_axis(Axis &A, const double *kv, int len, int p, int q, const double *gx, const double *gw)
{
    if (p < 0 || p > IGX_MAX_DEGREE) { set_error("degree %d out of range [0,%d]", p, IGX_MAX_DEGREE); return IGX_ERR_ARG; }
    if (len < 2 * (p + 1)) { set_error("knot vector too short (%d knots for degree %d)", len, p); return IGX_ERR_ARG; }
    for (int i = 1; i < len; ++i)
        if (kv[i] < kv[i - 1]) { set_error("knots should be increasing"); return IGX_ERR_ARG; }
    A.p = p; A.P = p + 1; A.q = q;
    A.kv.assign(kv, kv + len);
    A.N = len - p - 1;
    std::vector<int> k2m(len);
    A.mesh.clear();
    for (int i = 0; i < len; ++i) {
        if (i == 0 || kv[i] != kv[i - 1]) A.mesh.push_back(kv[i]);
        k2m[i] = (int)A.mesh.size() - 1;
    }
    A.n = (int)A.mesh.size() - 1;
    if (A.n < 1) { set_error("knot vector has no nonempty span"); return IGX_ERR_ARG; }
    A.G = A.n * q;
    A.span_knot.assign(A.n, 0);
    A.fa.assign(A.n, 0);
    A.simple = true;
    for (int i = 0; i + 1 < len; ++i)
        if (k2m[i] != k2m[i + 1]) A.span_knot[k2m[i]] = i;          // last knot index equal to mesh[s]
    for (int s = 0; s < A.n; ++s) {
        A.fa[s] = A.span_knot[s] - p;
        if (A.fa[s] < 0 || A.fa[s] + p >= A.N) { set_error("knot vector is not open (span %d)", s); return IGX_ERR_ARG; }
        if (s > 0 && A.fa[s] - A.fa[s - 1] != 1) A.simple = false;
    }
    A.mslo.resize(A.N); A.mshi.resize(A.N); A.jlo.resize(A.N); A.jhi.resize(A.N); A.rp.resize(A.N + 1);
    for (int i = 0; i < A.N; ++i) { A.mslo[i] = k2m[i]; A.mshi[i] = k2m[i + p + 1]; }
    auto overlap = [&](int i, int j) { return std::min(A.mshi[i], A.mshi[j]) > std::max(A.mslo[i], A.mslo[j]); };
    A.pair_i.clear(); A.pair_j.clear();
    A.rp[0] = 0;
    for (int i = 0; i < A.N; ++i) {
        if (A.mshi[i] <= A.mslo[i]) { set_error("basis function %d has empty support", i); return IGX_ERR_ARG; }
        int lo = i, hi = i + 1;
        while (lo > 0 && overlap(i, lo - 1)) --lo;
        while (hi < A.N && overlap(i, hi)) ++hi;
        A.jlo[i] = lo; A.jhi[i] = hi;
        A.rp[i + 1] = A.rp[i] + (hi - lo);
        for (int j = lo; j < hi; ++j) { A.pair_i.push_back(i); A.pair_j.push_back(j); }
    }
    A.S = A.rp[A.N];
    // iterated Gauss rule: nodes = outer(h, x) + m, weights = outer(h, w)
    A.nodes.resize(A.G); A.weights.resize(A.G);
    for (int s = 0; s < A.n; ++s) {
        const double a = A.mesh[s], b = A.mesh[s + 1];
        const double m = 0.5 * (a + b), h = 0.5 * (b - a);
        for (int l = 0; l < q; ++l) {
            const double hx = h * gx[l];
            A.nodes[s * q + l] = hx + m;
            A.weights[s * q + l] = h * gw[l];
        }
    }
    return IGX_OK;
}

static int upload_axis(Axis &A, hipStream_t st)
{
    int rc;
    if ((rc = dev_alloc_copy(&A.d_kv, A.kv.data(), A.kv.size(), st))) return rc;
    if ((rc = dev_alloc_copy(&A.d_nodes, A.nodes.data(), A.nodes.size(), st))) return rc;
    if ((rc = dev_alloc_copy(&A.d_w, A.weights.data(), A.weights.size(), st))) return rc;
    std::vector<int> ints;
    auto app = [&](const int *v, size_t n) { size_t o = ints.size(); ints.insert(ints.end(), v, v + n); return o; };
    const size_t o_fa = app(A.fa.data(), A.n), o_lo = app(A.mslo.data(), A.N), o_hi = app(A.mshi.data(), A.N);
    const size_t o_jl = app(A.jlo.data(), A.N), o_jh = app(A.jhi.data(), A.N), o_rp = app(A.rp.data(), A.N + 1);
    const size_t o_pi = app(A.pair_i.data(), A.S), o_pj = app(A.pair_j.data(), A.S);
    if ((rc = dev_alloc_copy(&A.d_ints, ints.data(), ints.size(), st))) return rc;
    IGX_HIP(hipMalloc((void **)&A.d_V, (size_t)A.G * A.P * 2 * sizeof(double)));
    IGX_HIP(hipMalloc((void **)&A.d_PI, (size_t)A.G * 4 * A.P * A.P * sizeof(double)));
    if ((rc = launch_basis_tables(st, A.d_kv, (int)A.kv.size(), A.p, A.d_nodes, (size_t)A.G, 1, nullptr, A.d_V, nullptr, nullptr))) return rc;
    if ((rc = launch_pi_tables(st, A.d_V, A.G, A.P, A.d_PI))) return rc;
    AxisDev &D = A.dev;
    D.p = A.p; D.P = A.P; D.N = A.N; D.n = A.n; D.q = A.q; D.G = A.G; D.S = A.S;
    D.nodes = A.d_nodes; D.w = A.d_w; D.V = A.d_V; D.PI = A.d_PI;
    D.fa = A.d_ints + o_fa; D.mslo = A.d_ints + o_lo; D.mshi = A.d_ints + o_hi;
    D.jlo = A.d_ints + o_jl; D.jhi = A.d_ints + o_jh; D.rp = A.d_ints + o_rp;
    D.pair_i = A.d_ints + o_pi; D.pair_j = A.d_ints + o_pj;
    IGX_HIP(hipStreamSynchronize(st));            // `ints` is a local buffer
    return IGX_OK;
}

static void free_axis(Axis &A)
{
    (void)hipFree(A.d_kv); (void)hipFree(A.d_nodes); (void)hipFree(A.d_w); (void)hipFree(A.d_V); (void)hipFree(A.d_PI); (void)hipFree(A.d_ints);
}

static int setup_geo_axis(GeoAxis &g, const double *kv, int len, int p, const double *d_nodes, int G, hipStream_t st)
{
    if (p < 0 || p > IGX_MAX_DEGREE || len < 2 * (p + 1)) { set_error("bad geometry knot vector (p=%d, %d knots)", p, len); return IGX_ERR_ARG; }
    g.p = p; g.P = p + 1; g.N = len - p - 1;
    g.kv.assign(kv, kv + len);
    int rc;
    if ((rc = dev_alloc_copy(&g.d_kv, kv, (size_t)len, st))) return rc;
    IGX_HIP(hipMalloc((void **)&g.d_V, std::max<size_t>(1, (size_t)G * g.P * 2) * sizeof(double)));
    IGX_HIP(hipMalloc((void **)&g.d_fa, std::max<size_t>(1, (size_t)G) * sizeof(int)));
    return launch_basis_tables(st, g.d_kv, len, p, d_nodes, (size_t)G, 1, nullptr, g.d_V, g.d_fa, nullptr);
}

static void free_geo_axis(GeoAxis &g) { (void)hipFree(g.d_kv); (void)hipFree(g.d_V); (void)hipFree(g.d_fa); }

static int load_vector_jet_run(igx_patch *pt, double *out);

// the sampled scalar coefficient of IGX_CONVDIFF: an affine coefficient is evaluated inside k_geoA and sampled into d_coeff only for the
// kernels that read the array (field kernels of the stage / entry-wise paths) -- 1.5 GB and a 17 ms launch at C5 that the fast
// chain never reads
static int ensure_coeff(igx_patch *pt)
{
    if (pt->coeff_sampled) return IGX_OK;
    if (!pt->coef_affine) { set_error("IGX_CONVDIFF needs igx_patch_set_coeff first"); return IGX_ERR_ARG; }
    const size_t n = (size_t)pt->dev.npts_loc;
    if (!pt->d_coeff && hipMalloc((void **)&pt->d_coeff, std::max<size_t>(1, n) * sizeof(double)) != hipSuccess) {
        pt->d_coeff = nullptr;
        set_error("hipMalloc of %.2f GB for the sampled coefficient failed", n * 8.0 / 1e9);
        return IGX_ERR_NOMEM;
    }
    if (int rc = launch_coeff_affine(pt->ctx->stream, pt, pt->coef_c, pt->d_coeff)) return rc;
    pt->coeff_sampled = true;
    return IGX_OK;
}

static int ensure_fields(igx_patch *pt, int kind)
{
    if (pt->fields_kind == kind) return IGX_OK;
    const int nF = igx_num_fields(pt->dim, kind, pt->dev.form_n);
    if (kind == IGX_FORM) {
        if ((!pt->d_formc && !pt->form_fn) || pt->dev.form_n == 0) { set_error("IGX_FORM needs igx_patch_set_form first"); return IGX_ERR_ARG; }
    }
    if (kind == IGX_CONVDIFF) {
        if (pt->dim != 3) { set_error("IGX_CONVDIFF is a 3D form"); return IGX_ERR_UNSUPPORTED; }
        if (pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("IGX_CONVDIFF needs a spline geometry (physical coordinates)"); return IGX_ERR_UNSUPPORTED; }
        if (int rc = ensure_coeff(pt)) return rc;
    }
    const size_t need = (size_t)nF * pt->dev.npts_loc;
    if (pt->fields_cap < need) {
        if (pt->d_fields) (void)hipFree(pt->d_fields);
        pt->d_fields = nullptr; pt->fields_cap = 0;
        hipError_t e = hipMalloc((void **)&pt->d_fields, std::max<size_t>(1, need) * sizeof(double));
        if (e != hipSuccess) { set_error("hipMalloc of %.2f GB for fields failed", need * 8.0 / 1e9); return IGX_ERR_NOMEM; }
        pt->fields_cap = need;
    }
    pt->fields_kind = -1;
    // (a form given as expressions has its own generated field kernel: no coefficient arrays)
    int rc = (kind == IGX_FORM && pt->form_fn) ? launch_form_fields(pt->ctx->stream, pt, pt->form_fn, pt->d_fields)
                                                : launch_geo_fields(pt->ctx->stream, pt, kind, pt->d_fields);
    if (rc) return rc;
    pt->fields_kind = kind;
    return IGX_OK;
}

// grow-only device workspace owned by the patch
template <class T>
static int ws_reserve(T **buf, size_t *cap, size_t n, const char *what)
{
    if (*cap >= n) return IGX_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    if (hipMalloc((void **)buf, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess) {
        *buf = nullptr;
        set_error("hipMalloc of %.3f GB for %s failed", n * sizeof(T) / 1e9, what);
        return IGX_ERR_NOMEM;
    }
    *cap = n;
    return IGX_OK;
}

} // namespace igx

using namespace igx;

// =============================================================================================
extern "C" {

int igx_version(void) { return IGX_VERSION; }
int igx_fused_stage_fits(int64_t c0max, int64_t S_mid, int64_t S_last, int64_t G_mid, int64_t G_last)
{
    return igx::fused_offsets_fit(c0max, S_mid, S_last, G_mid, G_last) ? 1 : 0;
}
const char *igx_last_error(void) { return g_err; }

igx_ctx *igx_create(int device_id)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        set_error("no HIP device available (%s); libigx has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_id < 0 || device_id >= ndev) { set_error("device %d out of range (have %d)", device_id, ndev); return nullptr; }
    if (hipSetDevice(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed", device_id); return nullptr; }
    igx_ctx *ctx = new (std::nothrow) igx_ctx();
    if (!ctx) return nullptr;
    ctx->device = device_id;
    if (hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || ctx->ncu < 1) ctx->ncu = 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete ctx; return nullptr; }
    for (auto &ev : ctx->ev) (void)hipEventCreate(&ev);
    return ctx;
}

void igx_destroy(igx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->ev) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int igx_sync(igx_ctx *ctx)
{
    if (!ctx) { set_error("null context"); return IGX_ERR_ARG; }
    IGX_HIP(hipStreamSynchronize(ctx->stream));
    return IGX_OK;
}

void *igx_stream(igx_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---------------------------------------------------------------------------------------------
int igx_active_deriv(igx_ctx *ctx, const double *kv, int kv_len, int p, const double *u, size_t nu, int numderiv, double *out)
{
    if (!ctx || !kv || !u || !out) { set_error("igx_active_deriv: null argument"); return IGX_ERR_ARG; }
    if (p < 0 || p > IGX_MAX_DEGREE || kv_len < 2 * (p + 1) || numderiv < 0) { set_error("igx_active_deriv: bad p/kv_len/numderiv"); return IGX_ERR_ARG; }
    if (nu == 0) return IGX_OK;
    IGX_HIP(hipSetDevice(ctx->device));
    double *d_kv = nullptr, *d_u = nullptr, *d_out = nullptr;
    const size_t nout = (size_t)(numderiv + 1) * (p + 1) * nu;
    // single exit: the temporaries are freed (after the stream has drained) on every path
    int rc = dev_alloc_copy(&d_kv, kv, (size_t)kv_len, ctx->stream);
    if (!rc) rc = dev_alloc_copy(&d_u, u, nu, ctx->stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMalloc((void **)&d_out, nout * sizeof(double));
    if (!rc && e == hipSuccess) rc = launch_basis_tables(ctx->stream, d_kv, kv_len, p, d_u, nu, numderiv, d_out, nullptr, nullptr, nullptr);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(out, d_out, nout * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
    (void)hipFree(d_kv); (void)hipFree(d_u); (void)hipFree(d_out);
    if (!rc && e != hipSuccess) { set_error("igx_active_deriv: %s", hipGetErrorString(e)); rc = IGX_ERR_HIP; }
    return rc;
}

int igx_find_spans(igx_ctx *ctx, const double *kv, int kv_len, int p, const double *u, size_t nu, int64_t *spans)
{
    if (!ctx || !kv || !u || !spans) { set_error("igx_find_spans: null argument"); return IGX_ERR_ARG; }
    if (p < 0 || p > IGX_MAX_DEGREE || kv_len < 2 * (p + 1)) { set_error("igx_find_spans: bad p/kv_len"); return IGX_ERR_ARG; }
    if (nu == 0) return IGX_OK;
    IGX_HIP(hipSetDevice(ctx->device));
    double *d_kv = nullptr, *d_u = nullptr;
    long long *d_sp = nullptr;
    int rc = dev_alloc_copy(&d_kv, kv, (size_t)kv_len, ctx->stream);
    if (!rc) rc = dev_alloc_copy(&d_u, u, nu, ctx->stream);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMalloc((void **)&d_sp, nu * sizeof(long long));
    if (!rc && e == hipSuccess) rc = launch_basis_tables(ctx->stream, d_kv, kv_len, p, d_u, nu, 0, nullptr, nullptr, nullptr, d_sp);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(spans, d_sp, nu * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
    (void)hipFree(d_kv); (void)hipFree(d_u); (void)hipFree(d_sp);
    if (!rc && e != hipSuccess) { set_error("igx_find_spans: %s", hipGetErrorString(e)); rc = IGX_ERR_HIP; }
    return rc;
}

int igx_grid_jacobian(igx_ctx *ctx, const igx_patch_desc *d, int ncomp, const double *const grid[IGX_MAX_DIM],
                      const int32_t ngrid[IGX_MAX_DIM], double *jac_out, double *eval_out)
{
    if (!ctx || !d || !grid || !ngrid) { set_error("igx_grid_jacobian: null argument"); return IGX_ERR_ARG; }
    const int dim = d->dim;
    if (dim != 2 && dim != 3) { set_error("igx_grid_jacobian: dim must be 2 or 3"); return IGX_ERR_ARG; }
    if (d->geo_kind != IGX_GEO_BSPLINE && d->geo_kind != IGX_GEO_NURBS) { set_error("igx_grid_jacobian: needs a spline geometry"); return IGX_ERR_ARG; }
    const bool nurbs = d->geo_kind == IGX_GEO_NURBS;
    const int nc = ncomp + (nurbs ? 1 : 0);
    if (ncomp < 1 || nc > MAX_COMP) { set_error("igx_grid_jacobian: %d components unsupported", ncomp); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    GeoAxis gax[3];
    double *d_grid[3] = {nullptr, nullptr, nullptr};
    int G[3] = {1, 1, 1};
    size_t nctrl = nc;
    int rc = IGX_OK;
    for (int k = 0; k < dim && !rc; ++k) {
        G[k] = ngrid[k];
        rc = dev_alloc_copy(&d_grid[k], grid[k], (size_t)G[k], st);
        if (!rc) rc = setup_geo_axis(gax[k], d->geo_kv[k], d->geo_kv_len[k], d->geo_p[k], d_grid[k], G[k], st);
        nctrl *= (size_t)gax[k].N;
    }
    double *d_ctrl = nullptr, *d_j = nullptr, *d_e = nullptr;
    const size_t npts = (size_t)G[0] * G[1] * G[2];
    if (!rc) rc = dev_alloc_copy(&d_ctrl, d->ctrl, nctrl, st);
    if (!rc && jac_out && hipMalloc((void **)&d_j, std::max<size_t>(1, npts * ncomp * dim) * sizeof(double)) != hipSuccess) rc = IGX_ERR_NOMEM;
    if (!rc && eval_out && hipMalloc((void **)&d_e, std::max<size_t>(1, npts * ncomp) * sizeof(double)) != hipSuccess) rc = IGX_ERR_NOMEM;
    if (!rc) rc = launch_grid_geo(st, dim, nc, nurbs, gax, G, d_ctrl, d_j, d_e);
    if (!rc && jac_out && hipMemcpyAsync(jac_out, d_j, npts * ncomp * dim * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) rc = IGX_ERR_HIP;
    if (!rc && eval_out && hipMemcpyAsync(eval_out, d_e, npts * ncomp * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) rc = IGX_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess && !rc) { set_error("igx_grid_jacobian: stream sync failed"); rc = IGX_ERR_HIP; }
    for (int k = 0; k < dim; ++k) { free_geo_axis(gax[k]); (void)hipFree(d_grid[k]); }
    (void)hipFree(d_ctrl); (void)hipFree(d_j); (void)hipFree(d_e);
    return rc;
}

// ---------------------------------------------------------------------------------------------
void igx_patch_destroy(igx_patch *pt)
{
    if (!pt) return;
    if (pt->twin) { igx_patch_destroy(pt->twin); pt->twin = nullptr; }
    (void)hipSetDevice(pt->ctx->device);
    (void)hipStreamSynchronize(pt->ctx->stream);
    for (int k = 0; k < 3; ++k) { free_axis(pt->ax[k]); free_geo_axis(pt->gax[k]); }
    (void)hipFree(pt->d_ctrl); (void)hipFree(pt->d_jac); (void)hipFree(pt->d_coeff); (void)hipFree(pt->d_formc); (void)hipFree(pt->d_fields); (void)hipFree(pt->d_data);
    (void)hipFree(pt->ftab_arr);
    (void)hipFree(pt->d_indices); (void)hipFree(pt->d_indptr); (void)hipFree(pt->d_pl0); (void)hipFree(pt->d_rl0_of); (void)hipFree(pt->d_steps); (void)hipFree(pt->d_ldesc);
    (void)hipFree(pt->d_pl0n); (void)hipFree(pt->d_stepsn); (void)hipFree(pt->d_qdesc); (void)hipFree(pt->d_qdescn);
    (void)hipFree(pt->d_K1); (void)hipFree(pt->d_K2);
    (void)hipFree(pt->d_geoa_tab); (void)hipFree(pt->d_geoa_tabn); (void)hipFree(pt->d_zeros); (void)hipFree(pt->d_triv); (void)hipFree(pt->d_tpairs);
    (void)hipFree(pt->d_ws_ij); (void)hipFree(pt->d_ws_out);
    (void)hipFree(pt->d_lv_f); (void)hipFree(pt->d_lv_t1); (void)hipFree(pt->d_lv_t2); (void)hipFree(pt->d_lv_o);
    delete pt;
}

igx_patch *igx_patch_create(igx_ctx *ctx, const igx_patch_desc *d)
{
    if (!ctx || !d) { set_error("igx_patch_create: null argument"); return nullptr; }
    const int dim = d->dim;
    if (dim != 2 && dim != 3) { set_error("igx_patch_create: dim must be 2 or 3 (got %d)", dim); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    igx_patch *pt = new (std::nothrow) igx_patch();
    if (pt) {                                   // chain choices: read once, here
        igx_knobs &k = pt->knobs;
        if (const char *e = getenv("IGX_PATH")) k.path = !strcmp(e, "fused") ? 1 : !strcmp(e, "unfused") ? 2 : !strcmp(e, "single") ? 3 : 0;
        if (const char *e = getenv("IGX_GEOA")) { k.geoa = strcmp(e, "0") != 0; k.geoa_mf = strcmp(e, "mfma") == 0; }
        if (const char *e = getenv("IGX_FINAL")) k.final_sel = !strcmp(e, "q") ? 1 : !strcmp(e, "valu") ? 2 : !strcmp(e, "mfma") ? 3 : 1;
        if (const char *e = getenv("IGX_ENTRIES")) k.entries_thread = !strcmp(e, "thread");
        k.poison = getenv("IGX_DEBUG_POISON") != nullptr;
        if (const char *e = getenv("IGX_BF")) k.bf = atoi(e);
        if (const char *e = getenv("IGX_PLACEMENT_TRIES")) k.placement_tries = std::max(1, std::min(16, atoi(e)));
        if (const char *e = getenv("IGX_STAGE_EVENTS")) k.stage_events = strcmp(e, "0") != 0;
    }
    if (pt) for (int k = 0; k < 16; ++k) pt->form_slot[k] = -1;
    if (!pt) return nullptr;
    pt->ctx = ctx;
    pt->dim = dim;
    hipStream_t st = ctx->stream;
    int rc = IGX_OK;

    // Gauss rule: nqp = max p + 1 shared by all axes (pyiga/assemblers.pyx:1338)
    int q = d->nqp;
    if (q <= 0) { q = 0; for (int k = 0; k < dim; ++k) q = std::max(q, d->p[k] + 1); }
    pt->nqp = q;
    std::vector<double> gx, gw;
    if (d->gauss_x && d->gauss_w) { gx.assign(d->gauss_x, d->gauss_x + q); gw.assign(d->gauss_w, d->gauss_w + q); }
    else gauss_legendre(q, gx, gw);

    for (int k = 0; k < dim && !rc; ++k) {
        if (!d->kv[k]) { set_error("igx_patch_create: kv[%d] is null", k); rc = IGX_ERR_ARG; break; }
        rc = setup_axis(pt->ax[k], d->kv[k], d->kv_len[k], d->p[k], q, gx.data(), gw.data());
        if (!rc) rc = upload_axis(pt->ax[k], st);
    }
    if (dim == 2) {            // neutral third axis for index arithmetic
        pt->ax[2].N = 1; pt->ax[2].n = 1; pt->ax[2].G = 1; pt->ax[2].S = 1; pt->ax[2].q = q;
        pt->ax[2].dev.N = 1; pt->ax[2].dev.n = 1; pt->ax[2].dev.G = 1; pt->ax[2].dev.S = 1; pt->ax[2].dev.q = q;
    }
    if (!rc && pt->knobs.stage_events < 0) {   // default: per-kernel events where the kernels run for milliseconds
        long long npts = 1;
        for (int k = 0; k < dim; ++k) npts *= pt->ax[k].G;
        pt->knobs.stage_events = dim == 3 && npts >= (1ll << 24);
    }
    // span box (on-demand assemblers): fields on a window of every axis, batched entries only
    for (int k = 0; k < dim && !rc; ++k) pt->boxed = pt->boxed || d->box_hi[k] != 0 || d->box_lo[k] != 0;
    if (!rc && pt->boxed) {
        for (int k = 0; k < dim && !rc; ++k)
            if (d->box_lo[k] < 0 || d->box_lo[k] >= d->box_hi[k] || d->box_hi[k] > pt->ax[k].n) {
                set_error("igx_patch_create: bad span box [%d,%d) on axis %d (%d spans)", d->box_lo[k], d->box_hi[k], k, pt->ax[k].n);
                rc = IGX_ERR_ARG;
            }
        if (!rc && (d->row0_lo != 0 || d->row0_hi != 0)) { set_error("igx_patch_create: a span box and a row slab exclude each other"); rc = IGX_ERR_ARG; }
    }
    // slab
    if (!rc) {
        const Axis &A0 = pt->ax[0];
        pt->r0_lo = d->row0_lo;
        pt->r0_hi = d->row0_hi > 0 ? d->row0_hi : A0.N;
        if (pt->r0_lo < 0 || pt->r0_hi > A0.N || pt->r0_lo >= pt->r0_hi) {
            set_error("igx_patch_create: bad row slab [%d,%d) for %d dofs", pt->r0_lo, pt->r0_hi, A0.N);
            rc = IGX_ERR_ARG;
        }
    }
    if (!rc) {
        const Axis &A0 = pt->ax[0], &A1 = pt->ax[1], &A2 = pt->ax[2];
        pt->s0_lo = A0.mslo[pt->r0_lo];
        pt->s0_hi = A0.mshi[pt->r0_hi - 1];
        if (pt->boxed) { pt->s0_lo = d->box_lo[0]; pt->s0_hi = d->box_hi[0]; }
        const long long plane = (long long)A1.N * A2.N;
        const long long Srest = (long long)A1.S * A2.S;
        pt->nrows_total = (long long)A0.N * plane;
        pt->row_lo = pt->r0_lo * plane;
        pt->row_hi = pt->r0_hi * plane;
        pt->nnz_off = (long long)A0.rp[pt->r0_lo] * Srest;
        pt->nnz = (long long)(A0.rp[pt->r0_hi] - A0.rp[pt->r0_lo]) * Srest;
        // the fused stage forms only the lower triangle: the rows of the p0 dof planes above the slab hold the sources
        // of the slab's upper-triangle entries (mirror pass), behind the owned values in the same buffer
        pt->nnz_ext = (long long)(A0.rp[std::min(pt->r0_hi + A0.p, A0.N)] - A0.rp[pt->r0_lo]) * Srest;
        // elements attributed to this slab: spans of axis 0 split proportionally to the owned dofs
        const long long sp_lo = (long long)A0.n * pt->r0_lo / A0.N, sp_hi = (long long)A0.n * pt->r0_hi / A0.N;
        pt->nelem_owned = (sp_hi - sp_lo) * A1.n * A2.n;
        if (pt->nnz > 0x7fffffffLL && !pt->boxed) {
            set_error("slab has %lld nonzeros; int32 CSR indices need < 2^31 (use more slabs)", pt->nnz);
            rc = IGX_ERR_UNSUPPORTED;
        }
        PatchDev &pd = pt->dev;
        pd.dim = dim;
        for (int k = 0; k < 3; ++k) pd.ax[k] = pt->ax[k].dev;
        pd.r0_lo = pt->r0_lo; pd.r0_hi = pt->r0_hi; pd.s0_lo = pt->s0_lo; pd.s0_hi = pt->s0_hi;
        pd.g0_lo = pt->s0_lo * q; pd.G0_loc = (pt->s0_hi - pt->s0_lo) * q;
        pd.b1 = pd.b2 = 0; pd.L1 = A1.G; pd.L2 = A2.G;
        if (pt->boxed) {
            pd.b1 = d->box_lo[1] * q; pd.L1 = (d->box_hi[1] - d->box_lo[1]) * q;
            if (dim == 3) { pd.b2 = d->box_lo[2] * q; pd.L2 = (d->box_hi[2] - d->box_lo[2]) * q; }
        }
        pd.npts_loc = (long long)pd.G0_loc * pd.L1 * pd.L2;
        pd.nnz_off = pt->nnz_off;
    }
    // geometry
    if (!rc) {
        pt->geo_kind = d->geo_kind;
        if (d->geo_kind == IGX_GEO_BSPLINE || d->geo_kind == IGX_GEO_NURBS) {
            pt->ncomp = dim + (d->geo_kind == IGX_GEO_NURBS ? 1 : 0);
            if (!d->ctrl) { set_error("igx_patch_create: ctrl is null"); rc = IGX_ERR_ARG; }
            size_t nctrl = pt->ncomp;
            for (int k = 0; k < dim && !rc; ++k) {
                if (!d->geo_kv[k]) { set_error("igx_patch_create: geo_kv[%d] is null", k); rc = IGX_ERR_ARG; break; }
                rc = setup_geo_axis(pt->gax[k], d->geo_kv[k], d->geo_kv_len[k], d->geo_p[k], pt->ax[k].d_nodes, pt->ax[k].G, st);
                nctrl *= (size_t)pt->gax[k].N;
                // the geometry must be defined on the parameter domain of the basis
                if (!rc && (pt->gax[k].kv.front() > pt->ax[k].kv.front() || pt->gax[k].kv.back() < pt->ax[k].kv.back())) {
                    set_error("geometry knot vector %d does not cover the parameter domain", k);
                    rc = IGX_ERR_ARG;
                }
            }
            if (!rc) rc = dev_alloc_copy(&pt->d_ctrl, d->ctrl, nctrl, st);
        } else if (d->geo_kind == IGX_GEO_JACOBIAN) {
            if (!d->jac) { set_error("igx_patch_create: jac is null"); rc = IGX_ERR_ARG; }
            else {
                // whole-grid array: the resident planes of axis 0; boxed patch: the array IS the box
                const size_t per_plane = (size_t)pt->dev.L1 * pt->dev.L2 * dim * dim;
                rc = dev_alloc_copy(&pt->d_jac, d->jac + (pt->boxed ? 0 : (size_t)pt->dev.g0_lo * per_plane), (size_t)pt->dev.G0_loc * per_plane, st);
            }
        } else { set_error("igx_patch_create: unknown geo_kind %d", d->geo_kind); rc = IGX_ERR_ARG; }
    }
    if (!rc) {
        pt->sumfact_ok = !pt->boxed && sumfact_supported(pt) != 0;
        if (pt->sumfact_ok) rc = sumfact_prepare(pt);
    }
    if (!rc && hipStreamSynchronize(st) != hipSuccess) { set_error("igx_patch_create: stream sync failed: %s", hipGetErrorString(hipGetLastError())); rc = IGX_ERR_HIP; }
    if (rc) { igx_patch_destroy(pt); return nullptr; }
    // Repeated knots on the last axis only, geometry given as a spline: the twin patch (mid and last axis exchanged) through which
    // the mass and the stiffness matrix are assembled (igx_internal.h, igx_patch::twin).  The twin is an ordinary patch -- same
    // parameter domain, same map, |det| of the Jacobian unchanged -- only its values land in THIS patch's CSR layout.  A twin that
    // cannot be created or whose fast chain does not serve the patch is dropped: the stage kernels take such a patch as before.
    if (dim == 3 && !pt->boxed && pt->sumfact_ok && pt->ax[1].simple && !pt->ax[2].simple && d->ctrl &&
        (d->geo_kind == IGX_GEO_BSPLINE || d->geo_kind == IGX_GEO_NURBS)) {
        igx_patch_desc e = *d;
        const int perm[3] = {0, 2, 1};
        for (int k = 0; k < 3; ++k) {
            e.p[k] = d->p[perm[k]]; e.kv[k] = d->kv[perm[k]]; e.kv_len[k] = d->kv_len[perm[k]];
            e.geo_p[k] = d->geo_p[perm[k]]; e.geo_kv[k] = d->geo_kv[perm[k]]; e.geo_kv_len[k] = d->geo_kv_len[perm[k]];
        }
        e.nqp = pt->nqp;
        const size_t n0 = (size_t)pt->gax[0].N, n1 = (size_t)pt->gax[1].N, n2 = (size_t)pt->gax[2].N, nc = (size_t)pt->ncomp;
        std::vector<double> ctrl(n0 * n1 * n2 * nc);
        for (size_t a = 0; a < n0; ++a)
            for (size_t b = 0; b < n1; ++b)
                for (size_t c = 0; c < n2; ++c)
                    for (size_t k = 0; k < nc; ++k) ctrl[((a * n2 + c) * n1 + b) * nc + k] = d->ctrl[((a * n1 + b) * n2 + c) * nc + k];
        e.ctrl = ctrl.data();
        const std::string keep = g_err;
        igx_patch *tw = igx_patch_create(ctx, &e);
        if (tw) {
            tw->is_twin = true;
            const int kinds = sumfact_twin_kinds(tw);
            if (kinds) { pt->twin = tw; pt->twin_kinds = kinds; }
            else igx_patch_destroy(tw);
        } else set_error("%s", keep.c_str());          // (the failure of an optional twin is not the caller's error)
    }
    return pt;
}

int igx_patch_get_info(const igx_patch *pt, igx_patch_info *info)
{
    if (!pt || !info) { set_error("igx_patch_get_info: null argument"); return IGX_ERR_ARG; }
    memset(info, 0, sizeof(*info));
    info->dim = pt->dim; info->nqp = pt->nqp;
    for (int k = 0; k < pt->dim; ++k) { info->ndofs[k] = pt->ax[k].N; info->nspans[k] = pt->ax[k].n; info->ngauss[k] = pt->ax[k].G; }
    info->nrows_total = pt->nrows_total; info->row_lo = pt->row_lo; info->row_hi = pt->row_hi;
    info->nnz = pt->nnz; info->nnz_offset = pt->nnz_off; info->nelem_owned = pt->nelem_owned;
    info->sumfact_ok = pt->sumfact_ok ? 1 : 0;
    return IGX_OK;
}

// The coefficient of the convection-diffusion form on the twin of a patch (igx_internal.h, igx_patch::twin): an affine one as
// its four numbers, a sampled one as the parent's array with the two last grid axes exchanged.
__global__ void k_swap_last_axes(const double *__restrict__ src, double *__restrict__ dst, const int L1, const int L2, const long long n)
{
    const long long per = (long long)L1 * L2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long g0 = i / per, r = i - g0 * per;
        const int g2 = (int)(r / L1), g1 = (int)(r - (long long)g2 * L1);        // dst is [g0][g2][g1]: consecutive threads write consecutive doubles
        dst[i] = src[g0 * per + (long long)g1 * L2 + g2];
    }
}
static int twin_coeff(igx_patch *pt)
{
    igx_patch *tw = pt->twin;
    if (!tw || !((pt->twin_kinds >> IGX_CONVDIFF) & 1)) return IGX_OK;
    tw->fields_kind = -1;
    tw->coef_affine = pt->coef_affine;
    tw->coeff_sampled = false;
    for (int k = 0; k < 4; ++k) tw->coef_c[k] = pt->coef_c[k];
    if (pt->coef_affine || !pt->d_coeff || !pt->coeff_sampled) return IGX_OK;
    const long long n = pt->dev.npts_loc;
    if (!tw->d_coeff) IGX_HIP(hipMalloc((void **)&tw->d_coeff, std::max<size_t>(1, (size_t)n) * sizeof(double)));
    if (n > 0) {
        k_swap_last_axes<<<dim3((unsigned)std::min<long long>((n + 255) / 256, 65536)), dim3(256), 0, pt->ctx->stream>>>(pt->d_coeff, tw->d_coeff, pt->dev.L1, pt->dev.L2, n);
        IGX_HIP(hipGetLastError());
        IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    }
    tw->coeff_sampled = true;
    return IGX_OK;
}

int igx_patch_set_coeff(igx_patch *pt, const double *coeff)
{
    if (!pt || !coeff) { set_error("igx_patch_set_coeff: null argument"); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    const size_t per_plane = (size_t)pt->dev.L1 * pt->dev.L2;
    const size_t n = (size_t)pt->dev.G0_loc * per_plane;
    if (!pt->d_coeff) IGX_HIP(hipMalloc((void **)&pt->d_coeff, std::max<size_t>(1, n) * sizeof(double)));
    IGX_HIP(hipMemcpyAsync(pt->d_coeff, coeff + (pt->boxed ? 0 : (size_t)pt->dev.g0_lo * per_plane), n * sizeof(double), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    pt->fields_kind = -1;
    pt->coef_affine = 0;
    pt->coeff_sampled = true;
    return twin_coeff(pt);
}

int igx_patch_set_coeff_affine(igx_patch *pt, const double c[4])
{
    if (!pt || !c) { set_error("igx_patch_set_coeff_affine: null argument"); return IGX_ERR_ARG; }
    if (pt->dim != 3) { set_error("igx_patch_set_coeff_affine: the coefficient belongs to the 3D convection-diffusion form"); return IGX_ERR_UNSUPPORTED; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    pt->fields_kind = -1;
    // (the fused geometry + axis-0 sweep evaluates the coefficient itself from the physical coordinates it has at hand; the
    // kernels that read sampled values -- entry-wise, stage kernels -- get them through ensure_coeff when they run)
    pt->coef_affine = 1;
    pt->coeff_sampled = false;
    for (int k = 0; k < 4; ++k) pt->coef_c[k] = c[k];
    return twin_coeff(pt);
}

int igx_patch_set_coeff_expr(igx_patch *pt, const char *expr, int *cache_hit)
{
    if (!pt || !expr) { set_error("igx_patch_set_coeff_expr: null argument"); return IGX_ERR_ARG; }
    if (pt->dim != 3) { set_error("igx_patch_set_coeff_expr: the coefficient belongs to the 3D convection-diffusion form"); return IGX_ERR_UNSUPPORTED; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    const size_t n = (size_t)pt->dev.npts_loc;
    double *buf = nullptr;                               // committed to the patch only when the kernel has run
    IGX_HIP(hipMalloc((void **)&buf, std::max<size_t>(1, n) * sizeof(double)));
    if (int rc = launch_coeff_expr(pt->ctx->stream, pt, expr, buf, cache_hit)) { (void)hipFree(buf); return rc; }
    if (pt->d_coeff) (void)hipFree(pt->d_coeff);
    pt->d_coeff = buf;
    pt->fields_kind = -1;
    pt->coef_affine = 0;
    pt->coeff_sampled = true;
    return twin_coeff(pt);
}

// a function given as a C expression, evaluated at the resident Gauss points into a device array (input of igx_load_vector_d)
int igx_patch_eval_expr_d(igx_patch *pt, const char *expr, int parametric, double *d_out, int *cache_hit)
{
    if (!pt || !expr || !d_out) { set_error("igx_patch_eval_expr_d: null argument"); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    return launch_coeff_expr(pt->ctx->stream, pt, expr, d_out, cache_hit, parametric != 0);
}

int igx_rtc_compile(const char *expr, const char *arch, char *path_out, int path_len, int *cache_hit)
{
    return rtc_compile_expr(expr, arch, path_out, path_len, cache_hit);
}

// coefficient table of IGX_FORM: validated into locals, committed to the patch only when everything (allocation, copies)
// has succeeded -- a failed call leaves the previous form in place.  `on_device`: the arrays are device pointers over the
// RESIDENT Gauss slab (igx_patch_gauss_slab), else host pointers over the full grid.
static int set_form_impl(igx_patch *pt, const double *const coef[16], bool on_device, const char *who)
{
    if (!pt || !coef) { set_error("%s: null argument", who); return IGX_ERR_ARG; }
    const int nj = pt->dim + 1;                   // jet size: value + dim derivatives
    for (int r = 0; r < 4; ++r)
        for (int s = 0; s < 4; ++s)
            if (coef[4 * r + s] && (r >= nj || s >= nj)) { set_error("%s: coefficient (%d,%d) does not exist in %dD", who, r, s, pt->dim); return IGX_ERR_ARG; }
    int slot[16], n = 0;
    for (int k = 0; k < 16; ++k) slot[k] = coef[k] ? n++ : -1;
    if (n == 0) { set_error("%s: all coefficients are absent", who); return IGX_ERR_ARG; }
    // parametric terms: the Jacobian mixes the three derivative directions of each jet block
    bool blk[2][2] = {{false, false}, {false, false}};      // [test is a derivative][trial is a derivative]
    for (int r = 0; r < 4; ++r)
        for (int s = 0; s < 4; ++s)
            if (coef[4 * r + s]) blk[r > 0][s > 0] = true;
    int form_ab[16], nt = 0;
    for (int a = 0; a < nj; ++a)
        for (int b = 0; b < nj; ++b)
            if (blk[a > 0][b > 0]) form_ab[nt++] = 4 * a + b;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    const size_t npts = (size_t)pt->dev.npts_loc, per_plane = npts / (size_t)pt->dev.G0_loc;
    double *d_new = nullptr;
    hipError_t e = hipMalloc((void **)&d_new, std::max<size_t>(1, (size_t)n * npts) * sizeof(double));
    if (e != hipSuccess) { set_error("hipMalloc of %.2f GB for the form coefficients failed", n * npts * 8.0 / 1e9); return IGX_ERR_NOMEM; }
    for (int k = 0; k < 16 && e == hipSuccess; ++k)
        if (coef[k])
            e = on_device ? hipMemcpyAsync(d_new + (size_t)slot[k] * npts, coef[k], npts * sizeof(double), hipMemcpyDeviceToDevice, pt->ctx->stream)
                          : hipMemcpyAsync(d_new + (size_t)slot[k] * npts, coef[k] + (pt->boxed ? 0 : (size_t)pt->dev.g0_lo * per_plane), npts * sizeof(double),
                                           hipMemcpyHostToDevice, pt->ctx->stream);
    const hipError_t es = hipStreamSynchronize(pt->ctx->stream);      // also on failure: no copy is in flight when d_new is freed
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) { (void)hipFree(d_new); set_error("%s: %s", who, hipGetErrorString(e)); return IGX_ERR_HIP; }
    (void)hipFree(pt->d_formc);
    pt->d_formc = d_new;
    pt->form_fn = nullptr;
    for (int k = 0; k < 16; ++k) { pt->form_slot[k] = slot[k]; pt->dev.form_ab[k] = k < nt ? form_ab[k] : 0; }
    pt->dev.form_n = nt;
    pt->dev.form_par = 0;
    pt->fields_kind = -1;
    return IGX_OK;
}

// Parametric jet form (include/igx.h): up to 16 terms  c_k * (slot mv_k of v) * (slot mu_k of u); the coefficients are host
// arrays over the full Gauss grid.  Committed only when every copy has succeeded.
int igx_patch_set_pform(igx_patch *pt, int n, const int *masks, const double *const *coef)
{
    if (!pt || !masks || !coef) { set_error("igx_patch_set_pform: null argument"); return IGX_ERR_ARG; }
    if (n < 1 || n > 16) { set_error("igx_patch_set_pform: %d terms (1..16 per call)", n); return IGX_ERR_ARG; }
    const int full = (1 << pt->dim) - 1;
    for (int k = 0; k < n; ++k) {
        if (!coef[k]) { set_error("igx_patch_set_pform: coefficient %d is null", k); return IGX_ERR_ARG; }
        if ((masks[2 * k] & ~full) || (masks[2 * k + 1] & ~full)) { set_error("igx_patch_set_pform: term %d names an axis the patch does not have", k); return IGX_ERR_ARG; }
    }
    if (pt->boxed) { set_error("igx_patch_set_pform: the patch holds a span box"); return IGX_ERR_UNSUPPORTED; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    const size_t npts = (size_t)pt->dev.npts_loc, per_plane = npts / (size_t)pt->dev.G0_loc;
    double *d_new = nullptr;
    hipError_t e = hipMalloc((void **)&d_new, std::max<size_t>(1, (size_t)n * npts) * sizeof(double));
    if (e != hipSuccess) { set_error("hipMalloc of %.2f GB for the form coefficients failed", n * npts * 8.0 / 1e9); return IGX_ERR_NOMEM; }
    for (int k = 0; k < n && e == hipSuccess; ++k)
        e = hipMemcpyAsync(d_new + (size_t)k * npts, coef[k] + (size_t)pt->dev.g0_lo * per_plane, npts * sizeof(double), hipMemcpyHostToDevice, pt->ctx->stream);
    const hipError_t es = hipStreamSynchronize(pt->ctx->stream);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) { (void)hipFree(d_new); set_error("igx_patch_set_pform: %s", hipGetErrorString(e)); return IGX_ERR_HIP; }
    (void)hipFree(pt->d_formc);
    pt->d_formc = d_new;
    pt->form_fn = nullptr;
    for (int k = 0; k < 16; ++k) { pt->form_slot[k] = k < n ? k : -1; pt->dev.form_ab[k] = k < n ? ((masks[2 * k] << 3) | masks[2 * k + 1]) : 0; }
    pt->dev.form_n = n;
    pt->dev.form_par = 1;
    pt->ftab.valid = false;
    if (pt->twin) pt->twin->ftab.valid = false;
    pt->fields_kind = -1;
    return IGX_OK;
}

// Derivative orders (0..2) held by the two slots of every axis' basis table; (0, 1) is what every built-in form expects, so
// anything else is accepted by IGX_FORM with a parametric jet form only (basis_orders_ok).
int igx_patch_set_basis_orders(igx_patch *pt, const int *slot0, const int *slot1)
{
    if (!pt || !slot0 || !slot1) { set_error("igx_patch_set_basis_orders: null argument"); return IGX_ERR_ARG; }
    for (int k = 0; k < pt->dim; ++k)
        if (slot0[k] < 0 || slot1[k] > 2 || slot0[k] >= slot1[k]) { set_error("igx_patch_set_basis_orders: axis %d: orders (%d, %d), expected 0 <= first < second <= 2", k, slot0[k], slot1[k]); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    // On EVERY exit the flag says what the tables hold: it drops to false before the first table is touched and becomes true
    // again only when all axes are back at (value, first derivative); an axis whose rewrite failed is marked as holding
    // nothing usable, so that the next successful call rewrites it.
    pt->basis_default = false;
    int rc = IGX_OK;
    for (int k = 0; k < pt->dim && rc == IGX_OK; ++k) {
        Axis &A = pt->ax[k];
        if (A.ord[0] != slot0[k] || A.ord[1] != slot1[k]) {
            A.ord[0] = A.ord[1] = -1;
            rc = launch_basis_tables(st, A.d_kv, (int)A.kv.size(), A.p, A.d_nodes, (size_t)A.G, slot1[k], nullptr, A.d_V, nullptr, nullptr, slot0[k], slot1[k]);
            if (rc == IGX_OK) rc = launch_pi_tables(st, A.d_V, A.G, A.P, A.d_PI);
            if (rc == IGX_OK) { A.ord[0] = slot0[k]; A.ord[1] = slot1[k]; }
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == IGX_OK) { set_error("igx_patch_set_basis_orders: synchronisation failed"); rc = IGX_ERR_HIP; }
    if (rc != IGX_OK) {
        for (int k = 0; k < pt->dim; ++k) pt->ax[k].ord[0] = pt->ax[k].ord[1] = -1;      // (the kernels may not have run: rewrite all next time)
        return rc;
    }
    bool def = true;
    for (int k = 0; k < pt->dim; ++k) def = def && pt->ax[k].ord[0] == 0 && pt->ax[k].ord[1] == 1;
    pt->basis_default = def;
    return IGX_OK;
}

static int basis_orders_ok(const igx_patch *pt, int kind, const char *who)
{
    if (pt->basis_default || (kind == IGX_FORM && pt->dev.form_par)) return IGX_OK;
    set_error("%s: the basis tables hold derivative orders other than (0, 1) (igx_patch_set_basis_orders): only a parametric jet form can be assembled", who);
    return IGX_ERR_ARG;
}

int igx_patch_set_form(igx_patch *pt, const double *const coef[16])
{
    if (pt) pt->ftab.valid = false;                      // (sampled coefficients: the stage kernels over the field arrays)
    if (pt && pt->twin) pt->twin->ftab.valid = false;
    return set_form_impl(pt, coef, false, "igx_patch_set_form");
}

// The coefficient table of IGX_FORM as C expressions in the physical coordinates.  Spline geometries: the expressions are
// compiled INTO the field kernel of the form (rtc.hip, igx_form_fields) -- nothing is evaluated or stored here, the
// coefficients never exist as arrays.  Otherwise (control lines beyond LDS): one generated kernel evaluates them at the
// resident Gauss points and the arrays take the way of igx_patch_set_form_d.
// The physical coefficient table of a form given as expressions, as the fast chain wants it: constants told from expressions
// (a constant is a number, possibly in parentheses: pyiga_amd.symbolic writes them so), equal entries found by their text.
static void capture_form_table(igx_patch *pt, const char *const expr[16])
{
    igx_patch::FormTable &T = pt->ftab;
    T = igx_patch::FormTable();
    (void)hipFree(pt->ftab_arr); pt->ftab_arr = nullptr; pt->ftab_ready = false;
    for (int k = 0; k < 16; ++k) {
        T.arr_of[k] = -1;
        if (!expr[k]) continue;
        T.present |= 1 << k;
        std::string e(expr[k]);
        // strip blanks and balanced outer parentheses
        auto strip = [](std::string x) {
            for (;;) {
                while (!x.empty() && isspace((unsigned char)x.front())) x.erase(x.begin());
                while (!x.empty() && isspace((unsigned char)x.back())) x.pop_back();
                if (x.size() >= 2 && x.front() == '(' && x.back() == ')') {
                    int depth = 0; bool outer = true;
                    for (size_t i = 0; i + 1 < x.size(); ++i) { depth += x[i] == '(' ? 1 : x[i] == ')' ? -1 : 0; if (depth == 0) { outer = false; break; } }
                    if (outer) { x = x.substr(1, x.size() - 2); continue; }
                }
                return x;
            }
        };
        const std::string b = strip(e);
        char *end = nullptr;
        const double v = b.empty() ? 0.0 : strtod(b.c_str(), &end);
        if (!b.empty() && end && *end == 0) { T.is_const |= 1 << k; T.cval[k] = v; }
        else T.expr[k] = e;
    }
    auto same = [&](int i, int j) {
        const bool pi = (T.present >> i) & 1, pj = (T.present >> j) & 1;
        if (pi != pj) return false;
        if (!pi) return true;
        const bool ci = (T.is_const >> i) & 1, cj = (T.is_const >> j) & 1;
        if (ci != cj) return false;
        return ci ? T.cval[i] == T.cval[j] : T.expr[i] == T.expr[j];
    };
    T.sym = true; T.blocksym = true;
    for (int r = 0; r < 4; ++r)
        for (int c = r + 1; c < 4; ++c)
            if (!same(4 * r + c, 4 * c + r)) { T.sym = false; if (r >= 1) T.blocksym = false; }
    // arrays: one per distinct expression text
    for (int k = 0; k < 16; ++k) {
        if (!((T.present >> k) & 1) || ((T.is_const >> k) & 1)) continue;
        for (int j = 0; j < k; ++j)
            if (T.arr_of[j] >= 0 && T.expr[j] == T.expr[k]) { T.arr_of[k] = T.arr_of[j]; break; }
        if (T.arr_of[k] < 0) T.arr_of[k] = T.narr++;
    }
    T.valid = true;
}

int igx_patch_set_form_expr(igx_patch *pt, const char *const expr[16], int *cache_hit)
{
    if (!pt || !expr) { set_error("igx_patch_set_form_expr: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_patch_set_form_expr: the patch holds a span box"); return IGX_ERR_UNSUPPORTED; }
    const char *list[16];
    int n = 0;
    for (int k = 0; k < 16; ++k)
        if (expr[k]) list[n++] = expr[k];
    if (n == 0) { set_error("igx_patch_set_form_expr: all coefficients are absent"); return IGX_ERR_ARG; }
    const int nj = pt->dim + 1;
    for (int r = 0; r < 4; ++r)
        for (int s = 0; s < 4; ++s)
            if (expr[4 * r + s] && (r >= nj || s >= nj)) { set_error("igx_patch_set_form_expr: coefficient (%d,%d) does not exist in %dD", r, s, pt->dim); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    capture_form_table(pt, expr);
    // (the table is one of PHYSICAL coefficients in physical coordinates: the twin of a patch with repeated knots on its last axis
    // -- igx_patch::twin -- takes it as it is and samples its functions on its own Gauss grid when its chain first runs)
    if (pt->twin) capture_form_table(pt->twin, expr);
    if (form_fields_applicable(pt)) {
        int form_ab[16];
        const int nt = form_terms(pt->dim, expr, form_ab);
        void *fn = nullptr;
        if (int rc = rtc_form_fields_function(pt, expr, nt, form_ab, &fn, cache_hit)) return rc;
        (void)hipFree(pt->d_formc);
        pt->d_formc = nullptr;
        pt->form_fn = fn;
        int j = 0;
        for (int k = 0; k < 16; ++k) { pt->form_slot[k] = expr[k] ? j++ : -1; pt->dev.form_ab[k] = k < nt ? form_ab[k] : 0; }
        pt->dev.form_n = nt;
        pt->dev.form_par = 0;
        pt->fields_kind = -1;
        return IGX_OK;
    }
    const size_t npts = (size_t)pt->dev.npts_loc;
    double *buf = nullptr;
    if (hipMalloc((void **)&buf, std::max<size_t>(1, (size_t)n * npts) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %.2f GB for the form coefficients failed", n * npts * 8.0 / 1e9); return IGX_ERR_NOMEM; }
    int rc = launch_form_exprs(pt->ctx->stream, pt, n, list, buf, cache_hit);
    if (!rc) {
        const double *d_coef[16];
        int j = 0;
        for (int k = 0; k < 16; ++k) d_coef[k] = expr[k] ? buf + (size_t)(j++) * npts : nullptr;
        rc = set_form_impl(pt, d_coef, true, "igx_patch_set_form_expr");
    }
    (void)hipFree(buf);
    return rc;
}

int igx_rtc_compile_form(int n, const char *const *expr, const char *arch, char *path_out, int path_len, int *cache_hit)
{
    return rtc_compile_form(n, expr, arch, path_out, path_len, cache_hit);
}

int igx_patch_form_generated(const igx_patch *pt) { return pt && pt->form_fn ? 1 : 0; }

int igx_rtc_compile_form_fields(int dim, int ncomp, const char *const expr[16], const char *arch, char *path_out, int path_len, int *cache_hit)
{
    return rtc_compile_form_fields(dim, ncomp, expr, arch, path_out, path_len, cache_hit);
}

int igx_patch_set_form_d(igx_patch *pt, const double *const d_coef[16])
{
    if (pt) pt->ftab.valid = false;                      // (sampled coefficients: the stage kernels over the field arrays)
    if (pt && pt->twin) pt->twin->ftab.valid = false;
    return set_form_impl(pt, d_coef, true, "igx_patch_set_form_d");
}

int igx_patch_gauss(const igx_patch *pt, int axis, double *nodes, double *weights)
{
    if (!pt || axis < 0 || axis >= pt->dim) { set_error("igx_patch_gauss: bad argument"); return IGX_ERR_ARG; }
    if (nodes) memcpy(nodes, pt->ax[axis].nodes.data(), pt->ax[axis].nodes.size() * sizeof(double));
    if (weights) memcpy(weights, pt->ax[axis].weights.data(), pt->ax[axis].weights.size() * sizeof(double));
    return IGX_OK;
}

int igx_pattern(igx_patch *pt, int32_t *indptr, int32_t *indices)
{
    if (!pt) { set_error("igx_pattern: null patch"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_pattern: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    const long long nrows = pt->row_hi - pt->row_lo;
    if (!pt->have_pattern) {
        if (!pt->d_indptr) IGX_HIP(hipMalloc((void **)&pt->d_indptr, (size_t)(nrows + 1) * sizeof(int32_t)));
        if (!pt->d_indices) {
            hipError_t e = hipMalloc((void **)&pt->d_indices, std::max<size_t>(1, (size_t)pt->nnz) * sizeof(int32_t));
            if (e != hipSuccess) { set_error("hipMalloc of %.2f GB for CSR indices failed", pt->nnz * 4.0 / 1e9); return IGX_ERR_NOMEM; }
        }
        int rc = launch_pattern(st, pt, pt->d_indptr, pt->d_indices);
        if (rc) return rc;
        pt->have_pattern = true;
    }
    if (indptr) IGX_HIP(hipMemcpyAsync(indptr, pt->d_indptr, (size_t)(nrows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (indices) IGX_HIP(hipMemcpyAsync(indices, pt->d_indices, (size_t)pt->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    IGX_HIP(hipStreamSynchronize(st));
    return IGX_OK;
}

int igx_assemble(igx_patch *pt, int kind, int algo, double *data_out)
{
    if (!pt) { set_error("igx_assemble: null patch"); return IGX_ERR_ARG; }
    if (kind < IGX_MASS || kind > IGX_FORM) { set_error("igx_assemble: unknown kind %d", kind); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_assemble: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    if (int rc = basis_orders_ok(pt, kind, "igx_assemble")) return rc;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    const bool was_auto = algo == IGX_ALGO_AUTO;
    if (algo == IGX_ALGO_AUTO) algo = (pt->sumfact_ok && sumfact_supports_kind(pt, kind)) ? IGX_ALGO_SUMFACT : IGX_ALGO_ENTRYWISE;
    if (algo == IGX_ALGO_SUMFACT && !sumfact_supports_kind(pt, kind)) { set_error("igx_assemble: sum factorisation does not support this form yet"); return IGX_ERR_UNSUPPORTED; }
    if (algo == IGX_ALGO_SUMFACT && !pt->sumfact_ok) { set_error("igx_assemble: sum factorisation does not support this patch (degree > %d)", IGX_MAX_SF_DEGREE); return IGX_ERR_UNSUPPORTED; }
    if (algo != IGX_ALGO_SUMFACT && algo != IGX_ALGO_ENTRYWISE) { set_error("igx_assemble: unknown algo %d", algo); return IGX_ERR_ARG; }
    if (!pt->d_data) {
        const size_t bytes = ((size_t)std::max(pt->nnz, pt->nnz_ext) + IGX_DUMP_PAD) * sizeof(double);   // + halo rows of the mirror pass / dump slots of masked stores
        hipError_t e = hipMalloc((void **)&pt->d_data, bytes);
        if (e != hipSuccess) { set_error("hipMalloc of %.2f GB for CSR values failed", pt->nnz * 8.0 / 1e9); return IGX_ERR_NOMEM; }
        // opt-in (IGX_PLACEMENT_TRIES): keep the candidate buffer on which the mirror pass of this patch runs fastest.  The
        // candidates are alive together (the driver must not hand the same pages out again); a failed allocation or probe ends
        // the search with what there is.
        if (pt->knobs.placement_tries > 1 && pt->knobs.bf == 2 && algo == IGX_ALGO_SUMFACT && igx_kind_symmetric(kind) && pt->sumfact_ok) {   // (only the chain with the mirror pass: IGX_BF=2)
            std::vector<double *> cand(1, pt->d_data);
            std::vector<float> ms(1, sumfact_probe_mirror(pt, pt->d_data));
            for (int k = 1; k < pt->knobs.placement_tries && ms[0] >= 0.0f; ++k) {
                double *b = nullptr;
                if (hipMalloc((void **)&b, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
                const float t = sumfact_probe_mirror(pt, b);
                if (t < 0.0f) { (void)hipFree(b); break; }
                cand.push_back(b); ms.push_back(t);
            }
            size_t best = 0;
            for (size_t k = 1; k < cand.size(); ++k)
                if (ms[k] < ms[best]) best = k;
            for (size_t k = 0; k < cand.size(); ++k)
                if (k != best) (void)hipFree(cand[k]);
            pt->d_data = cand[best];
            if (ms[0] >= 0.0f) {                         // (a patch without a mirror pass -- 2D, other paths -- has nothing to time)
                pt->placement_ms_best = ms[best];
                pt->placement_ms_worst = *std::max_element(ms.begin(), ms.end());
                pt->placement_tried = (int)cand.size();
            }
        }
    }
    if (pt->knobs.poison)                         // every value must be written exactly once
        IGX_HIP(hipMemsetAsync(pt->d_data, 0xFF, (size_t)pt->nnz * sizeof(double), st));
    memset(&pt->timing, 0, sizeof(pt->timing));
    pt->timing.algo_used = algo;
    hipEvent_t *ev = pt->ctx->ev;
    IGX_HIP(hipEventRecord(ev[0], st));
    pt->fields_kind = -1;                           // the timed path always recomputes the fields
    pt->last_path = 0;
    int rc = IGX_OK;
    if (algo != IGX_ALGO_SUMFACT || sumfact_needs_fields(pt, kind)) rc = ensure_fields(pt, kind);
    if (rc) return rc;
    pt->timing.n_launches = 1;
    // k_single2d: first and last event only (one launch); a chain without stage events likewise
    const bool one_launch = algo == IGX_ALGO_SUMFACT && pt->dim == 2 && sumfact_single_launch(pt, kind);
    const bool staged = !one_launch && pt->knobs.stage_events;
    if (staged) IGX_HIP(hipEventRecord(ev[1], st));
    if (algo == IGX_ALGO_SUMFACT) {
        rc = sumfact_assemble(pt, kind, pt->d_data);
        if (rc == IGX_ERR_UNSUPPORTED && was_auto) {
            // a shape the stage kernels refuse (a limit of their staging buffers): with IGX_ALGO_AUTO the entry-wise kernels take
            // over -- same device, same matrix; an explicit IGX_ALGO_SUMFACT keeps the error
            algo = IGX_ALGO_ENTRYWISE;
            pt->timing.algo_used = algo;
            pt->last_path = 0;
            rc = ensure_fields(pt, kind);
        }
        if (rc) return rc;
    }
    if (algo != IGX_ALGO_SUMFACT) {
        rc = launch_entries_csr(st, pt, kind, pt->d_data);
        if (rc) return rc;
        pt->timing.n_launches++;
        if (staged) IGX_HIP(hipEventRecord(ev[4], st));
    }
    IGX_HIP(hipEventRecord(ev[5], st));
    IGX_HIP(hipStreamSynchronize(st));
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("igx_assemble: kernel failure: %s", hipGetErrorString(e)); return IGX_ERR_HIP; }
    }
    (void)hipEventElapsedTime(&pt->timing.total_ms, ev[0], ev[5]);
    if (one_launch) pt->timing.stage1_ms = pt->timing.total_ms;
    if (!staged) {                              // only the whole interval was timed
    } else if (algo == IGX_ALGO_SUMFACT) {
        (void)hipEventElapsedTime(&pt->timing.fields_ms, ev[0], ev[1]);
        (void)hipEventElapsedTime(&pt->timing.stage0_ms, ev[1], ev[2]);
        (void)hipEventElapsedTime(&pt->timing.stage1_ms, ev[2], ev[3]);
        (void)hipEventElapsedTime(&pt->timing.final_ms, ev[3], ev[4]);
    } else {
        (void)hipEventElapsedTime(&pt->timing.fields_ms, ev[0], ev[1]);
        (void)hipEventElapsedTime(&pt->timing.entry_ms, ev[1], ev[4]);
    }
    if (data_out) {
        IGX_HIP(hipMemcpyAsync(data_out, pt->d_data, (size_t)pt->nnz * sizeof(double), hipMemcpyDeviceToHost, st));
        IGX_HIP(hipStreamSynchronize(st));
    }
    return IGX_OK;
}

int igx_assemble_kron3(igx_patch *p3, igx_patch *p2, int kind, const double *m0, const double *k0, double *data_out)
{
    if (!p3 || !p2 || !m0) { set_error("igx_assemble_kron3: null argument"); return IGX_ERR_ARG; }
    if (kind != IGX_MASS && kind != IGX_STIFFNESS) { set_error("igx_assemble_kron3: mass or stiffness"); return IGX_ERR_ARG; }
    if (kind == IGX_STIFFNESS && !k0) { set_error("igx_assemble_kron3: the stiffness form needs k0"); return IGX_ERR_ARG; }
    if (int rcb = basis_orders_ok(p2, kind, "igx_assemble_kron3")) return rcb;
    if (p3->dim != 3 || p2->dim != 2 || p3->boxed || p2->boxed) { set_error("igx_assemble_kron3: a 3D patch and the 2D patch of its cross-section"); return IGX_ERR_ARG; }
    if (p3->ctx != p2->ctx) { set_error("igx_assemble_kron3: both patches must live in one context"); return IGX_ERR_ARG; }
    for (int k = 0; k < 2; ++k) {
        const Axis &a = p3->ax[1 + k], &b = p2->ax[k];
        if (a.N != b.N || a.P != b.P || a.q != b.q || a.S != b.S || a.kv != b.kv) { set_error("igx_assemble_kron3: axis %d of the cross-section does not match (knots, degree or Gauss points per span)", k); return IGX_ERR_ARG; }
    }
    if (p2->r0_lo != 0 || p2->r0_hi != p2->ax[0].N) { set_error("igx_assemble_kron3: the cross-section patch must be whole"); return IGX_ERR_ARG; }
    // k_kron3 stages one 2D row (<= 128 entries) and one 1D row (<= 16 entries) per wave in LDS (kron.hip)
    if ((2 * p3->ax[1].p + 1) * (2 * p3->ax[2].p + 1) > 128 || 2 * p3->ax[0].p + 1 > 16) {
        set_error("igx_assemble_kron3: degrees (%d, %d, %d) beyond the row buffers of the expansion kernel", p3->ax[0].p, p3->ax[1].p, p3->ax[2].p);
        return IGX_ERR_UNSUPPORTED;
    }
    IGX_HIP(hipSetDevice(p3->ctx->device));
    hipStream_t st = p3->ctx->stream;
    const int C0 = 2 * p3->ax[0].p + 1, N0 = p3->ax[0].N;
    for (int i = 0; i < N0; ++i)
        if (p3->ax[0].jhi[i] - p3->ax[0].jlo[i] > C0) { set_error("igx_assemble_kron3: more than 2 p + 1 columns per row on axis 0"); return IGX_ERR_UNSUPPORTED; }
    if (!p3->d_data) {
        const size_t bytes = ((size_t)std::max(p3->nnz, p3->nnz_ext) + IGX_DUMP_PAD) * sizeof(double);
        if (hipMalloc((void **)&p3->d_data, bytes) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %.2f GB for CSR values failed", p3->nnz * 8.0 / 1e9); return IGX_ERR_NOMEM; }
    }
    if (p3->knobs.poison) IGX_HIP(hipMemsetAsync(p3->d_data, 0xFF, (size_t)p3->nnz * sizeof(double), st));
    // 2D matrices of the cross-section: the regular 2D path (mass first: its values move aside, the stiffness values stay in
    // the patch's own buffer)
    double *d_M2 = nullptr, *d_band = nullptr;
    const size_t n2 = (size_t)p2->nnz;
    int rc = igx_assemble(p2, IGX_MASS, IGX_ALGO_AUTO, nullptr);
    if (rc) return rc;
    const double *A2 = p2->d_data, *B2 = nullptr;
    if (kind == IGX_STIFFNESS) {
        if (hipMalloc((void **)&d_M2, std::max<size_t>(1, n2) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); set_error("igx_assemble_kron3: out of device memory"); return IGX_ERR_NOMEM; }
        if (hipMemcpyAsync(d_M2, p2->d_data, n2 * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(d_M2); set_error("igx_assemble_kron3: copy failed"); return IGX_ERR_HIP; }
        rc = igx_assemble(p2, IGX_STIFFNESS, IGX_ALGO_AUTO, nullptr);
        if (rc) { (void)hipFree(d_M2); return rc; }
        A2 = p2->d_data; B2 = d_M2;                         // K = M0 (x) K2D + K0 (x) M2D
    }
    const size_t nb = (size_t)N0 * C0;
    if (hipMalloc((void **)&d_band, 2 * nb * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_M2); set_error("igx_assemble_kron3: out of device memory"); return IGX_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(d_band, m0, nb * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_band + nb, kind == IGX_STIFFNESS ? k0 : m0, nb * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);       // (m0 / k0 belong to the caller: no copy in flight on return)
    memset(&p3->timing, 0, sizeof(p3->timing));
    p3->timing.algo_used = IGX_ALGO_SUMFACT;
    p3->last_path = 0;
    hipEvent_t *ev = p3->ctx->ev;
    if (e == hipSuccess) {
        (void)hipEventRecord(ev[0], st);
        rc = launch_kron3(st, p3, d_band, B2 ? d_band + nb : nullptr, C0, A2, B2);
        (void)hipEventRecord(ev[5], st);
        e = hipStreamSynchronize(st);
    }
    (void)hipFree(d_band); (void)hipFree(d_M2);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("igx_assemble_kron3: %s", hipGetErrorString(e)); return IGX_ERR_HIP; }
    (void)hipEventElapsedTime(&p3->timing.total_ms, ev[0], ev[5]);
    p3->timing.final_ms = p3->timing.total_ms;
    p3->timing.n_launches = 1;
    p3->last_path = IGX_PATH_KRON;
    if (data_out) {
        IGX_HIP(hipMemcpyAsync(data_out, p3->d_data, (size_t)p3->nnz * sizeof(double), hipMemcpyDeviceToHost, st));
        IGX_HIP(hipStreamSynchronize(st));
    }
    return IGX_OK;
}

int igx_patch_placement(const igx_patch *pt, int *tried, float *best_ms, float *worst_ms)
{
    if (!pt) { set_error("igx_patch_placement: null patch"); return IGX_ERR_ARG; }
    if (tried) *tried = pt->placement_tried;
    if (best_ms) *best_ms = pt->placement_ms_best;
    if (worst_ms) *worst_ms = pt->placement_ms_worst;
    return IGX_OK;
}

int igx_last_timing(const igx_patch *pt, igx_timing *t)
{
    if (!pt || !t) { set_error("igx_last_timing: null argument"); return IGX_ERR_ARG; }
    *t = pt->timing;
    return IGX_OK;
}

const double *igx_d_csr_data(const igx_patch *pt) { return pt ? pt->d_data : nullptr; }
const int32_t *igx_d_csr_indices(const igx_patch *pt) { return pt ? pt->d_indices : nullptr; }
const int32_t *igx_d_csr_indptr(const igx_patch *pt) { return pt ? pt->d_indptr : nullptr; }

int igx_entries_d(igx_patch *pt, int kind, const size_t *d_ij, size_t M, double *d_out)
{
    if (!pt || (M && (!d_ij || !d_out))) { set_error("igx_entries_d: null argument"); return IGX_ERR_ARG; }
    if (kind < IGX_MASS || kind > IGX_FORM) { set_error("igx_entries_d: unknown kind %d", kind); return IGX_ERR_ARG; }
    if (int rcb = basis_orders_ok(pt, kind, "igx_entries_d")) return rcb;
    if (M == 0) return IGX_OK;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    int rc = ensure_fields(pt, kind);
    if (rc) return rc;
    (void)hipEventRecord(pt->ctx->ev[6], st);
    rc = launch_entries_list(st, pt, kind, d_ij, M, d_out);
    (void)hipEventRecord(pt->ctx->ev[7], st);
    if (rc) return rc;
    IGX_HIP(hipStreamSynchronize(st));
    memset(&pt->timing, 0, sizeof(pt->timing));
    (void)hipEventElapsedTime(&pt->timing.entry_ms, pt->ctx->ev[6], pt->ctx->ev[7]);
    pt->timing.total_ms = pt->timing.entry_ms;
    pt->timing.algo_used = IGX_ALGO_ENTRYWISE;
    pt->timing.n_launches = 1;
    return IGX_OK;
}

int igx_entries(igx_patch *pt, int kind, const size_t *ij, size_t M, double *out)
{
    if (!pt || (M && (!ij || !out))) { set_error("igx_entries: null argument"); return IGX_ERR_ARG; }
    if (kind < IGX_MASS || kind > IGX_FORM) { set_error("igx_entries: unknown kind %d", kind); return IGX_ERR_ARG; }
    if (M == 0) return IGX_OK;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    int rc;
    if ((rc = ws_reserve(&pt->d_ws_ij, &pt->ws_ij_cap, 2 * M, "index pairs"))) return rc;
    if ((rc = ws_reserve(&pt->d_ws_out, &pt->ws_out_cap, M, "entry values"))) return rc;
    IGX_HIP(hipMemcpyAsync(pt->d_ws_ij, ij, 2 * M * sizeof(size_t), hipMemcpyHostToDevice, st));
    if ((rc = igx_entries_d(pt, kind, pt->d_ws_ij, M, pt->d_ws_out))) return rc;
    IGX_HIP(hipMemcpyAsync(out, pt->d_ws_out, M * sizeof(double), hipMemcpyDeviceToHost, st));
    IGX_HIP(hipStreamSynchronize(st));
    return IGX_OK;
}

} // extern "C"
// values of boxes of the reordered tensor (ACA consumer): no index upload, one launch pair, one copy back
int igx::entries_pair_boxes(igx_patch *pt, int kind, const PairBoxes &B, double *out)
{
    const size_t M = (size_t)B.off[B.n];
    if (M == 0) return IGX_OK;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    int rc;
    if ((rc = ws_reserve(&pt->d_ws_ij, &pt->ws_ij_cap, 2 * M, "index pairs"))) return rc;
    if ((rc = ws_reserve(&pt->d_ws_out, &pt->ws_out_cap, M, "entry values"))) return rc;
    if ((rc = launch_box_pairs(st, pt, B, pt->d_ws_ij))) return rc;
    if ((rc = igx_entries_d(pt, kind, pt->d_ws_ij, M, pt->d_ws_out))) return rc;
    IGX_HIP(hipMemcpyAsync(out, pt->d_ws_out, M * sizeof(double), hipMemcpyDeviceToHost, st));
    IGX_HIP(hipStreamSynchronize(st));
    return IGX_OK;
}

extern "C" {
int igx_patch_set_aca_batch(igx_patch *pt, long long max_entries)
{
    if (!pt || max_entries < 0) { set_error("igx_patch_set_aca_batch: bad argument"); return IGX_ERR_ARG; }
    pt->aca_batch = max_entries;
    return IGX_OK;
}

int igx_fast_assemble_stats(const igx_patch *pt, long long *requests, long long *entries, int *rank)
{
    if (!pt) { set_error("igx_fast_assemble_stats: null patch"); return IGX_ERR_ARG; }
    if (requests) *requests = pt->aca_requests;
    if (entries) *entries = pt->aca_entries;
    if (rank) *rank = pt->aca_rank;
    return IGX_OK;
}

int igx_fields(igx_patch *pt, int kind, double *out, int64_t *shape4)
{
    if (!pt) { set_error("igx_fields: null patch"); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    int rc = ensure_fields(pt, kind);
    if (rc) return rc;
    const int nF = igx_num_fields(pt->dim, kind, pt->dev.form_n);
    if (shape4) { shape4[0] = nF; shape4[1] = pt->dev.G0_loc; shape4[2] = pt->dev.L1; shape4[3] = pt->dim == 3 ? pt->dev.L2 : 1; }
    if (out) {
        IGX_HIP(hipMemcpyAsync(out, pt->d_fields, (size_t)nF * pt->dev.npts_loc * sizeof(double), hipMemcpyDeviceToHost, pt->ctx->stream));
        IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    }
    return IGX_OK;
}

static int lv_workspace(igx_patch *pt, size_t *n_out)
{
    const PatchDev &pd = pt->dev;
    const int dim = pt->dim;
    const size_t N1 = pt->ax[1].N, N2 = dim == 3 ? pt->ax[2].N : 1, G1 = pt->ax[1].G;
    *n_out = (size_t)(pt->r0_hi - pt->r0_lo) * N1 * N2;
    const size_t n_t1 = dim == 3 ? (size_t)pd.G0_loc * G1 * N2 : (size_t)pd.G0_loc * N1;
    const size_t n_t2 = dim == 3 ? (size_t)pd.G0_loc * N1 * N2 : 1;
    int rc;
    if ((rc = ws_reserve(&pt->d_lv_t1, &pt->lv_t1_cap, n_t1, "load-vector workspace"))) return rc;
    if ((rc = ws_reserve(&pt->d_lv_t2, &pt->lv_t2_cap, n_t2, "load-vector workspace"))) return rc;
    return ws_reserve(&pt->d_lv_o, &pt->lv_o_cap, *n_out, "load vector");
}

int igx_load_vector_d(igx_patch *pt, const double *d_fvals, double *d_out)
{
    if (!pt || !d_fvals || !d_out) { set_error("igx_load_vector_d: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_load_vector_d: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    if (int rcb = basis_orders_ok(pt, IGX_MASS, "igx_load_vector_d")) return rcb;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    int rc = ensure_fields(pt, IGX_MASS);           // W = gw0*gw1*gw2*|det J| on the resident Gauss slab
    if (rc) return rc;
    size_t n_out;
    if ((rc = lv_workspace(pt, &n_out))) return rc;
    (void)hipEventRecord(pt->ctx->ev[6], st);        // device time of the contractions, inputs resident
    int nl = pt->dim;
    rc = launch_load_vector(st, pt, d_fvals, pt->d_fields, d_out, pt->d_lv_t1, pt->d_lv_t2, -1, 0, &nl);
    (void)hipEventRecord(pt->ctx->ev[7], st);
    if (rc) return rc;
    IGX_HIP(hipStreamSynchronize(st));
    memset(&pt->timing, 0, sizeof(pt->timing));
    (void)hipEventElapsedTime(&pt->timing.total_ms, pt->ctx->ev[6], pt->ctx->ev[7]);
    pt->timing.algo_used = 3;                        // load vector
    pt->timing.n_launches = nl;
    return IGX_OK;
}

int igx_patch_last_path(const igx_patch *pt) { return pt ? pt->last_path : 0; }

int igx_patch_gauss_slab(const igx_patch *pt, int64_t *g0_lo, int64_t *g0_n)
{
    if (!pt) { set_error("igx_patch_gauss_slab: null patch"); return IGX_ERR_ARG; }
    if (g0_lo) *g0_lo = pt->dev.g0_lo;
    if (g0_n) *g0_n = pt->dev.G0_loc;
    return IGX_OK;
}

int igx_load_vector(igx_patch *pt, const double *fvals, double *out)
{
    if (!pt || !fvals || !out) { set_error("igx_load_vector: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_load_vector: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    const PatchDev &pd = pt->dev;
    const size_t npts = (size_t)pd.npts_loc, plane = npts / (size_t)pd.G0_loc;
    int rc;
    size_t n_out;
    if ((rc = ws_reserve(&pt->d_lv_f, &pt->lv_f_cap, npts, "function values"))) return rc;
    if ((rc = lv_workspace(pt, &n_out))) return rc;
    // the function values of the resident Gauss planes are contiguous in the full-grid array (axis 0 is slowest)
    IGX_HIP(hipMemcpyAsync(pt->d_lv_f, fvals + (size_t)pd.g0_lo * plane, npts * sizeof(double), hipMemcpyHostToDevice, st));
    if ((rc = igx_load_vector_d(pt, pt->d_lv_f, pt->d_lv_o))) return rc;
    IGX_HIP(hipMemcpyAsync(out, pt->d_lv_o, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    IGX_HIP(hipStreamSynchronize(st));
    return IGX_OK;
}

// device buffers that stay resident between calls (function values, index pairs, results)
void *igx_dev_alloc(igx_ctx *ctx, size_t bytes)
{
    if (!ctx) { set_error("igx_dev_alloc: null context"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { set_error("hipSetDevice failed"); return nullptr; }
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(1, bytes)) != hipSuccess) { set_error("hipMalloc of %.3f GB failed", bytes / 1e9); return nullptr; }
    return p;
}

void igx_dev_free(igx_ctx *ctx, void *d_ptr)
{
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ptr);
}

int igx_dev_upload(igx_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !src))) { set_error("igx_dev_upload: null argument"); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(ctx->device));
    IGX_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    IGX_HIP(hipStreamSynchronize(ctx->stream));
    return IGX_OK;
}

int igx_dev_download(igx_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !d_src))) { set_error("igx_dev_download: null argument"); return IGX_ERR_ARG; }
    IGX_HIP(hipSetDevice(ctx->device));
    IGX_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    IGX_HIP(hipStreamSynchronize(ctx->stream));
    return IGX_OK;
}

int igx_load_vector_jet(igx_patch *pt, const double *const coef[4], double *out)
{
    if (!pt || !coef || !out) { set_error("igx_load_vector_jet: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_load_vector_jet: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    if (int rcb = basis_orders_ok(pt, IGX_MASS, "igx_load_vector_jet")) return rcb;
    const int dim = pt->dim;
    for (int r = dim + 1; r < 4; ++r)
        if (coef[r]) { set_error("igx_load_vector_jet: coefficient %d does not exist in %dD", r, dim); return IGX_ERR_ARG; }
    // the linear functional  sum_r F_r D_r v  is column 0 of a jet form: its parametric coefficients
    // G_a = W * sum_r T[a][r] F_r come out of the same field kernel (fields_form)
    const double *table[16];
    for (int k = 0; k < 16; ++k) table[k] = nullptr;
    bool any = false;
    for (int r = 0; r < 4; ++r) { table[4 * r] = coef[r]; any = any || coef[r]; }
    if (!any) { set_error("igx_load_vector_jet: all coefficients are absent"); return IGX_ERR_ARG; }
    int rc = igx_patch_set_form(pt, table);
    if (rc) return rc;
    return load_vector_jet_run(pt, out);
}

// the same functional with its coefficients given as C expressions in x, y, z (igx_patch_set_form_expr): nothing sampled on the host
int igx_load_vector_jet_expr(igx_patch *pt, const char *const expr[4], double *out, int *cache_hit)
{
    if (!pt || !expr || !out) { set_error("igx_load_vector_jet_expr: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_load_vector_jet_expr: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    if (int rcb = basis_orders_ok(pt, IGX_MASS, "igx_load_vector_jet_expr")) return rcb;
    for (int r = pt->dim + 1; r < 4; ++r)
        if (expr[r]) { set_error("igx_load_vector_jet_expr: coefficient %d does not exist in %dD", r, pt->dim); return IGX_ERR_ARG; }
    const char *table[16];
    for (int k = 0; k < 16; ++k) table[k] = nullptr;
    for (int r = 0; r < 4; ++r) table[4 * r] = expr[r];
    int rc = igx_patch_set_form_expr(pt, table, cache_hit);
    if (rc) return rc;
    return load_vector_jet_run(pt, out);
}

int igx_load_vector_expr(igx_patch *pt, const char *expr, int parametric, double *out, int *cache_hit)
{
    if (!pt || !expr || !out) { set_error("igx_load_vector_expr: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_load_vector_expr: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    if (int rcb = basis_orders_ok(pt, IGX_MASS, "igx_load_vector_expr")) return rcb;
    // does the fused contraction kernel serve this patch, and is there a run-time compiler?  Asked BEFORE the weight field and
    // the workspaces are touched: IGX_ERR_UNSUPPORTED = shape not served (the caller takes igx_patch_eval_expr_d +
    // igx_load_vector_d), IGX_ERR_NORTC = no libhiprtc on this box (the caller samples the function itself)
    if (int rca = lv12_expr_applicable(pt, parametric)) return rca;
    IGX_HIP(hipSetDevice(pt->ctx->device));
    hipStream_t st = pt->ctx->stream;
    int rc = ensure_fields(pt, IGX_MASS);           // W = gw0*gw1*gw2*|det J| on the resident Gauss slab (kept between calls)
    if (rc) return rc;
    size_t n_out;
    if ((rc = lv_workspace(pt, &n_out))) return rc;
    (void)hipEventRecord(pt->ctx->ev[6], st);
    if ((rc = launch_lv12_expr(st, pt, expr, parametric, pt->d_fields, pt->d_lv_t2, cache_hit))) return rc;
    if ((rc = launch_lv_axis0(st, pt, pt->d_lv_t2, pt->d_lv_o, 0, 0))) return rc;
    (void)hipEventRecord(pt->ctx->ev[7], st);
    IGX_HIP(hipMemcpyAsync(out, pt->d_lv_o, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    IGX_HIP(hipStreamSynchronize(st));
    memset(&pt->timing, 0, sizeof(pt->timing));
    (void)hipEventElapsedTime(&pt->timing.total_ms, pt->ctx->ev[6], pt->ctx->ev[7]);
    pt->timing.algo_used = 3;
    pt->timing.n_launches = 2;
    return IGX_OK;
}

int igx_rtc_compile_load_vector(int P, int npass, int parametric, const char *expr, const char *arch, char *path_out, int path_len, int *cache_hit)
{
    return rtc_compile_lv12(P, npass, parametric, expr, arch, path_out, path_len, cache_hit);
}

} // extern "C"

// the contractions of a jet functional whose coefficients are the column 0 of the patch's form
static int igx::load_vector_jet_run(igx_patch *pt, double *out)
{
    const int dim = pt->dim;
    int rc = ensure_fields(pt, IGX_FORM);
    if (rc) return rc;
    hipStream_t st = pt->ctx->stream;
    const PatchDev &pd = pt->dev;
    const size_t npts = (size_t)pd.npts_loc;
    size_t n_out;
    if ((rc = lv_workspace(pt, &n_out))) return rc;
    double *d_t1 = pt->d_lv_t1, *d_t2 = pt->d_lv_t2, *d_o = pt->d_lv_o;
    auto cleanup = [&]() {};
    for (int k = 0; k < pd.form_n && rc == IGX_OK; ++k) {
        const int a = pd.form_ab[k] >> 2;                  // jet index of v; derivative a >= 1 acts on grid axis dim - a
        rc = launch_load_vector(st, pt, pt->d_fields + (size_t)k * npts, nullptr, d_o, d_t1, d_t2, a >= 1 ? dim - a : -1, k > 0);
    }
    hipError_t e = hipSuccess;
    if (rc == IGX_OK) e = hipMemcpyAsync(out, d_o, n_out * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) { set_error("igx_load_vector_jet: %s", hipGetErrorString(e)); return IGX_ERR_HIP; }
    return rc;
}


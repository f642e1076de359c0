// B-spline basis evaluation, geometry Jacobians and quadrature fields on the device.
//
//   findspan / active_deriv   restate pyiga/bspline_cy.pyx:13-27, 42-121 (Piegl-Tiller A2.3)
//   geometry evaluation       restates BSplineFunc.grid_eval/grid_jacobian (pyiga/bspline.py:874-921)
//                             and the NURBS quotient rule (pyiga/geometry.py:17-25,116-123)
//   fields                    restate precompute_fields (pyiga/assemblers.pyx:86-110,234-275,
//                             1223-1249,1389-1449): W = gw*|det J|,  B = W * Jinv Jinv^T
#include "igx_internal.h"
#include "geo_device.h"
#include <algorithm>

namespace igx {

// ---------------------------------------------------------------------------------------------
__device__ inline int dev_findspan(const double *kv, int n, int p, double u)
{
    if (u >= kv[n - p - 1]) return n - p - 2;      // last interval
    int a = 0, b = n - 1;
    while (b - a > 1) {
        int c = a + (b - a) / 2;
        if (kv[c] > u) b = c; else a = c;
    }
    return a;
}

// All active basis functions and derivatives up to `nd` at u.  res(k, r) for k<=nd, r<=p.
template <class Store>
__device__ inline int dev_active_deriv(const double *kv, int nk, int p, double u, int nd, Store res)
{
    double NDU[MAXP][MAXP];
    double left[MAXP], right[MAXP], abuf[2][MAXP + 1];
    const int span = dev_findspan(kv, nk, p, u);
    NDU[0][0] = 1.0;
    for (int j = 1; j <= p; ++j) {
        left[j - 1] = u - kv[span + 1 - j];
        right[j - 1] = kv[span + j] - u;
        double saved = 0.0;
        for (int r = 0; r < j; ++r) {
            NDU[j][r] = right[r] + left[j - r - 1];
            double temp = NDU[r][j - 1] / NDU[j][r];
            NDU[r][j] = saved + right[r] * temp;
            saved = left[j - r - 1] * temp;
        }
        NDU[j][j] = saved;
    }
    for (int j = 0; j <= p; ++j) res(0, j, NDU[j][p]);
    for (int r = 0; r <= p; ++r) {
        int s1 = 0, s2 = 1;
        abuf[0][0] = 1.0;
        int fac = p;
        for (int k = 1; k <= nd; ++k) {
            double *a1 = abuf[s1], *a2 = abuf[s2];
            const int rk = r - k, pk = p - k;
            double d = 0.0;
            if (pk < 0) { res(k, r, 0.0); continue; }
            if (r >= k) {
                a2[0] = a1[0] / NDU[pk + 1][rk];
                d = a2[0] * NDU[rk][pk];
            }
            const int j1 = (rk >= -1) ? 1 : -rk;
            const int j2 = (r - 1 <= pk) ? k - 1 : p - r;
            for (int j = j1; j <= j2; ++j) {
                a2[j] = (a1[j] - a1[j - 1]) / NDU[pk + 1][rk + j];
                d += a2[j] * NDU[rk + j][pk];
            }
            if (r <= pk) {
                a2[k] = -a1[k - 1] / NDU[pk + 1][r];
                d += a2[k] * NDU[r][pk];
            }
            res(k, r, d * fac);
            fac *= pk;
            int t = s1; s1 = s2; s2 = t;
        }
    }
    return span;
}

// V slots: (value, first derivative) unless the patch assembles a parametric jet form of higher order
// (igx_patch_set_basis_orders): then slot s holds the derivative of order os.
__global__ void k_basis_tables(const double *kv, int nk, int p, const double *u, size_t nu, int nd,
                               double *out, double *V, int *fa, long long *spans, int o0, int o1)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nu) return;
    const int P = p + 1;
    double *o = out, *v = V;
    int span = dev_active_deriv(kv, nk, p, u[i], nd, [=](int k, int r, double val) {
        if (o) o[((size_t)k * P + r) * nu + i] = val;
        if (v && k == o0) v[(i * P + r) * 2] = val;
        if (v && k == o1) v[(i * P + r) * 2 + 1] = val;
    });
    if (v && nd < o1)
        for (int r = 0; r < P; ++r) v[(i * P + r) * 2 + 1] = 0.0;
    if (fa) fa[i] = span - p;
    if (spans) spans[i] = span;
}

int launch_basis_tables(hipStream_t st, const double *d_kv, int nk, int p, const double *d_u, size_t nu,
                        int numderiv, double *d_out, double *d_V, int *d_fa, long long *d_spans, int o0, int o1)
{
    if (nu == 0) return IGX_OK;
    const int bs = 64;
    k_basis_tables<<<dim3((unsigned)((nu + bs - 1) / bs)), dim3(bs), 0, st>>>(d_kv, nk, p, d_u, nu, numderiv,
                                                                              d_out, d_V, d_fa, d_spans, o0, o1);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// PI[g][t][a][b] = V[g][b][tu] * V[g][a][tv],  t = tu + 2*tv
__global__ void k_pi_tables(const double *V, int G, int P, double *PI)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)G * 4 * P * P;
    if (idx >= total) return;
    int b = idx % P;
    int a = (idx / P) % P;
    int t = (idx / ((size_t)P * P)) % 4;
    size_t g = idx / ((size_t)4 * P * P);
    int tu = t & 1, tv = t >> 1;
    PI[idx] = V[(g * P + b) * 2 + tu] * V[(g * P + a) * 2 + tv];
}

int launch_pi_tables(hipStream_t st, const double *d_V, int G, int P, double *d_PI)
{
    size_t total = (size_t)G * 4 * P * P;
    k_pi_tables<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(d_V, G, P, d_PI);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// ---------------------------------------------------------------------------------------------
// geometry: homogeneous spline value + parametric derivatives at one grid point.
// jac[c][k]: component c, derivative along GRID AXIS k (not yet reordered to x,y,z).

template <int DIM>
__device__ inline void eval_geo(const GeoView &gv, const int g[3], double val[MAX_COMP], double jac[MAX_COMP][3])
{
    const int nc = gv.nc;
    for (int c = 0; c < MAX_COMP; ++c) {
        val[c] = 0.0;
        for (int k = 0; k < 3; ++k) jac[c][k] = 0.0;
    }
    const double *V0 = gv.V[0] + (size_t)g[0] * gv.P[0] * 2;
    const double *V1 = gv.V[1] + (size_t)g[1] * gv.P[1] * 2;
    const int f0 = gv.fa[0][g[0]], f1 = gv.fa[1][g[1]];
    if (DIM == 2) {
        for (int a0 = 0; a0 < gv.P[0]; ++a0) {
            double sv[MAX_COMP], sd[MAX_COMP];
            for (int c = 0; c < MAX_COMP; ++c) sv[c] = sd[c] = 0.0;
            const double *row = gv.ctrl + ((size_t)(f0 + a0) * gv.N[1] + f1) * nc;
            for (int a1 = 0; a1 < gv.P[1]; ++a1) {
                const double n1 = V1[a1 * 2], d1 = V1[a1 * 2 + 1];
                for (int c = 0; c < nc; ++c) {
                    const double cf = row[(size_t)a1 * nc + c];
                    sv[c] += n1 * cf;
                    sd[c] += d1 * cf;
                }
            }
            const double n0 = V0[a0 * 2], d0 = V0[a0 * 2 + 1];
            for (int c = 0; c < nc; ++c) {
                val[c] += n0 * sv[c];
                jac[c][0] += d0 * sv[c];
                jac[c][1] += n0 * sd[c];
            }
        }
    } else {
        const double *V2 = gv.V[2] + (size_t)g[2] * gv.P[2] * 2;
        const int f2 = gv.fa[2][g[2]];
        for (int a0 = 0; a0 < gv.P[0]; ++a0) {
            double tv[MAX_COMP], t1[MAX_COMP], t2[MAX_COMP];
            for (int c = 0; c < MAX_COMP; ++c) tv[c] = t1[c] = t2[c] = 0.0;
            for (int a1 = 0; a1 < gv.P[1]; ++a1) {
                double sv[MAX_COMP], sd[MAX_COMP];
                for (int c = 0; c < MAX_COMP; ++c) sv[c] = sd[c] = 0.0;
                const double *row = gv.ctrl + (((size_t)(f0 + a0) * gv.N[1] + (f1 + a1)) * gv.N[2] + f2) * nc;
                for (int a2 = 0; a2 < gv.P[2]; ++a2) {
                    const double n2 = V2[a2 * 2], d2 = V2[a2 * 2 + 1];
                    for (int c = 0; c < nc; ++c) {
                        const double cf = row[(size_t)a2 * nc + c];
                        sv[c] += n2 * cf;
                        sd[c] += d2 * cf;
                    }
                }
                const double n1 = V1[a1 * 2], d1 = V1[a1 * 2 + 1];
                for (int c = 0; c < nc; ++c) {
                    tv[c] += n1 * sv[c];
                    t1[c] += d1 * sv[c];
                    t2[c] += n1 * sd[c];
                }
            }
            const double n0 = V0[a0 * 2], d0 = V0[a0 * 2 + 1];
            for (int c = 0; c < nc; ++c) {
                val[c] += n0 * tv[c];
                jac[c][0] += d0 * tv[c];
                jac[c][1] += n0 * t1[c];
                jac[c][2] += n0 * t2[c];
            }
        }
    }
}

// Physical Jacobian Jm[r][c] = dG_r / d xi_c with c in (x,y,z) order, i.e. c = 0 differentiates
// along the LAST grid axis (pyiga/bspline.py:917-921); NURBS by the quotient rule.
template <int DIM>
__device__ inline void physical_jacobian(const GeoView &gv, bool nurbs, const int g[3], int ncomp,
                                         double Jm[MAX_COMP][3], double ev[MAX_COMP])
{
    double val[MAX_COMP], jac[MAX_COMP][3];
    eval_geo<DIM>(gv, g, val, jac);
    finish_jacobian<DIM>(val, jac, nurbs, ncomp, gv.nc, Jm, ev);
}

template <int DIM>
__global__ void k_grid_geo(GeoView gv, bool nurbs, int ncomp, int G0, int G1, int G2, double *jac_out, double *eval_out)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)G0 * G1 * (DIM == 3 ? G2 : 1);
    if (idx >= total) return;
    int g[3];
    if (DIM == 3) { g[2] = idx % G2; g[1] = (idx / G2) % G1; g[0] = idx / ((long long)G2 * G1); }
    else { g[1] = idx % G1; g[0] = idx / G1; g[2] = 0; }
    double Jm[MAX_COMP][3], ev[MAX_COMP];
    physical_jacobian<DIM>(gv, nurbs, g, ncomp, Jm, ev);
    if (jac_out)
        for (int r = 0; r < ncomp; ++r)
            for (int c = 0; c < DIM; ++c) jac_out[(idx * ncomp + r) * DIM + c] = Jm[r][c];
    if (eval_out)
        for (int r = 0; r < ncomp; ++r) eval_out[idx * ncomp + r] = ev[r];
}


int launch_grid_geo(hipStream_t st, int dim, int ncomp_total, bool nurbs, const GeoAxis gax[3], const int G[3],
                    const double *d_ctrl, double *d_jac, double *d_eval)
{
    GeoView gv = make_view(dim, gax, d_ctrl, ncomp_total);
    const int ncomp = nurbs ? ncomp_total - 1 : ncomp_total;
    long long total = (long long)G[0] * G[1] * (dim == 3 ? G[2] : 1);
    if (total == 0) return IGX_OK;
    dim3 grid((unsigned)((total + 127) / 128)), block(128);
    if (dim == 2) k_grid_geo<2><<<grid, block, 0, st>>>(gv, nurbs, ncomp, G[0], G[1], 1, d_jac, d_eval);
    else k_grid_geo<3><<<grid, block, 0, st>>>(gv, nurbs, ncomp, G[0], G[1], G[2], d_jac, d_eval);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// Affine coefficient  c(x) = c[0] + c[1] x + c[2] y + c[3] z  at the resident Gauss points, from the geometry map on the
// device (the host path samples a Python callable on the whole grid and ships one double per Gauss point).
template <int DIM>
__global__ void k_coeff_affine(GeoView gv, bool nurbs, int g0_lo, int G0loc, int G1, int G2, int b1, int b2, double c0, double c1, double c2, double c3, double *coeff)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)G0loc * G1 * (DIM == 3 ? G2 : 1);
    if (idx >= total) return;
    int g[3];                 // (G1, G2: extents of the resident window of axes 1, 2; b1, b2: its first Gauss indices)
    if (DIM == 3) { g[2] = b2 + idx % G2; g[1] = b1 + (idx / G2) % G1; g[0] = g0_lo + (int)(idx / ((long long)G2 * G1)); }
    else { g[1] = b1 + idx % G1; g[0] = g0_lo + (int)(idx / G1); g[2] = 0; }
    double Jm[MAX_COMP][3], ev[MAX_COMP];
    physical_jacobian<DIM>(gv, nurbs, g, DIM, Jm, ev);
    double v = c0 + c1 * ev[0] + c2 * ev[1];
    if (DIM == 3) v += c3 * ev[2];
    coeff[idx] = v;
}

int launch_coeff_affine(hipStream_t st, const igx_patch *pt, const double c[4], double *d_coeff)
{
    const int dim = pt->dim;
    if (pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("an affine coefficient needs a spline geometry (physical coordinates)"); return IGX_ERR_UNSUPPORTED; }
    GeoView gv = make_view(dim, pt->gax, pt->d_ctrl, pt->ncomp);
    const PatchDev &pd = pt->dev;
    const long long total = pd.npts_loc;
    if (total == 0) return IGX_OK;
    const int G1 = pd.L1, G2 = dim == 3 ? pd.L2 : 1;
    dim3 grid((unsigned)((total + 127) / 128)), block(128);
    const bool nurbs = pt->geo_kind == IGX_GEO_NURBS;
    if (dim == 2) k_coeff_affine<2><<<grid, block, 0, st>>>(gv, nurbs, pd.g0_lo, pd.G0_loc, G1, G2, pd.b1, pd.b2, c[0], c[1], c[2], c[3], d_coeff);
    else k_coeff_affine<3><<<grid, block, 0, st>>>(gv, nurbs, pd.g0_lo, pd.G0_loc, G1, G2, pd.b1, pd.b2, c[0], c[1], c[2], c[3], d_coeff);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// ---------------------------------------------------------------------------------------------
// geo_kind BSPLINE/NURBS: evaluate from the control net; JACOBIAN: read the user array slab.
template <int DIM, bool FORM>
__global__ void k_geo_fields(GeoView gv, int geo_kind, const double *jac_in, const double *coeff, const FormView fv, const PatchDev pd, int kind,
                             const double *w0, const double *w1, const double *w2,
                             int g0_lo, int G0loc, int G1, int G2, double *fields)
{
    const long long total = (long long)G0loc * G1 * (DIM == 3 ? G2 : 1);
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int g[3];                 // (G1, G2: extents of the resident window of axes 1, 2 -- the whole axes unless the patch is boxed)
    if (DIM == 3) { g[2] = pd.b2 + idx % G2; g[1] = pd.b1 + (idx / G2) % G1; g[0] = g0_lo + (int)(idx / ((long long)G2 * G1)); }
    else { g[1] = pd.b1 + idx % G1; g[0] = g0_lo + (int)(idx / G1); g[2] = 0; }
    double t[9];
    double ev[MAX_COMP] = {0.0, 0.0, 0.0, 0.0};
    if (geo_kind == IGX_GEO_JACOBIAN) {
        const double *src = jac_in + idx * (DIM * DIM);
        for (int k = 0; k < DIM * DIM; ++k) t[k] = src[k];
    } else {
        double Jm[MAX_COMP][3];
        physical_jacobian<DIM>(gv, geo_kind == IGX_GEO_NURBS, g, DIM, Jm, ev);
        for (int r = 0; r < DIM; ++r)
            for (int c = 0; c < DIM; ++c) t[r * DIM + c] = Jm[r][c];
    }
    double GW = w0[g[0]] * w1[g[1]];
    if (DIM == 3) GW = GW * w2[g[2]];
    // FORM is a separate instantiation: its 4x4 products would cost the other kinds a third of their occupancy
    if (FORM) fields_form<DIM>(t, GW, fv, pd.form_n, pd.form_ab, pd.form_par, fields, total, idx);
    else if (DIM == 3 && kind == IGX_CONVDIFF) fields_convdiff(t, GW, ev, coeff[idx], fields, total, idx);
    else fields_from_jac<DIM>(t, GW, kind, fields, total, idx);
}

// Line-wise variant for spline geometries: the tensor-product structure of the geometry map is
// exploited per grid line.  A block first contracts the control net with the basis functions of
// the leading axes for its line(s) -> "line coefficients" Lc[cL][comp][{value, d/d axis0, d/d axis1}]
// in LDS, then each thread evaluates its point with only (pL+1) * ncomp * (DIM+1) FMAs.
// (The per-point kernel above costs prod(p_k+1) * ncomp * (DIM+1) FMAs and is memory-latency bound
// on the control-net gathers.)
template <int DIM, int NC, bool FORM>
__global__ void __launch_bounds__(256) k_geo_fields_lines(GeoView gv, bool nurbs, int kind, const double *coeff, const FormView fv, const PatchDev pd,
                                                          const double *w0, const double *w1, const double *w2,
                                                          int g0_lo, int G0loc, int G1, int G2, int LPB, double *fields)
{
    extern __shared__ double Lc[];                        // [LPB][NgL][nc][DIM]
    const int LN = (DIM == 3) ? G2 : G1;                  // line length (last axis)
    const long long nlines = (DIM == 3) ? (long long)G0loc * G1 : G0loc;
    const long long total = nlines * LN;
    constexpr int nc = NC;                               // components incl. the NURBS weight (compile time: clean unrolling)
    const int NgL = gv.N[DIM - 1];
    const long long line0 = (long long)blockIdx.x * LPB;
    const int per_line = NgL * nc;

    // ---- phase 1: line coefficients
    for (int w = threadIdx.x; w < LPB * per_line; w += blockDim.x) {
        const int ll = w / per_line, rem = w - ll * per_line;
        const int cL = rem / nc, c = rem - cL * nc;
        const long long line = line0 + ll;
        if (line >= nlines) continue;
        double sv = 0.0, s0 = 0.0, s1 = 0.0;
        if (DIM == 2) {
            const int g0 = g0_lo + (int)line;
            const double *V0 = gv.V[0] + (size_t)g0 * gv.P[0] * 2;
            const int f0 = gv.fa[0][g0];
            for (int a0 = 0; a0 < gv.P[0]; ++a0) {
                const double cf = gv.ctrl[((size_t)(f0 + a0) * gv.N[1] + cL) * nc + c];
                sv += V0[a0 * 2] * cf;
                s0 += V0[a0 * 2 + 1] * cf;
            }
        } else {
            const int g0 = g0_lo + (int)(line / G1), g1 = (int)(line % G1);
            const double *V0 = gv.V[0] + (size_t)g0 * gv.P[0] * 2;
            const double *V1 = gv.V[1] + (size_t)g1 * gv.P[1] * 2;
            const int f0 = gv.fa[0][g0], f1 = gv.fa[1][g1];
            for (int a0 = 0; a0 < gv.P[0]; ++a0) {
                double tv = 0.0, t1 = 0.0;
                for (int a1 = 0; a1 < gv.P[1]; ++a1) {
                    const double cf = gv.ctrl[(((size_t)(f0 + a0) * gv.N[1] + (f1 + a1)) * gv.N[2] + cL) * nc + c];
                    tv += V1[a1 * 2] * cf;
                    t1 += V1[a1 * 2 + 1] * cf;
                }
                sv += V0[a0 * 2] * tv;
                s0 += V0[a0 * 2 + 1] * tv;
                s1 += V0[a0 * 2] * t1;
            }
        }
        double *dst = Lc + (size_t)w * DIM;
        dst[0] = sv; dst[1] = s0;
        if (DIM == 3) dst[2] = s1;
    }
    __syncthreads();

    // ---- phase 2: points of the line(s)
    const int PL = gv.P[DIM - 1];
    const int npts_blk = LPB * LN;
    for (int t = threadIdx.x; t < npts_blk; t += blockDim.x) {
        const int ll = t / LN, gL = t - ll * LN;
        const long long line = line0 + ll;
        if (line >= nlines) break;
        const double *VL = gv.V[DIM - 1] + (size_t)gL * PL * 2;
        const int fL = gv.fa[DIM - 1][gL];
        double val[MAX_COMP], jac[MAX_COMP][3];
#pragma unroll
        for (int c = 0; c < MAX_COMP; ++c) { val[c] = 0.0; jac[c][0] = jac[c][1] = jac[c][2] = 0.0; }
        const double *lc = Lc + ((size_t)ll * NgL + fL) * nc * DIM;
        for (int aL = 0; aL < PL; ++aL) {
            const double n = VL[aL * 2], d = VL[aL * 2 + 1];
#pragma unroll
            for (int c = 0; c < nc; ++c) {
                const double *e = lc + ((size_t)aL * nc + c) * DIM;
                val[c] += n * e[0];
                jac[c][0] += n * e[1];
                if (DIM == 3) jac[c][1] += n * e[2];
                jac[c][DIM - 1] += d * e[0];
            }
        }
        double Jm[MAX_COMP][3], ev[MAX_COMP];
        finish_jacobian<DIM>(val, jac, NC == DIM + 1, DIM, NC, Jm, ev);
        double tt[9];
        for (int r = 0; r < DIM; ++r)
            for (int c = 0; c < DIM; ++c) tt[r * DIM + c] = Jm[r][c];
        int g0, g1;
        if (DIM == 3) { g0 = g0_lo + (int)(line / G1); g1 = (int)(line % G1); }
        else { g0 = g0_lo + (int)line; g1 = gL; }
        double GW = w0[g0] * w1[g1];
        if (DIM == 3) GW = GW * w2[gL];
        if (FORM) fields_form<DIM>(tt, GW, fv, pd.form_n, pd.form_ab, pd.form_par, fields, total, line * LN + gL);
        else if (DIM == 3 && kind == IGX_CONVDIFF) fields_convdiff(tt, GW, ev, coeff[line * LN + gL], fields, total, line * LN + gL);
        else fields_from_jac<DIM>(tt, GW, kind, fields, total, line * LN + gL);
    }
}

int launch_geo_fields(hipStream_t st, const igx_patch *pt, int kind, double *d_fields)
{
    const int dim = pt->dim;
    GeoView gv = make_view(dim, pt->gax, pt->d_ctrl, pt->ncomp);
    const PatchDev &pd = pt->dev;
    const long long total = pd.npts_loc;
    if (total == 0) return IGX_OK;
    FormView fv{};
    fv.c = pt->d_formc;
    for (int k = 0; k < 16; ++k) fv.slot[k] = pt->form_slot[k];
    const int G1 = pd.L1, G2 = (dim == 3) ? pd.L2 : 1;
    if (pt->geo_kind != IGX_GEO_JACOBIAN && !pt->boxed) {      // (a boxed patch is small: the point-wise kernel)
        const int LN = (dim == 3) ? G2 : G1;
        // lines per block: a whole number of 256-thread passes over the block's points where a few lines give one
        // (576-point lines: 4 lines = 9 passes; one line would leave the third pass a quarter full)
        int LPB = std::max(1, 256 / LN);
        if (LN > 256)
            for (int l = 1; l <= 8; ++l)
                if ((l * LN) % 256 == 0) { LPB = l; break; }
        const size_t lds = (size_t)LPB * pt->gax[dim - 1].N * pt->ncomp * dim * sizeof(double);
        if (lds <= 64 * 1024) {
            const long long nlines = total / LN;
            dim3 grid((unsigned)((nlines + LPB - 1) / LPB)), block(256);
            const bool nurbs = pt->geo_kind == IGX_GEO_NURBS;
#define LAUNCH_LINES(D_, NC_) do { \
                if (kind == IGX_FORM) k_geo_fields_lines<D_, NC_, true><<<grid, block, lds, st>>>(gv, nurbs, kind, pt->d_coeff, fv, pd, pd.ax[0].w, pd.ax[1].w, D_ == 3 ? pd.ax[2].w : nullptr, pd.g0_lo, pd.G0_loc, G1, G2, LPB, d_fields); \
                else k_geo_fields_lines<D_, NC_, false><<<grid, block, lds, st>>>(gv, nurbs, kind, pt->d_coeff, fv, pd, pd.ax[0].w, pd.ax[1].w, D_ == 3 ? pd.ax[2].w : nullptr, pd.g0_lo, pd.G0_loc, G1, G2, LPB, d_fields); } while (0)
            if (dim == 2) { if (nurbs) LAUNCH_LINES(2, 3); else LAUNCH_LINES(2, 2); }
            else { if (nurbs) LAUNCH_LINES(3, 4); else LAUNCH_LINES(3, 3); }
#undef LAUNCH_LINES
            IGX_HIP(hipGetLastError());
            return IGX_OK;
        }
    }
    dim3 grid((unsigned)((total + 127) / 128)), block(128);
#define LAUNCH_PTS(D_, F_) k_geo_fields<D_, F_><<<grid, block, 0, st>>>(gv, pt->geo_kind, pt->d_jac, pt->d_coeff, fv, pd, kind, pd.ax[0].w, pd.ax[1].w, \
                                                                    D_ == 3 ? pd.ax[2].w : nullptr, pd.g0_lo, pd.G0_loc, G1, G2, d_fields)
    if (dim == 2) { if (kind == IGX_FORM) LAUNCH_PTS(2, true); else LAUNCH_PTS(2, false); }
    else { if (kind == IGX_FORM) LAUNCH_PTS(3, true); else LAUNCH_PTS(3, false); }
#undef LAUNCH_PTS
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}


// ---------------------------------------------------------------------------------------------
// 2D mass / stiffness in ONE launch (BASELINE configs 1, 2; pyiga/assemblers.pyx:86-135,234-349 fields + combine).
// The stage kernels need three dependent launches and two HBM intermediates for a problem whose whole output is a few
// MB: launch-bound.  Here a block of 1024 threads owns R0 x R1 rows (dofs i0 x i1) and keeps everything in LDS:
//   0. every table it needs (basis values of both axes on its Gauss window, supports, first active functions, pair
//      ranges) is staged ONCE -- the phases below touch no global memory except the control net and the output;
//   1. fields (W, or the upper triangle of W J^-1 J^-T) on the window: the geometry map line-wise (control net contracted
//      with the axis-0 basis per Gauss plane first, then (p_g + 1) (nc) products per point);
//   2. axis-0 sweep  K1[r][j0][y][g1] = sum_g0 (V0[j0][tu] * V0[i0][tv]) * field_y(g0, g1)  for all rows of the tile, every
//      column partner j0 (the FULL window: no mirror pass) and the terms y of the form;
//   3. contraction along axis 1 for the block's entries, written to their (contiguous) CSR positions:
//      entry = (T0 + T3) + (T1 + T2),  T_y = sum_g1 (V1[j1][tu] * V1[i1][tv]) * K1[r][j0][y][g1].
// Three barriers per block.  Every product of two basis values is formed once and the sums run over the support
// intersection in ascending order, so entry (j, i) repeats the arithmetic of entry (i, j) with T1 and T2 exchanged: the
// matrix is symmetric bit for bit.
struct Single2DArgs {
    PatchDev pd;
    GeoView gv;
    int geo_kind;
    const double *jac;         // IGX_GEO_JACOBIAN: resident slab of the user array
    int R0, R1, NG0, WIN;      // rows per block, LDS extents (planes of axis 0, window points of axis 1)
    int NCOL;                  // geometry control columns of axis 1 a window can touch (LDS extent of the line coefficients)
    double *data;
};

template <int KIND>
__global__ void __launch_bounds__(1024) k_single2d(const Single2DArgs A)
{
    constexpr int NF = KIND == IGX_MASS ? 1 : 3, NY = KIND == IGX_MASS ? 1 : 4;
    // term y: field, axis-0 type of u / v (0 value, 1 derivative); axis 1: y = 0 (d, d), 1 (v, d), 2 (d, v), 3 (v, v)
    extern __shared__ double lds[];
    const PatchDev &pd = A.pd;
    const AxisDev &A0 = pd.ax[0], &A1 = pd.ax[1];
    const int q = A0.q, P0 = A0.P, P1 = A1.P, p0 = P0 - 1, p1 = P1 - 1, NG0 = A.NG0, WIN = A.WIN, C0M = 2 * P0 - 1, NCOL = A.NCOL;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int i0a = pd.r0_lo + blockIdx.y * A.R0, i0b = min(i0a + A.R0, pd.r0_hi), nr0 = i0b - i0a;
    const int i1a = blockIdx.x * A.R1, i1b = min(i1a + A.R1, A1.N), nr1 = i1b - i1a;
    const int s0b = A0.mslo[i0a], g0b = s0b * q, nG0 = A0.mshi[i0b - 1] * q - g0b;
    const int s1b = A1.mslo[i1a], g1b = s1b * q, w1 = A1.mshi[i1b - 1] * q - g1b;
    // dofs whose tables are staged: the tile's rows and their column partners
    const int d0lo = max(i0a - p0, 0), nd0 = min(i0b + p0, A0.N) - d0lo;
    const int d1lo = max(i1a - p1, 0), nd1 = min(i1b + p1, A1.N) - d1lo;
    double *fld = lds;                                   // [NF][NG0][WIN]
    double *K1 = fld + NF * NG0 * WIN;                   // [R0][C0M][NY][WIN]
    double *V0s = K1 + A.R0 * C0M * NY * WIN;            // [NG0][P0][2]
    double *V1s = V0s + NG0 * P0 * 2;                    // [WIN][P1][2]
    double *Lc = V1s + WIN * P1 * 2;                     // [NG0][NCOL][MAX_COMP][2]: (value, d/d axis 0) of the net contracted along axis 0
    int *it = (int *)(Lc + NG0 * NCOL * MAX_COMP * 2);
    int *ms0 = it, *me0 = ms0 + (A.R0 + 2 * p0), *fa0 = me0 + (A.R0 + 2 * p0);          // supports of the staged dofs, first active per span
    int *jl0 = fa0 + (A.R0 + p0), *rp0s = jl0 + 2 * A.R0;                                    // (jlo, count) and rp0 of the tile's rows
    int *ms1 = rp0s + A.R0, *me1 = ms1 + (A.R1 + 2 * p1), *fa1 = me1 + (A.R1 + 2 * p1);
    int *jl1 = fa1 + (A.R1 + p1), *rp1s = jl1 + 2 * A.R1;
    // ---- 0. staging
    for (int i = tid; i < nG0 * P0 * 2; i += NT) V0s[i] = A0.V[(size_t)g0b * P0 * 2 + i];
    for (int i = tid; i < w1 * P1 * 2; i += NT) V1s[i] = A1.V[(size_t)g1b * P1 * 2 + i];
    for (int i = tid; i < nd0; i += NT) { ms0[i] = A0.mslo[d0lo + i]; me0[i] = A0.mshi[d0lo + i]; }
    for (int i = tid; i < nd1; i += NT) { ms1[i] = A1.mslo[d1lo + i]; me1[i] = A1.mshi[d1lo + i]; }
    for (int i = tid; i < nG0 / q; i += NT) fa0[i] = A0.fa[s0b + i];
    for (int i = tid; i < w1 / q; i += NT) fa1[i] = A1.fa[s1b + i];
    for (int i = tid; i < nr0; i += NT) { jl0[2 * i] = A0.jlo[i0a + i]; jl0[2 * i + 1] = A0.jhi[i0a + i] - A0.jlo[i0a + i]; rp0s[i] = A0.rp[i0a + i]; }
    for (int i = tid; i < nr1; i += NT) { jl1[2 * i] = A1.jlo[i1a + i]; jl1[2 * i + 1] = A1.jhi[i1a + i] - A1.jlo[i1a + i]; rp1s[i] = A1.rp[i1a + i]; }
    // ---- 1a. geometry: the control net contracted with the axis-0 basis of every plane of the window
    const bool spline = A.geo_kind != IGX_GEO_JACOBIAN;
    const int nc = A.gv.nc;
    int colb = 0, ncol = 0;
    if (spline) {
        colb = A.gv.fa[1][g1b];
        ncol = A.gv.fa[1][g1b + w1 - 1] + A.gv.P[1] - colb;
        for (int i = tid; i < nG0 * ncol * nc; i += NT) {
            const int a = i / (ncol * nc), rem = i - a * (ncol * nc), col = rem / nc, c = rem - col * nc;
            const int g0 = g0b + a, f0 = A.gv.fa[0][g0];
            const double *Vg = A.gv.V[0] + (size_t)g0 * A.gv.P[0] * 2;
            double sv = 0.0, sd = 0.0;
            for (int a0 = 0; a0 < A.gv.P[0]; ++a0) {
                const double cf = A.gv.ctrl[((size_t)(f0 + a0) * A.gv.N[1] + (colb + col)) * nc + c];
                sv += Vg[a0 * 2] * cf;
                sd += Vg[a0 * 2 + 1] * cf;
            }
            double *dst = Lc + ((size_t)(a * NCOL + col) * MAX_COMP + c) * 2;
            dst[0] = sv; dst[1] = sd;
        }
    }
    __syncthreads();
    // ---- 1b. fields on the window
    for (int idx = tid; idx < nG0 * w1; idx += NT) {
        const int a = idx / w1, b = idx - a * w1;
        const int g0 = g0b + a, g1 = g1b + b;
        double t[9];
        if (!spline) {
            const double *src = A.jac + ((size_t)(g0 - pd.g0_lo) * A1.G + g1) * 4;
            t[0] = src[0]; t[1] = src[1]; t[2] = src[2]; t[3] = src[3];
        } else {
            const double *Vg = A.gv.V[1] + (size_t)g1 * A.gv.P[1] * 2;
            const double *lc = Lc + (size_t)(a * NCOL + (A.gv.fa[1][g1] - colb)) * MAX_COMP * 2;
            double val[MAX_COMP], jac[MAX_COMP][3];
#pragma unroll
            for (int c = 0; c < MAX_COMP; ++c) { val[c] = 0.0; jac[c][0] = jac[c][1] = jac[c][2] = 0.0; }
            for (int a1 = 0; a1 < A.gv.P[1]; ++a1) {
                const double n1 = Vg[a1 * 2], d1 = Vg[a1 * 2 + 1];
#pragma unroll
                for (int c = 0; c < MAX_COMP; ++c)
                    if (c < nc) {
                        const double e0 = lc[(a1 * MAX_COMP + c) * 2], e1 = lc[(a1 * MAX_COMP + c) * 2 + 1];
                        val[c] += n1 * e0;
                        jac[c][0] += n1 * e1;             // d / d axis 0
                        jac[c][1] += d1 * e0;             // d / d axis 1
                    }
            }
            double Jm[MAX_COMP][3], ev[MAX_COMP];
            finish_jacobian<2>(val, jac, A.geo_kind == IGX_GEO_NURBS, 2, nc, Jm, ev);
            t[0] = Jm[0][0]; t[1] = Jm[0][1]; t[2] = Jm[1][0]; t[3] = Jm[1][1];
        }
        double f[6];
        fields_values<2>(t, A0.w[g0] * A1.w[g1], KIND, f);
#pragma unroll
        for (int k = 0; k < NF; ++k) fld[(k * NG0 + a) * WIN + b] = f[k];
    }
    __syncthreads();
    // ---- 2. sweep of axis 0 for every row of the tile (a thread: one column partner, one point of axis 1, every term)
    for (int idx = tid; idx < nr0 * C0M * w1; idx += NT) {
        const int rj = idx / w1, b = idx - rj * w1;
        const int r = rj / C0M, jj = rj - r * C0M;
        double acc[NY];
#pragma unroll
        for (int y = 0; y < NY; ++y) acc[y] = 0.0;
        if (jj < jl0[2 * r + 1]) {
            const int i0 = i0a + r, j0 = jl0[2 * r] + jj;
            const int slo = max(ms0[i0 - d0lo], ms0[j0 - d0lo]), shi = min(me0[i0 - d0lo], me0[j0 - d0lo]);
            const double *fp = fld + b;
            for (int s = slo; s < shi; ++s) {
                const int fa = fa0[s - s0b];
                const double *vu = V0s + (size_t)((s * q - g0b) * P0 + (j0 - fa)) * 2;
                const double *vv = V0s + (size_t)((s * q - g0b) * P0 + (i0 - fa)) * 2;
                for (int l = 0; l < q; ++l) {
                    const double u0 = vu[l * P0 * 2], u1 = vu[l * P0 * 2 + 1], v0 = vv[l * P0 * 2], v1 = vv[l * P0 * 2 + 1];
                    const double *f = fp + (size_t)(s * q + l - g0b) * WIN;
                    if (KIND == IGX_MASS) acc[0] = fma(u0 * v0, f[0], acc[0]);
                    else {
                        const double f1 = f[(size_t)NG0 * WIN];
                        acc[0] = fma(u0 * v0, f[0], acc[0]);                          // (U0, W0)[y] = (0,0), (1,0), (0,1), (1,1)
                        acc[1] = fma(u1 * v0, f1, acc[1]);
                        acc[2] = fma(u0 * v1, f1, acc[2]);
                        acc[3] = fma(u1 * v1, f[(size_t)2 * NG0 * WIN], acc[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int y = 0; y < NY; ++y) K1[((size_t)rj * NY + y) * WIN + b] = acc[y];
    }
    __syncthreads();
    // ---- 3. contraction along axis 1: entries of the rows (i0, i1) of the tile
    const long long S1 = A1.S;
    const int C1M = 2 * P1 - 1;
    for (int slot = tid; slot < nr0 * nr1 * C0M * C1M; slot += NT) {
        const int rr = slot / (C0M * C1M), e_ = slot - rr * (C0M * C1M);
        const int r = rr / nr1, r1 = rr - r * nr1;
        const int c0 = jl0[2 * r + 1], c1 = jl1[2 * r1 + 1];
        if (e_ >= c0 * c1) continue;
        const int jj = e_ / c1, i1 = i1a + r1, j1 = jl1[2 * r1] + (e_ - jj * c1);
        const int slo = max(ms1[i1 - d1lo], ms1[j1 - d1lo]), shi = min(me1[i1 - d1lo], me1[j1 - d1lo]);
        const double *kp0 = K1 + (size_t)(r * C0M + jj) * NY * WIN;
        double T[4] = {0.0, 0.0, 0.0, 0.0};
        for (int s = slo; s < shi; ++s) {
            const int fa = fa1[s - s1b];
            for (int l = 0; l < q; ++l) {
                const int b = s * q + l - g1b;
                const double *vi = V1s + (size_t)(b * P1 + (i1 - fa)) * 2, *vj = V1s + (size_t)(b * P1 + (j1 - fa)) * 2;
                const double vj0 = vj[0], vj1 = vj[1], vi0 = vi[0], vi1 = vi[1];
                if (KIND == IGX_MASS) T[0] = fma(vj0 * vi0, kp0[b], T[0]);
                else {
                    T[0] = fma(vj1 * vi1, kp0[b], T[0]);
                    T[1] = fma(vj0 * vi1, kp0[WIN + b], T[1]);
                    T[2] = fma(vj1 * vi0, kp0[2 * WIN + b], T[2]);
                    T[3] = fma(vj0 * vi0, kp0[3 * WIN + b], T[3]);
                }
            }
        }
        A.data[(long long)rp0s[r] * S1 - pd.nnz_off + (long long)c0 * rp1s[r1] + e_] = KIND == IGX_MASS ? T[0] : (T[0] + T[3]) + (T[1] + T[2]);
    }
}

bool single2d_supported(const igx_patch *pt, int kind)
{
    return pt->dim == 2 && (kind == IGX_MASS || kind == IGX_STIFFNESS) && pt->ax[0].q == pt->ax[1].q && pt->ax[0].P <= 6 && pt->ax[1].P <= 6;
}

// tile shape and LDS image of the single-launch kernel for this patch; false if even the smallest tile does not fit.
// The largest tile that fits LDS does the least redundant work (its Gauss window overlaps the neighbours' by p spans per
// side); a small patch is latency-bound instead, so the tile shrinks (down to 2 x 4 rows) while the grid still fits one
// resident round of the chip.
constexpr long long SINGLE2D_ROUND = 256;                // blocks of one round: one per CU
static bool single2d_plan(const igx_patch *pt, int kind, Single2DArgs &A, size_t &bytes, bool whole_patch = false)
{
    igx_patch::S2DPlan &M = pt->s2d[kind == IGX_MASS ? 0 : 1][whole_patch ? 1 : 0];
    if (M.valid) {
        A.R0 = M.R0; A.R1 = M.R1; A.NG0 = M.NG0; A.WIN = M.WIN; A.NCOL = M.NCOL; bytes = M.bytes;
        return M.ok != 0;
    }
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1];
    const int NF = kind == IGX_MASS ? 1 : 3, NY = kind == IGX_MASS ? 1 : 4, q = A0.q;
    const bool spline = pt->geo_kind != IGX_GEO_JACOBIAN;
    const long long nr0 = whole_patch ? A0.N : std::max(pt->r0_hi - pt->r0_lo, 0);
    static const int shapes[][2] = {{8, 8}, {6, 6}, {4, 4}, {2, 4}, {2, 2}, {1, 2}, {1, 1}};
    constexpr int NSHAPE = 7, SMALLEST_WANTED = 3;
    // geometry control columns of axis 1 under the Gauss window of a tile: the functions active at its first node, plus one
    // per geometry knot between its first and last node (the geometry's mesh may be finer than the space's)
    auto geo_columns = [&](int R1) {
        const GeoAxis &g = pt->gax[1];
        int most = 0;
        for (int i1a = 0; i1a < A1.N; i1a += R1) {
            const int i1b = std::min(i1a + R1, A1.N);
            const double x0 = A1.nodes[(size_t)A1.mslo[i1a] * q], x1 = A1.nodes[(size_t)A1.mshi[i1b - 1] * q - 1];
            const long long between = std::upper_bound(g.kv.begin(), g.kv.end(), x1) - std::upper_bound(g.kv.begin(), g.kv.end(), x0);
            most = std::max(most, (int)between + g.P);
        }
        return std::min(most, g.N);
    };
    auto image = [&](int k, Single2DArgs &S) {
        S.R0 = shapes[k][0]; S.R1 = shapes[k][1];
        S.NG0 = (S.R0 + A0.p) * q; S.WIN = (S.R1 + A1.p) * q;
        S.NCOL = spline ? geo_columns(S.R1) : 0;
        const size_t doubles = (size_t)NF * S.NG0 * S.WIN + (size_t)S.R0 * (2 * A0.P - 1) * NY * S.WIN + (size_t)S.NG0 * A0.P * 2 +
                               (size_t)S.WIN * A1.P * 2 + (size_t)S.NG0 * S.NCOL * MAX_COMP * 2;
        const size_t ints = (size_t)3 * (S.R0 + 2 * A0.p) + 3 * S.R0 + (size_t)3 * (S.R1 + 2 * A1.p) + 3 * S.R1 + 16;
        return doubles * sizeof(double) + ints * sizeof(int);
    };
    auto blocks = [&](int k) { return ((A1.N + shapes[k][1] - 1) / shapes[k][1]) * ((nr0 + shapes[k][0] - 1) / shapes[k][0]); };
    int k = 0;
    while (k < NSHAPE && (bytes = image(k, A)) > 150 * 1024) ++k;
    const bool ok = k < NSHAPE;
    if (ok) {
        while (k < SMALLEST_WANTED && blocks(k + 1) <= SINGLE2D_ROUND) ++k;
        bytes = image(k, A);
    }
    M.valid = 1; M.ok = ok; M.R0 = A.R0; M.R1 = A.R1; M.NG0 = A.NG0; M.WIN = A.WIN; M.NCOL = A.NCOL; M.bytes = bytes;
    return ok;
}

long long single2d_blocks(const igx_patch *pt, int kind, int *tile_rows, bool whole_patch)
{
    Single2DArgs A{};
    size_t bytes = 0;
    if (!single2d_plan(pt, kind, A, bytes, whole_patch)) return -1;
    if (tile_rows) *tile_rows = A.R0 * A.R1;
    const long long nr0 = whole_patch ? pt->ax[0].N : std::max(pt->r0_hi - pt->r0_lo, 0);
    return ((pt->ax[1].N + A.R1 - 1) / A.R1) * ((nr0 + A.R0 - 1) / A.R0);
}

int launch_single2d(hipStream_t st, igx_patch *pt, int kind, double *d_data)
{
    const Axis &A1 = pt->ax[1];
    Single2DArgs A{};
    size_t bytes = 0;
    if (!single2d_plan(pt, kind, A, bytes)) { set_error("single-launch 2D kernel: LDS image too large (%zu bytes)", bytes); return IGX_ERR_UNSUPPORTED; }
    A.pd = pt->dev;
    A.gv = make_view(2, pt->gax, pt->d_ctrl, pt->ncomp);
    A.geo_kind = pt->geo_kind; A.jac = pt->d_jac; A.data = d_data;
    const int nr0 = pt->r0_hi - pt->r0_lo;
    if (nr0 <= 0) return IGX_OK;
    dim3 grid((unsigned)((A1.N + A.R1 - 1) / A.R1), (unsigned)((nr0 + A.R0 - 1) / A.R0));
    const int nt = 1024;                                 // the phases of a block are short and serial: many threads, short latency
    if (kind == IGX_MASS) {
        if (bytes > 64 * 1024) IGX_HIP(hipFuncSetAttribute((const void *)k_single2d<IGX_MASS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        k_single2d<IGX_MASS><<<grid, dim3(nt), bytes, st>>>(A);
    } else {
        if (bytes > 64 * 1024) IGX_HIP(hipFuncSetAttribute((const void *)k_single2d<IGX_STIFFNESS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        k_single2d<IGX_STIFFNESS><<<grid, dim3(nt), bytes, st>>>(A);
    }
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

} // namespace igx

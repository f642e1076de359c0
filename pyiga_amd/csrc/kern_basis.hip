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

__global__ void k_basis_tables(const double *kv, int nk, int p, const double *u, size_t nu, int nd,
                               double *out, double *V, int *fa, long long *spans)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nu) return;
    const int P = p + 1;
    double *o = out, *v = V;
    int span = dev_active_deriv(kv, nk, p, u[i], nd, [=](int k, int r, double val) {
        if (o) o[((size_t)k * P + r) * nu + i] = val;
        if (v && k < 2) v[(i * P + r) * 2 + k] = val;
    });
    if (v && nd < 1)
        for (int r = 0; r < P; ++r) v[(i * P + r) * 2 + 1] = 0.0;
    if (fa) fa[i] = span - p;
    if (spans) spans[i] = span;
}

int launch_basis_tables(hipStream_t st, const double *d_kv, int nk, int p, const double *d_u, size_t nu,
                        int numderiv, double *d_out, double *d_V, int *d_fa, long long *d_spans)
{
    if (nu == 0) return IGX_OK;
    const int bs = 64;
    k_basis_tables<<<dim3((unsigned)((nu + bs - 1) / bs)), dim3(bs), 0, st>>>(d_kv, nk, p, d_u, nu, numderiv,
                                                                              d_out, d_V, d_fa, d_spans);
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
    if (FORM) fields_form<DIM>(t, GW, fv, pd.form_n, pd.form_ab, fields, total, idx);
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
        if (FORM) fields_form<DIM>(tt, GW, fv, pd.form_n, pd.form_ab, fields, total, line * LN + gL);
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
// MB: launch-bound.  Here a block owns R0 x R1 rows (dofs i0 x i1) and keeps everything in LDS:
//   1. fields (W, or the upper triangle of W J^-1 J^-T) on the Gauss window of its rows, straight from the geometry;
//   2. per row i0: axis-0 sweep  K1[j0][y][g1] = sum_g0 (V0[j0][tu] * V0[i0][tv]) * field_y(g0, g1)  for every column
//      partner j0 (the FULL window: no mirror pass) and the terms y of the form;
//   3. contraction along axis 1 for the block's entries, written to their (contiguous) CSR positions:
//      entry = (T0 + T3) + (T1 + T2),  T_y = sum_g1 (V1[j1][tu] * V1[i1][tv]) * K1[j0][y][g1].
// Every product of two basis values is formed once and the sums run over the support intersection in ascending order, so
// entry (j, i) repeats the arithmetic of entry (i, j) with T1 and T2 exchanged: the matrix is symmetric bit for bit.
struct Single2DArgs {
    PatchDev pd;
    GeoView gv;
    int geo_kind;
    const double *jac;         // IGX_GEO_JACOBIAN: resident slab of the user array
    int R0, R1, NG0, WIN;      // rows per block, LDS extents (planes of axis 0, window points of axis 1)
    double *data;
};

template <int KIND>
__global__ void __launch_bounds__(1024) k_single2d(const Single2DArgs A)
{
    constexpr int NF = KIND == IGX_MASS ? 1 : 3, NY = KIND == IGX_MASS ? 1 : 4;
    // term y: field, axis-0 type of u / v, axis-1 type of u / v (0 value, 1 derivative)
    // (axis 1: u d d v v... spelled out in step 3: y = 0 (d, d), 1 (v, d), 2 (d, v), 3 (v, v))
    constexpr int YF[4] = {0, 1, 1, 2}, U0[4] = {0, 1, 0, 1}, W0[4] = {0, 0, 1, 1};
    extern __shared__ double lds[];
    const PatchDev &pd = A.pd;
    const AxisDev &A0 = pd.ax[0], &A1 = pd.ax[1];
    const int q = A0.q, P0 = A0.P, P1 = A1.P, NG0 = A.NG0, WIN = A.WIN, C0M = 2 * P0 - 1;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int i0a = pd.r0_lo + blockIdx.y * A.R0, i0b = min(i0a + A.R0, pd.r0_hi);
    const int i1a = blockIdx.x * A.R1, i1b = min(i1a + A.R1, A1.N);
    const int g0b = A0.mslo[i0a] * q, nG0 = A0.mshi[i0b - 1] * q - g0b;
    const int g1b = A1.mslo[i1a] * q, w1 = A1.mshi[i1b - 1] * q - g1b;
    double *fld = lds;                                   // [NF][NG0][WIN]
    double *K1 = fld + NF * NG0 * WIN;                   // [C0M][NY][WIN]
    double *V1s = K1 + C0M * NY * WIN;                   // [WIN][P1][2]
    double *c0s = V1s + WIN * P1 * 2;                    // [C0M][NY][NG0]
    // ---- 1. fields on the window
    for (int idx = tid; idx < nG0 * w1; idx += NT) {
        const int a = idx / w1, b = idx - a * w1;
        const int g[3] = {g0b + a, g1b + b, 0};
        double t[9];
        if (A.geo_kind == IGX_GEO_JACOBIAN) {
            const double *src = A.jac + ((size_t)(g[0] - pd.g0_lo) * A1.G + g[1]) * 4;
            t[0] = src[0]; t[1] = src[1]; t[2] = src[2]; t[3] = src[3];
        } else {
            double Jm[MAX_COMP][3], ev[MAX_COMP];
            physical_jacobian<2>(A.gv, A.geo_kind == IGX_GEO_NURBS, g, 2, Jm, ev);
            t[0] = Jm[0][0]; t[1] = Jm[0][1]; t[2] = Jm[1][0]; t[3] = Jm[1][1];
        }
        double f[6];
        fields_values<2>(t, A0.w[g[0]] * A1.w[g[1]], KIND, f);
#pragma unroll
        for (int k = 0; k < NF; ++k) fld[(k * NG0 + a) * WIN + b] = f[k];
    }
    for (int idx = tid; idx < w1 * P1 * 2; idx += NT) V1s[idx] = A1.V[(size_t)g1b * P1 * 2 + idx];
    __syncthreads();
    const long long S1 = A1.S;
    for (int i0 = i0a; i0 < i0b; ++i0) {
        const int jl = A0.jlo[i0], c0 = A0.jhi[i0] - jl;
        const int slo_i = A0.mslo[i0], shi_i = A0.mshi[i0];
        // ---- 2a. axis-0 coefficients of the row's column partners on their common planes (0 elsewhere)
        for (int idx = tid; idx < c0 * NG0; idx += NT) {
            const int jj = idx / NG0, a = idx - jj * NG0;
            const int j0 = jl + jj, g0 = g0b + a, s = g0 / q;
            const bool on = a < nG0 && s >= max(slo_i, A0.mslo[j0]) && s < min(shi_i, A0.mshi[j0]);
            double vu[2] = {0.0, 0.0}, vv[2] = {0.0, 0.0};
            if (on) {
                const int fa = A0.fa[s];
                const double *u = A0.V + ((size_t)g0 * P0 + (j0 - fa)) * 2, *v = A0.V + ((size_t)g0 * P0 + (i0 - fa)) * 2;
                vu[0] = u[0]; vu[1] = u[1]; vv[0] = v[0]; vv[1] = v[1];
            }
#pragma unroll
            for (int y = 0; y < NY; ++y) c0s[(jj * NY + y) * NG0 + a] = KIND == IGX_MASS ? vu[0] * vv[0] : vu[U0[y]] * vv[W0[y]];
        }
        __syncthreads();
        // ---- 2b. sweep of axis 0
        for (int idx = tid; idx < c0 * NY * w1; idx += NT) {
            const int jy = idx / w1, b = idx - jy * w1;
            const int jj = jy / NY, y = jy - jj * NY;
            const int j0 = jl + jj;
            const int alo = max(slo_i, A0.mslo[j0]) * q - g0b, ahi = min(shi_i, A0.mshi[j0]) * q - g0b;
            const double *cf = c0s + (size_t)jy * NG0;
            const double *fp = fld + (size_t)(KIND == IGX_MASS ? 0 : YF[y]) * NG0 * WIN + b;
            double acc = 0.0;
            for (int a = alo; a < ahi; ++a) acc = fma(cf[a], fp[(size_t)a * WIN], acc);
            K1[(size_t)jy * WIN + b] = acc;
        }
        __syncthreads();
        // ---- 3. contraction along axis 1, entries of the rows (i0, i1a .. i1b)
        const int C1M = 2 * P1 - 1, per_row = c0 * C1M;
        const long long base0 = (long long)A0.rp[i0] * S1 - pd.nnz_off;
        for (int slot = tid; slot < (i1b - i1a) * per_row; slot += NT) {
            const int r = slot / per_row, e = slot - r * per_row;
            const int i1 = i1a + r;
            const int jl1 = A1.jlo[i1], c1 = A1.jhi[i1] - jl1;
            if (e >= c0 * c1) continue;
            const int jj = e / c1, j1 = jl1 + (e - jj * c1);
            const int slo = max(A1.mslo[i1], A1.mslo[j1]), shi = min(A1.mshi[i1], A1.mshi[j1]);
            double T[4] = {0.0, 0.0, 0.0, 0.0};
            for (int s = slo; s < shi; ++s) {
                const int fa = A1.fa[s];
                for (int l = 0; l < q; ++l) {
                    const int b = s * q + l - g1b;
                    const double *vi = V1s + (size_t)(b * P1 + (i1 - fa)) * 2, *vj = V1s + (size_t)(b * P1 + (j1 - fa)) * 2;
                    const double vj0 = vj[0], vj1 = vj[1], vi0 = vi[0], vi1 = vi[1];
                    if (KIND == IGX_MASS) T[0] = fma(vj0 * vi0, K1[(size_t)jj * WIN + b], T[0]);
                    else {
                        const double *kp = K1 + (size_t)jj * NY * WIN + b;
                        T[0] = fma(vj1 * vi1, kp[0], T[0]);
                        T[1] = fma(vj0 * vi1, kp[WIN], T[1]);
                        T[2] = fma(vj1 * vi0, kp[2 * WIN], T[2]);
                        T[3] = fma(vj0 * vi0, kp[3 * WIN], T[3]);
                    }
                }
            }
            A.data[base0 + (long long)c0 * A1.rp[i1] + e] = KIND == IGX_MASS ? T[0] : (T[0] + T[3]) + (T[1] + T[2]);
        }
        __syncthreads();
    }
}

bool single2d_supported(const igx_patch *pt, int kind)
{
    return pt->dim == 2 && (kind == IGX_MASS || kind == IGX_STIFFNESS) && pt->ax[0].q == pt->ax[1].q && pt->ax[0].P <= 6 && pt->ax[1].P <= 6;
}

int launch_single2d(hipStream_t st, igx_patch *pt, int kind, double *d_data)
{
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1];
    Single2DArgs A{};
    A.pd = pt->dev;
    A.gv = make_view(2, pt->gax, pt->d_ctrl, pt->ncomp);
    A.geo_kind = pt->geo_kind; A.jac = pt->d_jac; A.data = d_data;
    const int NF = kind == IGX_MASS ? 1 : 3, NY = kind == IGX_MASS ? 1 : 4, q = A0.q;
    static const int shapes[][2] = {{4, 16}, {2, 16}, {2, 8}, {1, 8}, {1, 4}, {1, 2}, {1, 1}};
    size_t bytes = 0;
    for (const auto &sh : shapes) {
        A.R0 = sh[0]; A.R1 = sh[1];
        A.NG0 = (A.R0 + A0.p) * q; A.WIN = (A.R1 + A1.p) * q;
        bytes = ((size_t)NF * A.NG0 * A.WIN + (size_t)(2 * A0.P - 1) * NY * A.WIN + (size_t)A.WIN * A1.P * 2 + (size_t)(2 * A0.P - 1) * NY * A.NG0) * sizeof(double);
        if (bytes <= 72 * 1024) break;
    }
    if (bytes > 150 * 1024) { set_error("single-launch 2D kernel: LDS image too large (%zu bytes)", bytes); return IGX_ERR_UNSUPPORTED; }
    const int nr0 = pt->r0_hi - pt->r0_lo;
    if (nr0 <= 0) return IGX_OK;
    dim3 grid((unsigned)((A1.N + A.R1 - 1) / A.R1), (unsigned)((nr0 + A.R0 - 1) / A.R0));
    int nt = 1024;                                       // the phases of a block are short and serial: many threads, short latency
#ifdef IGX_ABLATE
    if (const char *e = getenv("IGX_S2D_NT")) nt = atoi(e);
#endif
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

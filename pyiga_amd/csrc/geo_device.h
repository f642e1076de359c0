// Device helpers shared by the geometry/fields kernels (kern_basis.hip) and the fused
// geometry + stage-A sweep (geoa.hip).
#pragma once
#include "igx_internal.h"

namespace igx {

// geometry basis restricted to the grid nodes + control net
struct GeoView {
    const double *V[3];   // [G][P][2]
    const int *fa[3];     // [G]
    int P[3], N[3];       // active count, number of control points per axis
    const double *ctrl;   // (N0,N1[,N2], nc)
    int nc;               // components incl. weight
};

// homogeneous value/derivatives -> physical Jacobian Jm[r][c] = dG_r / d xi_c with c in (x,y,z)
// order, i.e. c = 0 differentiates along the LAST grid axis (pyiga/bspline.py:917-921); NURBS by
// the quotient rule (pyiga/geometry.py:17-25).
template <int DIM>
__device__ inline void finish_jacobian(const double val[MAX_COMP], const double jac[MAX_COMP][3], bool nurbs,
                                       int ncomp, int nc, double Jm[MAX_COMP][3], double ev[MAX_COMP])
{
    if (nurbs) {
        // quotient rule (V'W - V W') / W^2 with ONE reciprocal: an f64 division costs ~30 VALU
        // instructions on gfx950 and the reference formula has d*d + d of them per point
        const double W = val[nc - 1];
        const double iW = 1.0 / W, iW2 = iW * iW;
        for (int r = 0; r < ncomp; ++r) {
            ev[r] = val[r] * iW;
            for (int c = 0; c < DIM; ++c) {
                const int k = DIM - 1 - c;
                Jm[r][c] = (jac[r][k] * W - val[r] * jac[nc - 1][k]) * iW2;
            }
        }
    } else {
        for (int r = 0; r < ncomp; ++r) {
            ev[r] = val[r];
            for (int c = 0; c < DIM; ++c) Jm[r][c] = jac[r][DIM - 1 - c];
        }
    }
}

// quadrature fields of one point from its Jacobian (row-major t[]): f[0] = W for the mass form, the upper
// triangle of  W * JacInv JacInv^T  (row-major) for the stiffness form.  Returns the number of fields.
template <int DIM>
__device__ inline int fields_values(const double t[9], double GW, int kind, double f[6])
{
    if (DIM == 2) {
        const double det = t[0] * t[3] - t[1] * t[2];
        const double W = GW * fabs(det);
        if (kind == IGX_MASS) { f[0] = W; return 1; }
        const double inv = 1.0 / det;
        const double J0 = inv * t[3], J1 = inv * -t[1], J2 = inv * -t[2], J3 = inv * t[0];
        f[0] = W * (J0 * J0 + J1 * J1);
        f[1] = W * (J0 * J2 + J1 * J3);
        f[2] = W * (J2 * J2 + J3 * J3);
        return 3;
    } else {
        const double t3 = t[4] * t[8] - t[5] * t[7];
        const double t4 = t[3] * t[8] - t[5] * t[6];
        const double t5 = t[3] * t[7] - t[4] * t[6];
        const double det = (t[0] * t3 - t[1] * t4) + t[2] * t5;
        const double W = GW * fabs(det);
        if (kind == IGX_MASS) { f[0] = W; return 1; }
        const double inv = 1.0 / det;
        double JI[9];
        JI[0] = inv * t3;
        JI[1] = inv * -(t[1] * t[8] - t[2] * t[7]);
        JI[2] = inv * (t[1] * t[5] - t[2] * t[4]);
        JI[3] = inv * -t4;
        JI[4] = inv * (t[0] * t[8] - t[2] * t[6]);
        JI[5] = inv * -(t[0] * t[5] - t[2] * t[3]);
        JI[6] = inv * t5;
        JI[7] = inv * -(t[0] * t[7] - t[1] * t[6]);
        JI[8] = inv * (t[0] * t[4] - t[1] * t[3]);
        f[0] = W * ((JI[0] * JI[0] + JI[1] * JI[1]) + JI[2] * JI[2]);
        f[1] = W * ((JI[0] * JI[3] + JI[1] * JI[4]) + JI[2] * JI[5]);
        f[2] = W * ((JI[0] * JI[6] + JI[1] * JI[7]) + JI[2] * JI[8]);
        f[3] = W * ((JI[3] * JI[3] + JI[4] * JI[4]) + JI[5] * JI[5]);
        f[4] = W * ((JI[3] * JI[6] + JI[4] * JI[7]) + JI[5] * JI[8]);
        f[5] = W * ((JI[6] * JI[6] + JI[7] * JI[7]) + JI[8] * JI[8]);
        return 6;
    }
}

// ... stored as fields[f*stride + pt]
template <int DIM>
__device__ inline void fields_from_jac(const double t[9], double GW, int kind, double *fields, long long stride, long long pt)
{
    double f[6];
    const int nf = fields_values<DIM>(t, GW, kind, f);
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (k < nf) fields[k * stride + pt] = f[k];
}

// Fields of the convection-diffusion form (3D): c*B (upper triangle, as stiffness) and
// beta_a = W * sum_r JacInv[a][r] * b_r with b = (y, -x, 1) at the physical point ev.
__device__ inline void fields_convdiff(const double t[9], double GW, const double ev[MAX_COMP], double c,
                                       double *fields, long long stride, long long pt)
{
    const double t3 = t[4] * t[8] - t[5] * t[7];
    const double t4 = t[3] * t[8] - t[5] * t[6];
    const double t5 = t[3] * t[7] - t[4] * t[6];
    const double det = (t[0] * t3 - t[1] * t4) + t[2] * t5;
    const double W = GW * fabs(det);
    const double inv = 1.0 / det;
    double JI[9];
    JI[0] = inv * t3;
    JI[1] = inv * -(t[1] * t[8] - t[2] * t[7]);
    JI[2] = inv * (t[1] * t[5] - t[2] * t[4]);
    JI[3] = inv * -t4;
    JI[4] = inv * (t[0] * t[8] - t[2] * t[6]);
    JI[5] = inv * -(t[0] * t[5] - t[2] * t[3]);
    JI[6] = inv * t5;
    JI[7] = inv * -(t[0] * t[7] - t[1] * t[6]);
    JI[8] = inv * (t[0] * t[4] - t[1] * t[3]);
    const double cW = c * W;
    fields[pt] = cW * ((JI[0] * JI[0] + JI[1] * JI[1]) + JI[2] * JI[2]);
    fields[stride + pt] = cW * ((JI[0] * JI[3] + JI[1] * JI[4]) + JI[2] * JI[5]);
    fields[2 * stride + pt] = cW * ((JI[0] * JI[6] + JI[1] * JI[7]) + JI[2] * JI[8]);
    fields[3 * stride + pt] = cW * ((JI[3] * JI[3] + JI[4] * JI[4]) + JI[5] * JI[5]);
    fields[4 * stride + pt] = cW * ((JI[3] * JI[6] + JI[4] * JI[7]) + JI[5] * JI[8]);
    fields[5 * stride + pt] = cW * ((JI[6] * JI[6] + JI[7] * JI[7]) + JI[8] * JI[8]);
    const double b0 = ev[1], b1 = -ev[0], b2 = 1.0;
    fields[6 * stride + pt] = W * ((JI[0] * b0 + JI[1] * b1) + JI[2] * b2);
    fields[7 * stride + pt] = W * ((JI[3] * b0 + JI[4] * b1) + JI[5] * b2);
    fields[8 * stride + pt] = W * ((JI[6] * b0 + JI[7] * b1) + JI[8] * b2);
}

// IGX_FORM: parametric jet coefficients  M = W * T P T^t,  T = diag(1, JacInv)  (JacInv[a][r] = d xi_a / d x_r),
// so that  sum_rs P_rs D_r v D_s u  (physical jets)  =  sum_ab M_ab Dhat_a v Dhat_b u  (parametric jets).
// Only the terms listed in form_ab are stored (field t <-> form_ab[t] = 4 a + b).
// form_par: the coefficients are parametric already (igx_patch_set_pform): field k = Gauss weight * coefficient k.
template <int DIM>
__device__ inline void fields_form(const double t[9], double GW, const FormView &fv, const int form_n, const int *form_ab, const int form_par,
                                   double *fields, long long stride, long long pt)
{
    if (form_par) {
        for (int k = 0; k < form_n; ++k) fields[(long long)k * stride + pt] = GW * fv.c[(long long)k * stride + pt];
        return;
    }
    constexpr int NJ = DIM + 1;
    double T[4][4];
    for (int r = 0; r < 4; ++r)
        for (int s = 0; s < 4; ++s) T[r][s] = 0.0;
    T[0][0] = 1.0;
    double det;
    if (DIM == 2) {
        det = t[0] * t[3] - t[1] * t[2];
        const double inv = 1.0 / det;
        T[1][1] = inv * t[3]; T[1][2] = inv * -t[1];
        T[2][1] = inv * -t[2]; T[2][2] = inv * t[0];
    } else {
        const double t3 = t[4] * t[8] - t[5] * t[7];
        const double t4 = t[3] * t[8] - t[5] * t[6];
        const double t5 = t[3] * t[7] - t[4] * t[6];
        det = (t[0] * t3 - t[1] * t4) + t[2] * t5;
        const double inv = 1.0 / det;
        T[1][1] = inv * t3;
        T[1][2] = inv * -(t[1] * t[8] - t[2] * t[7]);
        T[1][3] = inv * (t[1] * t[5] - t[2] * t[4]);
        T[2][1] = inv * -t4;
        T[2][2] = inv * (t[0] * t[8] - t[2] * t[6]);
        T[2][3] = inv * -(t[0] * t[5] - t[2] * t[3]);
        T[3][1] = inv * t5;
        T[3][2] = inv * -(t[0] * t[7] - t[1] * t[6]);
        T[3][3] = inv * (t[0] * t[4] - t[1] * t[3]);
    }
    const double W = GW * fabs(det);
    double P[4][4];
#pragma unroll
    for (int r = 0; r < NJ; ++r)
#pragma unroll
        for (int s = 0; s < NJ; ++s) {
            const int sl = fv.slot[4 * r + s];
            P[r][s] = sl >= 0 ? fv.c[(long long)sl * stride + pt] : 0.0;
        }
    double TP[4][4];                                   // T P: rows of T act on the test index
#pragma unroll
    for (int a = 0; a < NJ; ++a)
#pragma unroll
        for (int s = 0; s < NJ; ++s) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < NJ; ++r) v = fma(T[a][r], P[r][s], v);
            TP[a][s] = v;
        }
    for (int k = 0; k < form_n; ++k) {
        const int a = form_ab[k] >> 2, b = form_ab[k] & 3;
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < NJ; ++s) v = fma(TP[a][s], T[b][s], v);
        fields[(long long)k * stride + pt] = W * v;
    }
}

inline GeoView make_view(int dim, const GeoAxis gax[3], const double *d_ctrl, int nc)
{
    GeoView gv{};
    for (int k = 0; k < 3; ++k) {
        gv.V[k] = k < dim ? gax[k].d_V : nullptr;
        gv.fa[k] = k < dim ? gax[k].d_fa : nullptr;
        gv.P[k] = k < dim ? gax[k].P : 1;
        gv.N[k] = k < dim ? gax[k].N : 1;
    }
    gv.ctrl = d_ctrl;
    gv.nc = nc;
    return gv;
}

} // namespace igx

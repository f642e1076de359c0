// CSR pattern and the entry-wise ("gather") quadrature kernels.
//
// One thread = one matrix entry, summed over the intersection of the two supports with the loops
// nested axis 0 outermost and a single accumulator -- the reference's own order:
//   entry_impl / combine      pyiga/assemblers.pyx:116-172 (M2D), 281-349 (S2D), 1255-1322 (M3D), 1455-1540 (S3D)
//   from_seq{2,3}             pyiga/assemble_tools_cy.pyx:36-49
//   multi_entries             pyiga/genericasm.pxi:353-436, 677-758
// The pattern kernel replaces MLStructure.nonzero + COO->CSR (pyiga/mlmatrix.py:113-130,
// pyiga/assemble.py:742-752) by writing canonical CSR directly.
//
// These kernels accept any degree <= IGX_MAX_DEGREE and any open knot vector (repeated interior
// knots included); they are the general path and the parity anchor for the sum-factorised path.
#include "igx_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace igx {

struct Dof3 { int i[3]; };

template <int DIM>
__device__ inline bool unravel(const PatchDev &pd, size_t I, int out[3])
{
    if (DIM == 2) {
        out[1] = (int)(I % pd.ax[1].N);
        size_t r = I / pd.ax[1].N;
        out[0] = (int)r; out[2] = 0;
        return r < (size_t)pd.ax[0].N;
    }
    out[2] = (int)(I % pd.ax[2].N); I /= pd.ax[2].N;
    out[1] = (int)(I % pd.ax[1].N);
    size_t r = I / pd.ax[1].N;
    out[0] = (int)r;
    return r < (size_t)pd.ax[0].N;
}

// value of entry (i, j); fields are structure-of-arrays over the resident Gauss slab
template <int DIM, int KIND>
__device__ inline double entry_value(const PatchDev &pd, const double *fields, const int i[3], const int j[3])
{
    int glo[3], ghi[3];
    for (int k = 0; k < DIM; ++k) {
        const AxisDev &A = pd.ax[k];
        const int lo = max(A.mslo[i[k]], A.mslo[j[k]]);
        const int hi = min(A.mshi[i[k]], A.mshi[j[k]]);
        if (lo >= hi) return 0.0;                     // no intersection of supports
        glo[k] = lo * A.q; ghi[k] = hi * A.q;
    }
    // Gauss points outside the resident slab cannot be evaluated here
    if (glo[0] < pd.g0_lo || ghi[0] > pd.g0_lo + pd.G0_loc) return __builtin_nan("");
    if (glo[1] < pd.b1 || ghi[1] > pd.b1 + pd.L1) return __builtin_nan("");          // (boxed patches: every axis)
    if (DIM == 3 && (glo[2] < pd.b2 || ghi[2] > pd.b2 + pd.L2)) return __builtin_nan("");
    const AxisDev &A0 = pd.ax[0], &A1 = pd.ax[1], &A2 = pd.ax[2];
    const long long stride = pd.npts_loc;
    double r = 0.0;
    for (int g0 = glo[0]; g0 < ghi[0]; ++g0) {
        const int f0 = A0.fa[g0 / A0.q];
        const double *u0 = A0.V + ((size_t)g0 * A0.P + (j[0] - f0)) * 2;
        const double *v0 = A0.V + ((size_t)g0 * A0.P + (i[0] - f0)) * 2;
        for (int g1 = glo[1]; g1 < ghi[1]; ++g1) {
            const int f1 = A1.fa[g1 / A1.q];
            const double *u1 = A1.V + ((size_t)g1 * A1.P + (j[1] - f1)) * 2;
            const double *v1 = A1.V + ((size_t)g1 * A1.P + (i[1] - f1)) * 2;
            if (DIM == 2) {
                const long long pt = (long long)(g0 - pd.g0_lo) * pd.L1 + (g1 - pd.b1);
                if (KIND == IGX_MASS) {
                    r += (((u0[0] * u1[0]) * (v0[0] * v1[0])) * fields[pt]);
                } else if (KIND == IGX_FORM) {
                    // jets in parametric (x, y) order: 0 = value, 1 = d/dx (last grid axis), 2 = d/dy
                    const double Du[3] = {u0[0] * u1[0], u0[0] * u1[1], u0[1] * u1[0]};
                    const double Dv[3] = {v0[0] * v1[0], v0[0] * v1[1], v0[1] * v1[0]};
                    double e = 0.0;
                    for (int t = 0; t < pd.form_n; ++t) {
                        const int ab = pd.form_ab[t];
                        double dv = Dv[0], du = Du[0];
                        if (pd.form_par) {           // masks over the grid axes: slot 1 of the table on the axes of the mask
                            dv = v0[(ab >> 3) & 1] * v1[(ab >> 4) & 1];
                            du = u0[ab & 1] * u1[(ab >> 1) & 1];
                        } else {
#pragma unroll
                            for (int c = 1; c < 3; ++c) { if ((ab >> 2) == c) dv = Dv[c]; if ((ab & 3) == c) du = Du[c]; }
                        }
                        e += (fields[t * stride + pt] * du) * dv;
                    }
                    r += e;
                } else {
                    const double f_0 = fields[pt], f_1 = fields[stride + pt], f_2 = fields[2 * stride + pt];
                    const double du10 = u0[0] * u1[1], du01 = u0[1] * u1[0];
                    const double dv10 = v0[0] * v1[1], dv01 = v0[1] * v1[0];
                    r += ((((f_0 * du10) + (f_1 * du01)) * dv10) + (((f_1 * du10) + (f_2 * du01)) * dv01));
                }
            } else {
                for (int g2 = glo[2]; g2 < ghi[2]; ++g2) {
                    const int f2 = A2.fa[g2 / A2.q];
                    const double *u2 = A2.V + ((size_t)g2 * A2.P + (j[2] - f2)) * 2;
                    const double *v2 = A2.V + ((size_t)g2 * A2.P + (i[2] - f2)) * 2;
                    const long long pt = ((long long)(g0 - pd.g0_lo) * pd.L1 + (g1 - pd.b1)) * pd.L2 + (g2 - pd.b2);
                    if (KIND == IGX_MASS) {
                        r += (((u0[0] * u1[0] * u2[0]) * (v0[0] * v1[0] * v2[0])) * fields[pt]);
                    } else if (KIND == IGX_FORM) {
                        // jets in parametric (x, y, z) order: index 0 = value, 1 = d/dx (last grid axis), 2 = d/dy, 3 = d/dz
                        const double Du[4] = {u0[0] * u1[0] * u2[0], u0[0] * u1[0] * u2[1], u0[0] * u1[1] * u2[0], u0[1] * u1[0] * u2[0]};
                        const double Dv[4] = {v0[0] * v1[0] * v2[0], v0[0] * v1[0] * v2[1], v0[0] * v1[1] * v2[0], v0[1] * v1[0] * v2[0]};
                        double e = 0.0;
                        for (int t = 0; t < pd.form_n; ++t) {
                            const int ab = pd.form_ab[t];
                            double dv = Dv[0], du = Du[0];
                            if (pd.form_par) {
                                dv = v0[(ab >> 3) & 1] * v1[(ab >> 4) & 1] * v2[(ab >> 5) & 1];
                                du = u0[ab & 1] * u1[(ab >> 1) & 1] * u2[(ab >> 2) & 1];
                            } else {
#pragma unroll
                                for (int c = 1; c < 4; ++c) { if ((ab >> 2) == c) dv = Dv[c]; if ((ab & 3) == c) du = Du[c]; }
                            }
                            e += (fields[t * stride + pt] * du) * dv;
                        }
                        r += e;
                    } else {
                        const double f_0 = fields[pt], f_1 = fields[stride + pt], f_2 = fields[2 * stride + pt];
                        const double f_3 = fields[3 * stride + pt], f_4 = fields[4 * stride + pt], f_5 = fields[5 * stride + pt];
                        const double du100 = u0[0] * u1[0] * u2[1];
                        const double du010 = u0[0] * u1[1] * u2[0];
                        const double du001 = u0[1] * u1[0] * u2[0];
                        const double dv100 = v0[0] * v1[0] * v2[1];
                        const double dv010 = v0[0] * v1[1] * v2[0];
                        const double dv001 = v0[1] * v1[0] * v2[0];
                        double e = ((((((f_0 * du100) + (f_1 * du010)) + (f_2 * du001)) * dv100)
                                     + ((((f_1 * du100) + (f_3 * du010)) + (f_4 * du001)) * dv010))
                                    + ((((f_2 * du100) + (f_4 * du010)) + (f_5 * du001)) * dv001));
                        if (KIND == IGX_CONVDIFF) {      // + (beta . du) * v, fields 6..8 = beta in (x,y,z) order
                            const double f_6 = fields[6 * stride + pt], f_7 = fields[7 * stride + pt], f_8 = fields[8 * stride + pt];
                            e += (((f_6 * du100) + (f_7 * du010)) + (f_8 * du001)) * (v0[0] * v1[0] * v2[0]);
                        }
                        r += e;
                    }
                }
            }
        }
    }
    return r;
}

template <int DIM, int KIND>
__global__ void k_entries_list(PatchDev pd, const double *fields, const size_t *ij, size_t M, double *out)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    int i[3], j[3];
    const bool ok_i = unravel<DIM>(pd, ij[2 * k], i), ok_j = unravel<DIM>(pd, ij[2 * k + 1], j);
    const bool ok = ok_i && ok_j;
    out[k] = ok ? entry_value<DIM, KIND>(pd, fields, i, j) : 0.0;
}

// Wave-per-entry form for batched requests at p >= 2 (multi_entries on arbitrary pairs): the 64 lanes of a
// wave split the Gauss points of the support intersection -- lane = point of the (axis 1 [, axis 2]) box, contiguous along
// the last axis, so the field loads of a wave are runs of (overlap * q) doubles -- and walk axis 0 together; the partial
// sums are added with wave shuffles.  The summation order differs from entry_value (one thread, the reference's order);
// k_entries_list keeps that order for p = 1 and as the parity anchor (IGX_ENTRIES=thread).
template <int DIM, int KIND>
__global__ void __launch_bounds__(256) k_entries_wave(PatchDev pd, const double *fields, const size_t *ij, size_t M, double *out)
{
    const size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (k >= M) return;
    int i[3], j[3];
    const bool ok = unravel<DIM>(pd, ij[2 * k], i) && unravel<DIM>(pd, ij[2 * k + 1], j);
    int glo[3] = {0, 0, 0}, ng[3] = {1, 1, 1};
    bool empty = !ok;
    if (ok)
        for (int a = 0; a < DIM; ++a) {
            const AxisDev &A = pd.ax[a];
            const int lo = max(A.mslo[i[a]], A.mslo[j[a]]), hi = min(A.mshi[i[a]], A.mshi[j[a]]);
            if (lo >= hi) empty = true;
            glo[a] = lo * A.q; ng[a] = (hi - lo) * A.q;
        }
    if (empty) { if (lane == 0) out[k] = 0.0; return; }
    if (glo[0] < pd.g0_lo || glo[0] + ng[0] > pd.g0_lo + pd.G0_loc || glo[1] < pd.b1 || glo[1] + ng[1] > pd.b1 + pd.L1 ||
        (DIM == 3 && (glo[2] < pd.b2 || glo[2] + ng[2] > pd.b2 + pd.L2))) { if (lane == 0) out[k] = __builtin_nan(""); return; }
    const AxisDev &A0 = pd.ax[0], &A1 = pd.ax[1], &A2 = pd.ax[2];
    const long long stride = pd.npts_loc;
    const int nbox = ng[1] * (DIM == 3 ? ng[2] : 1);
    double r = 0.0;
    for (int idx = lane; idx < nbox; idx += 64) {
        const int g1 = glo[1] + (DIM == 3 ? idx / ng[2] : idx), g2 = DIM == 3 ? glo[2] + idx % ng[2] : 0;
        const int f1 = A1.fa[g1 / A1.q];
        const double *u1 = A1.V + ((size_t)g1 * A1.P + (j[1] - f1)) * 2, *v1 = A1.V + ((size_t)g1 * A1.P + (i[1] - f1)) * 2;
        double u2v = 1.0, u2d = 0.0, v2v = 1.0, v2d = 0.0;
        if (DIM == 3) {
            const int f2 = A2.fa[g2 / A2.q];
            const double *u2 = A2.V + ((size_t)g2 * A2.P + (j[2] - f2)) * 2, *v2 = A2.V + ((size_t)g2 * A2.P + (i[2] - f2)) * 2;
            u2v = u2[0]; u2d = u2[1]; v2v = v2[0]; v2d = v2[1];
        }
        for (int g0 = glo[0]; g0 < glo[0] + ng[0]; ++g0) {
            const int f0 = A0.fa[g0 / A0.q];
            const double *u0 = A0.V + ((size_t)g0 * A0.P + (j[0] - f0)) * 2, *v0 = A0.V + ((size_t)g0 * A0.P + (i[0] - f0)) * 2;
            const long long pt = DIM == 3 ? ((long long)(g0 - pd.g0_lo) * pd.L1 + (g1 - pd.b1)) * pd.L2 + (g2 - pd.b2)
                                          : (long long)(g0 - pd.g0_lo) * pd.L1 + (g1 - pd.b1);
            if (DIM == 2) {
                if (KIND == IGX_MASS) r += (((u0[0] * u1[0]) * (v0[0] * v1[0])) * fields[pt]);
                else {
                    const double f_0 = fields[pt], f_1 = fields[stride + pt], f_2 = fields[2 * stride + pt];
                    const double du10 = u0[0] * u1[1], du01 = u0[1] * u1[0], dv10 = v0[0] * v1[1], dv01 = v0[1] * v1[0];
                    r += ((((f_0 * du10) + (f_1 * du01)) * dv10) + (((f_1 * du10) + (f_2 * du01)) * dv01));
                }
            } else {
                if (KIND == IGX_MASS) r += (((u0[0] * u1[0] * u2v) * (v0[0] * v1[0] * v2v)) * fields[pt]);
                else {
                    const double f_0 = fields[pt], f_1 = fields[stride + pt], f_2 = fields[2 * stride + pt];
                    const double f_3 = fields[3 * stride + pt], f_4 = fields[4 * stride + pt], f_5 = fields[5 * stride + pt];
                    const double du100 = u0[0] * u1[0] * u2d, du010 = u0[0] * u1[1] * u2v, du001 = u0[1] * u1[0] * u2v;
                    const double dv100 = v0[0] * v1[0] * v2d, dv010 = v0[0] * v1[1] * v2v, dv001 = v0[1] * v1[0] * v2v;
                    double e = ((((((f_0 * du100) + (f_1 * du010)) + (f_2 * du001)) * dv100)
                                 + ((((f_1 * du100) + (f_3 * du010)) + (f_4 * du001)) * dv010))
                                + ((((f_2 * du100) + (f_4 * du010)) + (f_5 * du001)) * dv001));
                    if (KIND == IGX_CONVDIFF) {
                        const double f_6 = fields[6 * stride + pt], f_7 = fields[7 * stride + pt], f_8 = fields[8 * stride + pt];
                        e += (((f_6 * du100) + (f_7 * du010)) + (f_8 * du001)) * (v0[0] * v1[0] * v2v);
                    }
                    r += e;
                }
            }
        }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) r += __shfl_xor(r, sft);
    if (lane == 0) out[k] = r;
}

// index pairs (ravelled row, column) of boxes of the reordered tensor, from the resident per-axis pair tables
__global__ void k_box_pairs(PatchDev pd, PairBoxes B, size_t *ij)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B.off[B.n]) return;
    int b = 0;
    while (b + 1 < B.n && idx >= B.off[b + 1]) ++b;
    long long l = idx - B.off[b];
    int r[3];
    r[2] = B.lo[b][2] + (int)(l % B.len[b][2]); l /= B.len[b][2];
    r[1] = B.lo[b][1] + (int)(l % B.len[b][1]); l /= B.len[b][1];
    r[0] = B.lo[b][0] + (int)l;
    size_t I = 0, J = 0;
    for (int k = 0; k < pd.dim; ++k) {
        I = I * (size_t)pd.ax[k].N + (size_t)pd.ax[k].pair_i[r[k]];
        J = J * (size_t)pd.ax[k].N + (size_t)pd.ax[k].pair_j[r[k]];
    }
    ij[2 * idx] = I; ij[2 * idx + 1] = J;
}

int launch_box_pairs(hipStream_t st, const igx_patch *pt, const PairBoxes &B, size_t *d_ij)
{
    const long long M = B.off[B.n];
    if (M == 0) return IGX_OK;
    k_box_pairs<<<dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st>>>(pt->dev, B, d_ij);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

int launch_entries_list(hipStream_t st, const igx_patch *pt, int kind, const size_t *d_ij, size_t M, double *d_out)
{
    if (M == 0) return IGX_OK;
    const PatchDev &pd = pt->dev;
    // wave-per-entry for the fixed forms at p >= 2 (IGX_ENTRIES=thread keeps the one-thread form, the reference's summation order)
    int pmax = 0;
    for (int k = 0; k < pt->dim; ++k) pmax = std::max(pmax, pt->ax[k].p);
    if (kind != IGX_FORM && pmax >= 2 && !pt->knobs.entries_thread && M < (1ull << 25)) {
        dim3 gridw((unsigned)((M + 3) / 4)), blockw(256);
        if (pt->dim == 2) {
            if (kind == IGX_MASS) k_entries_wave<2, IGX_MASS><<<gridw, blockw, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
            else k_entries_wave<2, IGX_STIFFNESS><<<gridw, blockw, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        } else {
            if (kind == IGX_MASS) k_entries_wave<3, IGX_MASS><<<gridw, blockw, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
            else if (kind == IGX_CONVDIFF) k_entries_wave<3, IGX_CONVDIFF><<<gridw, blockw, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
            else k_entries_wave<3, IGX_STIFFNESS><<<gridw, blockw, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        }
        IGX_HIP(hipGetLastError());
        return IGX_OK;
    }
    dim3 grid((unsigned)((M + 127) / 128)), block(128);
    if (pt->dim == 2) {
        if (kind == IGX_MASS) k_entries_list<2, IGX_MASS><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        else if (kind == IGX_FORM) k_entries_list<2, IGX_FORM><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        else k_entries_list<2, IGX_STIFFNESS><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
    } else {
        if (kind == IGX_MASS) k_entries_list<3, IGX_MASS><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        else if (kind == IGX_CONVDIFF) k_entries_list<3, IGX_CONVDIFF><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        else if (kind == IGX_FORM) k_entries_list<3, IGX_FORM><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
        else k_entries_list<3, IGX_STIFFNESS><<<grid, block, 0, st>>>(pd, pt->d_fields, d_ij, M, d_out);
    }
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// ---------------------------------------------------------------------------------------------
// Row geometry helpers.  Row I = (i0,i1[,i2]); its columns are the product of the per-axis ranges
// [jlo_k, jhi_k) in lexicographic order (canonical CSR), so
//   indptr(I) = rp0[i0]*S1*S2 + c0*(rp1[i1]*S2 + c1*rp2[i2])        (c_k = jhi_k - jlo_k)
template <int DIM>
__device__ inline long long row_start(const PatchDev &pd, const int i[3], int c[3])
{
    for (int k = 0; k < DIM; ++k) c[k] = pd.ax[k].jhi[i[k]] - pd.ax[k].jlo[i[k]];
    if (DIM == 2)
        return (long long)pd.ax[0].rp[i[0]] * pd.ax[1].S + (long long)c[0] * pd.ax[1].rp[i[1]];
    return igx_rowptr3(pd.ax[0].rp, pd.ax[1].rp, pd.ax[2].rp, pd.ax[1].S, pd.ax[2].S, c[0], c[1], i[0], i[1], i[2]);
}

// position of column j in row i (j must be inside the row's pattern)
template <int DIM>
__device__ inline long long entry_pos(const PatchDev &pd, const int i[3], const int j[3])
{
    int c[3];
    long long base = row_start<DIM>(pd, i, c);
    long long off = j[0] - pd.ax[0].jlo[i[0]];
    off = off * c[1] + (j[1] - pd.ax[1].jlo[i[1]]);
    if (DIM == 3) off = off * c[2] + (j[2] - pd.ax[2].jlo[i[2]]);
    return base + off - pd.nnz_off;
}

template <int DIM>
__global__ void k_pattern(PatchDev pd, long long row_lo, long long nrows, int maxrow, int32_t *indptr, int32_t *indices)
{
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long lr = tid / maxrow;
    int o = (int)(tid % maxrow);
    if (lr > nrows) return;
    if (lr == nrows) {                               // closing indptr entry
        if (o == 0 && indptr) {
            // total nnz of the owned rows = start of the (virtual) next row
            int i[3], c[3];
            long long I = row_lo + nrows - 1;
            unravel<DIM>(pd, (size_t)I, i);
            long long st = row_start<DIM>(pd, i, c);
            long long len = (long long)c[0] * c[1] * (DIM == 3 ? c[2] : 1);
            indptr[nrows] = (int32_t)(st + len - pd.nnz_off);
        }
        return;
    }
    int i[3], c[3];
    unravel<DIM>(pd, (size_t)(row_lo + lr), i);
    long long st = row_start<DIM>(pd, i, c) - pd.nnz_off;
    const int len = c[0] * c[1] * (DIM == 3 ? c[2] : 1);
    if (o == 0 && indptr) indptr[lr] = (int32_t)st;
    if (o >= len || !indices) return;
    int j[3];
    int rem = o;
    if (DIM == 3) { j[2] = pd.ax[2].jlo[i[2]] + rem % c[2]; rem /= c[2]; }
    j[1] = pd.ax[1].jlo[i[1]] + rem % c[1]; rem /= c[1];
    j[0] = pd.ax[0].jlo[i[0]] + rem;
    long long J = (long long)j[0] * pd.ax[1].N + j[1];
    if (DIM == 3) J = J * pd.ax[2].N + j[2];
    indices[st + o] = (int32_t)J;
}

static int max_row_len(const igx_patch *pt)
{
    int m = 1;
    for (int k = 0; k < pt->dim; ++k) {
        int mk = 0;
        for (int i = 0; i < pt->ax[k].N; ++i) mk = std::max(mk, pt->ax[k].jhi[i] - pt->ax[k].jlo[i]);
        m *= mk;
    }
    return m;
}

int launch_pattern(hipStream_t st, const igx_patch *pt, int32_t *d_indptr, int32_t *d_indices)
{
    const long long nrows = pt->row_hi - pt->row_lo;
    const int maxrow = max_row_len(pt);
    const long long total = (nrows + 1) * maxrow;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (pt->dim == 2) k_pattern<2><<<grid, block, 0, st>>>(pt->dev, pt->row_lo, nrows, maxrow, d_indptr, d_indices);
    else k_pattern<3><<<grid, block, 0, st>>>(pt->dev, pt->row_lo, nrows, maxrow, d_indptr, d_indices);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// ---------------------------------------------------------------------------------------------
// Whole-matrix entry-wise assembly: thread (row, offset) computes the lower-triangle entry and
// writes it and its mirror (assemble_entries(symmetric=True), pyiga/assemble.py:742-752).
// Rows r0_lo..r0_hi are owned; additionally the lower entries of the rows in the p planes above
// the slab whose COLUMN is owned are computed so that their mirror lands in an owned row
// (zero-communication multi-GPU scheme, DESIGN.md section "multi-GPU").
template <int DIM, int KIND>
__global__ void k_entries_csr(PatchDev pd, const double *fields, long long row_first, long long nrows_scan,
                              int maxrow, bool symmetric, double *data)
{
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long lr = tid / maxrow;
    int o = (int)(tid % maxrow);
    if (lr >= nrows_scan) return;
    int i[3], c[3], j[3];
    unravel<DIM>(pd, (size_t)(row_first + lr), i);
    for (int k = 0; k < DIM; ++k) c[k] = pd.ax[k].jhi[i[k]] - pd.ax[k].jlo[i[k]];
    const int len = c[0] * c[1] * (DIM == 3 ? c[2] : 1);
    if (o >= len) return;
    int rem = o;
    if (DIM == 3) { j[2] = pd.ax[2].jlo[i[2]] + rem % c[2]; rem /= c[2]; } else j[2] = 0;
    j[1] = pd.ax[1].jlo[i[1]] + rem % c[1]; rem /= c[1];
    j[0] = pd.ax[0].jlo[i[0]] + rem;
    // lower triangle only: J <= I  <=>  (j0,j1,j2) <=lex (i0,i1,i2)
    bool lower = true, diag = true;
    for (int k = 0; k < DIM; ++k) {
        if (j[k] != i[k]) { lower = j[k] < i[k]; diag = false; break; }
    }
    const bool own_row = i[0] >= pd.r0_lo && i[0] < pd.r0_hi;
    if (!symmetric) {                                // non-symmetric form: every entry of the owned rows, no mirror
        if (own_row) data[entry_pos<DIM>(pd, i, j)] = entry_value<DIM, KIND>(pd, fields, i, j);
        return;
    }
    if (!lower) return;
    const bool own_col = j[0] >= pd.r0_lo && j[0] < pd.r0_hi;
    if (!own_row && !own_col) return;
    const double v = entry_value<DIM, KIND>(pd, fields, i, j);
    if (own_row) data[entry_pos<DIM>(pd, i, j)] = v;
    if (own_col && !diag) data[entry_pos<DIM>(pd, j, i)] = v;
}

int launch_entries_csr(hipStream_t st, const igx_patch *pt, int kind, double *d_data)
{
    const PatchDev &pd = pt->dev;
    long long plane = (long long)pt->ax[1].N * (pt->dim == 3 ? pt->ax[2].N : 1);
    // scan rows of the owned planes plus the planes above that can still couple to owned columns
    int scan_hi = pt->r0_hi;
    while (scan_hi < pt->ax[0].N && pt->ax[0].jlo[scan_hi] < pt->r0_hi) ++scan_hi;
    const long long row_first = (long long)pt->r0_lo * plane;
    const long long nrows_scan = (long long)(scan_hi - pt->r0_lo) * plane;
    const int maxrow = max_row_len(pt);
    const long long total = nrows_scan * maxrow;
    if (total == 0) return IGX_OK;
    dim3 grid((unsigned)((total + 127) / 128)), block(128);
    if (pt->dim == 2) {
        if (kind == IGX_MASS) k_entries_csr<2, IGX_MASS><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, true, d_data);
        else if (kind == IGX_FORM) k_entries_csr<2, IGX_FORM><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, false, d_data);
        else k_entries_csr<2, IGX_STIFFNESS><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, true, d_data);
    } else {
        if (kind == IGX_MASS) k_entries_csr<3, IGX_MASS><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, true, d_data);
        else if (kind == IGX_CONVDIFF) k_entries_csr<3, IGX_CONVDIFF><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, false, d_data);
        else if (kind == IGX_FORM) k_entries_csr<3, IGX_FORM><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, false, d_data);
        else k_entries_csr<3, IGX_STIFFNESS><<<grid, block, 0, st>>>(pd, pt->d_fields, row_first, nrows_scan, maxrow, true, d_data);
    }
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

} // namespace igx

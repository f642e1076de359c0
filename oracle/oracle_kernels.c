/*
 * CPU oracle kernels -- TEST INFRASTRUCTURE ONLY (see oracle/iga_oracle.py).
 *
 * Plain-C restatement of the reference's entry-wise quadrature sums:
 *   from_seq{2,3}            pyiga/assemble_tools_cy.pyx:36-49
 *   intersect_intervals      pyiga/assemble_tools_cy.pyx:145-156
 *   entry_impl               pyiga/assemblers.pyx:140-172,317-349,1281-1322,1499-1540
 *   combine                  pyiga/assemblers.pyx:116-135,281-312,1255-1276,1455-1494
 *   multi_entries_chunk      pyiga/genericasm.pxi:691-703 (+ thread chunks :742-757)
 *
 * One matrix entry = one scalar accumulator over the tensor Gauss grid restricted
 * to the intersection of the two supports, loops nested axis 0 outermost, exactly
 * as in the generated Cython.  Never linked into the product library.
 */
#include <stddef.h>
#include <stdint.h>
#include <omp.h>

typedef struct {
    int dim, kind, nder, F;
    size_t ndofs[3], ng[3];
    const int64_t *ms[3];
    const double *C[3];
    const double *fields;
} orc_ctx;

static inline const double *Cptr(const orc_ctx *c, int ax, size_t dof, size_t g)
{
    return c->C[ax] + (dof * c->ng[ax] + g) * (size_t)c->nder;
}

static double entry2(const orc_ctx *c, const size_t i[2], const size_t j[2])
{
    size_t sta[2], end[2];
    for (int k = 0; k < 2; ++k) {
        int64_t a = c->ms[k][2 * j[k]], b = c->ms[k][2 * j[k] + 1];
        int64_t a2 = c->ms[k][2 * i[k]], b2 = c->ms[k][2 * i[k] + 1];
        int64_t lo = a > a2 ? a : a2, hi = b < b2 ? b : b2;
        if (lo >= hi) return 0.0;
        sta[k] = (size_t)lo; end[k] = (size_t)hi;
    }
    const double *u0 = Cptr(c, 0, j[0], sta[0]), *u1 = Cptr(c, 1, j[1], sta[1]);
    const double *v0 = Cptr(c, 0, i[0], sta[0]), *v1 = Cptr(c, 1, i[1], sta[1]);
    const size_t n0 = end[0] - sta[0], n1 = end[1] - sta[1];
    const int F = c->F;
    double r = 0.0;
    if (c->kind == 0) {
        for (size_t i0 = 0; i0 < n0; ++i0)
            for (size_t i1 = 0; i1 < n1; ++i1) {
                const double *f = c->fields + ((sta[0] + i0) * c->ng[1] + (sta[1] + i1)) * F;
                r += (((u0[i0] * u1[i1]) * (v0[i0] * v1[i1])) * f[0]);
            }
    } else if (c->kind == 3) {
        /* general first-order-jet form in 2D: f = [P(16, rows/cols 0..2 used), W, JacInv(4) row-major [param][phys]] */
        for (size_t i0 = 0; i0 < n0; ++i0)
            for (size_t i1 = 0; i1 < n1; ++i1) {
                const double *f = c->fields + ((sta[0] + i0) * c->ng[1] + (sta[1] + i1)) * F;
                const double du10 = u0[2 * i0 + 0] * u1[2 * i1 + 1], du01 = u0[2 * i0 + 1] * u1[2 * i1 + 0];
                const double dv10 = v0[2 * i0 + 0] * v1[2 * i1 + 1], dv01 = v0[2 * i0 + 1] * v1[2 * i1 + 0];
                double Gu[3], Gv[3];
                Gu[0] = u0[2 * i0 + 0] * u1[2 * i1 + 0];
                Gv[0] = v0[2 * i0 + 0] * v1[2 * i1 + 0];
                for (int r_ = 0; r_ < 2; ++r_) {
                    Gu[1 + r_] = (f[17 + r_] * du10) + (f[19 + r_] * du01);
                    Gv[1 + r_] = (f[17 + r_] * dv10) + (f[19 + r_] * dv01);
                }
                double e = 0.0;
                for (int r_ = 0; r_ < 3; ++r_)
                    for (int s_ = 0; s_ < 3; ++s_) e += (f[4 * r_ + s_] * Gu[s_]) * Gv[r_];
                r += e * f[16];
            }
    } else {
        for (size_t i0 = 0; i0 < n0; ++i0)
            for (size_t i1 = 0; i1 < n1; ++i1) {
                const double *f = c->fields + ((sta[0] + i0) * c->ng[1] + (sta[1] + i1)) * F;
                double du10 = u0[2 * i0 + 0] * u1[2 * i1 + 1];
                double du01 = u0[2 * i0 + 1] * u1[2 * i1 + 0];
                double dv10 = v0[2 * i0 + 0] * v1[2 * i1 + 1];
                double dv01 = v0[2 * i0 + 1] * v1[2 * i1 + 0];
                r += ((((f[0] * du10) + (f[1] * du01)) * dv10) + (((f[1] * du10) + (f[2] * du01)) * dv01));
            }
    }
    return r;
}

static double entry3(const orc_ctx *c, const size_t i[3], const size_t j[3])
{
    size_t sta[3], end[3];
    for (int k = 0; k < 3; ++k) {
        int64_t a = c->ms[k][2 * j[k]], b = c->ms[k][2 * j[k] + 1];
        int64_t a2 = c->ms[k][2 * i[k]], b2 = c->ms[k][2 * i[k] + 1];
        int64_t lo = a > a2 ? a : a2, hi = b < b2 ? b : b2;
        if (lo >= hi) return 0.0;
        sta[k] = (size_t)lo; end[k] = (size_t)hi;
    }
    const double *u0 = Cptr(c, 0, j[0], sta[0]), *u1 = Cptr(c, 1, j[1], sta[1]), *u2 = Cptr(c, 2, j[2], sta[2]);
    const double *v0 = Cptr(c, 0, i[0], sta[0]), *v1 = Cptr(c, 1, i[1], sta[1]), *v2 = Cptr(c, 2, i[2], sta[2]);
    const size_t n0 = end[0] - sta[0], n1 = end[1] - sta[1], n2 = end[2] - sta[2];
    const int F = c->F;
    double r = 0.0;
    if (c->kind == 0) {
        for (size_t i0 = 0; i0 < n0; ++i0)
            for (size_t i1 = 0; i1 < n1; ++i1) {
                const double *f = c->fields + (((sta[0] + i0) * c->ng[1] + (sta[1] + i1)) * c->ng[2] + sta[2]) * F;
                for (size_t i2 = 0; i2 < n2; ++i2)
                    r += (((u0[i0] * u1[i1] * u2[i2]) * (v0[i0] * v1[i1] * v2[i2])) * f[i2 * F]);
            }
    } else {
        for (size_t i0 = 0; i0 < n0; ++i0)
            for (size_t i1 = 0; i1 < n1; ++i1) {
                const double *fb = c->fields + (((sta[0] + i0) * c->ng[1] + (sta[1] + i1)) * c->ng[2] + sta[2]) * F;
                for (size_t i2 = 0; i2 < n2; ++i2) {
                    const double *f = fb + i2 * F;
                    double du100 = (u0[2 * i0 + 0] * u1[2 * i1 + 0] * u2[2 * i2 + 1]);
                    double du010 = (u0[2 * i0 + 0] * u1[2 * i1 + 1] * u2[2 * i2 + 0]);
                    double du001 = (u0[2 * i0 + 1] * u1[2 * i1 + 0] * u2[2 * i2 + 0]);
                    double dv100 = (v0[2 * i0 + 0] * v1[2 * i1 + 0] * v2[2 * i2 + 1]);
                    double dv010 = (v0[2 * i0 + 0] * v1[2 * i1 + 1] * v2[2 * i2 + 0]);
                    double dv001 = (v0[2 * i0 + 1] * v1[2 * i1 + 0] * v2[2 * i2 + 0]);
                    if (c->kind == 3) {
                        /* general first-order-jet form, evaluated the way the generated code of such a vform
                           does (pyiga/codegen/cython.py:325-387): physical gradients through JacInv, then the
                           integrand, then the weight.  f = [P(16), W, JacInv(9)], JacInv row-major [param][phys] */
                        double Gu[4], Gv[4];
                        Gu[0] = (u0[2 * i0 + 0] * u1[2 * i1 + 0] * u2[2 * i2 + 0]);
                        Gv[0] = (v0[2 * i0 + 0] * v1[2 * i1 + 0] * v2[2 * i2 + 0]);
                        for (int r_ = 0; r_ < 3; ++r_) {
                            Gu[1 + r_] = ((f[17 + r_] * du100) + (f[20 + r_] * du010)) + (f[23 + r_] * du001);
                            Gv[1 + r_] = ((f[17 + r_] * dv100) + (f[20 + r_] * dv010)) + (f[23 + r_] * dv001);
                        }
                        double e = 0.0;
                        for (int r_ = 0; r_ < 4; ++r_)
                            for (int s_ = 0; s_ < 4; ++s_) e += (f[4 * r_ + s_] * Gu[s_]) * Gv[r_];
                        r += e * f[16];
                    } else if (c->kind == 1) {
                        r += ((((((f[0] * du100) + (f[1] * du010)) + (f[2] * du001)) * dv100)
                               + ((((f[1] * du100) + (f[3] * du010)) + (f[4] * du001)) * dv010))
                              + ((((f[2] * du100) + (f[4] * du010)) + (f[5] * du001)) * dv001));
                    } else {
                        /* physical gradient of u, component r: sum_a JacInv[a][r] * du_a */
                        const double gu0 = ((f[5] * du100) + (f[8] * du010)) + (f[11] * du001);
                        const double gu1 = ((f[6] * du100) + (f[9] * du010)) + (f[12] * du001);
                        const double gu2 = ((f[7] * du100) + (f[10] * du010)) + (f[13] * du001);
                        const double gv0 = ((f[5] * dv100) + (f[8] * dv010)) + (f[11] * dv001);
                        const double gv1 = ((f[6] * dv100) + (f[9] * dv010)) + (f[12] * dv001);
                        const double gv2 = ((f[7] * dv100) + (f[10] * dv010)) + (f[13] * dv001);
                        const double vv = (v0[2 * i0 + 0] * v1[2 * i1 + 0] * v2[2 * i2 + 0]);
                        r += ((((((f[0] * gu0) * gv0) + ((f[0] * gu1) * gv1)) + ((f[0] * gu2) * gv2))
                               + ((((f[2] * gu0) + (-f[1] * gu1)) + gu2) * vv)) * f[4]);
                    }
                }
            }
    }
    return r;
}

int orc_entries(int dim, int kind, const size_t *ndofs, const size_t *ng,
                const int64_t *ms0, const int64_t *ms1, const int64_t *ms2,
                const double *C0, const double *C1, const double *C2, int nder,
                const double *fields, int F,
                const size_t *idx, size_t M, double *out, int nthreads)
{
    if (dim != 2 && dim != 3) return 1;
    orc_ctx c;
    c.dim = dim; c.kind = kind; c.nder = nder; c.F = F;
    for (int k = 0; k < dim; ++k) { c.ndofs[k] = ndofs[k]; c.ng[k] = ng[k]; }
    c.ms[0] = ms0; c.ms[1] = ms1; c.ms[2] = ms2;
    c.C[0] = C0; c.C[1] = C1; c.C[2] = C2;
    c.fields = fields;
    if (nthreads < 1) nthreads = 1;
    /* The reference hands one contiguous chunk per thread to a pool (chunk_tasks(), assemble_tools_cy.pyx:387-391); entries
       near the patch boundary have smaller support intersections, so equal chunks are unequal work.  The values do not
       depend on the schedule (every entry is summed by one thread, in the reference's order); the baseline timing gets the
       balanced schedule so that it is not understated by the port. */
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
    for (ptrdiff_t k = 0; k < (ptrdiff_t)M; ++k) {
        size_t I = idx[2 * k], J = idx[2 * k + 1];
        if (dim == 2) {
            size_t i[2], j[2];
            i[1] = I % c.ndofs[1]; i[0] = I / c.ndofs[1];
            j[1] = J % c.ndofs[1]; j[0] = J / c.ndofs[1];
            out[k] = (i[0] < c.ndofs[0] && j[0] < c.ndofs[0]) ? entry2(&c, i, j) : 0.0;
        } else {
            size_t i[3], j[3];
            i[2] = I % c.ndofs[2]; I /= c.ndofs[2]; i[1] = I % c.ndofs[1]; i[0] = I / c.ndofs[1];
            j[2] = J % c.ndofs[2]; J /= c.ndofs[2]; j[1] = J % c.ndofs[1]; j[0] = J / c.ndofs[1];
            out[k] = (i[0] < c.ndofs[0] && j[0] < c.ndofs[0]) ? entry3(&c, i, j) : 0.0;
        }
    }
    return 0;
}

// Low-rank (adaptive cross approximation) assembly as a CONSUMER of batched entries.
//
// The reference's one native component (pyiga/fastasm.cc:294-494 `aca`, `aca_3d`; :505-760 reordering and inflation;
// driver pyiga/fast_assemble_cy.pyx:101-113) pulls single entries A(i, j) through a C callback.  Here the same algorithm
// runs on the host and asks the device for whole rows, columns, fibres -- and, where they are small, whole slices -- of the
// REORDERED matrix at a time (entries_pair_boxes: the index pairs of a request are generated on the device from the
// resident per-axis pair tables, one launch per request, only the values come back):
//
//   2D:  X[r0][r1]     = A[(i0,i1),(j0,j1)],      r_k = index of the 1D pair (i_k, j_k) with overlapping supports
//   3D:  X[r0][r1][r2] = A[(i0,i1,i2),(j0,j1,j2)]
//
// X has low rank for smooth geometries; ACA with partial pivoting builds it from a few crosses (2D) or from fibre x slice
// crosses whose slices are themselves approximated by 2D ACA with the current approximation as starting value (3D).  The
// result is inflated to the canonical CSR values of the patch.  Control flow, stopping rules (tolerance reached
// `tolcount` times, `skipcount` tiny pivots) and defaults follow the reference; the pseudo-random restart uses a fixed
// linear congruential sequence (reproducible) instead of rand().
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <vector>
#include "igx_internal.h"

namespace igx {

namespace {

struct Requests {
    igx_patch *pt;
    int kind, dim;
    long long nreq = 0, nent = 0;
    unsigned lcg = 12345u;

    size_t S(int k) const { return (size_t)pt->ax[k].S; }
    unsigned next_random() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; }

    // one box of the reordered tensor: axes in `full` (bit k) run over all their pairs, the others are fixed at r[k]; the index
    // pairs are generated on the device (launch_box_pairs), only the values travel
    int box(unsigned full, const size_t r[3], double *out)
    {
        PairBoxes B{};
        B.n = 1;
        long long n = 1;
        for (int k = 0; k < 3; ++k) {
            const bool f = k < dim && ((full >> k) & 1u);
            B.lo[0][k] = f || k >= dim ? 0 : (int)r[k];
            B.len[0][k] = f ? (int)S(k) : 1;
            n *= B.len[0][k];
        }
        B.off[0] = 0; B.off[1] = n;
        ++nreq; nent += n;
        return entries_pair_boxes(pt, kind, B, out);
    }
    // entries along axis `free_ax`, the other pair indices fixed: out[r] for r < S(free_ax)
    int line(int free_ax, const size_t r[3], double *out) { return box(1u << free_ax, r, out); }
};

size_t argmax_abs(const double *v, size_t n)
{
    size_t arg = 0;
    double mx = std::fabs(v[0]);
    for (size_t i = 1; i < n; ++i)
        if (std::fabs(v[i]) > mx) { mx = std::fabs(v[i]); arg = i; }
    return arg;
}

// ACA of an m x n matrix given by row and column generators; X (row-major, m x n) holds the starting value and receives
// the approximation.  Returns the number of crosses added, or -1 when a generator failed.
int aca_matrix(size_t m, size_t n, const std::function<int(size_t, double *)> &gen_row, const std::function<int(size_t, double *)> &gen_col,
               double *X, double tol, int maxiter, int max_skip, int max_tol, int verbose, Requests &rq)
{
    std::vector<double> col(m), row(n);
    int skipcount = 0, tolcount = 0, k = 0;
    size_t i = m / 2;
    while (true) {
        if (k >= maxiter) { if (verbose >= 1) printf("Maximum iteration count reached; aborting (%d it.)\n", k); break; }
        if (gen_row(i, row.data())) return -1;
        for (size_t l = 0; l < n; ++l) row[l] -= X[i * n + l];                 // error row
        const size_t j0 = argmax_abs(row.data(), n);
        const double e = std::fabs(row[j0]);
        if (e < 1e-15) {                                                        // tiny row: try another one
            if (verbose >= 2) printf("Skipping row %zu\n", i);
            i = rq.next_random() % m;
            if (++skipcount >= max_skip) { if (verbose >= 1) printf("Skipped %d times; stopping (%d it.)\n", skipcount, k); break; }
            continue;
        } else if (e < tol) {
            if (++tolcount >= max_tol) { if (verbose >= 1) printf("Desired tolerance reached %d times; stopping (%d it.)\n", tolcount, k); break; }
        } else skipcount = tolcount = 0;
        if (verbose >= 2) printf("%zu\t%zu\t%g\n", i, j0, e);
        if (gen_col(j0, col.data())) return -1;
        for (size_t l = 0; l < m; ++l) col[l] = (col[l] - X[l * n + j0]) / row[j0];   // scaled error column
        for (size_t a = 0; a < m; ++a) {                                        // rank-1 correction
            const double c = col[a];
            if (c == 0.0) continue;
            double *xr = X + a * n;
            for (size_t b = 0; b < n; ++b) xr[b] += c * row[b];
        }
        ++k;
        col[i] = 0.0;                                                           // the error vanishes there now
        i = argmax_abs(col.data(), m);
    }
    return k;
}

} // namespace

} // namespace igx

using namespace igx;

extern "C" int igx_fast_assemble(igx_patch *pt, int kind, double tol, int maxiter, int skipcount, int tolcount, int verbose,
                                 double *data_out, int *rank_out, long long *entries_out)
{
    if (!pt || !data_out) { set_error("igx_fast_assemble: null argument"); return IGX_ERR_ARG; }
    if (pt->boxed) { set_error("igx_fast_assemble: the patch holds a span box (batched entries only)"); return IGX_ERR_UNSUPPORTED; }
    if (kind != IGX_MASS && kind != IGX_STIFFNESS) { set_error("igx_fast_assemble: mass and stiffness forms only"); return IGX_ERR_UNSUPPORTED; }
    if (pt->row_lo != 0 || pt->row_hi != pt->nrows_total) { set_error("igx_fast_assemble: needs the whole patch, not a row slab"); return IGX_ERR_UNSUPPORTED; }
    if (maxiter < 1 || skipcount < 1 || tolcount < 1) { set_error("igx_fast_assemble: bad iteration parameters"); return IGX_ERR_ARG; }
    const int dim = pt->dim;
    Requests rq{pt, kind, dim};
    const size_t n0 = rq.S(0), n1 = rq.S(1), n2 = dim == 3 ? rq.S(2) : 1;
    if (n0 == 0 || n1 == 0 || n2 == 0) return IGX_OK;
    std::vector<double> X;
    try { X.assign(n0 * n1 * n2, 0.0); } catch (...) { set_error("igx_fast_assemble: %.2f GB of host memory for the reordered tensor", n0 * n1 * n2 * 8.0 / 1e9); return IGX_ERR_NOMEM; }
    int rank = 0;
    const bool batch_all = dim == 2 && (long long)(n0 * n1) <= pt->aca_batch;        // 2D: the whole matrix in one request
    const bool batch_slice = dim == 3 && (long long)(n1 * n2) <= pt->aca_batch;     // 3D: a slice in one request
    if (batch_all) {
        const size_t r[3] = {0, 0, 0};
        if (rq.box(3u, r, X.data())) return IGX_ERR_HIP;
        if (verbose >= 1) printf("%zu x %zu pairs fetched in one request (exact)\n", n0, n1);
    } else if (dim == 2) {
        auto row = [&](size_t i, double *out) { const size_t r[3] = {i, 0, 0}; return rq.line(1, r, out); };
        auto col = [&](size_t j, double *out) { const size_t r[3] = {0, j, 0}; return rq.line(0, r, out); };
        rank = aca_matrix(n0, n1, row, col, X.data(), tol, maxiter, skipcount, tolcount, verbose, rq);
        if (rank < 0) return IGX_ERR_HIP;
    } else {
        // fibre x slice crosses (pyiga/fastasm.cc:385-494)
        std::vector<double> col(n0), mat(n1 * n2);
        size_t I[3] = {n0 / 2, n1 / 2, n2 / 2};
        int skips = 0, tols = 0;
        while (true) {
            if (rank >= maxiter) { if (verbose >= 1) printf("Maximum iteration count reached; aborting (%d outer it.)\n", rank); break; }
            if (rq.line(0, I, col.data())) return IGX_ERR_HIP;                 // error fibre through (., I1, I2)
            for (size_t l = 0; l < n0; ++l) col[l] -= X[(l * n1 + I[1]) * n2 + I[2]];
            const size_t i0 = argmax_abs(col.data(), n0);
            const double e = std::fabs(col[i0]);
            if (e < 1e-15) {
                if (verbose >= 2) printf("Skipping...\n");
                I[1] = rq.next_random() % n1; I[2] = rq.next_random() % n2;
                if (++skips >= skipcount) { if (verbose >= 1) printf("Skipped %d times; stopping (%d outer it.)\n", skips, rank); break; }
                continue;
            } else if (e < tol) {
                if (++tols >= tolcount) { if (verbose >= 1) printf("Desired tolerance reached %d times; stopping (%d outer it.)\n", tols, rank); break; }
            } else skips = tols = 0;
            I[0] = i0;
            if (verbose >= 2) printf("%zu\t%zu\t%zu\t%g\n", I[0], I[1], I[2], e);
            // the slice A[i0, :, :] by 2D ACA, starting from the current approximation
            if (batch_slice) {                                                  // exact slice, one request
                const size_t r[3] = {i0, 0, 0};
                if (rq.box(6u, r, mat.data())) return IGX_ERR_HIP;
            } else {
                std::copy(X.begin() + i0 * n1 * n2, X.begin() + (i0 + 1) * n1 * n2, mat.begin());
                auto srow = [&](size_t i, double *out) { const size_t r[3] = {i0, i, 0}; return rq.line(2, r, out); };
                auto scol = [&](size_t j, double *out) { const size_t r[3] = {i0, 0, j}; return rq.line(1, r, out); };
                if (aca_matrix(n1, n2, srow, scol, mat.data(), tol, maxiter, skipcount, tolcount, std::min(verbose, 1), rq) < 0) return IGX_ERR_HIP;
            }
            for (size_t a = 0; a < n1 * n2; ++a) mat[a] -= X[i0 * n1 * n2 + a];     // error slice
            const double pivot = col[i0];
            for (size_t l = 0; l < n0; ++l) col[l] /= pivot;
            for (size_t l = 0; l < n0; ++l) {
                const double c = col[l];
                if (c == 0.0) continue;
                double *xs = X.data() + l * n1 * n2;
                for (size_t a = 0; a < n1 * n2; ++a) xs[a] += c * mat[a];
            }
            ++rank;
            mat[I[1] * n2 + I[2]] = 0.0;
            const size_t am = argmax_abs(mat.data(), n1 * n2);
            I[1] = am / n2; I[2] = am % n2;
        }
    }
    // inflate: canonical CSR order = rows lexicographic, columns lexicographic inside the row = pair indices ascending
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1];
    size_t pos = 0;
    if (dim == 2) {
        for (int i0 = 0; i0 < A0.N; ++i0)
            for (int i1 = 0; i1 < A1.N; ++i1)
                for (int r0 = A0.rp[i0]; r0 < A0.rp[i0 + 1]; ++r0)
                    for (int r1 = A1.rp[i1]; r1 < A1.rp[i1 + 1]; ++r1) data_out[pos++] = X[(size_t)r0 * n1 + r1];
    } else {
        const Axis &A2 = pt->ax[2];
        for (int i0 = 0; i0 < A0.N; ++i0)
            for (int i1 = 0; i1 < A1.N; ++i1)
                for (int i2 = 0; i2 < A2.N; ++i2)
                    for (int r0 = A0.rp[i0]; r0 < A0.rp[i0 + 1]; ++r0)
                        for (int r1 = A1.rp[i1]; r1 < A1.rp[i1 + 1]; ++r1)
                            for (int r2 = A2.rp[i2]; r2 < A2.rp[i2 + 1]; ++r2) data_out[pos++] = X[((size_t)r0 * n1 + r1) * n2 + r2];
    }
    if (rank_out) *rank_out = rank;
    if (entries_out) *entries_out = rq.nent;
    pt->aca_requests = rq.nreq; pt->aca_entries = rq.nent; pt->aca_rank = rank;
    if (verbose >= 1) {
        printf("ACA: %d crosses, %lld entries in %lld batched requests (matrix has %lld)\n", rank, rq.nent, rq.nreq, (long long)pos);
        fflush(stdout);
    }
    return IGX_OK;
}

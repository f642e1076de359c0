/*
 * igx.h -- C ABI of libigx: MI355X (gfx950) tensor-product IgA assembly.
 *
 * This is the drop-in boundary for the hot path of c-f-h/pyiga
 *     pyiga.assemble.stiffness()/mass()  with a geometry map
 * (SURVEY.md section 8b).  Plain pointers and sizes only; no C++ or torch types.
 * Every entry point returns 0 on success and a non-zero code on failure
 * (message via igx_last_error()); nothing throws across the ABI.  All
 * `const double*` / `const size_t*` arguments are HOST pointers unless the name
 * starts with `d_`.  Host output buffers are allocated by the caller.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   igx_active_deriv / igx_find_spans  <- pyiga/bspline_cy.pyx:13-27,126-145
 *   igx_patch_create                   <- *Assembler{2,3}D.__init__
 *                                         pyiga/assemblers.pyx:38-80,186-228,1170-1217,1336-1383
 *                                         (quadrature.py:3-23, bspline.py:129-136,629-660,
 *                                          geometry.py:17-25,116-123, tensor.py:97-128,
 *                                          precompute_fields assemblers.pyx:86-110,234-275,1223-1249,1389-1449)
 *   igx_grid_jacobian / igx_grid_eval  <- BSplineFunc/NurbsFunc.grid_jacobian/grid_eval
 *                                         pyiga/bspline.py:874-921, pyiga/geometry.py:103-123
 *   igx_pattern                        <- MLStructure.from_kvs + nonzero
 *                                         pyiga/mlmatrix.py:59-65,113-130,420-440; mlmatrix_cy.pyx:189-289
 *                                         (emits canonical CSR directly instead of COO pairs)
 *   igx_entries                        <- BaseAssembler{2,3}D.multi_entries / entry
 *                                         pyiga/genericasm.pxi:353-436,677-758
 *                                         (entry_impl + combine, assemblers.pyx:116-172,281-349,1255-1322,1455-1540)
 *   igx_patch_set_coeff + IGX_CONVDIFF <- the assembler pyiga.compile.compile_vform generates for the form
 *                                         (pyiga/assemble.py:837-897, pyiga/codegen/cython.py:325-387,673-701)
 *   igx_patch_set_form + IGX_FORM      <- assemble.assemble(<form string>, ...) for scalar forms that are bilinear in
 *                                         (u, grad u) x (v, grad v): pyiga/assemble.py:837-897, pyiga/vform.py:1804-1885
 *   igx_load_vector                    <- inner_products / *FunctionalAssembler*.assemble_vector
 *                                         pyiga/assemble.py:288-340, pyiga/assemblers.pyx:883-1156,2204-2500,
 *                                         pyiga/genericasm.pxi:438-456,762-778
 *   igx_assemble                       <- assemble_entries(asm, symmetric=True)
 *                                         pyiga/assemble.py:703-754 (multi_entries + COO->CSR + mirror)
 */
#ifndef IGX_H
#define IGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGX_VERSION 101            /* 0.1.1: igx_patch_desc.box_lo / box_hi */
#define IGX_MAX_DIM 3
#define IGX_MAX_DEGREE 15          /* basis evaluation / entry-wise kernels */
#define IGX_MAX_SF_DEGREE 7        /* sum-factorised fast path */

typedef struct igx_ctx igx_ctx;       /* one per GPU: device id + HIP stream */
typedef struct igx_patch igx_patch;   /* device-resident state of one assembler */

enum { IGX_MASS = 0, IGX_STIFFNESS = 1,
       /* (inner(c*grad(u),grad(v)) + inner((x[1],-x[0],1.0),grad(u))*v)*dx -- the custom (vform) case of
          BASELINE config 5, non-symmetric, 3D only; c is set with igx_patch_set_coeff */
       IGX_CONVDIFF = 2,
       /* general scalar bilinear form in the first-order jets of u and v (2D: r,s = 0..2):
            a(u,v) = integral of  sum_{r,s=0..3} P_rs(x) * D_r v * D_s u ,   D_0 = identity, D_1..3 = d/dx, d/dy, d/dz (physical)
          P_rs are coefficient fields set with igx_patch_set_form: the block r,s >= 1 is a diffusion tensor
          (inner(dot(K,grad(u)),grad(v)): P = K), row 0 a convection vector (inner(b,grad(u))*v), column 0 its
          adjoint (u*inner(b,grad(v))), P_00 a reaction coefficient (c*u*v).  Non-symmetric.  What the
          reference compiles from a form string (pyiga/vform.py, pyiga/codegen/cython.py) for this class. */
       IGX_FORM = 3 };
enum { IGX_GEO_BSPLINE = 0, IGX_GEO_NURBS = 1, IGX_GEO_JACOBIAN = 2 };
/* algorithm selector for igx_assemble */
enum { IGX_ALGO_AUTO = 0,      /* sum-factorised when the patch supports it, else entry-wise */
       IGX_ALGO_ENTRYWISE = 1, /* one thread per matrix entry, the reference's own summation order */
       IGX_ALGO_SUMFACT = 2 }; /* global sum factorisation (stage kernels) */

enum { IGX_OK = 0, IGX_ERR_ARG = 1, IGX_ERR_HIP = 2, IGX_ERR_UNSUPPORTED = 3, IGX_ERR_NOMEM = 4,
       IGX_ERR_NORTC = 5,      /* an entry point that compiles at run time, on a box without libhiprtc: take the sampled-array path */
       IGX_ERR_COMPILE = 6 };  /* the expression handed to the run-time compiler did not compile (igx_last_error has the log) */

/* Knot vectors are in the reference's (z, y, x) order: axis 0 is the slowest dof index. */
typedef struct {
    int32_t dim;                       /* 2 or 3 */
    int32_t p[IGX_MAX_DIM];            /* spline degree per axis */
    int32_t kv_len[IGX_MAX_DIM];       /* number of knots per axis */
    const double *kv[IGX_MAX_DIM];     /* open knot vectors */

    int32_t geo_kind;                  /* IGX_GEO_* */
    int32_t geo_p[IGX_MAX_DIM];        /* geometry degrees (BSPLINE/NURBS) */
    int32_t geo_kv_len[IGX_MAX_DIM];
    const double *geo_kv[IGX_MAX_DIM];
    const double *ctrl;                /* control net, C order, shape (Ng0,..,Ng{d-1}, dim [+1 weight, premultiplied]) */
    const double *jac;                 /* IGX_GEO_JACOBIAN: (G0,..,G{d-1}, d, d), last axis = d/d(x,y,z) */

    int32_t nqp;                       /* Gauss points per span; 0 -> max(p)+1 as in the reference */
    const double *gauss_x;             /* optional nqp nodes on [-1,1] (NULL -> built-in Gauss-Legendre) */
    const double *gauss_w;             /* optional nqp weights */

    /* Row slab for multi-GPU: this patch owns the dof planes row0_lo <= i0 < row0_hi of axis 0
       (row0_hi = 0 means "all").  Only those CSR rows are produced. */
    int32_t row0_lo, row0_hi;

    /* Span box for on-demand assemblers (pyiga/codegen/cython.py:541-559, callers pyiga/_hdiscr.py:37-56,204-209): the
       geometry-dependent fields are kept only on the spans box_lo[k] <= s < box_hi[k] of every axis (all box_hi zero = whole
       patch).  A boxed patch serves igx_entries / igx_entries_d for pairs whose common support lies inside the box (others:
       NaN) and nothing else; IGX_GEO_JACOBIAN arrays and sampled coefficients are then given on the Gauss grid of the BOX.
       Present since igx_version() >= 101. */
    int32_t box_lo[IGX_MAX_DIM], box_hi[IGX_MAX_DIM];
} igx_patch_desc;

typedef struct {
    int32_t dim, nqp;
    int32_t ndofs[IGX_MAX_DIM], nspans[IGX_MAX_DIM], ngauss[IGX_MAX_DIM];
    int64_t nrows_total;               /* prod(ndofs) */
    int64_t row_lo, row_hi;            /* owned global rows [row_lo, row_hi) */
    int64_t nnz;                       /* nonzeros in the owned rows */
    int64_t nnz_offset;                /* global indptr[row_lo] */
    int64_t nelem_owned;               /* elements attributed to this slab (for throughput) */
    int32_t sumfact_ok;                /* 1 if the fast path supports this patch */
    int32_t reserved;
} igx_patch_info;

/* Device time of the last igx_assemble (HIP events on the ctx stream), milliseconds.  total_ms is always measured.  The
   per-kernel fields need events between the kernels and a marker costs ~5 us of stream time: they are recorded for 3D
   patches of >= 2^24 Gauss points (kernels of milliseconds) or when the patch was created under IGX_STAGE_EVENTS=1, and
   are 0 otherwise (a single-launch 2D assembly reports its one kernel as stage1_ms = total_ms). */
typedef struct {
    float total_ms;
    float fields_ms, stage0_ms, stage1_ms, final_ms, entry_ms;
    int32_t algo_used;                 /* IGX_ALGO_ENTRYWISE / IGX_ALGO_SUMFACT; 3 after igx_load_vector (total_ms = its contractions) */
    int32_t n_launches;
} igx_timing;

/* kernels the last igx_assemble(.., IGX_ALGO_SUMFACT) of a patch ran (igx_patch_last_path) */
#define IGX_PATH_GEOA    1   /* geometry + axis-0 sweep fused (k_geoA): timing.fields_ms ~ 0, stage0_ms = k_geoA */
#define IGX_PATH_FUSED   2   /* sweep + final stage fused (k_bf): timing.stage1_ms = k_bf */
#define IGX_PATH_MIRROR  4   /* upper triangle by the transposing mirror pass: timing.final_ms = k_mirror */
#define IGX_PATH_SINGLE  8   /* 2D mass / stiffness in ONE launch (k_single2d: fields, sweep and contraction in LDS): timing.stage1_ms */
#define IGX_PATH_BOTH   32   /* the fused stage wrote both triangles of a symmetric form itself (k_bf3): no mirror pass; timing.stage1_ms = k_bf3 */
#define IGX_PATH_BF3    64   /* the fused stage ran as k_bf3 (fused3.hip: entry rings per line, store duty on the sweeper waves) */
#define IGX_PATH_TWIN  128   /* repeated knots on the last axis: the chain ran on the patch with mid and last axis exchanged, k_bf3 stored to this patch's layout */
#define IGX_PATH_KRON   16   /* separable geometry: 2D matrices of the cross-section expanded by k_kron3 (igx_assemble_kron3): timing.final_ms */

int         igx_version(void);
const char *igx_last_error(void);

igx_ctx *igx_create(int device_id);                 /* NULL on failure */
void     igx_destroy(igx_ctx *ctx);
int      igx_sync(igx_ctx *ctx);
void    *igx_stream(igx_ctx *ctx);                  /* hipStream_t of this context */

/* --- B-spline evaluation (bspline_cy.pyx) ------------------------------------------------- */
/* out has shape (numderiv+1, p+1, nu), C order, exactly like pyiga.bspline_cy.active_deriv */
int igx_active_deriv(igx_ctx *ctx, const double *kv, int kv_len, int p,
                     const double *u, size_t nu, int numderiv, double *out);
/* spans[i] = pyx_findspan(kv, p, u[i]) */
int igx_find_spans(igx_ctx *ctx, const double *kv, int kv_len, int p,
                   const double *u, size_t nu, int64_t *spans);

/* --- geometry on a tensor grid (bspline.py:874-921, geometry.py:103-123) ------------------- */
/* desc uses only dim, geo_kind (BSPLINE/NURBS), geo_p, geo_kv_len, geo_kv, ctrl.
   ncomp = number of output components (geo.dim).  jac_out: (n0,..,n{d-1}, ncomp, d); eval_out:
   (n0,..,n{d-1}, ncomp).  Either output may be NULL. */
int igx_grid_jacobian(igx_ctx *ctx, const igx_patch_desc *desc, int ncomp,
                      const double *const grid[IGX_MAX_DIM], const int32_t ngrid[IGX_MAX_DIM],
                      double *jac_out, double *eval_out);

/* --- patch = assembler state on the device -------------------------------------------------- */
igx_patch *igx_patch_create(igx_ctx *ctx, const igx_patch_desc *desc);   /* NULL on failure */
void       igx_patch_destroy(igx_patch *patch);
int        igx_patch_get_info(const igx_patch *patch, igx_patch_info *info);

/* Scalar coefficient field of IGX_CONVDIFF on the FULL tensor Gauss grid (G0 x G1 x G2, C order, host
   pointer; what pyiga.utils.grid_eval_transformed(diff_coeff, gaussgrid, geo) returns).  Copied. */
int igx_patch_set_coeff(igx_patch *patch, const double *coeff);
/* The same coefficient given as an affine function of the PHYSICAL coordinates, c(x) = c[0] + c[1] x + c[2] y + c[3] z:
   evaluated on the device through the geometry map, nothing is sampled or shipped by the host. */
int igx_patch_set_coeff_affine(igx_patch *patch, const double c[4]);

/* The same coefficient as an EXPRESSION in the physical coordinates: `expr` is a C expression in the doubles x, y, z (and
   pi), e.g. "1.0 + x * x + 0.5 * sin(pi * z)".  The library emits a kernel for it, compiles it with hiprtc for the device
   of the patch, keeps the code object on disk under the hash of its source ($IGX_CACHE_DIR, default ~/.cache/igx) and
   evaluates it on the resident Gauss points from the geometry map -- nothing is sampled on the host.  *cache_hit (may be
   NULL) = 1 when the code object came from the cache.  Spline geometries only.  The analogue of the reference's run-time
   compiled assemblers and their source-hash module cache (pyiga/compile.py:58-73,120-132), for the one input the jet-form
   kernels cannot express themselves. */
int igx_patch_set_coeff_expr(igx_patch *patch, const char *expr, int *cache_hit);

/* A function given as a C expression in x, y, z (same grammar), evaluated at the RESIDENT Gauss points into the device array
   d_out (igx_patch_gauss_slab planes x G1 [x G2]) -- the input of igx_load_vector_d, without sampling the function on the
   host and uploading it (the reference samples f on the Gauss grid: pyiga/assemble.py:283-309).  parametric = 0: x, y, z are
   the physical coordinates (spline geometry needed); 1: the parametric coordinates of the Gauss points. */
int igx_patch_eval_expr_d(igx_patch *patch, const char *expr, int parametric, double *d_out, int *cache_hit);

/* Host only (no device, no patch): compile the coefficient kernel of `expr` for the architecture `arch` ("gfx950") into the
   cache, or find it there; path_out (may be NULL) receives the file name.  What igx_patch_set_coeff_expr does first. */
int igx_rtc_compile(const char *expr, const char *arch, char *path_out, int path_len, int *cache_hit);

/* Coefficients of IGX_FORM: coef[4*r + s] is P_rs on the FULL tensor Gauss grid (G0 x G1 x G2, C order, host
   pointer) or NULL for an absent (zero) coefficient; r = jet index of the test function v, s = of the trial
   function u.  Copied.  Replaces the previous form of the patch. */
int igx_patch_set_form(igx_patch *patch, const double *const coef[16]);
/* The same with coefficient arrays that already live on the device, each over the RESIDENT Gauss slab
   (igx_patch_gauss_slab planes x G1 [x G2], C order): no host staging of full-grid arrays.  A failed call of either
   variant leaves the previously set form untouched. */
int igx_patch_set_form_d(igx_patch *patch, const double *const d_coef[16]);

/* The same table given as C expressions in the physical coordinates x, y, z (and pi; the grammar of
   igx_patch_set_coeff_expr): expr[4*r + s] or NULL.  The FIELD kernel of the form is generated with the expressions inside
   -- geometry map, Jacobian, coefficients and the transformation to the parametric jet coefficients in one pass over the
   resident Gauss points; the coefficients never exist as arrays -- compiled for the device with hiprtc and cached on disk
   under the hash of its source.  The reference generates the field loop of a form with its inputs fused in and caches the
   compiled module the same way (pyiga/codegen/cython.py:673-701 generate_precomp, :325-387; pyiga/compile.py:58-73,
   120-132); nothing is sampled on the host.  Needs a spline geometry.  (A geometry whose control lines exceed the LDS of a
   block takes a generated kernel that evaluates the coefficients into arrays first.)  The call compiles; the kernel runs
   with the next operation that needs the fields.  *cache_hit (may be NULL): 1 if the code object came from the cache. */
int igx_patch_set_form_expr(igx_patch *patch, const char *const expr[16], int *cache_hit);
/* 1 if the patch's IGX_FORM is served by a generated field kernel (no coefficient arrays on the device), else 0. */
int igx_patch_form_generated(const igx_patch *patch);
/* Host only: compile the kernel that evaluates the n expressions into arrays for `arch` into the cache. */
int igx_rtc_compile_form(int n, const char *const *expr, const char *arch, char *path_out, int path_len, int *cache_hit);
/* Host only: compile the field kernel of the form expr[16] for a dim-dimensional geometry with ncomp components
   (ncomp = dim + 1: NURBS) for `arch` into the cache (what igx_patch_set_form_expr does). */
int igx_rtc_compile_form_fields(int dim, int ncomp, const char *const expr[16], const char *arch, char *path_out, int path_len, int *cache_hit);

/* Parametric jet form for IGX_FORM -- forms with second derivatives (hess, Dx(.., times=2)) and parametric derivatives
   (parametric=True) of the reference (pyiga/vform.py:592-625 physical Hessians from parametric ones, :1518-1586 Dx / grad /
   hess), which its code generator differentiates symbolically (pyiga/vform.py:540-607) and compiles per form:

       a(u, v) = sum_k  integral over the parameter domain of  c_k(xi) * D^(mv_k) v(xi) * D^(mu_k) u(xi)  d xi

   with n <= 16 terms per call.  masks[2k] = mv_k, masks[2k+1] = mu_k: bit a set = on GRID axis a the function enters with
   slot 1 of that axis' basis table, else with slot 0.  The slots hold the derivative orders set by igx_patch_set_basis_orders
   ((0, 1) = value / first derivative by default), so one call covers the terms whose derivative orders fit one choice of two
   orders per axis; the host (pyiga_amd/pforms.py) splits a form into such passes and adds the results.  coef[k]: host array
   over the FULL Gauss grid, WITHOUT the Gauss weights (the geometry factors -- |det J|, J^-1, the Hessian of the geometry map --
   are the caller's: they are part of c_k).  Copied; replaces the previous form of the patch (either kind). */
int igx_patch_set_pform(igx_patch *patch, int n, const int *masks, const double *const *coef);
/* Derivative orders (0 <= slot0[a] < slot1[a] <= 2, one pair per axis) held by the two slots of the basis tables.  With
   anything but (0, 1) on every axis the patch assembles a parametric jet form only (every other call fails with
   IGX_ERR_ARG until the default is restored). */
int igx_patch_set_basis_orders(igx_patch *patch, const int *slot0, const int *slot1);

/* Gauss grid and weights of axis k (host copies; length ngauss[k]) */
int igx_patch_gauss(const igx_patch *patch, int axis, double *nodes, double *weights);

/* Resident Gauss planes of axis 0: a row slab keeps only the planes its rows touch, [*g0_lo, *g0_lo + *g0_n).
   Device-side inputs (igx_load_vector_d) are laid out on this slab. */
int igx_patch_gauss_slab(const igx_patch *patch, int64_t *g0_lo, int64_t *g0_n);

/* IGX_PATH_* bits of the last sum-factorised assembly of this patch (0: none yet / entry-wise) */
int igx_patch_last_path(const igx_patch *patch);

/* Canonical CSR pattern of the owned rows.  indptr has (row_hi-row_lo+1) entries, LOCAL
   (indptr[0] = 0); indices has nnz entries (global column ids).  Either may be NULL.
   The pattern is also kept on the device. */
int igx_pattern(igx_patch *patch, int32_t *indptr, int32_t *indices);

/* Assemble all CSR values of the owned rows into device memory (symmetric: lower triangle
   computed, strict lower part mirrored -> exactly symmetric, as assemble_entries(symmetric=True);
   IGX_CONVDIFF is non-symmetric: every entry is computed, as assemble_entries(symmetric=False)).
   If data_out != NULL the nnz values are also copied to the host buffer. */
int igx_assemble(igx_patch *patch, int kind, int algo, double *data_out);
int igx_last_timing(const igx_patch *patch, igx_timing *t);

/* Device pointers of the most recent results (valid until the patch is destroyed) */
const double  *igx_d_csr_data(const igx_patch *patch);
const int32_t *igx_d_csr_indices(const igx_patch *patch);
const int32_t *igx_d_csr_indptr(const igx_patch *patch);

/* Low-rank (adaptive cross approximation) assembly of the mass / stiffness matrix of a whole patch: the algorithm of
   pyiga/fastasm.cc:294-494,701-760 (fast_assemble_2d/3d, reached from mass_fast/stiffness_fast, pyiga/assemble.py:1063-1101)
   on the host, pulling whole rows / columns / fibres of the reordered matrix from the device with batched entry requests
   instead of one callback per entry.  data_out: nnz values in the canonical CSR order of igx_pattern (approximate to
   `tol`).  rank_out: crosses added; entries_out: entries actually evaluated.  Defaults of the reference: tol 1e-10,
   maxiter 100, skipcount 3, tolcount 3; verbose 0..2 prints the reference's progress lines to stdout. */
int igx_fast_assemble(igx_patch *patch, int kind, double tol, int maxiter, int skipcount, int tolcount, int verbose,
                      double *data_out, int *rank_out, long long *entries_out);
/* Request granularity of igx_fast_assemble.  A request is one launch: the index pairs of whole lines / slices of the
   reordered tensor are generated on the device from resident per-axis tables (nothing is uploaded), only the values come
   back.  A slice (3D) or the whole matrix (2D) with at most `max_entries` entries is fetched exactly in ONE request instead
   of being approximated line by line (on this hardware a launch costs as much as ~2000 entries); 0 = always line by line (the
   reference's access pattern).  Default 65536. */
int igx_patch_set_aca_batch(igx_patch *patch, long long max_entries);
/* Counters of the last igx_fast_assemble of the patch: batched requests (launches), entries evaluated, crosses. */
int igx_fast_assemble_stats(const igx_patch *patch, long long *requests, long long *entries, int *rank);

/* multi_entries: ij is M x 2 (row, col) of ravelled dof indices; out[k] = 0.0 for pairs whose
   supports do not intersect.  Works for any pair, inside or outside the owned slab provided the
   fields of the pair's Gauss points are resident (always true for a full patch). */
int igx_entries(igx_patch *patch, int kind, const size_t *ij, size_t M, double *out);

/* The same with the index pairs and the results in device memory (no copies; pyiga/genericasm.pxi:722-758). */
int igx_entries_d(igx_patch *patch, int kind, const size_t *d_ij, size_t M, double *d_out);

/* Device buffers that stay resident between calls (function values, index pairs, results): plain hipMalloc'ed memory on the
   context's GPU; uploads / downloads are synchronous on the context's stream. */
void *igx_dev_alloc(igx_ctx *ctx, size_t bytes);                               /* NULL on failure */
void  igx_dev_free(igx_ctx *ctx, void *d_ptr);
int   igx_dev_upload(igx_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int   igx_dev_download(igx_ctx *ctx, void *dst, const void *d_src, size_t bytes);

/* Load vector of the owned rows: out[i] = sum over the Gauss grid of  B_i * f * gw0*gw1*gw2*|det J|
   (inner_products(kvs, f, geo=geo), L2FunctionalAssembler*.assemble_vector()).  fvals: the function on the
   FULL tensor Gauss grid (G0 x G1 [x G2], C order, host), i.e. utils.grid_eval(f, gaussgrid) or
   grid_eval_transformed(f, gaussgrid, geo); out: (row0_hi-row0_lo) x N1 [x N2] doubles (host). */
int igx_load_vector(igx_patch *patch, const double *fvals, double *out);

/* The same with the function values of the RESIDENT Gauss slab (G0_local x G1 [x G2], C order) and the result in device
   memory: no transfer, workspace owned by the patch (pyiga/assemble.py:288-340 with f already sampled). */
int igx_load_vector_d(igx_patch *patch, const double *d_fvals, double *d_out);

/* Linear functional in the first-order jet of v:  out[i] = integral of  sum_r F_r(x) D_r v_i  dx  (D_0 = id, D_1.. =
   physical derivatives).  coef[r]: F_r on the FULL tensor Gauss grid (host) or NULL.  Arity-1 form strings such as
   'inner(b, grad(v)) * dx' (pyiga/assemble.py:837-897).  Replaces the IGX_FORM coefficients of the patch. */
int igx_load_vector_jet(igx_patch *patch, const double *const coef[4], double *out);
/* The same functional with its coefficients given as C expressions in the physical coordinates x, y, z (expr[r] or NULL; the
   grammar and the run-time compilation of igx_patch_set_form_expr): nothing is sampled on the host. */
int igx_load_vector_jet_expr(igx_patch *patch, const char *const expr[4], double *out, int *cache_hit);
/* Load vector of a scalar function given as ONE C expression in x, y, z -- the physical coordinates, or with parametric != 0
   the parametric ones (pyiga/assemble.py:288-340, inner_products with f_physical = True / False).  3D patches that the fused
   contraction kernel serves (equal degrees 1..5 on the last two axes, at most 640 Gauss points per line): a generated variant
   of that kernel evaluates the function at the points of its grid line -- the function values never exist as an array; the
   weight field of the patch is computed once and kept.  Anything else: IGX_ERR_UNSUPPORTED (igx_patch_eval_expr_d +
   igx_load_vector_d serve it with one full-grid array).  A function of the physical coordinates needs a spline geometry. */
int igx_load_vector_expr(igx_patch *patch, const char *expr, int parametric, double *out, int *cache_hit);
/* Host only: compile that kernel (P = degree + 1 of the last two axes, npass = max(2, ceil(dofs of the last axis / 64)) <= 4). */
int igx_rtc_compile_load_vector(int P, int npass, int parametric, const char *expr, const char *arch, char *path_out, int path_len, int *cache_hit);

/* Precomputed fields (W or upper triangle of B) of the owned Gauss slab: out has shape
   (F, G0_local, G1[, G2]) (structure-of-arrays).  For tests of precompute_fields. */
int igx_fields(igx_patch *patch, int kind, double *out, int64_t *shape4);

/* Mass / stiffness matrix of a 3D patch whose geometry is SEPARABLE along axis 0 -- G(xi0, xi1, xi2) = (g(xi1, xi2), z(xi0)),
   an extruded cross-section -- as the sum of Kronecker products
       mass:  M0 (x) M2D            stiffness:  M0 (x) K2D + K0 (x) M2D
   patch2: the 2D patch of the cross-section (axes 1, 2 of patch3: same knot vectors and the SAME number of Gauss points
   per span; the map g); m0, k0 (host, [ndofs0][2 p0 + 1], row i0, column jlo(i0) + k, zeros where no column): the 1D matrices
   int phi_i phi_j |z'| and int phi_i' phi_j' / |z'| on the Gauss rule of axis 0 of patch3 (k0 unused for the mass form).
   The library assembles the 2D matrices on patch2 and expands them into the canonical CSR values of the owned rows of
   patch3 (device-resident like igx_assemble; data_out as there, may be NULL).  The caller establishes separability (the
   Python front-end inspects the control net: geometry.split_axis0).  What the reference does for geo = None with 1D
   matrices (pyiga/assemble.py:125-190,236-282), extended to separable geometry maps. */
int igx_assemble_kron3(igx_patch *patch3, igx_patch *patch2, int kind, const double *m0, const double *k0, double *data_out);

/* Opt-in buffer placement (environment IGX_PLACEMENT_TRIES=n at igx_patch_create, 3D symmetric forms): the first
   igx_assemble allocates up to n candidate buffers for the CSR values, times the mirror pass of the patch on each and keeps
   the fastest (the pass follows where the driver put the buffer physically: +-10 % between allocations, stable over the life
   of a buffer).  Reports how many candidates were timed (0: the search did not run) and the device time of the pass on the
   kept and on the slowest one.  (No counterpart in the reference.) */
int igx_patch_placement(const igx_patch *patch, int *tried, float *best_ms, float *worst_ms);

/* Planning query, host arithmetic only (no device, no patch): 1 if the fused sweep + contraction stage and the mirror pass
   may address a patch of these sizes with their 32-bit buffer offsets, 0 if the library takes the stage kernels with 64-bit
   addressing instead.  c0max = 2 p0 + 1 columns of axis 0 per row (1 in 2D), S_mid / S_last = number of 1D index pairs
   (i, j) of the mid / last axis, G_mid / G_last = their Gauss points.  The limits: a row block of one outer row
   (c0max * S_mid * S_last values) and one slice of the sweep intermediate (G_mid * G_last values) below 2^31 bytes.
   (No counterpart in the reference: its index type is size_t throughout, pyiga/assemble_tools_cy.pyx:44-49.) */
int igx_fused_stage_fits(int64_t c0max, int64_t S_mid, int64_t S_last, int64_t G_mid, int64_t G_last);

#ifdef __cplusplus
}
#endif
#endif /* IGX_H */

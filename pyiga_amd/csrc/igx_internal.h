// Internal declarations shared by the libigx translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "igx.h"

namespace igx {

constexpr int MAXP = IGX_MAX_DEGREE + 1;      // max active functions per span
constexpr int MAX_COMP = 4;                   // geometry components incl. NURBS weight

void set_error(const char *fmt, ...);
#define IGX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            igx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return IGX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Device-side views (POD, passed to kernels by value)
struct AxisDev {
    int p, P, N, n, q, G;     // degree, p+1, ndofs, nspans, Gauss pts/span, G = n*q
    int S;                    // number of 1D (i,j) pairs with overlapping support
    const double *nodes;      // [G]
    const double *w;          // [G]
    const double *V;          // [G][P][2]   (value, derivative) of the P active functions
    const double *PI;         // [G][4][P][P] products  PI[t][a][b] = V[b][tu]*V[a][tv], t = tu + 2*tv
                              //              (a: test function i, b: trial function j)
    const int *fa;            // [n]   first active dof of span s
    const int *mslo, *mshi;   // [N]   mesh-span support [mslo, mshi) of dof i
    const int *jlo, *jhi;     // [N]   1D column range of row i
    const int *rp;            // [N+1] exclusive prefix sum of (jhi - jlo): 1D pair index base
    const int *pair_i, *pair_j; // [S]  inverse of the pair index: r -> (i, j)
};

struct PatchDev {
    int dim;
    AxisDev ax[3];
    int r0_lo, r0_hi;         // owned dof planes of axis 0
    int s0_lo, s0_hi;         // resident span range of axis 0
    int g0_lo;                // first resident Gauss index of axis 0 (= s0_lo * q)
    int G0_loc;               // number of resident Gauss planes
    long long npts_loc;       // resident Gauss points = G0_loc * L1 [* L2]
    int b1, b2, L1, L2;       // resident Gauss window of axes 1, 2: first index and extent (whole axes unless the patch is boxed)
    long long nnz_off;        // global indptr[row_lo]
    // IGX_FORM: the terms that are present, in field order; ab = 4 * (jet index of v) + (jet index of u),
    // jet index 0 = value, 1..3 = PARAMETRIC derivative in (x, y, z) order (x = last grid axis)
    // form_par (igx_patch_set_pform): the fields ARE the parametric coefficients (times the Gauss weight); term k differentiates
    // v on the grid axes of mask form_ab[k] >> 3 and u on those of mask form_ab[k] & 7 ("differentiates" = takes slot 1 of the
    // axis' basis table, whose two slots hold the derivative orders chosen with igx_patch_set_basis_orders)
    int form_n;
    int form_ab[16];
    int form_par;
};

// index of Gauss point (g0, g1, g2) in the resident field arrays
__host__ __device__ inline long long field_index(const PatchDev &pd, int g0, int g1, int g2)
{
    const long long l = (long long)(g0 - pd.g0_lo) * pd.L1 + (g1 - pd.b1);
    return pd.dim == 3 ? l * pd.L2 + (g2 - pd.b2) : l;
}

// IGX_FORM coefficients on the resident Gauss slab
struct FormView {
    const double *c;          // [ncoef][npts_loc]
    int slot[16];             // physical coefficient 4*r+s -> row of c, or -1
};

// global CSR row pointer of row (i0,i1,i2); see DESIGN.md "CSR pattern without index arrays"
__host__ __device__ inline long long igx_rowptr3(const int *rp0, const int *rp1, const int *rp2,
                                                 long long S1, long long S2, int c0, int c1,
                                                 int i0, int i1, int i2)
{
    return (long long)rp0[i0] * S1 * S2 + (long long)c0 * ((long long)rp1[i1] * S2 + (long long)c1 * rp2[i2]);
}

// ---------------------------------------------------------------------------------------------
// Host-side state
struct Axis {
    int p = 0, P = 0, N = 0, n = 0, q = 0, G = 0, S = 0;
    bool simple = true;                       // all interior knots have multiplicity 1
    std::vector<double> kv, mesh, nodes, weights;
    std::vector<int> span_knot, fa, mslo, mshi, jlo, jhi, rp, pair_i, pair_j;
    // device
    double *d_kv = nullptr, *d_nodes = nullptr, *d_w = nullptr, *d_V = nullptr, *d_PI = nullptr;
    int *d_ints = nullptr;                    // one allocation: fa | mslo | mshi | jlo | jhi | rp | pair_i | pair_j
    int ord[2] = {0, 1};                      // derivative orders in the two slots of V (and PI)
    AxisDev dev{};
};

struct GeoAxis {                              // geometry basis restricted to the Gauss nodes
    int p = 0, P = 0, N = 0;
    std::vector<double> kv;
    double *d_kv = nullptr;
    double *d_V = nullptr;                    // [G][P][2]
    int *d_fa = nullptr;                      // [G] first active control index at node g
};

struct Plan;                                   // sum-factorisation plan (sumfact.hip)

} // namespace igx

struct igx_ctx {
    int device = 0;
    int ncu = 0;                              // compute units of THIS context's device (filled by igx_create): launch geometry
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
};

// Kernel-chain choices of a patch, read from the environment ONCE when the patch is created (igx_patch_create): every
// assembly of the patch takes the same path, whatever happens to the environment afterwards.  None selects a CPU path.
struct igx_knobs {
    int path = 0;                             // IGX_PATH: 0 default (fused in 3D; 2D: one launch for small patches, else stage kernels), 1 fused, 2 unfused, 3 single (2D)
    int geoa = 1;                             // IGX_GEOA=0: separate field and axis-0 sweep kernels
    int geoa_mf = 0;                          // IGX_GEOA=mfma: the FP64 matrix-core form of the axis-0 sweep inside k_geoA at p = 3, 4 (built and
                                              // measured in round 4: 5.08 against 4.50 ms for the vector form at C4 -- DESIGN.md section 3; off by default)
    int final_sel = 0;                        // IGX_FINAL: 0 default, 1 q, 2 valu, 3 mfma (any choice implies the stage kernels)
    int entries_thread = 0;                   // IGX_ENTRIES=thread: one thread per entry (the reference's summation order)
    int poison = 0;                           // IGX_DEBUG_POISON: NaN-fill the CSR values before an assembly (tests)
    int bf = 0;                               // IGX_BF=2: symmetric 3D forms through k_bf2 + the mirror pass (the chain of rounds 3-4) instead of k_bf3
    int placement_tries = 1;                  // IGX_PLACEMENT_TRIES=n (opt-in, 3D symmetric forms): the CSR value buffer is the fastest of n
                                              // allocations under the mirror pass, timed once at its first assembly
    int stage_events = -1;                    // IGX_STAGE_EVENTS: events between the kernels of a chain (per-kernel device times in
                                              // igx_last_timing).  A marker costs ~5 us of stream time: default on for 3D patches of
                                              // >= 2^24 Gauss points (kernels of milliseconds), off below and in 2D (kernels of
                                              // 5-40 us), where only the whole interval is timed
};

struct igx_patch;
static inline void stage_event(const igx_patch *pt, int k, hipStream_t st);

struct igx_patch {
    igx_ctx *ctx = nullptr;
    igx_knobs knobs;
    bool basis_default = true;                // every axis' table holds (value, first derivative) (igx_patch_set_basis_orders)
    bool boxed = false;                       // fields only on a span box (igx_patch_desc.box_*): batched entries only
    int dim = 0, nqp = 0;
    igx::Axis ax[3];
    // geometry
    int geo_kind = 0, ncomp = 0;              // ncomp = dim (+1 for NURBS)
    igx::GeoAxis gax[3];
    double *d_ctrl = nullptr;
    double *d_jac = nullptr;                  // IGX_GEO_JACOBIAN: resident slab of the user array
    double *d_formc = nullptr;                // IGX_FORM: physical coefficient fields [n][npts_loc]
    int form_slot[16];                        //   4*r+s -> row of d_formc or -1
    void *form_fn = nullptr;                  // IGX_FORM given as expressions: the generated field kernel (rtc.hip); no d_formc then
    // ... and its PHYSICAL coefficient table as the fast chain reads it (k_geoA<FORM = 2 | 3> evaluates the fields inside the sweep;
    // sumfact.hip: form_table_plan): entry 4 r + s is a constant, an expression (sampled into ftab_arr when the chain first runs),
    // or absent
    struct FormTable {
        bool valid = false, sym = false, blocksym = false;    // sym: P_rs == P_sr for all r, s; blocksym: for r, s >= 1
        int present = 0, is_const = 0;                        // bit masks over 4 r + s
        double cval[16] = {0};
        std::string expr[16];
        int arr_of[16];                                       // index of the entry's array in ftab_arr, or -1
        int narr = 0;
    } ftab;
    double *ftab_arr = nullptr;                               // [narr][npts_loc] sampled non-constant entries (lazily)
    bool ftab_ready = false;
    // Repeated knots on the LAST axis only (round 6): k_bf3 contracts an axis of single knots and sweeps one that may have repeated
    // ones, so such a patch is assembled through its TWIN -- the same patch with mid and last axis exchanged (knot vectors, control
    // net), created with it -- whose k_bf3 stores straight into the CSR layout of this patch (fused3.hip, TR).  twin_kinds: bit per
    // IGX_* kind the twin's fast chain serves; is_twin: this patch is one (its values belong to its owner's layout).
    igx_patch *twin = nullptr;
    int twin_kinds = 0;
    bool is_twin = false;
    double *d_coeff = nullptr;                // IGX_CONVDIFF: scalar coefficient on the resident Gauss slab
    int coef_affine = 0;                      // ... set by igx_patch_set_coeff_affine: k_geoA evaluates it from the geometry map
    bool coeff_sampled = false;               // d_coeff holds the values of the current coefficient (an affine one is sampled only when a kernel that reads the array runs: ensure_coeff)
    double coef_c[4] = {0, 0, 0, 0};
    // slab
    int r0_lo = 0, r0_hi = 0, s0_lo = 0, s0_hi = 0;
    long long row_lo = 0, row_hi = 0, nnz = 0, nnz_off = 0, nrows_total = 0, nelem_owned = 0;
    igx::PatchDev dev{};
    // fields cache
    int fields_kind = -1;
    double *d_fields = nullptr;               // [F][npts_loc]
    size_t fields_cap = 0;
    // CSR
    double *d_data = nullptr;
    int32_t *d_indices = nullptr, *d_indptr = nullptr;
    bool have_pattern = false;
    // sum factorisation workspaces
    bool sumfact_ok = false;
    int *d_pl0 = nullptr;                     // [npairs0][2] processed lower pairs (i0,j0) of axis 0
    int *d_rl0_of = nullptr;                  // [S0] 1D pair index -> compact processed-pair index or -1
    int npairs0 = 0;
    // non-symmetric forms (built on first use): all pairs (i0, j0) of the owned rows i0
    int *d_pl0n = nullptr, *d_stepsn = nullptr;
    int npairs0n = -1;
    // 32-byte line descriptors of the quadrature-lane final kernel (symmetric / non-symmetric pair list)
    int *d_qdesc = nullptr, *d_qdescn = nullptr;
    long long n_qdesc = 0, n_qdescn = 0;
    std::vector<int> h_pl0;                   // host copy of d_pl0 (the line descriptors are built lazily from it)
    bool desc_built = false;
    int *d_ldesc = nullptr;                   // [n_ldesc][4] line descriptors of the final stage
    int n_ldesc = 0;
    bool ldesc_ok = false;
    int *d_steps = nullptr;                   // flush-step tables of the sweeps (one allocation)
    const int *stepA_ptr = nullptr, *stepA_rec = nullptr, *stepB_ptr = nullptr, *stepB_rec = nullptr;
    float placement_ms_best = 0, placement_ms_worst = 0;   // IGX_PLACEMENT_TRIES: mirror pass on the kept / the slowest candidate
    int placement_tried = 0;
    double *d_geoa_tab = nullptr;             // per-plane records of axis 0 for k_geoA (geoa.hip), built on first use
    int geoa_mf = 0;                          // ... and whether they carry the row tables of the matrix-core sweep
    double *d_geoa_tabn = nullptr;            // the same for the non-symmetric sweep (16-int flush steps), built on first use
    double *d_K1 = nullptr, *d_K2 = nullptr;
    size_t K1_cap = 0, K2_cap = 0;
    // persistent workspaces of the batched-entry and load-vector entry points (grow-only, freed with the patch)
    size_t *d_ws_ij = nullptr; double *d_ws_out = nullptr;
    size_t ws_ij_cap = 0, ws_out_cap = 0;
    long long aca_batch = 65536;              // igx_patch_set_aca_batch
    struct S2DPlan { int valid = 0, ok = 0, R0 = 0, R1 = 0, NG0 = 0, WIN = 0, NCOL = 0; size_t bytes = 0; };
    mutable S2DPlan s2d[2][2];                // tile of the single-launch 2D kernel [mass | stiffness][resident rows | whole patch], worked out once (kern_basis.hip)
    long long aca_requests = 0, aca_entries = 0; int aca_rank = 0;      // igx_fast_assemble_stats
    double *d_lv_f = nullptr, *d_lv_t1 = nullptr, *d_lv_t2 = nullptr, *d_lv_o = nullptr;
    size_t lv_f_cap = 0, lv_t1_cap = 0, lv_t2_cap = 0, lv_o_cap = 0;
    // fused sweep + final stage (fused.hip)
    long long nnz_ext = 0;                    // values of the owned rows + the p0 halo planes above them (mirror sources)
    double *d_zeros = nullptr;                // a row of zeros: input of absent sweep slots
    int *d_triv = nullptr;                    // one-dof outer axis of the 2D case: pl0 {0,0} | rp0 {0,1} | jlo0 {0} | jhi0 {1}
    int *d_tpairs = nullptr;                  // [ntp][2] mirror targets: outer pairs (i0 owned, j0 >= i0)
    int ntp = 0;
    int last_path = 0;                        // kernels of the last sum-factorised assembly: IGX_PATH_* bits
    igx_timing timing{};
};

static inline void stage_event(const igx_patch *pt, int k, hipStream_t st)
{
    if (pt->knobs.stage_events) (void)hipEventRecord(pt->ctx->ev[k], st);
}


// ---------------------------------------------------------------------------------------------
// kernel launchers (defined in the kern_*.hip files)
namespace igx {
int launch_basis_tables(hipStream_t st, const double *d_kv, int nk, int p, const double *d_u, size_t nu,
                        int numderiv, double *d_out_nd_p_n /* (nd+1,P,nu) or null */,
                        double *d_V /* [nu][P][2] or null */, int *d_fa /* [nu] or null */,
                        long long *d_spans /* or null */, int o0 = 0, int o1 = 1 /* derivative orders of the two V slots */);
int launch_pi_tables(hipStream_t st, const double *d_V, int G, int P, double *d_PI);
int launch_geo_fields(hipStream_t st, const igx_patch *pt, int kind, double *d_fields);
// 2D mass / stiffness in one launch, no intermediates (kern_basis.hip)
bool single2d_supported(const igx_patch *pt, int kind);
long long single2d_blocks(const igx_patch *pt, int kind, int *tile_rows, bool whole_patch);   // launch grid of the single-launch kernel (-1: no tile fits), rows per tile; for the resident rows or as if the patch were whole
int launch_single2d(hipStream_t st, igx_patch *pt, int kind, double *d_data);
int launch_grid_geo(hipStream_t st, int dim, int ncomp_total, bool nurbs, const GeoAxis gax[3],
                    const int G[3], const double *d_ctrl, double *d_jac, double *d_eval);
int launch_fields_dump(hipStream_t st, const igx_patch *pt);
int launch_coeff_affine(hipStream_t st, const igx_patch *pt, const double c[4], double *d_coeff);
// Kronecker expansion of a separable geometry (kron.hip)
int launch_kron3(hipStream_t st, igx_patch *p3, const double *d_a0, const double *d_b0, int C0, const double *d_A2, const double *d_B2);
// run-time compiled coefficient expressions (rtc.hip)
int launch_coeff_expr(hipStream_t st, igx_patch *pt, const char *expr, double *d_coeff, int *cache_hit, bool parametric = false /* x, y, z = parametric coordinates */);
int lv12_expr_applicable(const igx_patch *pt, int parametric);   // IGX_OK | IGX_ERR_UNSUPPORTED (shape) | IGX_ERR_NORTC (no libhiprtc)
int rtc_compile_expr(const char *expr, const char *arch, char *path_out, int path_len, int *cache_hit);
int launch_form_exprs(hipStream_t st, igx_patch *pt, int n_expr, const char *const *expr, double *d_out /* [n_expr][npts_loc] */, int *cache_hit);
int rtc_compile_form(int n_expr, const char *const *expr, const char *arch, char *path_out, int path_len, int *cache_hit);
// field kernel of a form given as expressions (geometry + coefficients + jet transformation in one generated kernel)
int rtc_form_fields_function(igx_patch *pt, const char *const expr[16], int nterms, const int *form_ab, void **fn_out, int *cache_hit, bool parametric = false);
int launch_form_fields(hipStream_t st, const igx_patch *pt, void *fn, double *d_fields);
bool form_fields_applicable(const igx_patch *pt);
// 3D load vector with the function inside the first two contractions (rtc.hip) + helpers of kern_vector.hip
int launch_lv12_expr(hipStream_t st, igx_patch *pt, const char *expr, int parametric, const double *d_W, double *d_t2, int *cache_hit);
int rtc_compile_lv12(int P, int npass, int parametric, const char *expr, const char *arch, char *path_out, int path_len, int *cache_hit);
bool lv12_shape(const igx_patch *pt, int *clen_out, int *nch_out, size_t *lds_out);
int launch_lv_axis0(hipStream_t st, const igx_patch *pt, const double *d_t2, double *d_out, int deriv0, int accumulate);
int form_terms(int dim, const char *const expr[16], int form_ab[16]);
int rtc_compile_form_fields(int dim, int ncomp, const char *const expr[16], const char *arch, char *path_out, int path_len, int *cache_hit);
int launch_pattern(hipStream_t st, const igx_patch *pt, int32_t *d_indptr, int32_t *d_indices);
int launch_entries_list(hipStream_t st, const igx_patch *pt, int kind, const size_t *d_ij, size_t M, double *d_out);
// boxes of the reordered tensor X[r0][r1][r2] (r_k = index of a 1D pair (i_k, j_k) with overlapping supports): the entries
// lo[b][k] <= r_k < lo[b][k] + len[b][k] of every box, box after box, lexicographic inside a box
struct PairBoxes { int n; int lo[8][3], len[8][3]; long long off[9]; };
int launch_box_pairs(hipStream_t st, const igx_patch *pt, const PairBoxes &B, size_t *d_ij);
int entries_pair_boxes(igx_patch *pt, int kind, const PairBoxes &B, double *out_host);   // (igx_api.hip: index pairs built on the device)
int launch_entries_csr(hipStream_t st, const igx_patch *pt, int kind, double *d_data);
int launch_load_vector(hipStream_t st, const igx_patch *pt, const double *d_f, const double *d_W, double *d_out,
                       double *d_t1, double *d_t2, int deriv_axis = -1, int accumulate = 0, int *n_launches = nullptr /* 2: k_lv12 + axis 0, else one per axis */);
inline int igx_num_fields(int dim, int kind, int form_n = 0)
{
    return kind == IGX_MASS ? 1 : (kind == IGX_CONVDIFF ? 9 : (kind == IGX_FORM ? form_n : dim * (dim + 1) / 2));
}
constexpr size_t IGX_DUMP_PAD = 1024 * 16 + 16;   // doubles behind the CSR values: 1024 dump lines of the final stage
inline bool igx_kind_symmetric(int kind) { return kind != IGX_CONVDIFF && kind != IGX_FORM; }
// fused sweep + final stage and mirror pass (fused.hip)
struct BFInputs {
    int pad_stiff3 = 0;                       // treat the slot set as the full first-order set (absent slots read the zero row): general forms
    int tr = 0;                               // the patch is the axis-exchanged twin of the caller's (igx_patch::twin): k_bf3 stores to the caller's CSR layout
    const Axis *mid, *last;                   // swept axis, last (contiguous) axis
    int slot_n[4][4];                         // [last-axis type y][mid-axis type t1]: number of input arrays (<= 2)
    const double *slot_ptr[4][4][2];          // their device pointers: array[slice][g_mid - gmid_lo][g_last]
    long long slice_stride;                   // doubles between slices (outer pairs)
    const double *zeros;                      // >= G_last zeros
    int gmid_lo;
    const int *pl0;                           // [npairs][2] outer pairs (device)
    int npairs;
    const int *rp0, *jlo0, *jhi0;             // outer axis tables (device)
    int sym, mid_lo, mid_hi;
    int span_hi;                              // resident spans of the mid axis end here
};
struct MirrorInputs {
    const Axis *mid, *last;
    const int *rp0, *jlo0, *jhi0;
    const int *tpairs;
    int ntp;
    int i1_lo, i1_hi;                         // target rows of the mid axis
};
int fused_supported(const BFInputs &in);
// 32-bit buffer offsets of k_bf2 / k_mirror2 (fused.hip): row block of an outer row and a K1 slice below 2^31 bytes
bool fused_offsets_fit(long long c0max, long long S_mid, long long S_last, long long G_mid, long long G_last);
int launch_bf(hipStream_t st, const igx_patch *pt, const BFInputs &in, double *d_data);
int launch_mirror(hipStream_t st, const igx_patch *pt, const MirrorInputs &in, double *d_data);
// k_bf3 (fused3.hip): the fused stage of the symmetric forms that writes both triangles itself (no mirror pass)
bool fused3_supported(const BFInputs &in);
bool fused3_offsets_fit(int dim, int p0, int p1, int p2, long long S_mid, long long S_last, long long N_last);
bool fused3_degrees(int P1, int P2, int Q, bool sym3d, bool mid_simple);
bool fused3_tr_fits(int p0, int p1, int p2, long long S_mid, long long S_last);
int launch_bf3(hipStream_t st, const igx_patch *pt, const BFInputs &in, double *d_data);
// fused geometry + stage A (geoa.hip)
bool geoA_supported(const igx_patch *pt, int kind, int nslots);
// table of a general first-order form for k_geoA (FORM = 2 | 3): see GeoAArgs (geoa.hip)
struct GeoAForm {
    int sym;                                  // symmetric table: lower pairs of axis 0, symmetric flush
    double pc[16];
    const double *pa[16];
    int pmask;
    int fslot[16];
    int nsrc[8], sfield[8][4], stype[8][4];
};
int launch_geoA(hipStream_t st, igx_patch *pt, int kind, int nslots, const int *slot_field, const int *slot_type,
                double *const *slot_out, long long slice_stride, int chunk_len, int nchunks, const int *slot_xfield = nullptr, const int *slot_xtype = nullptr,
                const GeoAForm *form = nullptr);
bool geoA_form_supported(const igx_patch *pt);
bool sumfact_needs_fields(const igx_patch *pt, int kind);
int sumfact_twin_kinds(const igx_patch *tw);                    // IGX_* kinds (bit mask) the fast chain of an axis-exchanged twin serves
bool sumfact_single_launch(const igx_patch *pt, int kind);     // the 2D single-launch kernel will run (no stage events inside)
// device time of the mirror pass of this patch on `buf` (access pattern only: the values are whatever the buffer holds);
// < 0 when the patch has no such pass
float sumfact_probe_mirror(igx_patch *pt, double *buf);
int sumfact_supported(const igx_patch *pt);
int sumfact_prepare(igx_patch *pt);
int sumfact_supports_kind(const igx_patch *pt, int kind);
int sumfact_assemble(igx_patch *pt, int kind, double *d_data);
} // namespace igx

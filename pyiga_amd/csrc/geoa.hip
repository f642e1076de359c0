// Fused geometry + stage A (3D, spline/NURBS geometry, symmetric fixed forms).
//
// Stage A sweeps axis 0:   K1[x][pair (i0,j0)][g1][g2] = sum_{g0} PI0[g0][t_x][a][b] * field_{f_x}(g0, g1, g2)
// and the fields are pointwise functions of the geometry Jacobian.  The separate kernels wrote the fields
// (6 x npts doubles at 3D stiffness) and read them back; here a block evaluates them on the fly:
//
//   block  = NS waves on the SAME 64 points (g1, g2) of the plane; wave w owns sweep slot x = w (one K1 array:
//            15 accumulators for the lower triangle of the (p+1)^2 pair window)
//   batch  = NS consecutive Gauss planes of axis 0: wave w evaluates the geometry of plane base + w for the 64 points
//            and puts the fields into LDS; after the barrier every wave sweeps its slot over the NS planes
//            (double-buffered LDS: one barrier per batch)
//   geometry: G(g0, g1, g2) = sum_a0 N_a0(g0) C_a0(g1, g2): the column coefficients C (value, d/d1, d/d2 of the
//            homogeneous control net contracted along axes 1, 2) of the block's 64 points live in LDS and change only
//            when the axis-0 span of the GEOMETRY changes; a point costs (p0g+1) * nc * 4 FMAs + the Jacobian algebra
//            (a batch that straddles a span boundary is evaluated span by span)
//   coefficients PI0 and the flush tables are wave-uniform: scalar loads
//
// HBM traffic: the K1 arrays, written once (17 GB at C4 instead of 12.6 + 12.6 + 17).
// Reference: pyiga/assemble_tools_cy.pyx (vform kernels evaluate the same fields per quadrature point),
// geometry evaluation pyiga/bspline.py:917-921, pyiga/geometry.py:17-25.
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include "geo_device.h"

namespace igx {

constexpr int GA_MAXS = 8;
#ifndef GA_ONEDIV
#define GA_ONEDIV 1           // stiffness fields from the unscaled quotient-rule matrix: one division per point
#endif
#ifndef GA_SPREAD
#define GA_SPREAD 0           // 1: the K1 stores of a span end go out one per plane of the next span (vector form; measured below)
#endif
#ifndef GA_READBOTH
#define GA_READBOTH 1         // both basis rows of a plane are read from LDS even when they are the same row (no register copies)
#endif
#ifndef GA_DPP
#define GA_DPP 1              // axis-0 sweep with the pair products V_a[tv] V_b[tu] of a plane in ONE register (lane m of every row of
#endif                        // 16 lanes = pair m) and v_fmac_f64 ... row_newbcast:m -- no multiply and no broadcast reads per pair
typedef const double __attribute__((address_space(4))) *cdp;      // uniform tables: scalar loads
typedef const int __attribute__((address_space(4))) *cip;

struct GeoAArgs {
    GeoView gv;
    const double *w0, *w1, *w2;
    int nurbs, kind;
    int G1, G2;
    long long NPL, stride;      // points of a plane; doubles between the K1 slices of consecutive pairs
    const double *tab;          // [G0][GA_REC] per-plane records (k_geoa_table): everything uniform a plane needs
    const int *steps;           // [nsteps][8]
    int s_lo, s_hi, q, chunk_len;
    int ntiles;                 // 64-point tiles of the plane
#ifdef IGX_ABLATE
    int dbg;                    // ablation (IGX_GEOA_DBG, -DIGX_ABLATE builds only): 1 no geometry, 2 no stores, 4 no sweep arithmetic
#endif
    int field[GA_MAXS], type[GA_MAXS];
    double *out[GA_MAXS];
    // non-symmetric forms (FORM = 1, the convection-diffusion form): optional second source of a slot (out = sum_g PI0[type] field
    // + PI0[xtype] xfield; -1: none), the scalar coefficient of the diffusion part -- sampled on the resident Gauss slab
    // (plane g0_lo first) or affine in the physical coordinates, c = cf[0] + cf[1] x + cf[2] y + cf[3] z
    int xfield[GA_MAXS], xtype[GA_MAXS];
    const double *coeff;
    int g0_lo, coef_affine;
    double cf[4];
    int soff_ok;                // every K1 slice offset (pairs x stride x 8 bytes) fits 32 bits: scalar-offset stores
    // general first-order forms (FORM = 2 non-symmetric, 3 symmetric; IGX_FORM given as a coefficient table, igx_patch_set_form_expr):
    // a(u, v) = int sum_{r,s} P_rs D_r v D_s u, D_0 = id, D_1.. = d/dx, d/dy, d/dz.  Entry 4 r + s of the PHYSICAL table is a
    // constant (pc), an array sampled on the resident Gauss slab (pa, plane g0_lo first) or absent (bit of pmask clear); the
    // kernel transforms it at every point, F = W T P T^T with T = diag(1, J^-1) (fields_form, geo_device.h), keeps the entries the
    // form needs -- fslot[4 a + b]: place of F_ab among the fields of a point in LDS, or -1 -- and array x sums up to four sources
    // (axis-0 type, field) per plane.
    double pc[16];
    const double *pa[16];
    int pmask, amask, fmask;    // bits 4 r + s: entry present | entry is an array;  bits 4 a + b: F_ab is kept (fslot >= 0)
    int fslot[16];
    int nsrc[GA_MAXS], sfield[GA_MAXS][4], stype[GA_MAXS][4];
    int nslots;                 // arrays the form has (<= 8): the sweep waves beyond them sweep nothing and store nothing
};
constexpr int GA_NFT = 13;      // fields of a point a table form may keep in LDS: ten for a symmetric table with every entry, thirteen for
                                // reaction + convection both ways + a symmetric diffusion block (sumfact.hip: form_table_plan)

typedef int int8v __attribute__((ext_vector_type(8)));

// switches of the timing experiments (work left out, wrong results by construction): compiled only into -DIGX_ABLATE builds
#ifdef IGX_ABLATE
#define GA_OFF(bit) (A.dbg & (bit))
#else
#define GA_OFF(bit) false
#endif

// Diagnostic build (-DIGX_GA_STAMP, never the shipped library): shader cycles per wave in the sections of the loop
#ifdef IGX_GA_STAMP
__device__ unsigned long long g_ga_stamp[2048 * 8 * 6];
#define GA_T(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define GA_T(i)
#endif

// acc += pv[lane M of this lane's row of 16] * bv: the DP-ALU form of DPP (gfx90a and later: row_newbcast is the one DPP control
// the FP64 instructions take).  All 64 lanes are active wherever this is used.
template <int M>
__device__ __forceinline__ void fmac_rowbc(double &acc, const double pv, const double bv)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(pv), "v"(bv), "n"(M));
}
// pair m = a (a + 1) / 2 + b (b <= a) of the lower triangle <-> accumulator acc[a][b]
template <int P, int A_ = 0, int B_ = 0>
__device__ __forceinline__ void sweep_lower_rowbc(double (&acc)[P][P], const double (&pv)[(P * (P + 1) / 2 + 15) / 16], const double bv)
{
    if constexpr (A_ < P) {
        constexpr int m = A_ * (A_ + 1) / 2 + B_;
        fmac_rowbc<(m & 15)>(acc[A_][B_], pv[m >> 4], bv);
        if constexpr (B_ < A_) sweep_lower_rowbc<P, A_, B_ + 1>(acc, pv, bv);
        else sweep_lower_rowbc<P, A_ + 1, 0>(acc, pv, bv);
    }
}

// the full window of a non-symmetric form: pair m = a P + b <-> acc[a][b], products in NPV registers of 16 pairs each
template <int P, int M = 0>
__device__ __forceinline__ void sweep_full_rowbc(double (&acc)[P][P], const double (&pv)[(P * P + 15) / 16], const double bv)
{
    if constexpr (M < P * P) {
        fmac_rowbc<(M & 15)>(acc[M / P][M % P], pv[M >> 4], bv);
        sweep_full_rowbc<P, M + 1>(acc, pv, bv);
    }
}

// Per-plane record of axis 0 (doubles): everything wave-uniform that a Gauss plane g needs, built once per patch so that
// the staging inside the sweep is one coalesced copy without dependent loads or branches:
//   [0, 12)   basis of the SPACE at the plane: [value | derivative][active function a < P <= 6]
//   [12, 18)  basis of the GEOMETRY: (N, N') of its <= 3 active functions;  [18] quadrature weight;  [19] first active
//             control index (as a double)
//   [20, 24)  eight ints: number of flush steps after this plane (0 unless it ends a span) | first step | K1 slots of the
//             first step's P pairs
//   matrix-core sweep (k_geoA<.., MF = true>): the lower pairs (i0, j0) that are live on a span keep a ROW of the 16 x 16
//   accumulator tile for their whole life -- class delta = i0 - j0 owns P - delta rows, pair (j0 + delta, j0) sits in row
//   base[delta] + j0 mod (P - delta) -- so nothing moves when a dof leaves:
//   [24, 32)  sixteen ints: row -> a | b << 4 | 256 (local indices of the row's pair on this plane's span), 0: row not live
//   [32, 40)  sixteen ints: row -> K1 slot of the row's pair if it is COMPLETE after this plane (last plane of its span),
//             -2: complete but not stored (neither its row nor its column is owned by the slab), -1: not complete
//   [40]      two ints: smallest stored K1 slot of the plane (the store offsets are relative to it: 32 bits) | unused
constexpr int GA_REC = 42;
constexpr int GA_ROWS = 16;

__global__ void k_geoa_table(const double *V0s, int P, const double *V0g, const int *fa0g, int P0G, const double *w0, int q,
                             const int *step_ptr, const int *steps, int G0, double *tab, const int *rows, int nonsym)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G0) return;
    double *r = tab + (size_t)g * GA_REC;
    for (int k = 0; k < GA_REC; ++k) r[k] = 0.0;
    for (int a = 0; a < P; ++a) { r[a] = V0s[((size_t)g * P + a) * 2]; r[6 + a] = V0s[((size_t)g * P + a) * 2 + 1]; }
    for (int e = 0; e < 2 * P0G; ++e) r[12 + e] = V0g[(size_t)g * P0G * 2 + e];
    r[18] = w0[g];
    r[19] = (double)fa0g[g];
    int *ri = (int *)(r + 20);
    const int s = g / q;
    if (g - s * q == q - 1) {
        const int st0 = step_ptr[s], st1 = step_ptr[s + 1];
        ri[0] = st1 - st0; ri[1] = st0;
        if (st1 > st0 && !nonsym)
            for (int a = 0; a < P; ++a) ri[2 + a] = steps[(size_t)st0 * 8 + a];
    }
    // row tables of the matrix-core sweep (host-built per span: [n][2 GA_ROWS + 2] ints)
    int *ra = (int *)(r + 24), *rs = (int *)(r + 32), *rb = (int *)(r + 40);
    for (int m = 0; m < GA_ROWS; ++m) { ra[m] = 0; rs[m] = -1; }
    rb[0] = 0; rb[1] = 0;
    if (nonsym) {
        // table of a non-symmetric form: the 16 ints of the first flush step after the plane (K1 slots of the pairs
        // (d + a, d) | (d, d + a)) take the place of the row table
        if (g - s * q == q - 1 && step_ptr[s + 1] > step_ptr[s])
            for (int m = 0; m < 16; ++m) ra[m] = steps[(size_t)step_ptr[s] * 16 + m];
    } else if (rows) {
        const int *sr = rows + (size_t)s * (2 * GA_ROWS + 2);
        for (int m = 0; m < GA_ROWS; ++m) ra[m] = sr[m];
        if (g - s * q == q - 1) {
            for (int m = 0; m < GA_ROWS; ++m) rs[m] = sr[GA_ROWS + m];
            rb[0] = sr[2 * GA_ROWS];
        }
    }
}

// Uniform tables (sweep coefficients, axis-0 geometry basis, flush records) are NOT read with scalar loads inside the loop:
// the scalar cache holds 16 KB, the tables are larger, and a scalar load that misses it queues in the L2 behind the K1
// store stream -- tools/ubench/k1_store.hip: 3.1 ms of arithmetic with such loads become 9.4 ms when the stores are on,
// while the same arithmetic without them overlaps the stores completely.  The block stages the table rows of a batch in
// LDS with vector loads issued a whole batch ahead.
// K1 store of one completed pair: (descriptor of the wave's array: scalar) + (slice offset of the pair: scalar, 32 bits) +
// (this lane's point: a constant) -- no vector instruction for the address (the flat form costs a 64-bit multiply-add per
// store).  Arrays of 2 GB and more (soff_ok = 0, decided on the host) keep 64-bit addresses.
struct K1Store {
    __amdgpu_buffer_rsrc_t rs;
    int voff, ok;
    __device__ __forceinline__ K1Store(double *base, const long long pt, const int tile, const int soff_ok)
    {
        const long long t0 = (long long)tile * 64;
        rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + t0), (short)0, 0x7ffffff0, 0x00020000);
        voff = (int)(pt - t0) * 8;
#ifdef GA_K1_FLAT
        ok = 0;                                              // (A/B build: 64-bit addresses everywhere)
#else
        ok = soff_ok;
#endif
    }
    __device__ __forceinline__ void store(double *out, const int slot, const long long stride, const double x) const
    {
        if (ok) {
            typedef int i2s __attribute__((ext_vector_type(2)));
            i2s v; v.x = __double2loint(x); v.y = __double2hiint(x);
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, (int)((unsigned)slot * (unsigned)(stride * 8)), 0);
        } else out[(long long)slot * stride] = x;
    }
};

// FORM = 1 (non-symmetric): the block carries GA_NGW extra GEOMETRY waves -- the sweep of the full (p+1)^2 window holds 72
// accumulator registers at p = 5 and the slots with two sources sweep twice as long as the others, so the eight sweep waves
// only sweep and four more waves evaluate the planes of the next batch beside them (two planes each).
#ifndef GA_NGW_N
#define GA_NGW_N 4
#endif
constexpr int GA_NGW = GA_NGW_N;
// geometry waves of a block: none (FORM = 0: every sweep wave evaluates the plane of its number), GA_NGW beside the eight sweep waves
// of the convection-diffusion form (two planes each), eight for the table forms (their evaluation is twice as long: one plane each,
// sixteen waves of 128 registers)
#ifndef GA_NGW_T
#define GA_NGW_T 8
#endif
#ifndef GA_T2_UNROLL_P
#define GA_T2_UNROLL_P 5      // non-symmetric table forms: the planes of a batch unrolled up to this P (functions per span): 10.3-10.7 -> 9.8-10.3 ms at C4 size
#endif
#ifndef GA_MASS8
#define GA_MASS8 1            // the mass form (one array) on the eight-wave block
#endif
#ifndef GA_F1_UNROLL_P
#define GA_F1_UNROLL_P 6      // convection-diffusion form: the planes of a batch unrolled up to this P (functions per span): 6.81 -> 6.37 ms at C5
#endif
constexpr int geoa_ngw(int FORM) { return FORM >= 2 ? GA_NGW_T : FORM == 1 ? GA_NGW : 0; }
constexpr int geoa_wpe(int NS, int FORM) { return FORM >= 1 ? (geoa_ngw(FORM) > 4 ? 4 : 3) : NS >= 8 ? 4 : 1; }
constexpr int geoa_threads(int NS, int FORM) { return (NS + geoa_ngw(FORM)) * 64; }
// D2 (round 6): the same kernel for 2D patches -- a "plane" is the line of Gauss points of axis 1, a block's 64 points lie on it,
// NS = 4 arrays (2D stiffness: (axis-0 type, field) = (0, B00) (2, B01) (1, B01) (3, B11)) or 1 (mass), NC = 2 | 3 components.
// Replaces the field kernel + k_stageA of the 2D chain (two launches and 50 MB of field traffic at BASELINE config 2).
template <int P, int NS, int P0G, int NC, bool MF, int FORM = 0, bool D2 = false>
__global__ void __launch_bounds__(geoa_threads(NS, FORM)) __attribute__((amdgpu_waves_per_eu(geoa_wpe(NS, FORM), FORM >= 1 ? geoa_wpe(NS, FORM) : 4)))
k_geoA(const GeoAArgs A)
{
    static_assert(!D2 || (FORM == 0 && !MF), "2D: the symmetric fixed forms, vector sweep");
    constexpr int NGW = geoa_ngw(FORM);                   // geometry waves (0: every sweep wave evaluates the plane of its own number)
    constexpr bool SYMW = FORM == 0 || FORM == 3;         // lower triangle of the pair window (symmetric forms)
    static_assert(!MF || (NS == 8 && P * (P + 1) / 2 <= GA_ROWS), "matrix-core sweep: eight slots, at most 16 live pairs");
    static_assert(!(MF && FORM), "the matrix-core sweep serves the symmetric forms");
    constexpr int NF = D2 ? 3 : FORM == 1 ? 9 : FORM == 2 ? GA_NFT : FORM == 3 ? 10 : 6;   // fields of a point in LDS (convection-diffusion: c B (6) + beta (3))
    constexpr int PV = (P + 1) & ~1;                      // basis row in registers, padded to an even length
    constexpr int NT = (NS + NGW) * 64;                   // threads
    constexpr int RECW = (MF || FORM >= 1) ? GA_REC : 24; // doubles of a plane record that the kernel uses (the row tables: matrix-core / non-symmetric sweeps only)
    constexpr int NRC = NS * RECW;                        // doubles of a batch of plane records
    constexpr int NTS = NGW ? NGW * 64 : NT;              // threads that stage them (the geometry waves where they exist)
    constexpr int KRC = (NRC + NTS - 1) / NTS;            // ... per thread
    // fields of two batches of planes, [buffer][plane][field][point]; the matrix-core sweep reads four planes with one
    // instruction (16 lanes each): its plane stride is padded so that they fall on different banks
    constexpr int FST = NF * 64 + (MF ? 16 : 0);
    __shared__ double fld_[2 * NS * FST];
#define FLD(buf_, j_, k_, ln_) fld_[((buf_) * NS + (j_)) * FST + (k_) * 64 + (ln_)]
    // matrix-core sweep: products V_b[tu] V_a[tv] of the 16 rows, per plane of a batch and type, rows permuted to
    // [row mod 4][row / 4] (the four rows of a lane are neighbours), plane stride padded like the fields'
    constexpr int ATS = 4 * GA_ROWS + 16;
    __shared__ __attribute__((aligned(16))) double atb[MF ? 2 * NS * ATS : 2];
    // pair-product sweep (GA_DPP): products of the lower pairs per plane of a batch and type, [buffer][plane][type][NPV * 16]
    constexpr int NPR = P * (P + 1) / 2, NPV = (NPR + 15) / 16;
    constexpr bool DPS = GA_DPP && !MF && FORM == 0 && NPV == 1;
    constexpr int NPVF = (P * P + 15) / 16;               // ... of the full window (non-symmetric forms)
    constexpr bool DPF = GA_DPP && FORM >= 1;
    constexpr int NPW = FORM == 3 ? NPV : NPVF;           // product registers of a (plane, type): lower triangle or full window
    __shared__ double prd_[DPS ? 2 * NS * 4 * NPV * 16 : DPF ? 2 * NS * 4 * NPW * 16 : 2];
    __shared__ double Cs[P0G * NC][3][64];                // column coefficients of the block's points, geometry span f0_blk
    __shared__ __attribute__((aligned(16))) double rec[3][NS][RECW];   // plane records of three batches: swept | evaluated | arriving
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // table forms: the 16 constants / array pointers / field slots are read from LDS where they are used (as kernel arguments in
    // scalar registers they are 80 of the 102 a wave has, hoisted out of every loop: spills)
    __shared__ double tabc[FORM >= 2 ? 16 : 1];
    __shared__ const double *tabp[FORM >= 2 ? 16 : 1];
    __shared__ int tabf[FORM >= 2 ? 16 : 1];
    if constexpr (FORM >= 2) {
        if (tid < 16) { tabc[tid] = A.pc[tid]; tabp[tid] = A.pa[tid]; tabf[tid] = A.fslot[tid]; }
    }
    // consecutive block ids go to different XCDs: give each XCD a contiguous range of point tiles
    const int per_xcd = gridDim.x >> 3;                   // the grid is padded to a multiple of 8 blocks
    int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
#ifdef IGX_ABLATE
    if (A.dbg & 8) tile = blockIdx.x;
#endif
    if (tile >= A.ntiles) tile = A.ntiles - 1;            // surplus blocks redo the last tile (same values again: harmless)
    long long pt = (long long)tile * 64 + lane;
    if (pt >= A.NPL) pt = A.NPL - 1;                      // lanes past the end redo the last point (and store it again: harmless)
    const int g1 = D2 ? (int)pt : (int)(pt / A.G2), g2 = D2 ? 0 : (int)(pt - (long long)g1 * A.G2);
    const int q = A.q;
    const int own_lo = A.s_lo + blockIdx.y * A.chunk_len;
    const int own_hi = min(own_lo + A.chunk_len, A.s_hi);
    const int s_begin = max(A.s_lo, own_lo - (P - 1));
    const int g_begin = s_begin * q, g_end = own_hi * q, g_last = g_end - 1;

    // ---- table staging: a straight copy of the batch's plane records, requested at the top of an iteration and written to
    // LDS at its end
    const GeoView &gv = A.gv;
    double rc_reg[KRC];
    auto stage_load = [&](const int gb) {
#pragma unroll
        for (int k = 0; k < KRC; ++k) {
            const int i = tid - (NT - NTS) + k * NTS, j = i / RECW;
            if (i >= 0 && i < NRC) rc_reg[k] = A.tab[(size_t)min(gb + j, g_last) * GA_REC + (i - j * RECW)];
        }
    };
    auto stage_store = [&](const int slot) {
#pragma unroll
        for (int k = 0; k < KRC; ++k) {
            const int i = tid - (NT - NTS) + k * NTS;
            if (i >= 0 && i < NRC) (&rec[slot][0][0])[i] = rc_reg[k];
        }
    };

    // ---- geometry
    const double GW1 = A.w1[g1], GW2 = D2 ? 1.0 : A.w2[g2];
    // the prologue loads are complete before the loop: a wait for them inside it would also drain the K1 stores
    asm volatile("" :: "v"(GW1), "v"(GW2));
    int f0_blk = -1;
    // the block's waves share the work: wave w contracts the control net along axes 1, 2 for its (a0, component) pairs
    auto columns = [&](const int f0) {
        if constexpr (D2) {                               // 2D: the control net contracted along axis 1 (value, d/d1)
            const double *V1 = gv.V[1] + (size_t)g1 * gv.P[1] * 2;
            const int f1 = gv.fa[1][g1];
            for (int e = w; e < P0G * NC; e += NS) {
                const int a0 = e / NC, c = e - a0 * NC;
                double sv = 0.0, s1 = 0.0;
                for (int a1 = 0; a1 < gv.P[1]; ++a1) {
                    const double cf = gv.ctrl[((size_t)(f0 + a0) * gv.N[1] + (f1 + a1)) * NC + c];
                    sv = fma(V1[a1 * 2], cf, sv);
                    s1 = fma(V1[a1 * 2 + 1], cf, s1);
                }
                Cs[e][0][lane] = sv; Cs[e][1][lane] = s1;
            }
            return;
        }
        const double *V1 = gv.V[1] + (size_t)g1 * gv.P[1] * 2, *V2 = gv.V[2] + (size_t)g2 * gv.P[2] * 2;
        const int f1 = gv.fa[1][g1], f2 = gv.fa[2][g2];
        for (int e = NGW ? w - NS : w; e >= 0 && e < P0G * NC; e += NGW ? NGW : NS) {
            const int a0 = e / NC, c = e - a0 * NC;
            double sv = 0.0, s1 = 0.0, s2 = 0.0;
            for (int a1 = 0; a1 < gv.P[1]; ++a1)
                for (int a2 = 0; a2 < gv.P[2]; ++a2) {
                    const double cf = gv.ctrl[(((size_t)(f0 + a0) * gv.N[1] + (f1 + a1)) * gv.N[2] + (f2 + a2)) * NC + c];
                    sv = fma(V1[a1 * 2] * V2[a2 * 2], cf, sv);
                    s1 = fma(V1[a1 * 2 + 1] * V2[a2 * 2], cf, s1);
                    s2 = fma(V1[a1 * 2] * V2[a2 * 2 + 1], cf, s2);
                }
            Cs[e][0][lane] = sv; Cs[e][1][lane] = s1; Cs[e][2][lane] = s2;
        }
    };
    // fields of plane j of the batch in buffer gbuf at this lane's point -> fld[buf][j]
    auto evaluate = [&](const int gbuf, const int buf, const int jp, const int gpl) {
        const double *gt = &rec[gbuf][jp][12];
        double V0[2 * P0G];
#pragma unroll
        for (int e = 0; e < 2 * P0G; ++e) V0[e] = gt[e];
        const double gw0 = gt[6];
        double val[MAX_COMP], jac[MAX_COMP][3];
#pragma unroll
        for (int c = 0; c < MAX_COMP; ++c) { val[c] = 0.0; jac[c][0] = jac[c][1] = jac[c][2] = 0.0; }
#pragma unroll
        for (int a0 = 0; a0 < P0G; ++a0) {
            const double n = V0[a0 * 2], d = V0[a0 * 2 + 1];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double c0 = Cs[a0 * NC + c][0][lane], c1 = Cs[a0 * NC + c][1][lane];
                val[c] = fma(n, c0, val[c]);
                jac[c][0] = fma(d, c0, jac[c][0]);
                jac[c][1] = fma(n, c1, jac[c][1]);
                if constexpr (!D2) jac[c][2] = fma(n, Cs[a0 * NC + c][2][lane], jac[c][2]);
            }
        }
        double GW = gw0 * GW1;
        GW = GW * GW2;
        if constexpr (D2) {
            // 2D: Jacobian (quotient rule for a NURBS map), W or the upper triangle of W J^-1 J^-T (pyiga/assemblers.pyx:86-110, 234-275)
            double Jm2[MAX_COMP][3], ev2[MAX_COMP];
            finish_jacobian<2>(val, jac, NC == 3, 2, NC, Jm2, ev2);
            double tt2[9] = {Jm2[0][0], Jm2[0][1], Jm2[1][0], Jm2[1][1], 0.0, 0.0, 0.0, 0.0, 0.0};
            double f2[6];
            fields_values<2>(tt2, GW, A.kind, f2);
            if (A.kind == IGX_MASS) FLD(buf, jp, 0, lane) = f2[0];
            else {
#pragma unroll
                for (int k = 0; k < 3; ++k) FLD(buf, jp, k, lane) = f2[k];
            }
            return;
        }
        if constexpr (FORM == 1) {
            // convection-diffusion form: c W JacInv JacInv^T (upper triangle) and beta_a = W sum_r JacInv[a][r] b_r, b = (y, -x, 1)
            // (fields_convdiff, geo_device.h), with ONE division: M = quotient-rule numerator (J = M / w^2; M = J, w = 1 for a
            // polynomial map), D = w^5 |det M|:   c W J^-1 J^-T = c GW w^3 / D adj(M) adj(M)^T,
            // beta = GW det(M) / D adj(M) (Y, -X, w) with the homogeneous coordinates (X, Y, Z, w) of the point
            double t[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    t[r * 3 + c] = NC == 4 ? jac[r][2 - c] * val[3] - val[r] * jac[3][2 - c] : jac[r][2 - c];
            double a[9];
            a[0] = t[4] * t[8] - t[5] * t[7];
            a[1] = -(t[1] * t[8] - t[2] * t[7]);
            a[2] = t[1] * t[5] - t[2] * t[4];
            a[3] = -(t[3] * t[8] - t[5] * t[6]);
            a[4] = t[0] * t[8] - t[2] * t[6];
            a[5] = -(t[0] * t[5] - t[2] * t[3]);
            a[6] = t[3] * t[7] - t[4] * t[6];
            a[7] = -(t[0] * t[7] - t[1] * t[6]);
            a[8] = t[0] * t[4] - t[1] * t[3];
            const double det = (t[0] * a[0] + t[1] * a[3]) + t[2] * a[6];
            const double wh = NC == 4 ? val[3] : 1.0, wh2 = wh * wh;
            const double inv = 1.0 / (NC == 4 ? (wh2 * wh2) * (wh * fabs(det)) : fabs(det));
            // c w^3: affine coefficient from the homogeneous coordinates, or the sampled value of the point
            double cw3;
            if (A.coef_affine) cw3 = (((A.cf[0] * wh + A.cf[1] * val[0]) + A.cf[2] * val[1]) + A.cf[3] * val[2]) * (NC == 4 ? wh2 : 1.0);
            else cw3 = A.coeff[(long long)(gpl - A.g0_lo) * A.NPL + pt] * (NC == 4 ? wh2 * wh : 1.0);
            const double g_inv = GW * inv;
            const double sc = cw3 * g_inv, bs = det * g_inv;
            FLD(buf, jp, 0, lane) = sc * ((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
            FLD(buf, jp, 1, lane) = sc * ((a[0] * a[3] + a[1] * a[4]) + a[2] * a[5]);
            FLD(buf, jp, 2, lane) = sc * ((a[0] * a[6] + a[1] * a[7]) + a[2] * a[8]);
            FLD(buf, jp, 3, lane) = sc * ((a[3] * a[3] + a[4] * a[4]) + a[5] * a[5]);
            FLD(buf, jp, 4, lane) = sc * ((a[3] * a[6] + a[4] * a[7]) + a[5] * a[8]);
            FLD(buf, jp, 5, lane) = sc * ((a[6] * a[6] + a[7] * a[7]) + a[8] * a[8]);
            const double X = val[0], Y = val[1];
            FLD(buf, jp, 6, lane) = bs * ((a[0] * Y - a[1] * X) + a[2] * wh);
            FLD(buf, jp, 7, lane) = bs * ((a[3] * Y - a[4] * X) + a[5] * wh);
            FLD(buf, jp, 8, lane) = bs * ((a[6] * Y - a[7] * X) + a[8] * wh);
            return;
        }
        if constexpr (FORM >= 2) {
            // general first-order form: physical Jacobian and point, T = diag(1, J^-1), F = W T P T^T (fields_form, geo_device.h)
            double t[9];                                   // J[r][c] = d G_r / d xi_c, c = 0 the LAST grid axis; NURBS: quotient rule, one reciprocal
            if constexpr (NC == 4) {
                const double iW = 1.0 / val[3], iW2 = iW * iW;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) t[r * 3 + c] = (jac[r][2 - c] * val[3] - val[r] * jac[3][2 - c]) * iW2;
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) t[r * 3 + c] = jac[r][2 - c];
            }
            const double t3 = t[4] * t[8] - t[5] * t[7], t4 = t[3] * t[8] - t[5] * t[6], t5 = t[3] * t[7] - t[4] * t[6];
            const double det = (t[0] * t3 - t[1] * t4) + t[2] * t5;
            const double inv = 1.0 / det;
            double T[4][4];
            T[1][1] = inv * t3; T[1][2] = inv * -(t[1] * t[8] - t[2] * t[7]); T[1][3] = inv * (t[1] * t[5] - t[2] * t[4]);
            T[2][1] = inv * -t4; T[2][2] = inv * (t[0] * t[8] - t[2] * t[6]); T[2][3] = inv * -(t[0] * t[5] - t[2] * t[3]);
            T[3][1] = inv * t5; T[3][2] = inv * -(t[0] * t[7] - t[1] * t[6]); T[3][3] = inv * (t[0] * t[4] - t[1] * t[3]);
            const double W = GW * fabs(det);
            // the physical table at this point: wave-uniform choices (constant | sampled array | absent)
            const long long ip = (long long)(gpl - A.g0_lo) * A.NPL + pt;
            // (the constants without a test -- an absent entry is the constant 0 -- so that the sixteen LDS reads are in flight
            // together; the sampled entries, where the form has any, requested one after the other before the first is used)
            double Pm[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) Pm[r][c] = tabc[4 * r + c];
            if (A.amask) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if ((A.amask >> (4 * r + c)) & 1) Pm[r][c] = tabp[4 * r + c][ip];
            }
            // row by row of T P (rows of T act on the test index): few values live at a time
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                double TPa[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    TPa[c] = a == 0 ? Pm[0][c] : fma(T[a][3], Pm[3][c], fma(T[a][2], Pm[2][c], T[a][1] * Pm[1][c]));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (!((A.fmask >> (4 * a + b)) & 1)) continue;
                    const double v = b == 0 ? TPa[0] : fma(TPa[3], T[b][3], fma(TPa[2], T[b][2], TPa[1] * T[b][1]));
                    FLD(buf, jp, tabf[4 * a + b], lane) = W * v;
                }
            }
            return;
        }
        if (GA_ONEDIV && NS == 8 && A.kind != IGX_MASS) {
            // stiffness fields with ONE division (an f64 division is 12 vector instructions).  With the unscaled quotient-rule
            // matrix M = V'W - V W' (J = M / W^2; M = J for a polynomial geometry):
            //   GW |det J| J^-1 J^-T = GW adj(J) adj(J)^T / |det J| = GW / (W^2 |det M|) * adj(M) adj(M)^T
            double t[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    t[r * 3 + c] = NC == 4 ? jac[r][2 - c] * val[3] - val[r] * jac[3][2 - c] : jac[r][2 - c];
            double a[9];
            a[0] = t[4] * t[8] - t[5] * t[7];
            a[1] = -(t[1] * t[8] - t[2] * t[7]);
            a[2] = t[1] * t[5] - t[2] * t[4];
            a[3] = -(t[3] * t[8] - t[5] * t[6]);
            a[4] = t[0] * t[8] - t[2] * t[6];
            a[5] = -(t[0] * t[5] - t[2] * t[3]);
            a[6] = t[3] * t[7] - t[4] * t[6];
            a[7] = -(t[0] * t[7] - t[1] * t[6]);
            a[8] = t[0] * t[4] - t[1] * t[3];
            const double det = (t[0] * a[0] + t[1] * a[3]) + t[2] * a[6];
            const double sc = GW / (NC == 4 ? (val[3] * val[3]) * fabs(det) : fabs(det));
            FLD(buf, jp, 0, lane) = sc * ((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
            FLD(buf, jp, 1, lane) = sc * ((a[0] * a[3] + a[1] * a[4]) + a[2] * a[5]);
            FLD(buf, jp, 2, lane) = sc * ((a[0] * a[6] + a[1] * a[7]) + a[2] * a[8]);
            FLD(buf, jp, 3, lane) = sc * ((a[3] * a[3] + a[4] * a[4]) + a[5] * a[5]);
            FLD(buf, jp, 4, lane) = sc * ((a[3] * a[6] + a[4] * a[7]) + a[5] * a[8]);
            FLD(buf, jp, 5, lane) = sc * ((a[6] * a[6] + a[7] * a[7]) + a[8] * a[8]);
            return;
        }
        double Jm[MAX_COMP][3], ev[MAX_COMP];
        finish_jacobian<3>(val, jac, NC == 4, 3, NC, Jm, ev);
        double tt[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) tt[r * 3 + c] = Jm[r][c];
        double f[6];
        fields_values<3>(tt, GW, A.kind, f);
        const int nf = A.kind == IGX_MASS ? 1 : 6;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < nf) FLD(buf, jp, k, lane) = f[k];
    };
    // all waves: evaluate the batch that starts at plane gn (table rows in gts[gbuf]).  The column coefficients belong to
    // one span of the geometry's axis 0; a batch that straddles span boundaries is evaluated span by span (uniform control:
    // every wave of the block takes the same way)
    auto next_batch = [&](const int gn, const int gbuf, const int buf) {
        if (gn >= g_end) return;
        const int jl = min(NS - 1, g_last - gn);
        const int last = __builtin_amdgcn_readfirstlane((int)rec[gbuf][jl][19]);
        int cur = __builtin_amdgcn_readfirstlane((int)rec[gbuf][0][19]);
        for (;;) {
            if (cur != f0_blk) {
                columns(cur);
                f0_blk = cur;
                __syncthreads();
            }
            if (NGW == 0) {
                if (w <= jl && __builtin_amdgcn_readfirstlane((int)rec[gbuf][w][19]) == cur && !GA_OFF(1)) evaluate(gbuf, buf, w, gn + w);
            } else if (FORM >= 2) {
                // table forms: the evaluation is twice the stiffness form's -- the geometry waves take the planes 0 .. NGW-1 of a
                // batch, the sweep waves NGW .. NS-1 (the arrays with the fewest sources) the plane of their own number
                const int jp = w >= NS ? w - NS : (w >= NGW ? w : -1);
                if (jp >= 0 && jp <= jl && __builtin_amdgcn_readfirstlane((int)rec[gbuf][jp][19]) == cur && !GA_OFF(1)) evaluate(gbuf, buf, jp, gn + jp);
            } else if (w >= NS) {
                for (int jp = w - NS; jp <= jl; jp += NGW)
                    if (__builtin_amdgcn_readfirstlane((int)rec[gbuf][jp][19]) == cur && !GA_OFF(1)) evaluate(gbuf, buf, jp, gn + jp);
            }
            if (cur == last) break;
            int nxt = last;
            for (int j = jl; j >= 0; --j) {
                const int v = __builtin_amdgcn_readfirstlane((int)rec[gbuf][j][19]);
                if (v > cur) nxt = v;
            }
            cur = nxt;
            __syncthreads();                              // the columns are rewritten next
        }
    };

    // the same walk over the geometry spans of a batch for waves that evaluate nothing (FORM = 1: the sweep waves): the
    // barriers of next_batch, nothing else
    auto next_batch_barriers = [&](const int gn, const int gbuf) {
        if (gn >= g_end) return;
        const int jl = min(NS - 1, g_last - gn);
        const int last = __builtin_amdgcn_readfirstlane((int)rec[gbuf][jl][19]);
        int cur = __builtin_amdgcn_readfirstlane((int)rec[gbuf][0][19]);
        for (;;) {
            if (cur != f0_blk) { f0_blk = cur; __syncthreads(); }
            if (cur == last) break;
            int nxt = last;
            for (int j = jl; j >= 0; --j) {
                const int v = __builtin_amdgcn_readfirstlane((int)rec[gbuf][j][19]);
                if (v > cur) nxt = v;
            }
            cur = nxt;
            __syncthreads();
        }
    };

    // ---- matrix-core sweep (p = 3, 4 stiffness): the wave's K1 array as a 16 x 64 tile of v_mfma_f64_16x16x4_f64 accumulators --
    // rows = live lower pairs (fixed row per pair, see GA_REC), columns = the 64 points in four 16-point tiles; lane
    // (g = lane / 16, n = lane % 16) holds rows g, g + 4, g + 8, g + 12 of point 16 nt + n in acc[nt][0..3] (measured layout:
    // tools/ubench/mfma_layout.hip).  Four consecutive planes of one span are ONE instruction per tile, A = the products of
    // the 16 rows at the four planes, B = the field at 4 planes x 16 points; planes that do not fill a group of four inside
    // their span and batch take 16 vector multiply-adds in the same layout.  The instruction adds its four planes in
    // ascending order with one rounding each (checked bit for bit against an fma chain), so a pair's sum does not depend on
    // how the batches cut its spans: row slabs and sweep chunks still reproduce the whole patch bit for bit.
    if constexpr (MF) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        typedef double d4 __attribute__((ext_vector_type(4)));
        typedef int i2v __attribute__((ext_vector_type(2)));
        const int t = A.type[w], fi = A.field[w];
        const int g = lane >> 4, n = lane & 15;
        d4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = d4{0.0, 0.0, 0.0, 0.0};
        const long long tile0 = (long long)tile * 64;
        const bool full = tile0 + 64 <= A.NPL;                 // (uniform) no point of the tile lies past the plane
        const unsigned stride8 = (unsigned)(A.stride * 8);      // bytes between K1 slices (spread of a span's slots x stride8 < 2^31: host)
        constexpr unsigned OOB = 0x7ffffff8u;
        const double *const outw = A.out[w] + tile0;           // (no scalar load of a kernel argument inside the loop: see above)
        asm volatile("" :: "s"(outw));
        // products of the batch that starts at plane gn (records in slot rsl) -> atb[bufn]: wave w = plane w, lane = (type, row)
        auto products = [&](const int gn, const int rsl, const int bufn) {
            if (gn + w > g_last) return;
            const int ty = lane >> 4, m = lane & 15;
            const int ab = ((const int *)&rec[rsl][w][24])[m];
            const double va = rec[rsl][w][6 * (ty >> 1) + (ab & 15)], vb = rec[rsl][w][6 * (ty & 1) + ((ab >> 4) & 15)];
            atb[(bufn * NS + w) * ATS + ty * GA_ROWS + (m & 3) * 4 + (m >> 2)] = (ab & 256) ? vb * va : 0.0;
        };
        stage_load(g_begin); stage_store(0);
        stage_load(g_begin + NS); stage_store(1);
        __syncthreads();
        next_batch(g_begin, 0, 0);
        products(g_begin, 0, 0);
        __syncthreads();
        int it = 0, l = 0, sp = s_begin, rs = 0;
        for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
            const int buf = it & 1;
            const int rn = rs == 2 ? 0 : rs + 1, ra = rn == 2 ? 0 : rn + 1;
            stage_load(gb + 2 * NS);
            const int jend = min(NS, g_end - gb);
            const double *ab = &atb[buf * NS * ATS + t * GA_ROWS];
            // four planes of one span: one matrix instruction per tile
            auto unit4 = [&](const int j) {
                const double am = ab[(j + g) * ATS + (n & 3) * 4 + (n >> 2)];
                double bm[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bm[nt] = FLD(buf, j + g, fi, 16 * nt + n);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm[nt], acc[nt], 0, 0, 0);
            };
            // one plane: 16 vector multiply-adds in the accumulator layout
            auto unit1 = [&](const int j) {
                const d2 *cr = (const d2 *)&ab[j * ATS + 4 * g];
                const d2 c01 = cr[0], c23 = cr[1];
                double b1[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b1[nt] = FLD(buf, j, fi, 16 * nt + n);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[nt][0] = fma(c01.x, b1[nt], acc[nt][0]); acc[nt][1] = fma(c01.y, b1[nt], acc[nt][1]);
                    acc[nt][2] = fma(c23.x, b1[nt], acc[nt][2]); acc[nt][3] = fma(c23.y, b1[nt], acc[nt][3]);
                }
            };
            // end of span sp (its last plane is plane jl of the batch): the rows of the leaving dofs are complete -- stored
            // (quarter-wave runs of 128 B per row and tile; lanes without a row to store carry an out-of-range offset) and
            // cleared under an exec mask
            auto flush = [&](const int jl) {
                const bool write = sp >= own_lo && !GA_OFF(2);
                const int *rsl = (const int *)&rec[rs][jl][32];
                const int sbase = __builtin_amdgcn_readfirstlane(((const int *)&rec[rs][jl][40])[0]);
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(outw + (long long)sbase * A.stride), (short)0, 0x7ffffff0, 0x00020000);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int slot = rsl[g + 4 * v];
                    const bool done = slot != -1;
                    const unsigned off = (slot >= 0 && write) ? (unsigned)(slot - sbase) * stride8 + n * 8u : OOB;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const unsigned o = full || tile0 + 16 * nt + n < A.NPL ? off + nt * 128u : OOB;
                        i2v x; x.x = __double2loint(acc[nt][v]); x.y = __double2hiint(acc[nt][v]);
                        __builtin_amdgcn_raw_buffer_store_b64(x, rsrc, (int)o, 0, 0);
                    }
                    const unsigned long long dm = __builtin_amdgcn_ballot_w64(done);
                    unsigned long long sv;
                    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tv_mov_b64 %[a], 0\n\tv_mov_b64 %[b], 0\n\tv_mov_b64 %[c], 0\n\tv_mov_b64 %[d], 0\n\ts_mov_b64 exec, %[sv]"
                                 : [a] "+v"(acc[0][v]), [b] "+v"(acc[1][v]), [c] "+v"(acc[2][v]), [d] "+v"(acc[3][v]), [sv] "=&s"(sv) : [m] "s"(dm) : "scc");
                }
                ++sp;
            };
            // The batch cuts the spans anywhere: (A) the planes that finish the span begun in the last batch, one by one;
            // (B) whole spans -- one group of four, the rest one by one; (C) the first planes of the span the batch ends in.
            // (Each phase updates the accumulators in one way only: no register copies where the paths meet.)
            int j = 0;
            if (l != 0) {
                while (j < jend && l < P) { unit1(j); ++j; ++l; }
                if (l == P) { flush(j - 1); l = 0; }
            }
            while (j + P <= jend) {
                unit4(j);
#pragma unroll
                for (int e = 4; e < P; ++e) unit1(j + e);
                j += P;
                flush(j - 1);
            }
            if (j < jend) {
                if (j + 4 <= jend) { unit4(j); j += 4; l = 4; }
                while (j < jend) { unit1(j); ++j; ++l; }
            }
            next_batch(gb + NS, rn, buf ^ 1);
            products(gb + NS, rn, buf ^ 1);
            stage_store(ra);
            rs = rn;
            __syncthreads();
        }
        return;
    }
    // ---- general first-order forms (coefficient table; FORM = 2: full pair window, FORM = 3: symmetric table, lower triangle): the
    // block structure of the convection-diffusion form below -- geometry waves evaluate the next batch beside eight sweep waves --
    // with up to four sources per array and the symmetric flush where the table is symmetric
    if constexpr (FORM >= 2) {
        static_assert(DPF, "the table forms sweep with row-broadcast multiply-adds");
        constexpr int PW = NPW * 16;
        // (one (type, pair) per geometry thread, worked out once; the planes of the batch in turn)
        static_assert(4 * PW <= NGW * 64, "one product per geometry thread and plane");
        const int pr_r = tid - NS * 64, pr_ty = pr_r / PW, pr_m = pr_r - pr_ty * PW;
        int pr_a, pr_b;
        bool pr_ok;
        if (SYMW) {                                       // m = a (a + 1) / 2 + b, b <= a
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= pr_m) ++a;
            pr_ok = a < P;
            pr_b = min(pr_m - a * (a + 1) / 2, P - 1); pr_a = min(a, P - 1);
        } else { pr_a = min(pr_m / P, P - 1); pr_b = pr_m - (pr_m / P) * P; pr_ok = pr_m < P * P; }
        pr_a += 6 * (pr_ty >> 1); pr_b += 6 * (pr_ty & 1);
        const bool pr_on = pr_r >= 0 && pr_r < 4 * PW;
        auto products_tab = [&](const int rsl, const int bufn) {
            if (pr_on) {
#pragma unroll
                for (int jp = 0; jp < NS; ++jp) {
                    const double x = rec[rsl][jp][pr_a] * rec[rsl][jp][pr_b];
                    prd_[(bufn * NS + jp) * (4 * PW) + pr_r] = pr_ok ? x : 0.0;
                }
            }
        };
        if (w >= NS) {
            // (the plane records of batch it + 2 are REQUESTED an iteration before they are written to LDS: the round trip of the
            // request lies beside a whole iteration, not inside one -- with one block per CU nothing else hides it)
            stage_load(g_begin); stage_store(0);
            stage_load(g_begin + NS); stage_store(1);
            stage_load(g_begin + 2 * NS);
            __syncthreads();
            next_batch(g_begin, 0, 0);
            products_tab(0, 0);
            __syncthreads();
            int it = 0, rs = 0;
            for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
                const int rn = rs == 2 ? 0 : rs + 1, ra = rn == 2 ? 0 : rn + 1;
                stage_store(ra);
                stage_load(gb + 3 * NS);
                next_batch(gb + NS, rn, (it & 1) ^ 1);
                products_tab(rn, (it & 1) ^ 1);
                rs = rn;
                __syncthreads();
            }
            return;
        }
        const bool live = w < A.nslots;
        const int nsrc = live ? A.nsrc[w] : 1;
        // (types and fields of the sources in scalar registers, read ONCE: a scalar load inside the plane loop waits on the
        // counter the LDS reads use and drains them; every use below names its source by a constant)
        int sty[4], sfl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { sty[k] = __builtin_amdgcn_readfirstlane(A.stype[w][k]); sfl[k] = __builtin_amdgcn_readfirstlane(A.sfield[w][k]); }
        double *const out = A.out[w] + pt;
        const K1Store k1s(A.out[w], pt, tile, A.soff_ok);
        double acc[P][P];
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
        __syncthreads();
        next_batch(g_begin, 0, 0);                        // (the planes NGW .. NS-1 of a batch are evaluated here: see next_batch)
        __syncthreads();
        int it = 0, l = 0, sp = s_begin, rs = 0;
        for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
            const int buf = it & 1;
            const int rn = rs == 2 ? 0 : rs + 1;
            auto operands = [&](double (&pv)[NPW], double &bv, const int j, const int ty, const int ff) {
                const double *pr_ = &prd_[((buf * NS + j) * 4 + ty) * PW + (lane & 15)];
#pragma unroll
                for (int e = 0; e < NPW; ++e) pv[e] = pr_[16 * e];
                bv = FLD(buf, j, ff, lane);
            };
            double pv0[NPW], bv0;
            operands(pv0, bv0, 0, sty[0], sfl[0]);
            // (the symmetric window is 15 registers at p = 4: the planes of a batch unrolled, so that the waits for the LDS reads
            // are counted and the operands of the next plane stay in flight under this plane's arithmetic)
            auto plane = [&](const int j) __attribute__((always_inline)) -> bool {
                if (gb + j >= g_end) return false;
                // the first source of the next plane is requested before this plane's arithmetic; further sources as they come
                double pn[NPW], bn;
                operands(pn, bn, j + 1 < NS ? j + 1 : j, sty[0], sfl[0]);
                // (the further sources of this plane: all requested before the first multiply-add)
                double pv1[NPW], pv2[NPW], pv3[NPW], bv1 = 0.0, bv2 = 0.0, bv3 = 0.0;
                if (nsrc > 1) operands(pv1, bv1, j, sty[1], sfl[1]);
                if (nsrc > 2) operands(pv2, bv2, j, sty[2], sfl[2]);
                if (nsrc > 3) operands(pv3, bv3, j, sty[3], sfl[3]);
                asm volatile("" ::: "memory");
                if constexpr (SYMW) sweep_lower_rowbc<P>(acc, pv0, bv0);
                else sweep_full_rowbc<P>(acc, pv0, bv0);
                if (nsrc > 1) { if constexpr (SYMW) sweep_lower_rowbc<P>(acc, pv1, bv1); else sweep_full_rowbc<P>(acc, pv1, bv1); }
                if (nsrc > 2) { if constexpr (SYMW) sweep_lower_rowbc<P>(acc, pv2, bv2); else sweep_full_rowbc<P>(acc, pv2, bv2); }
                if (nsrc > 3) { if constexpr (SYMW) sweep_lower_rowbc<P>(acc, pv3, bv3); else sweep_full_rowbc<P>(acc, pv3, bv3); }
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b < (SYMW ? a + 1 : P); ++b) asm volatile("" : "+v"(acc[a][b]));
                asm volatile("" ::: "memory");
#pragma unroll
                for (int e = 0; e < NPW; ++e) pv0[e] = pn[e];
                bv0 = bn;
                if (++l < q) return true;
                const bool write = sp >= own_lo && live && !GA_OFF(2);
                const int *fr = (const int *)&rec[rs][j][20];
                const int nst = __builtin_amdgcn_readfirstlane(fr[0]), st0 = __builtin_amdgcn_readfirstlane(fr[1]);
                for (int st = st0; st < st0 + nst; ++st) {
                    if constexpr (SYMW) {
                        int pr[P];
                        if (st == st0) {
#pragma unroll
                            for (int a = 0; a < P; ++a) pr[a] = __builtin_amdgcn_readfirstlane(fr[2 + a]);
                        } else {
                            const int8v rec8 = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 8);
#pragma unroll
                            for (int a = 0; a < P; ++a) pr[a] = rec8[a];
                        }
#pragma unroll
                        for (int a = 0; a < P; ++a)
                            if (pr[a] >= 0 && write) k1s.store(out, pr[a], A.stride, acc[a][0]);
#pragma unroll
                        for (int a = 0; a < P - 1; ++a)
#pragma unroll
                            for (int b = 0; b <= a; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
                        for (int b = 0; b < P; ++b) acc[P - 1][b] = 0.0;
                    } else {
                        int pr[16];
                        if (st == st0) {
                            const int *f16 = (const int *)&rec[rs][j][24];
#pragma unroll
                            for (int a = 0; a < P; ++a) pr[a] = __builtin_amdgcn_readfirstlane(f16[a]);
#pragma unroll
                            for (int a = 1; a < P; ++a) pr[8 + a] = __builtin_amdgcn_readfirstlane(f16[8 + a]);
                        } else {
                            const int8v r0 = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 16);
                            const int8v r1 = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 16 + 8);
#pragma unroll
                            for (int a = 0; a < P; ++a) { pr[a] = r0[a]; pr[8 + a] = r1[a]; }
                        }
#pragma unroll
                        for (int a = 0; a < P; ++a)
                            if (pr[a] >= 0 && write) k1s.store(out, pr[a], A.stride, acc[a][0]);
#pragma unroll
                        for (int a = 1; a < P; ++a)
                            if (pr[8 + a] >= 0 && write) k1s.store(out, pr[8 + a], A.stride, acc[0][a]);
#pragma unroll
                        for (int a = 0; a < P - 1; ++a)
#pragma unroll
                            for (int b = 0; b < P - 1; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
                        for (int b = 0; b < P; ++b) { acc[P - 1][b] = 0.0; acc[b][P - 1] = 0.0; }
                    }
                }
                l = 0; ++sp;
                return true;
            };
            if constexpr ((SYMW && P <= 5) || (!SYMW && P <= GA_T2_UNROLL_P)) {
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    if (!plane(j)) break;
            } else {
#pragma unroll 1
                for (int j = 0; j < NS; ++j)
                    if (!plane(j)) break;
            }
            next_batch(gb + NS, rn, buf ^ 1);
            rs = rn;
            __syncthreads();
        }
        return;
    }
    // ---- non-symmetric forms: the full (p+1)^2 pair window, up to two sources per slot, both pair families flushed
    if constexpr (FORM == 1) {
        // The geometry waves stage the plane records and evaluate the next batch; the sweep waves sweep: two loops that meet
        // only at the barriers, so that no value of one role is live in the other (the pair window of a sweep wave alone is 72
        // registers at p = 5; anything spilled around it is reloaded behind a vmcnt(0), i.e. behind the K1 stores).
        // pair products of a batch (GA_DPP): V_a[tv] V_b[tu] per (plane, type, pair m = a P + b), by the geometry waves
        // (one (type, pair) per thread, its place in the record and in the product image worked out ONCE; the planes of the batch in
        // turn -- the index arithmetic of six products per thread and batch was as long as the evaluation of a plane)
        constexpr int PWF = NPVF * 16;
        static_assert(!DPF || 4 * PWF <= NGW * 64, "one product per geometry thread and plane");
        const int pr_r = tid - NS * 64, pr_ty = pr_r / PWF, pr_m = pr_r - pr_ty * PWF;
        const int pr_a = 6 * (pr_ty >> 1) + min(pr_m / P, P - 1), pr_b = 6 * (pr_ty & 1) + (pr_m - (pr_m / P) * P);
        const bool pr_on = pr_r >= 0 && pr_r < 4 * PWF, pr_ok = pr_m < P * P;
        auto products_full = [&](const int rsl, const int bufn) {
            if constexpr (DPF) {
                if (pr_on) {
#pragma unroll
                    for (int jp = 0; jp < NS; ++jp) {
                        const double x = rec[rsl][jp][pr_a] * rec[rsl][jp][pr_b];
                        prd_[(bufn * NS + jp) * (4 * PWF) + pr_r] = pr_ok ? x : 0.0;
                    }
                }
            }
        };
        if (w >= NS) {
            // (records of batch it + 2 requested an iteration before they are written to LDS: see the table forms above)
            stage_load(g_begin); stage_store(0);
            stage_load(g_begin + NS); stage_store(1);
            stage_load(g_begin + 2 * NS);
            __syncthreads();
            next_batch(g_begin, 0, 0);
            products_full(0, 0);
            __syncthreads();
            int it = 0, rs = 0;
            for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
                const int rn = rs == 2 ? 0 : rs + 1, ra = rn == 2 ? 0 : rn + 1;
                stage_store(ra);
                stage_load(gb + 3 * NS);
                next_batch(gb + NS, rn, (it & 1) ^ 1);
                products_full(rn, (it & 1) ^ 1);
                rs = rn;
                __syncthreads();
            }
            return;
        }
        typedef double d2 __attribute__((ext_vector_type(2)));
        const int t = A.type[w], fi = A.field[w], xt = A.xtype[w], xf = A.xfield[w];
#ifdef GA_PRIO2
        if (xf >= 0) __builtin_amdgcn_s_setprio(GA_PRIO2);  // (experiment) the waves with two sources close every batch
#endif
        double *const out = A.out[w] + pt;
        const K1Store k1s(A.out[w], pt, tile, A.soff_ok);
        double acc[P][P];
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
        auto basis_row = [&](double (&v)[PV], const int buf, const int j, const int d) {
            const d2 *row = (const d2 *)&rec[buf][j][6 * d];
#pragma unroll
            for (int k = 0; k < P / 2; ++k) { const d2 x = row[k]; v[2 * k] = x.x; v[2 * k + 1] = x.y; }
            if (P & 1) v[P - 1] = rec[buf][j][6 * d + P - 1];
        };
        // acc[a][b] += V_b[tu] (V_a[tv] f): a = test function (row), b = trial function
        auto source = [&](const int rsl, const int buf, const int j, const int ty, const int ff) {
            const double bv = FLD(buf, j, ff, lane);
            double c[P];
            {
                double va[PV];
                basis_row(va, rsl, j, ty >> 1);
#pragma unroll
                for (int a = 0; a < P; ++a) c[a] = va[a] * bv;
            }
            // (the row of the trial functions is requested after the products: one row of basis values in registers at a time)
#pragma unroll
            for (int a = 0; a < P; ++a) asm volatile("" : "+v"(c[a]));
            asm volatile("" ::: "memory");
            double vb[PV];
            basis_row(vb, rsl, j, ty & 1);
#pragma unroll
            for (int a = 0; a < P; ++a)
#pragma unroll
                for (int b = 0; b < P; ++b) acc[a][b] = fma(vb[b], c[a], acc[a][b]);
#pragma unroll
            for (int a = 0; a < P; ++a)
#pragma unroll
                for (int b = 0; b < P; ++b) asm volatile("" : "+v"(acc[a][b]));
            asm volatile("" ::: "memory");
        };
        __syncthreads();
        next_batch_barriers(g_begin, 0);
        __syncthreads();
        int it = 0, l = 0, sp = s_begin, rs = 0;
        for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
            const int buf = it & 1;
            const int rn = rs == 2 ? 0 : rs + 1;
            // (GA_DPP: the operands of a plane -- the products of the slot's one or two types and the field values -- are requested
            // one plane ahead; a source is P P row-broadcast multiply-adds)
            double pv1[NPVF], pv2[NPVF], bv1 = 0.0, bv2 = 0.0;
            auto operands = [&](double (&pv)[NPVF], double &bv, const int j, const int ty, const int ff) {
                const double *pr_ = &prd_[((buf * NS + j) * 4 + ty) * (NPVF * 16) + (lane & 15)];
#pragma unroll
                for (int k = 0; k < NPVF; ++k) pv[k] = pr_[16 * k];
                bv = FLD(buf, j, ff, lane);
            };
            if constexpr (DPF) {
                operands(pv1, bv1, 0, t, fi);
                if (xf >= 0) operands(pv2, bv2, 0, xt, xf);
            }
            auto plane = [&](const int j) __attribute__((always_inline)) -> bool {
                if (gb + j >= g_end) return false;
                if constexpr (DPF) {
                    double pn1[NPVF], pn2[NPVF], bn1, bn2 = 0.0;
                    const int jn = j + 1 < NS ? j + 1 : j;
                    operands(pn1, bn1, jn, t, fi);
                    if (xf >= 0) operands(pn2, bn2, jn, xt, xf);
                    asm volatile("" ::: "memory");
                    if (!GA_OFF(4)) {
                        sweep_full_rowbc<P>(acc, pv1, bv1);
                        if (xf >= 0) sweep_full_rowbc<P>(acc, pv2, bv2);
                    } else acc[0][0] += (pv1[0] + bv1) + (pv2[0] + bv2);
#pragma unroll
                    for (int a = 0; a < P; ++a)
#pragma unroll
                        for (int b = 0; b < P; ++b) asm volatile("" : "+v"(acc[a][b]));
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k = 0; k < NPVF; ++k) { pv1[k] = pn1[k]; pv2[k] = pn2[k]; }
                    bv1 = bn1; bv2 = bn2;
                } else {
                    source(rs, buf, j, t, fi);
                    if (xf >= 0) source(rs, buf, j, xt, xf);
                }
                if (++l < q) return true;
                const bool write = sp >= own_lo && !GA_OFF(2);
                const int *fr = (const int *)&rec[rs][j][20];
                const int nst = __builtin_amdgcn_readfirstlane(fr[0]), st0 = __builtin_amdgcn_readfirstlane(fr[1]);
                for (int st = st0; st < st0 + nst; ++st) {
                    int pr[16];
                    if (st == st0) {
                        const int *f16 = (const int *)&rec[rs][j][24];
#pragma unroll
                        for (int a = 0; a < P; ++a) pr[a] = __builtin_amdgcn_readfirstlane(f16[a]);
#pragma unroll
                        for (int a = 1; a < P; ++a) pr[8 + a] = __builtin_amdgcn_readfirstlane(f16[8 + a]);
                    } else {                              // (several dofs leave at the end of the axis)
                        const int8v r0 = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 16);
                        const int8v r1 = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 16 + 8);
#pragma unroll
                        for (int a = 0; a < P; ++a) { pr[a] = r0[a]; pr[8 + a] = r1[a]; }
                    }
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (pr[a] >= 0 && write) k1s.store(out, pr[a], A.stride, acc[a][0]);
#pragma unroll
                    for (int a = 1; a < P; ++a)
                        if (pr[8 + a] >= 0 && write) k1s.store(out, pr[8 + a], A.stride, acc[0][a]);
#pragma unroll
                    for (int a = 0; a < P - 1; ++a)
#pragma unroll
                        for (int b = 0; b < P - 1; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
                    for (int b = 0; b < P; ++b) { acc[P - 1][b] = 0.0; acc[b][P - 1] = 0.0; }
                }
                l = 0; ++sp;
                return true;
            };
            // (the planes of a batch unrolled where the registers allow it: the waits for the LDS reads are counted then and the
            // next plane's operands stay in flight under this plane's arithmetic)
            if constexpr (DPF && P <= GA_F1_UNROLL_P) {
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    if (!plane(j)) break;
            } else {
#pragma unroll 1
                for (int j = 0; j < NS; ++j)
                    if (!plane(j)) break;
            }
            next_batch_barriers(gb + NS, rn);
            rs = rn;
            __syncthreads();
        }
        return;
    }
    // ---- pair-product sweep (GA_DPP; symmetric forms, at most 16 lower pairs: p <= 4): the products V_a[tv](g0) V_b[tu](g0) of a
    // plane are functions of the plane alone -- one thread computes one of them per batch (plane, type, pair) -> LDS; a sweep
    // wave reads ITS type's 16 products of a plane into one register (lane m of each row of 16 = pair m) and the field value of
    // its point, and updates the pair window with P (P + 1) / 2 row-broadcast multiply-adds: per plane and wave 15 FP64
    // instructions and 2 LDS reads at p = 4 instead of 5 + 15 and 7.  (acc += (V_a V_b) f instead of acc += V_b (V_a f): the
    // same roundings per (plane, pair) wherever the batches cut, so slabs and chunks still reproduce the patch bit for bit.)
    if constexpr (DPS) {
        // (a form with fewer arrays than waves -- mass: one -- runs on the same block: every wave evaluates its plane of the batch,
        // the waves without an array sweep and store nothing.  Mass at C4's size: 2.27 ms on one-wave blocks -> see DESIGN 3.1)
        const bool live = w < A.nslots;
        const int t = A.type[w], fi = A.field[w];
        double *const out = A.out[w] + pt;
        const K1Store k1s(A.out[w], pt, tile, A.soff_ok);
        double acc[P][P];
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
        auto products = [&](const int rsl, const int bufn) {
            for (int i = tid; i < NS * 64; i += NT) {
                const int jp = i >> 6, ty = (i >> 4) & 3, m = i & 15;
                const int a = m >= 10 ? 4 : m >= 6 ? 3 : m >= 3 ? 2 : m >= 1 ? 1 : 0, b = m - a * (a + 1) / 2;
                const double x = rec[rsl][jp][6 * (ty >> 1) + a] * rec[rsl][jp][6 * (ty & 1) + b];
                prd_[((bufn * NS + jp) * 4 + ty) * 16 + m] = m < NPR ? x : 0.0;
            }
        };
        const double *const pw = &prd_[t * 16 + (lane & 15)];
#ifdef IGX_GA_STAMP
        unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#endif
        stage_load(g_begin); stage_store(0);
        stage_load(g_begin + NS); stage_store(1);
        __syncthreads();
        next_batch(g_begin, 0, 0);
        products(0, 0);
        __syncthreads();
        GA_T(0);
        int it = 0, l = 0, sp = s_begin, rs = 0;
        for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
            const int buf = it & 1;
            const int rn = rs == 2 ? 0 : rs + 1, ra = rn == 2 ? 0 : rn + 1;
            stage_load(gb + 2 * NS);
            double pv[1] = {pw[(buf * NS) * 64]};
            double bv = FLD(buf, 0, fi, lane);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (gb + j >= g_end || !live) break;
                // the operands of the next plane are requested before this plane's arithmetic
                const int jn = j + 1 < NS ? j + 1 : j;
                const double pvn = pw[(buf * NS + jn) * 64], bvn = FLD(buf, jn, fi, lane);
                asm volatile("" ::: "memory");
                if (!GA_OFF(4)) sweep_lower_rowbc<P>(acc, pv, bv);
                else acc[0][0] += pv[0] + bv;
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) asm volatile("" : "+v"(acc[a][b]));
                asm volatile("" ::: "memory");
                pv[0] = pvn; bv = bvn;
                GA_T(1);
                if (++l < q) continue;
                const bool write = sp >= own_lo && !GA_OFF(2);
                const int *fr = (const int *)&rec[rs][j][20];
                const int nst = __builtin_amdgcn_readfirstlane(fr[0]), st0 = __builtin_amdgcn_readfirstlane(fr[1]);
                for (int st = st0; st < st0 + nst; ++st) {
                    int pr[P];
                    if (st == st0) {
#pragma unroll
                        for (int a = 0; a < P; ++a) pr[a] = __builtin_amdgcn_readfirstlane(fr[2 + a]);
                    } else {                              // (several dofs leave at the end of the axis)
                        const int8v rec8 = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 8);
#pragma unroll
                        for (int a = 0; a < P; ++a) pr[a] = rec8[a];
                    }
#pragma unroll
                    for (int a = 0; a < P; ++a)
                        if (pr[a] >= 0 && write) k1s.store(out, pr[a], A.stride, acc[a][0]);
#pragma unroll
                    for (int a = 0; a < P - 1; ++a)
#pragma unroll
                        for (int b = 0; b <= a; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
                    for (int b = 0; b < P; ++b) acc[P - 1][b] = 0.0;
                }
                l = 0; ++sp;
                GA_T(2);
            }
            GA_T(1);
            next_batch(gb + NS, rn, buf ^ 1);
            products(rn, buf ^ 1);
            GA_T(0);
            stage_store(ra);
            rs = rn;
            __syncthreads();
            GA_T(3);
        }
#ifdef IGX_GA_STAMP
        if (lane == 0 && blockIdx.x < 2048 && blockIdx.y == 0)
            for (int i = 0; i < 4; ++i) g_ga_stamp[(blockIdx.x * 8 + (w & 7)) * 6 + i] = st_[i];
#endif
        return;
    }
    // ---- sweep state of this wave
    const int t = A.type[w], fi = A.field[w];
    double *const out = A.out[w] + pt;
    const K1Store k1s(A.out[w], pt, tile, A.soff_ok);
    double acc[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
#if GA_SPREAD
    double park[P];
    int pslot[P];
    bool have = false;
#pragma unroll
    for (int a = 0; a < P; ++a) { park[a] = 0.0; pslot[a] = -1; }
#endif
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int tu = t & 1, tv = t >> 1;
    auto basis_row = [&](double (&v)[PV], const int buf, const int j, const int d) {
        const d2 *row = (const d2 *)&rec[buf][j][6 * d];
#pragma unroll
        for (int k = 0; k < P / 2; ++k) { const d2 x = row[k]; v[2 * k] = x.x; v[2 * k + 1] = x.y; }
        if (P & 1) v[P - 1] = rec[buf][j][6 * d + P - 1];
    };

#ifdef IGX_GA_STAMP
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#endif
    // prologue: plane records of batches 0 and 1, fields of batch 0
    stage_load(g_begin); stage_store(0);
    stage_load(g_begin + NS); stage_store(1);
    __syncthreads();
    next_batch(g_begin, 0, 0);
    __syncthreads();
    GA_T(0);
    int it = 0, l = 0, sp = s_begin;                      // plane gb + j = point l of span sp
    int rs = 0;                                           // record slot of the batch being swept = it % 3
    for (int gb = g_begin; gb < g_end; gb += NS, ++it) {
        const int buf = it & 1;
        const int rn = rs == 2 ? 0 : rs + 1, ra = rn == 2 ? 0 : rn + 1;     // slots of batches it + 1 (evaluated), it + 2 (arriving)
        // Order of an iteration: request the records of batch it + 2 -> sweep of this batch (K1 stores) -> geometry of the
        // next batch -> records to LDS -> barrier.  The compiler cannot count the stores of the flush (they sit behind
        // branches), so the wait for the record loads is a vmcnt(0): with the geometry evaluation between the last store and
        // that wait the stores are half an iteration old by then, instead of draining at full HBM latency once per batch
        // in front of the barrier.
        stage_load(gb + 2 * NS);
        double bv = FLD(buf, 0, fi, lane);
        double va[PV], vb[PV];                            // V[.][tv] (test functions, rows a), V[.][tu] (trial functions, columns b)
        basis_row(va, rs, 0, tv);
        if (GA_READBOTH || tu != tv) basis_row(vb, rs, 0, tu);
        else {
#pragma unroll
            for (int k = 0; k < PV; ++k) vb[k] = va[k];
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (gb + j >= g_end) break;
            // the rows of the next plane replace these right after their last use: the LDS reads are in flight under the FMAs
            const int jn = j + 1 < NS ? j + 1 : j;
            const double bvn = FLD(buf, jn, fi, lane);
            double c[P];
#pragma unroll
            for (int a = 0; a < P; ++a) c[a] = va[a] * bv;
#pragma unroll
            for (int a = 0; a < P; ++a) asm volatile("" : "+v"(c[a]));
            asm volatile("" ::: "memory");
            basis_row(va, rs, jn, tv);
            if (!GA_OFF(4)) {
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[a][b] = fma(vb[b], c[a], acc[a][b]);
            } else acc[0][0] += c[0] + vb[0];
#pragma unroll
            for (int a = 0; a < P; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) asm volatile("" : "+v"(acc[a][b]));
            asm volatile("" ::: "memory");
            if (GA_READBOTH || tu != tv) basis_row(vb, rs, jn, tu);
            else {                                        // same row: no second broadcast read
#pragma unroll
                for (int k = 0; k < PV; ++k) vb[k] = va[k];
            }
            bv = bvn;
#if GA_SPREAD
            // the completed pairs of the LAST span go out one per plane of this one (q >= P planes: all are out before the
            // next span end parks new ones): the store path sees a steady trickle instead of a burst of P stores per wave
            if (have) {
#pragma unroll
                for (int a = 0; a < P; ++a)
                    if (l == a && pslot[a] >= 0) k1s.store(out, pslot[a], A.stride, park[a]);
            }
#endif
            GA_T(1);                                      // sweep arithmetic (+ parked stores)
            if (++l < q) continue;
            // dofs that leave the active set after span sp: their pairs are complete
            const bool write = sp >= own_lo && !GA_OFF(2);
            const int *fr = (const int *)&rec[rs][j][20];
            const int nst = __builtin_amdgcn_readfirstlane(fr[0]), st0 = __builtin_amdgcn_readfirstlane(fr[1]);
#if GA_SPREAD
            have = false;
#endif
            for (int st = st0; st < st0 + nst; ++st) {
                int pr[P];
                if (st == st0) {
#pragma unroll
                    for (int a = 0; a < P; ++a) pr[a] = __builtin_amdgcn_readfirstlane(fr[2 + a]);
                } else {                                  // (several dofs leave at the end of the axis)
                    const int8v rec = *(const int8v __attribute__((address_space(4))) *)((cip)A.steps + (size_t)st * 8);
#pragma unroll
                    for (int a = 0; a < P; ++a) pr[a] = rec[a];
                }
#if GA_SPREAD
                if (nst == 1 && gb + j + 1 < g_end) {      // (one dof leaves and the sweep goes on: park; else store at once)
#pragma unroll
                    for (int a = 0; a < P; ++a) { park[a] = acc[a][0]; pslot[a] = write ? pr[a] : -1; }
                    have = true;
                } else
#endif
#pragma unroll
                for (int a = 0; a < P; ++a)
                    if (pr[a] >= 0 && write) k1s.store(out, pr[a], A.stride, acc[a][0]);
#pragma unroll
                for (int a = 0; a < P - 1; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
                for (int b = 0; b < P; ++b) acc[P - 1][b] = 0.0;
            }
            l = 0; ++sp;
            GA_T(2);                                      // flush
        }
        GA_T(1);
        next_batch(gb + NS, rn, buf ^ 1);
        GA_T(0);                                          // geometry
        stage_store(ra);
        rs = rn;
        __syncthreads();
        GA_T(3);                                          // barrier
    }
#ifdef IGX_GA_STAMP
    if (lane == 0 && blockIdx.x < 2048 && blockIdx.y == 0)
        for (int i = 0; i < 4; ++i) g_ga_stamp[(blockIdx.x * 8 + (w & 7)) * 6 + i] = st_[i];
#endif
}

template <int P, int NS, int P0G>
static int launch_geoA_2d(hipStream_t st, const GeoAArgs &A, int nc, dim3 grid)
{
    if (nc == 3) k_geoA<P, NS, P0G, 3, false, 0, true><<<grid, dim3(geoa_threads(NS, 0)), 0, st>>>(A);
    else k_geoA<P, NS, P0G, 2, false, 0, true><<<grid, dim3(geoa_threads(NS, 0)), 0, st>>>(A);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

template <int P, int NS, int P0G, bool MF = false, int FORM = 0>
static int launch_geoA_k(hipStream_t st, const GeoAArgs &A, int nc, dim3 grid)
{
    if (nc == 4) k_geoA<P, NS, P0G, 4, MF, FORM><<<grid, dim3(geoa_threads(NS, FORM)), 0, st>>>(A);
    else k_geoA<P, NS, P0G, 3, MF, FORM><<<grid, dim3(geoa_threads(NS, FORM)), 0, st>>>(A);
    IGX_HIP(hipGetLastError());
#ifdef IGX_GA_STAMP
    {
        static std::vector<unsigned long long> h(2048 * 8 * 6);
        IGX_HIP(hipStreamSynchronize(st));
        IGX_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_ga_stamp), h.size() * sizeof(unsigned long long)));
        const int nb = std::min<unsigned>(grid.x, 2048);
        for (int w = 0; w < NS; ++w) {
            double t[4] = {0, 0, 0, 0};
            for (int b = 0; b < nb; ++b) for (int i = 0; i < 4; ++i) t[i] += h[(b * 8 + w) * 6 + i];
            fprintf(stderr, "k_geoA stamp: wave %d  geometry %.0f  sweep %.0f  flush %.0f  barrier %.0f  (x100 ns per block)\n", w, t[0] / nb, t[1] / nb, t[2] / nb, t[3] / nb);
        }
    }
#endif
    return IGX_OK;
}

template <int P, int NS>
static int launch_geoA_g(hipStream_t st, const GeoAArgs &A, int nc, int p0g, dim3 grid)
{
    switch (p0g) {
    case 2: return launch_geoA_k<P, NS, 2>(st, A, nc, grid);
    case 3: return launch_geoA_k<P, NS, 3>(st, A, nc, grid);
    }
    return IGX_ERR_UNSUPPORTED;
}

bool geoA_supported(const igx_patch *pt, int kind, int nslots)
{
    if (pt->dim == 2) {
        // 2D (round 6): mass (one array) and stiffness (four), spline maps of degree 1 or 2 along axis 0, degrees 1 .. 4 (the
        // pair products of a plane in one register); a resident row slab is a span range of axis 0 like in 3D
        if ((kind != IGX_STIFFNESS && kind != IGX_MASS) || pt->boxed) return false;
        if (pt->geo_kind != IGX_GEO_BSPLINE && pt->geo_kind != IGX_GEO_NURBS) return false;
        if (nslots != (kind == IGX_MASS ? 1 : 4)) return false;
        const int P = pt->ax[0].P, p0g = pt->gax[0].P;
        if (P < 2 || P > 5 || p0g < 2 || p0g > 3) return false;
        const long long gspans = pt->gax[0].N - pt->gax[0].P + 1;
        return 2 * gspans <= (long long)pt->ax[0].G;
    }
    if (pt->dim != 3 || (kind != IGX_STIFFNESS && kind != IGX_MASS && kind != IGX_CONVDIFF)) return false;
    if (pt->geo_kind != IGX_GEO_BSPLINE && pt->geo_kind != IGX_GEO_NURBS) return false;
    if (nslots != (kind == IGX_MASS ? 1 : 8)) return false;
    const int P = pt->ax[0].P;
    if (P < 2 || P > 6) return false;
    // the convection-diffusion form (non-symmetric, eight merged slots): kernels for p = 2 .. 5, whole Gauss planes resident
    if (kind == IGX_CONVDIFF && (P < 3 || pt->boxed)) return false;
    const int p0g = pt->gax[0].P;
    if (p0g < 2 || p0g > 3) return false;
    // the column coefficients are recomputed (by the whole block, with barriers) at every span boundary of the geometry's
    // axis 0: worth it while the geometry is coarser than the quadrature grid.  Decided on the whole axis, not on the
    // resident slab: every slab of a patch takes the same path (bit-identical row blocks).
    const long long gspans = pt->gax[0].N - pt->gax[0].P + 1;
    return 2 * gspans <= (long long)pt->ax[0].G;
}

// general first-order forms (coefficient table): 3D spline geometries, degrees 2 .. 5 on axis 0, the whole Gauss slab resident
bool geoA_form_supported(const igx_patch *pt)
{
    if (pt->dim != 3 || pt->boxed || !pt->knobs.geoa) return false;
    if (pt->geo_kind != IGX_GEO_BSPLINE && pt->geo_kind != IGX_GEO_NURBS) return false;
    const int P = pt->ax[0].P, p0g = pt->gax[0].P;
    if (P < 3 || P > 6 || p0g < 2 || p0g > 3) return false;
    const long long gspans = pt->gax[0].N - pt->gax[0].P + 1;
    return 2 * gspans <= (long long)pt->ax[0].G;
}

int launch_geoA(hipStream_t st, igx_patch *pt, int kind, int nslots, const int *slot_field, const int *slot_type,
                double *const *slot_out, long long slice_stride, int chunk_len, int nchunks, const int *slot_xfield, const int *slot_xtype,
                const GeoAForm *form)
{
    const bool nonsym = kind == IGX_CONVDIFF || (form && !form->sym);
    if (nonsym) {
        if (kind == IGX_CONVDIFF && !pt->coef_affine && !pt->coeff_sampled) { set_error("IGX_CONVDIFF needs igx_patch_set_coeff first"); return IGX_ERR_ARG; }
        if (!pt->d_stepsn) { set_error("internal: flush records of the non-symmetric form are missing"); return IGX_ERR_UNSUPPORTED; }
        if (!pt->d_geoa_tabn) {                          // per-plane records with the 16-int flush steps of the non-symmetric sweep
            double *tab = nullptr;
            const Axis &A0n = pt->ax[0];
            if (hipMalloc((void **)&tab, (size_t)A0n.G * GA_REC * sizeof(double)) != hipSuccess) { set_error("per-plane table (non-symmetric form): out of device memory"); return IGX_ERR_NOMEM; }
            k_geoa_table<<<dim3((A0n.G + 127) / 128), dim3(128), 0, st>>>(A0n.d_V, A0n.P, pt->gax[0].d_V, pt->gax[0].d_fa, pt->gax[0].P, pt->dev.ax[0].w, A0n.q,
                                                                           pt->stepA_ptr, pt->d_stepsn, A0n.G, tab, nullptr, 1);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) { (void)hipFree(tab); set_error("per-plane table (non-symmetric form): %s", hipGetErrorString(e)); return IGX_ERR_HIP; }
            pt->d_geoa_tabn = tab;
        }
    }
    const PatchDev &pd = pt->dev;
    const Axis &A0 = pt->ax[0];
    if (!nonsym && !pt->d_geoa_tab) {                    // per-plane records of axis 0: once per patch
        // row tables of the matrix-core sweep, per span: [16] row -> a | b << 4 | 256, [16] row -> K1 slot of the pair when
        // it is complete after the span (-2: complete, not processed by this slab; -1: not complete), [2] smallest slot
        std::vector<int> rows;
        bool mf_ok = pt->dim == 3 && A0.P >= 2 && A0.P * (A0.P + 1) / 2 <= GA_ROWS;
        if (mf_ok) {
            const int P = A0.P, W = 2 * GA_ROWS + 2;
            std::vector<int> slot_of(A0.S, -1);
            for (size_t r = 0; r + 1 < pt->h_pl0.size(); r += 2) {
                const int i0 = pt->h_pl0[r], j0 = pt->h_pl0[r + 1];
                slot_of[A0.rp[i0] + (j0 - A0.jlo[i0])] = (int)(r / 2);
            }
            int base[8] = {0};
            for (int d = 1; d < P; ++d) base[d] = base[d - 1] + (P - (d - 1));
            rows.assign((size_t)A0.n * W, 0);
            for (int sp = 0; sp < A0.n; ++sp) {
                int *r = &rows[(size_t)sp * W];
                for (int m = 0; m < GA_ROWS; ++m) r[GA_ROWS + m] = -1;
                const int f = A0.fa[sp], fnext = sp + 1 < A0.n ? A0.fa[sp + 1] : f + P;
                int lo = -1, hi = -1;
                for (int j0 = f; j0 < f + P; ++j0)
                    for (int i0 = j0; i0 < f + P; ++i0) {
                        const int d = i0 - j0, m = base[d] + j0 % (P - d);
                        r[m] = (i0 - f) | ((j0 - f) << 4) | 256;
                        if (j0 < fnext) {
                            const int sl = slot_of[A0.rp[i0] + (j0 - A0.jlo[i0])];
                            r[GA_ROWS + m] = sl >= 0 ? sl : -2;
                            if (sl >= 0) { lo = lo < 0 ? sl : std::min(lo, sl); hi = std::max(hi, sl); }
                        }
                    }
                r[2 * GA_ROWS] = std::max(lo, 0);
                // store offsets relative to the smallest slot of the span are 32-bit
                if (lo >= 0 && ((long long)(hi - lo) * slice_stride + 64) * 8 >= 0x7ff00000LL) mf_ok = false;
            }
        }
        int *d_rows = nullptr;
        if (mf_ok) {
            IGX_HIP(hipMalloc((void **)&d_rows, rows.size() * sizeof(int)));
            if (hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { (void)hipFree(d_rows); set_error("row tables of the matrix-core sweep: upload failed"); return IGX_ERR_HIP; }
        }
        double *tab = nullptr;                           // committed to the patch only when the build has been launched
        if (hipMalloc((void **)&tab, (size_t)A0.G * GA_REC * sizeof(double)) != hipSuccess) { (void)hipFree(d_rows); set_error("per-plane table of the fused geometry + axis-0 sweep: out of device memory"); return IGX_ERR_NOMEM; }
        k_geoa_table<<<dim3((A0.G + 127) / 128), dim3(128), 0, st>>>(A0.d_V, A0.P, pt->gax[0].d_V, pt->gax[0].d_fa, pt->gax[0].P, pd.ax[0].w, A0.q,
                                                                     pt->stepA_ptr, pt->stepA_rec, A0.G, tab, d_rows, 0);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // (`rows` and d_rows are released below)
        (void)hipFree(d_rows);
        if (e != hipSuccess) {
            (void)hipFree(tab);
            set_error("per-plane table of the fused geometry + axis-0 sweep: %s", hipGetErrorString(e));
            return IGX_ERR_HIP;
        }
        pt->d_geoa_tab = tab;
        pt->geoa_mf = mf_ok ? 1 : 0;
    }
    GeoAArgs A{};
    const bool d2 = pt->dim == 2;
    A.gv = make_view(pt->dim, pt->gax, pt->d_ctrl, pt->ncomp);
    A.w0 = pd.ax[0].w; A.w1 = pd.ax[1].w; A.w2 = d2 ? nullptr : pd.ax[2].w;
    A.nurbs = pt->geo_kind == IGX_GEO_NURBS; A.kind = kind;
    A.G1 = pd.ax[1].G; A.G2 = d2 ? 1 : pd.ax[2].G;
    A.NPL = (long long)A.G1 * A.G2; A.stride = slice_stride;
    // (the hardware's range check adds the scalar offset to the lane offset before it compares with the length of the
    // descriptor: a slice offset at or above 2^31 - 16 would have its stores DROPPED -- found in round 4 by the WRITE_SIZE counter
    // of C5 reading 6 of 8 arrays' worth; C4's largest offset is 2.127e9, just below)
    A.soff_ok = (long long)(nonsym ? pt->npairs0n : pt->npairs0) * slice_stride * 8 + 1024 < 0x7ffffff0LL ? 1 : 0;
    A.tab = nonsym ? pt->d_geoa_tabn : pt->d_geoa_tab; A.steps = nonsym ? pt->d_stepsn : pt->stepA_rec;
    A.coeff = pt->d_coeff; A.g0_lo = pd.g0_lo; A.coef_affine = pt->coef_affine;
    for (int k = 0; k < 4; ++k) A.cf[k] = pt->coef_c[k];
    A.s_lo = pt->s0_lo; A.s_hi = pt->s0_hi; A.q = A0.q; A.chunk_len = chunk_len;
#ifdef IGX_ABLATE
    { const char *e = getenv("IGX_GEOA_DBG"); A.dbg = e ? atoi(e) : 0; }
#endif
    for (int x = 0; x < nslots; ++x) {
        A.field[x] = slot_field ? slot_field[x] : 0; A.type[x] = slot_type ? slot_type[x] : 0; A.out[x] = slot_out[x];
        A.xfield[x] = slot_xfield ? slot_xfield[x] : -1; A.xtype[x] = slot_xtype ? slot_xtype[x] : 0;
    }
    A.nslots = nslots;
    for (int x = nslots; x < GA_MAXS; ++x) A.out[x] = A.out[0];  // (waves without an array: never stored through)
    A.ntiles = (int)((A.NPL + 63) / 64);
    dim3 grid((unsigned)((A.ntiles + 7) / 8 * 8), nchunks);     // the kernel permutes the tiles over the XCDs
    const int nc = pt->ncomp, p0g = pt->gax[0].P;
    if (d2) {
#define GEOA_2D(PV) case PV: return nslots == 1 ? (p0g == 2 ? launch_geoA_2d<PV, 1, 2>(st, A, nc, grid) : launch_geoA_2d<PV, 1, 3>(st, A, nc, grid)) \
                                                : (p0g == 2 ? launch_geoA_2d<PV, 4, 2>(st, A, nc, grid) : launch_geoA_2d<PV, 4, 3>(st, A, nc, grid))
        switch (A0.P) { GEOA_2D(2); GEOA_2D(3); GEOA_2D(4); GEOA_2D(5); }
#undef GEOA_2D
        set_error("fused geometry + stage A (2D): unsupported degree %d", A0.p);
        return IGX_ERR_UNSUPPORTED;
    }
    if (form) {
        for (int k = 0; k < 16; ++k) { A.pc[k] = form->pc[k]; A.pa[k] = form->pa[k]; A.fslot[k] = form->fslot[k]; }
        A.pmask = form->pmask; A.nslots = nslots; A.amask = 0; A.fmask = 0;
        for (int k = 0; k < 16; ++k) { if (form->pa[k]) A.amask |= 1 << k; if (form->fslot[k] >= 0) A.fmask |= 1 << k; }
        for (int x = 0; x < GA_MAXS; ++x) {
            A.nsrc[x] = x < nslots ? form->nsrc[x] : 1;
            for (int k = 0; k < 4; ++k) { A.sfield[x][k] = x < nslots ? form->sfield[x][k] : 0; A.stype[x][k] = x < nslots ? form->stype[x][k] : 0; }
            if (x >= nslots) A.out[x] = A.out[0];
        }
#define GEOA_T(PV, FV) case PV: return p0g == 2 ? launch_geoA_k<PV, 8, 2, false, FV>(st, A, nc, grid) : p0g == 3 ? launch_geoA_k<PV, 8, 3, false, FV>(st, A, nc, grid) : IGX_ERR_UNSUPPORTED
        if (form->sym) switch (A0.P) { GEOA_T(3, 3); GEOA_T(4, 3); GEOA_T(5, 3); GEOA_T(6, 3); }
        else switch (A0.P) { GEOA_T(3, 2); GEOA_T(4, 2); GEOA_T(5, 2); GEOA_T(6, 2); }
#undef GEOA_T
        set_error("fused geometry + stage A (general form): unsupported degree %d", A0.p);
        return IGX_ERR_UNSUPPORTED;
    }
    if (nonsym) {
#define GEOA_N(PV) case PV: return p0g == 2 ? launch_geoA_k<PV, 8, 2, false, 1>(st, A, nc, grid) : p0g == 3 ? launch_geoA_k<PV, 8, 3, false, 1>(st, A, nc, grid) : IGX_ERR_UNSUPPORTED
        switch (A0.P) {
            GEOA_N(3); GEOA_N(4); GEOA_N(5); GEOA_N(6);
        }
#undef GEOA_N
        set_error("fused geometry + stage A (non-symmetric): unsupported degree %d", A0.p);
        return IGX_ERR_UNSUPPORTED;
    }
    // matrix-core sweep (opt-in, IGX_GEOA=mfma): eight slots (3D stiffness), p = 3, 4 (15 / 10 live pairs on 16 rows; below
    // that the vector form issues fewer cycles, above it the pairs do not fit one tile).  One choice per patch: every slab
    // and chunk agrees.
    // (geometries of degree 1 along axis 0 only: with three column sets the LDS image leaves one block per CU)
    if (nslots == 8 && pt->geoa_mf && pt->knobs.geoa_mf && (A0.P == 5 || A0.P == 4) && A0.q == A0.P && p0g == 2) {
        if (A0.P == 5) return launch_geoA_k<5, 8, 2, true>(st, A, nc, grid);
        return launch_geoA_k<4, 8, 2, true>(st, A, nc, grid);
    }
    // (mass -- one array -- on the eight-wave block where the pair-product sweep applies (p <= 4): the geometry of eight planes in
    // parallel; the one-wave block of rounds 2-5 above that)
#define GEOA_P(PV) case PV: return (nslots == 1 && !(GA_MASS8 && PV <= 5)) ? launch_geoA_g<PV, 1>(st, A, nc, p0g, grid) : launch_geoA_g<PV, 8>(st, A, nc, p0g, grid)
    switch (A0.P) {
        GEOA_P(2); GEOA_P(3); GEOA_P(4); GEOA_P(5); GEOA_P(6);
    }
#undef GEOA_P
    set_error("fused geometry + stage A: unsupported degree %d", A0.p);
    return IGX_ERR_UNSUPPORTED;
}

} // namespace igx

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
#ifndef GA_READBOTH
#define GA_READBOTH 1         // both basis rows of a plane are read from LDS even when they are the same row (no register copies)
#endif
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
};

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

// Per-plane record of axis 0 (doubles): everything wave-uniform that a Gauss plane g needs, built once per patch so that
// the staging inside the sweep is one coalesced copy without dependent loads or branches:
//   [0, 12)   basis of the SPACE at the plane: [value | derivative][active function a < P <= 6]
//   [12, 18)  basis of the GEOMETRY: (N, N') of its <= 3 active functions;  [18] quadrature weight;  [19] first active
//             control index (as a double)
//   [20, 24)  eight ints: number of flush steps after this plane (0 unless it ends a span) | first step | K1 slots of the
//             first step's P pairs
constexpr int GA_REC = 24;

__global__ void k_geoa_table(const double *V0s, int P, const double *V0g, const int *fa0g, int P0G, const double *w0, int q,
                             const int *step_ptr, const int *steps, int G0, double *tab)
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
        if (st1 > st0)
            for (int a = 0; a < P; ++a) ri[2 + a] = steps[(size_t)st0 * 8 + a];
    }
}

// Uniform tables (sweep coefficients, axis-0 geometry basis, flush records) are NOT read with scalar loads inside the loop:
// the scalar cache holds 16 KB, the tables are larger, and a scalar load that misses it queues in the L2 behind the K1
// store stream -- tools/ubench/k1_store.hip: 3.1 ms of arithmetic with such loads become 9.4 ms when the stores are on,
// while the same arithmetic without them overlaps the stores completely.  The block stages the table rows of a batch in
// LDS with vector loads issued a whole batch ahead.
template <int P, int NS, int P0G, int NC>
__global__ void __launch_bounds__(NS * 64) __attribute__((amdgpu_waves_per_eu(NS >= 8 ? 4 : 1, 4)))
k_geoA(const GeoAArgs A)
{
    constexpr int PV = (P + 1) & ~1;                      // basis row in registers, padded to an even length
    constexpr int NT = NS * 64;                           // threads
    constexpr int NRC = NS * GA_REC;                      // doubles of a batch of plane records
    constexpr int KRC = (NRC + NT - 1) / NT;              // ... per thread
    __shared__ double fld[2][NS][6][64];                  // fields of two batches of planes
    __shared__ double Cs[P0G * NC][3][64];                // column coefficients of the block's points, geometry span f0_blk
    __shared__ __attribute__((aligned(16))) double rec[3][NS][GA_REC];   // plane records of three batches: swept | evaluated | arriving
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // consecutive block ids go to different XCDs: give each XCD a contiguous range of point tiles
    const int per_xcd = gridDim.x >> 3;                   // the grid is padded to a multiple of 8 blocks
    int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
#ifdef IGX_ABLATE
    if (A.dbg & 8) tile = blockIdx.x;
#endif
    if (tile >= A.ntiles) tile = A.ntiles - 1;            // surplus blocks redo the last tile (same values again: harmless)
    long long pt = (long long)tile * 64 + lane;
    if (pt >= A.NPL) pt = A.NPL - 1;                      // lanes past the end redo the last point (and store it again: harmless)
    const int g1 = (int)(pt / A.G2), g2 = (int)(pt - (long long)g1 * A.G2);
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
            const int i = tid + k * NT, j = i / GA_REC;
            if (i < NRC) rc_reg[k] = A.tab[(size_t)min(gb + j, g_last) * GA_REC + (i - j * GA_REC)];
        }
    };
    auto stage_store = [&](const int slot) {
#pragma unroll
        for (int k = 0; k < KRC; ++k) {
            const int i = tid + k * NT;
            if (i < NRC) (&rec[slot][0][0])[i] = rc_reg[k];
        }
    };

    // ---- geometry
    const double GW1 = A.w1[g1], GW2 = A.w2[g2];
    // the prologue loads are complete before the loop: a wait for them inside it would also drain the K1 stores
    asm volatile("" :: "v"(GW1), "v"(GW2));
    int f0_blk = -1;
    // the block's waves share the work: wave w contracts the control net along axes 1, 2 for its (a0, component) pairs
    auto columns = [&](const int f0) {
        const double *V1 = gv.V[1] + (size_t)g1 * gv.P[1] * 2, *V2 = gv.V[2] + (size_t)g2 * gv.P[2] * 2;
        const int f1 = gv.fa[1][g1], f2 = gv.fa[2][g2];
        for (int e = w; e < P0G * NC; e += NS) {
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
    auto evaluate = [&](const int gbuf, const int buf) {
        const double *gt = &rec[gbuf][w][12];
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
                const double c0 = Cs[a0 * NC + c][0][lane], c1 = Cs[a0 * NC + c][1][lane], c2 = Cs[a0 * NC + c][2][lane];
                val[c] = fma(n, c0, val[c]);
                jac[c][0] = fma(d, c0, jac[c][0]);
                jac[c][1] = fma(n, c1, jac[c][1]);
                jac[c][2] = fma(n, c2, jac[c][2]);
            }
        }
        double GW = gw0 * GW1;
        GW = GW * GW2;
        if (GA_ONEDIV && NS == 8) {
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
            fld[buf][w][0][lane] = sc * ((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
            fld[buf][w][1][lane] = sc * ((a[0] * a[3] + a[1] * a[4]) + a[2] * a[5]);
            fld[buf][w][2][lane] = sc * ((a[0] * a[6] + a[1] * a[7]) + a[2] * a[8]);
            fld[buf][w][3][lane] = sc * ((a[3] * a[3] + a[4] * a[4]) + a[5] * a[5]);
            fld[buf][w][4][lane] = sc * ((a[3] * a[6] + a[4] * a[7]) + a[5] * a[8]);
            fld[buf][w][5][lane] = sc * ((a[6] * a[6] + a[7] * a[7]) + a[8] * a[8]);
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
            if (k < nf) fld[buf][w][k][lane] = f[k];
    };
    // all waves: evaluate the batch that starts at plane gn (table rows in gts[gbuf]).  The column coefficients belong to
    // one span of the geometry's axis 0; a batch that straddles span boundaries is evaluated span by span (uniform control:
    // every wave of the block takes the same way)
    auto next_batch = [&](const int gn, const int gbuf, const int buf) {
        if (gn >= g_end) return;
        const int jl = min(NS - 1, g_last - gn);
        const int mine = w <= jl ? __builtin_amdgcn_readfirstlane((int)rec[gbuf][w][19]) : -1;
        const int last = __builtin_amdgcn_readfirstlane((int)rec[gbuf][jl][19]);
        int cur = __builtin_amdgcn_readfirstlane((int)rec[gbuf][0][19]);
        for (;;) {
            if (cur != f0_blk) {
                columns(cur);
                f0_blk = cur;
                __syncthreads();
            }
            if (mine == cur && !GA_OFF(1)) evaluate(gbuf, buf);
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

    // ---- sweep state of this wave
    const int t = A.type[w], fi = A.field[w];
    double *const out = A.out[w] + pt;
    double acc[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
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
        double bv = fld[buf][0][fi][lane];
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
            const double bvn = fld[buf][jn][fi][lane];
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
            GA_T(1);                                      // sweep arithmetic (+ parked stores)
            if (++l < q) continue;
            // dofs that leave the active set after span sp: their pairs are complete
            const bool write = sp >= own_lo && !GA_OFF(2);
            const int *fr = (const int *)&rec[rs][j][20];
            const int nst = __builtin_amdgcn_readfirstlane(fr[0]), st0 = __builtin_amdgcn_readfirstlane(fr[1]);
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
#pragma unroll
                for (int a = 0; a < P; ++a)
                    if (pr[a] >= 0 && write) out[(long long)pr[a] * A.stride] = acc[a][0];
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
static int launch_geoA_k(hipStream_t st, const GeoAArgs &A, int nc, dim3 grid)
{
    if (nc == 4) k_geoA<P, NS, P0G, 4><<<grid, dim3(NS * 64), 0, st>>>(A);
    else k_geoA<P, NS, P0G, 3><<<grid, dim3(NS * 64), 0, st>>>(A);
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
    if (pt->dim != 3 || (kind != IGX_STIFFNESS && kind != IGX_MASS)) return false;
    if (pt->geo_kind != IGX_GEO_BSPLINE && pt->geo_kind != IGX_GEO_NURBS) return false;
    if (nslots != (kind == IGX_MASS ? 1 : 8)) return false;
    const int P = pt->ax[0].P;
    if (P < 2 || P > 6) return false;
    const int p0g = pt->gax[0].P;
    if (p0g < 2 || p0g > 3) return false;
    // the column coefficients are recomputed (by the whole block, with barriers) at every span boundary of the geometry's
    // axis 0: worth it while the geometry is coarser than the quadrature grid.  Decided on the whole axis, not on the
    // resident slab: every slab of a patch takes the same path (bit-identical row blocks).
    const long long gspans = pt->gax[0].N - pt->gax[0].P + 1;
    return 2 * gspans <= (long long)pt->ax[0].G;
}

int launch_geoA(hipStream_t st, igx_patch *pt, int kind, int nslots, const int *slot_field, const int *slot_type,
                double *const *slot_out, long long slice_stride, int chunk_len, int nchunks)
{
    const PatchDev &pd = pt->dev;
    const Axis &A0 = pt->ax[0];
    if (!pt->d_geoa_tab) {                               // per-plane records of axis 0: once per patch
        double *tab = nullptr;                           // committed to the patch only when the build has been launched
        IGX_HIP(hipMalloc((void **)&tab, (size_t)A0.G * GA_REC * sizeof(double)));
        k_geoa_table<<<dim3((A0.G + 127) / 128), dim3(128), 0, st>>>(A0.d_V, A0.P, pt->gax[0].d_V, pt->gax[0].d_fa, pt->gax[0].P, pd.ax[0].w, A0.q,
                                                                     pt->stepA_ptr, pt->stepA_rec, A0.G, tab);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            (void)hipFree(tab);
            set_error("per-plane table of the fused geometry + axis-0 sweep: %s", hipGetErrorString(e));
            return IGX_ERR_HIP;
        }
        pt->d_geoa_tab = tab;
    }
    GeoAArgs A{};
    A.gv = make_view(3, pt->gax, pt->d_ctrl, pt->ncomp);
    A.w0 = pd.ax[0].w; A.w1 = pd.ax[1].w; A.w2 = pd.ax[2].w;
    A.nurbs = pt->geo_kind == IGX_GEO_NURBS; A.kind = kind;
    A.G1 = pd.ax[1].G; A.G2 = pd.ax[2].G;
    A.NPL = (long long)A.G1 * A.G2; A.stride = slice_stride;
    A.tab = pt->d_geoa_tab; A.steps = pt->stepA_rec;
    A.s_lo = pt->s0_lo; A.s_hi = pt->s0_hi; A.q = A0.q; A.chunk_len = chunk_len;
#ifdef IGX_ABLATE
    { const char *e = getenv("IGX_GEOA_DBG"); A.dbg = e ? atoi(e) : 0; }
#endif
    for (int x = 0; x < nslots; ++x) { A.field[x] = slot_field[x]; A.type[x] = slot_type[x]; A.out[x] = slot_out[x]; }
    A.ntiles = (int)((A.NPL + 63) / 64);
    dim3 grid((unsigned)((A.ntiles + 7) / 8 * 8), nchunks);     // the kernel permutes the tiles over the XCDs
    const int nc = pt->ncomp, p0g = pt->gax[0].P;
#define GEOA_P(PV) case PV: return nslots == 1 ? launch_geoA_g<PV, 1>(st, A, nc, p0g, grid) : launch_geoA_g<PV, 8>(st, A, nc, p0g, grid)
    switch (A0.P) {
        GEOA_P(2); GEOA_P(3); GEOA_P(4); GEOA_P(5); GEOA_P(6);
    }
#undef GEOA_P
    set_error("fused geometry + stage A: unsupported degree %d", A0.p);
    return IGX_ERR_UNSUPPORTED;
}

} // namespace igx

// Fused "sweep + final" stage of the sum-factorised assembly, and the mirror pass.
//
// The unfused chain (sumfact.hip) writes the intermediate of the second contraction, K2[y][r0][r1][g2], to HBM and
// reads it back in the final stage: 31 GB of the 110 GB the C4 chain moved in round 1, and its final stage was
// bound by the write path of scattered 72-byte runs (direct + mirrored).  Here one kernel does both contractions:
//
//   k_bf     block = (outer pair r0 = (i0, j0), chunk of rows of the swept "mid" axis, tile of rows of the last axis)
//            * SWEEPER waves (role = last-axis type y, lane = Gauss point g2 of the tile window) walk the mid axis span by
//              span exactly like k_stageB -- (p+1)x(p+1) active dof pairs in registers, window shifted when a dof
//              leaves -- but take the basis values from SGPRs (rank-1 form PI[t][a][b] = V[b][tu] V[a][tv]: 10 scalar
//              coefficients and 30..70 FMAs per point instead of 25 per term through LDS);
//            * the completed K2 lines of the leaving dof d -- column (d+a, d), row (d, d+a) -- go to LDS, never to HBM;
//            * CONTRACTOR waves (lane = (row i2 of the tile, line)) contract them with the last axis' basis into runs of
//              2p+1 entries and park them in an LDS ring until the whole CSR row SEGMENT
//                  row (i0, i1, i2), columns (j0, j1 = i1-p .. i1+p, j2 = i2-p .. i2+p)   = (2p+1)^2 contiguous doubles
//              is complete; segments are stored whole (648 B at p = 4: the pattern that writes at 4.8 TB/s in
//              profiles/r01_ubench_write_patterns.txt).
//            Only the lower triangle is formed for symmetric forms (pairs j0 <= i0; in a diagonal pair j1 <= i1, ...).
//   k_mirror target-driven transposing copy: upper-triangle segments are gathered from the lower triangle through an LDS
//            tile, again written as whole segments -> exactly symmetric output, as assemble_entries(symmetric=True)
//            (pyiga/assemble.py:742-752).
//
// Semantics follow combine()/entry_impl (pyiga/assemblers.pyx:1455-1540); summation order differs from the reference
// like any sum-factorised order does (parity to rounding, tests/test_gpu_parity.py).
//
// Requirements of k_bf (checked on the host, otherwise the unfused kernels run): mid and last axis have single interior
// knots, the same degree and q = p + 1 Gauss points per span.
#include "igx_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace igx {

typedef const double __attribute__((address_space(4))) *cdp;
typedef const int __attribute__((address_space(4))) *cip;

constexpr int BF_TL = 128;            // Gauss points of the last axis per block = lanes of the sweep (2 waves per role)

template <int P>
struct BFGeom {
    static constexpr int p = P - 1, W = 2 * P - 1;
    static constexpr int SPANS = BF_TL / P;                // spans of a tile window
    static constexpr int RMAX = SPANS - p;                 // rows of the last axis per tile
    static constexpr int NCW = (RMAX * W + 63) / 64;       // contractor waves: one lane per (row, line)
};

struct BFArgs {
    // input arrays of the sweep: In(y, t1, i)[slice][g_mid][g_last]; absent slots point at a row of zeros (strides 0)
    const double *sp[4][4][2];
    long long ss[4][4][2];        // doubles between slices (outer pairs)
    int rs[4][4][2];              // doubles between rows of the mid axis (G_last, or 0 for the zero row)
    int gmid_lo;                  // first resident Gauss index of the mid axis
    int G2;                       // Gauss points of the last axis
    const double *V1, *V2;        // basis tables [G][P][2] of the mid / last axis
    int n1, N1, n2, N2;           // spans / dofs of the mid and the last axis
    const int *rp1, *rp2;         // [N+1] pair prefix sums of the mid / last axis
    const int *pl0;               // [npairs][2] outer pairs (i0, j0)
    const int *rp0, *jlo0, *jhi0; // outer axis tables (a trivial one-dof axis in 2D)
    long long S1, S2, nnz_off;
    double *data;
    int sym;                      // symmetric form: in a diagonal outer pair only the lower triangle is formed
    int R2, ntiles;               // rows per tile of the last axis
    int mrows, nmchunks;          // rows per chunk of the mid axis
    int mid_lo, mid_hi;           // rows of the mid axis to produce
    int span_hi;                  // spans of the mid axis below this one are resident (2D row slabs; else n1)
    int npairs;
};

__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One Gauss point of the sweep for the role whose mid-axis types are MASKY (bit t1 = tu + 2 tv):
//   acc[a][b] += sum_{tu} V[b][tu] * (sum_{tv} V[a][tv] * kt[tu + 2 tv])
template <int P, int MASKY>
__device__ __forceinline__ void sweep_point(double (&acc)[P][P], const double (&kt)[4], const double (&v)[P][2])
{
#pragma unroll
    for (int tu = 0; tu < 2; ++tu) {
        const bool h0 = (MASKY >> tu) & 1, h1 = (MASKY >> (tu + 2)) & 1;
        if (!h0 && !h1) continue;
#pragma unroll
        for (int a = 0; a < P; ++a) {
            double c;
            if (h0 && h1) c = fma(v[a][1], kt[tu + 2], v[a][0] * kt[tu]);
            else if (h0) c = v[a][0] * kt[tu];
            else c = v[a][1] * kt[tu + 2];
#pragma unroll
            for (int b = 0; b < P; ++b) acc[a][b] = fma(v[b][tu], c, acc[a][b]);
        }
    }
}

template <int P, int NY, int MASKY, int NA>
__device__ __forceinline__ void bf_sweeper(const BFArgs &A, const int y, const int r0, const int g2l, const int g2,
                                           const int s_begin, const int rhi, double *lines, const int LS)
{
    constexpr int p = P - 1, TL = BF_TL;
    cdp V1 = (cdp)A.V1;
    double acc[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
    // uniform row pointers of the present slots
    const double *base[4][NA];
    int rs[4][NA];
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            base[t1][i] = A.sp[y][t1][i] + (long long)r0 * A.ss[y][t1][i] - (long long)A.gmid_lo * A.rs[y][t1][i];
            rs[t1][i] = A.rs[y][t1][i];
        }
    const int n_sw = min(A.n1, A.span_hi);           // spans that exist and are resident
    const int t_lastc = min(n_sw, rhi) - 1;          // last span that is swept
    double kv[P][4][NA];                             // values of one span; each is reloaded right after its use
    {
        const int s = min(s_begin, t_lastc);
#pragma unroll
        for (int l = 0; l < P; ++l)
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1)
                if ((MASKY >> t1) & 1)
#pragma unroll
                    for (int i = 0; i < NA; ++i) kv[l][t1][i] = (base[t1][i] + (long long)(s * P + l) * rs[t1][i])[g2];
    }
    for (int t = s_begin; t < rhi + 2; ++t) {
        bar_lds();                                   // B1
        if (t < rhi && t < n_sw) {
            const int tn = min(t + 1, t_lastc);
            cdp cf = V1 + (size_t)t * P * P * 2;
#pragma unroll
            for (int l = 0; l < P; ++l) {
                double v[P][2];
#pragma unroll
                for (int b = 0; b < P; ++b) { v[b][0] = cf[(l * P + b) * 2]; v[b][1] = cf[(l * P + b) * 2 + 1]; }
                double kt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t1 = 0; t1 < 4; ++t1)
                    if ((MASKY >> t1) & 1) {
                        kt[t1] = kv[l][t1][0];
                        if (NA == 2) kt[t1] += kv[l][t1][1];
#pragma unroll
                        for (int i = 0; i < NA; ++i) kv[l][t1][i] = (base[t1][i] + (long long)(tn * P + l) * rs[t1][i])[g2];
                    }
                sweep_point<P, MASKY>(acc, kt, v);
                // keep the points apart: the coefficient loads of point l+1 may not be hoisted above the FMAs of point l
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b < P; ++b) asm volatile("" : "+v"(acc[a][b]));
            }
        }
        bar_lds();                                   // B2: the contractors have read the previous lines
        if (t < rhi) {
            // dof t leaves: column (t+a, t) and row (t, t+a) of the pair window are complete
            double *ln = lines + y * TL + g2l;
#pragma unroll
            for (int a = 0; a < P; ++a) ln[a * LS] = acc[a][0];
#pragma unroll
            for (int a = 1; a < P; ++a) ln[(p + a) * LS] = acc[0][a];
#pragma unroll
            for (int a = 0; a < P - 1; ++a)
#pragma unroll
                for (int b = 0; b < P - 1; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
            for (int a = 0; a < P; ++a) { acc[a][P - 1] = 0.0; acc[P - 1][a] = 0.0; }
        }
    }
}

template <int P, int NY, int MASK, int NA>
__global__ void __launch_bounds__((NY * (BF_TL / 64) + BFGeom<P>::NCW) * 64) k_bf(const BFArgs A)
{
    using Gm = BFGeom<P>;
    constexpr int p = P - 1, W = 2 * P - 1, TL = BF_TL, NLG = TL / 64, NSW = NY * NLG, NCW = Gm::NCW, RMAX = Gm::RMAX;
    constexpr int LS = NY * TL + 2;                       // doubles per line (all types), padded against bank conflicts
    constexpr int OFF_RING = (W * LS + 1) & ~1;
    constexpr int OFF_CUR = (OFF_RING + (P + 1) * p * RMAX * W + 1) & ~1;
    constexpr int OFF_V2 = (OFF_CUR + 2 * P * RMAX * W + 1) & ~1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lines = lds;                 // [W][LS]: lines 0..p = pairs (d+a, d), lines p+a = pairs (d, d+a) of the last flush
    double *ring = lds + OFF_RING;       // [P+1][p][RMAX][W]: entries of the pairs (i1, j1 < i1), row slot i1 mod (P+1)
    double *cur = lds + OFF_CUR;         // [2][P][RMAX][W]:   entries of the pairs (d, d .. d+p), slot d & 1
    double *V2s = lds + OFF_V2;          // [TL][P][2]: last-axis basis values on the tile window

    cip pl0 = (cip)A.pl0, rp0 = (cip)A.rp0, jlo0 = (cip)A.jlo0, jhi0 = (cip)A.jhi0, rp1 = (cip)A.rp1, rp2 = (cip)A.rp2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // XCD-aware order: the tiles of one (pair, chunk) share K1 halo lines -> consecutive logical ids on one XCD
    unsigned bid = blockIdx.x;
    {
        const unsigned per = gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    const int tile = (int)(bid % A.ntiles);
    const int mch = (int)((bid / A.ntiles) % A.nmchunks);
    const int r0 = (int)(bid / ((unsigned)A.ntiles * A.nmchunks));
    const int i0 = pl0[2 * r0], j0 = pl0[2 * r0 + 1];
    const bool diag0 = A.sym && i0 == j0;
    const int row_lo = tile * A.R2, row_hi = min(row_lo + A.R2, A.N2), nrows = row_hi - row_lo;
    const int sp_lo = max(row_lo - p, 0), sp_hi = min(row_hi - 1, A.n2 - 1) + 1;
    const int win0 = sp_lo * P, nwin = (sp_hi - sp_lo) * P;
    const int rlo = A.mid_lo + mch * A.mrows, rhi = min(rlo + A.mrows, A.mid_hi);
    const int s_begin = max(rlo - p, 0);

    for (int idx = threadIdx.x; idx < nwin * P * 2; idx += blockDim.x) V2s[idx] = A.V2[(size_t)win0 * P * 2 + idx];
    // the barrier B1 of the first iteration orders these writes before their first use

    if (wave < NSW) {
        // ---------------- sweepers: role y, lane group lg
        const int y = wave % NY, lg = wave / NY;
        const int g2l = lg * 64 + lane;
        const int g2 = min(win0 + g2l, A.G2 - 1);
        if (NY == 1) bf_sweeper<P, NY, MASK & 15, NA>(A, 0, r0, g2l, g2, s_begin, rhi, lines, LS);
        else {
            switch (y) {
            case 0: bf_sweeper<P, NY, MASK & 15, NA>(A, 0, r0, g2l, g2, s_begin, rhi, lines, LS); break;
            case 1: bf_sweeper<P, NY, (MASK >> 4) & 15, NA>(A, 1, r0, g2l, g2, s_begin, rhi, lines, LS); break;
            case 2: bf_sweeper<P, NY, (MASK >> 8) & 15, NA>(A, 2, r0, g2l, g2, s_begin, rhi, lines, LS); break;
            default: bf_sweeper<P, NY, (MASK >> 12) & 15, NA>(A, 3, r0, g2l, g2, s_begin, rhi, lines, LS); break;
            }
        }
        return;
    }

    // ---------------- contractors: lane = (row r of the tile, line k9)
    const int cw = wave - NSW;
    const int idx = cw * 64 + lane;
    const bool cval = idx < nrows * W;
    const int r = cval ? idx / W : 0, k9 = cval ? idx - r * W : 0;
    const int i2 = row_lo + r;
    const int slo = max(i2 - p, 0), nsp = min(i2, A.n2 - 1) + 1 - slo;
    const int gl0 = (slo - sp_lo) * P;                    // window index of the first Gauss point of row i2's support
    const int la = k9 <= p ? k9 : k9 - p;                 // line k9: pair (d + la, d) for k9 <= p, else (d, d + la)
    const int c0i = jhi0[i0] - jlo0[i0];
    const long long S12 = A.S1 * A.S2;

    for (int t = s_begin; t < rhi + 2; ++t) {
        bar_lds();                                        // B1: the lines of flush t-1 are in LDS
        // ---- whole segments of row dd2 = t - 2 (its entries were completed in the previous iteration)
        const int dd2 = t - 2;
        if (dd2 >= rlo && dd2 < rhi) {
            const int jl1 = max(dd2 - p, 0), c1 = min(dd2 + p, A.N1 - 1) + 1 - jl1;
            const double *rg = ring + (size_t)((dd2 % (P + 1)) * p) * RMAX * W;
            const double *cu = cur + (size_t)((dd2 & 1) * P) * RMAX * W;
            for (int rr = cw; rr < nrows; rr += NCW) {
                const int i2r = row_lo + rr;
                const int jl2 = max(i2r - p, 0), c2r = min(i2r + p, A.N2 - 1) + 1 - jl2;
                const int ne = diag0 ? (dd2 - jl1) * c2r + (i2r - jl2 + 1) : c1 * c2r;
                double *dst = A.data + ((long long)rp0[i0] * S12 + (long long)c0i * ((long long)rp1[dd2] * A.S2 + (long long)c1 * rp2[i2r])
                                        - A.nnz_off + (long long)(j0 - jlo0[i0]) * c1 * c2r);
                for (int e = lane; e < ne; e += 64) {
                    const int m = e / c2r, o = e - m * c2r;
                    const int j1 = jl1 + m;
                    const double v = (j1 < dd2) ? rg[((p - (dd2 - j1)) * RMAX + rr) * W + o] : cu[((j1 - dd2) * RMAX + rr) * W + o];
                    dst[e] = v;
                }
            }
        }
        // ---- contract the lines of flush dd = t - 1 with the last axis
        const int dd = t - 1;
        if (dd >= s_begin && dd < rhi) {
            const int row1 = k9 <= p ? dd + la : dd, col1 = k9 <= p ? dd : dd + la;
            const bool lv = cval && row1 >= rlo && row1 < rhi && col1 < A.N1 && !(diag0 && col1 > row1);
            if (lv) {
                double accv[W];
#pragma unroll
                for (int o = 0; o < W; ++o) accv[o] = 0.0;
                const double *ln = lines + k9 * LS + gl0;
                const double *vs = V2s + gl0 * P * 2;
#pragma unroll
                for (int kk = 0; kk < P; ++kk) {
                    if (kk < nsp) {
                        const int a2 = i2 - (slo + kk);   // local index of the test function in this span
#pragma unroll
                        for (int l = 0; l < P; ++l) {
                            const int g = kk * P + l;
                            const double va0 = vs[(g * P + a2) * 2], va1 = vs[(g * P + a2) * 2 + 1];
                            double cu0, cu1 = 0.0;
                            if (NY == 1) cu0 = va0 * ln[g];
                            else {
                                cu0 = fma(va1, ln[2 * TL + g], va0 * ln[g]);           // types 0, 2
                                cu1 = fma(va1, ln[3 * TL + g], va0 * ln[TL + g]);      // types 1, 3
                            }
#pragma unroll
                            for (int b = 0; b < P; ++b) {
                                if (NY == 1) accv[kk + b] = fma(vs[(g * P + b) * 2], cu0, accv[kk + b]);
                                else accv[kk + b] = fma(vs[(g * P + b) * 2], cu0, fma(vs[(g * P + b) * 2 + 1], cu1, accv[kk + b]));
                            }
                        }
                    }
                }
                double *dste = (col1 < row1) ? ring + (size_t)((((row1 % (P + 1)) * p + (p - la)) * RMAX + r)) * W
                                             : cur + (size_t)((((dd & 1) * P + (col1 - row1)) * RMAX + r)) * W;
#pragma unroll
                for (int o = 0; o < W; ++o) dste[o] = accv[o];
            }
        }
        bar_lds();                                        // B2: lines may be overwritten, entries are visible
    }
}

// ---------------------------------------------------------------------------------------------
// Mirror pass: upper-triangle entries from the lower triangle.  Block = (target outer pair (i0, j0 >= i0), chunk of
// rows i2 of the last axis); it walks the rows i1 of the mid axis.  Per step the source runs
//     row (j0, j1, j2), columns (i0, i1, i2' in chunk)        for the j1 of the segment and j2 around the chunk
// are gathered into an LDS tile (reads: 72-byte runs whose neighbours in memory are the runs of the next steps, served
// by the XCD's L2), and the target segments  row (i0, i1, i2), columns (j0, j1, j2)  are written whole.
struct MirrorArgs {
    double *data;
    long long nnz_off, S1, S2;
    const int *rp0, *jlo0, *jhi0, *rp1, *jlo1, *jhi1, *rp2, *jlo2, *jhi2;
    int N1, N2;
    int i1_lo, i1_hi;            // target rows of the mid axis
    const int *tpairs;           // [ntp][2] target pairs
    int ntp, RC, nchunks;
    int W2max, NJ2max;           // LDS tile geometry
};

__global__ void __launch_bounds__(256) k_mirror(const MirrorArgs M)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    cip rp0 = (cip)M.rp0, jlo0 = (cip)M.jlo0, jhi0 = (cip)M.jhi0, rp1 = (cip)M.rp1, jlo1 = (cip)M.jlo1, jhi1 = (cip)M.jhi1;
    int *t_jlo2 = (int *)lds, *t_c2 = t_jlo2 + M.N2, *t_rp2 = t_c2 + M.N2;
    double *T = lds + ((3 * M.N2 + 1) / 2 + 1);
    for (int i = threadIdx.x; i < M.N2; i += blockDim.x) { t_jlo2[i] = M.jlo2[i]; t_c2[i] = M.jhi2[i] - M.jlo2[i]; t_rp2[i] = M.rp2[i]; }
    __syncthreads();
    unsigned bid = blockIdx.x;
    {
        const unsigned per = gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    const int chunk = (int)(bid % M.nchunks), tp = (int)(bid / M.nchunks);
    const int i0 = ((cip)M.tpairs)[2 * tp], j0 = ((cip)M.tpairs)[2 * tp + 1];
    const bool diag = i0 == j0;
    const int c0i = jhi0[i0] - jlo0[i0], c0j = jhi0[j0] - jlo0[j0];
    const int cl = chunk * M.RC, ch = min(cl + M.RC, M.N2);
    const int j2lo = t_jlo2[cl], j2hi = t_jlo2[ch - 1] + t_c2[ch - 1], nj2 = j2hi - j2lo;
    const int W2 = M.W2max, NJ2 = M.NJ2max;
    const long long S12 = M.S1 * M.S2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (int i1 = M.i1_lo; i1 < M.i1_hi; ++i1) {
        const int jl1i = jlo1[i1], c1i = jhi1[i1] - jl1i;
        const int a1 = diag ? i1 : jl1i, b1 = jhi1[i1], nj1 = b1 - a1;
        // ---- gather: T[j1 - a1][j2 - j2lo][i2' - jlo2[j2]]
        const int total = nj1 * nj2 * W2;
        for (int f = threadIdx.x; f < total; f += blockDim.x) {
            const int op = f % W2, rest = f / W2;
            const int j2r = rest % nj2, j1i = rest / nj2;
            const int j1 = a1 + j1i, j2 = j2lo + j2r;
            const int jl2 = t_jlo2[j2], c2 = t_c2[j2];
            const int i2p = jl2 + op;
            double v = 0.0;
            if (op < c2 && i2p >= cl && i2p < ch && !(diag && j1 == i1 && j2 <= i2p)) {
                const int c1j = jhi1[j1] - jlo1[j1];
                const long long src = (long long)rp0[j0] * S12 + (long long)c0j * ((long long)rp1[j1] * M.S2 + (long long)c1j * t_rp2[j2])
                                      + ((long long)(i0 - jlo0[j0]) * c1j + (i1 - jlo1[j1])) * c2 + op - M.nnz_off;
                v = M.data[src];
            }
            T[(j1i * NJ2 + j2r) * W2 + op] = v;
        }
        __syncthreads();
        // ---- whole target segments
        for (int i2 = cl + wave; i2 < ch; i2 += nwaves) {
            const int jl2 = t_jlo2[i2], c2 = t_c2[i2];
            const long long dst0 = (long long)rp0[i0] * S12 + (long long)c0i * ((long long)rp1[i1] * M.S2 + (long long)c1i * t_rp2[i2])
                                   + ((long long)(j0 - jlo0[i0]) * c1i + (a1 - jl1i)) * c2 - M.nnz_off;
            const int ne = nj1 * c2;
            for (int e = lane; e < ne; e += 64) {
                const int j1i = e / c2, o = e - j1i * c2;
                const int j2 = jl2 + o;
                if (diag && j1i == 0 && j2 <= i2) continue;
                M.data[dst0 + e] = T[(j1i * NJ2 + (j2 - j2lo)) * W2 + (i2 - t_jlo2[j2])];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host side
template <int P, int NY, int MASK, int NA>
static int launch_bf_k(hipStream_t st, const BFArgs &A, unsigned nblocks)
{
    using Gm = BFGeom<P>;
    constexpr int W = 2 * P - 1, p = P - 1, LS = NY * BF_TL + 2;
    constexpr int OFF_RING = (W * LS + 1) & ~1;
    constexpr int OFF_CUR = (OFF_RING + (P + 1) * p * Gm::RMAX * W + 1) & ~1;
    constexpr int OFF_V2 = (OFF_CUR + 2 * P * Gm::RMAX * W + 1) & ~1;
    constexpr size_t lds = (size_t)(OFF_V2 + BF_TL * P * 2) * sizeof(double);
    static_assert(lds <= 160 * 1024, "k_bf: LDS");
    IGX_HIP(hipFuncSetAttribute((const void *)k_bf<P, NY, MASK, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_bf<P, NY, MASK, NA><<<dim3(nblocks), dim3((NY * (BF_TL / 64) + Gm::NCW) * 64), lds, st>>>(A);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

constexpr int BF_MASK_MASS = 0x0001, BF_MASK_STIFF3 = 0x135F, BF_MASK_STIFF2 = 0x1248, BF_MASK_ALL = 0xFFFF;

template <int P>
static int launch_bf_p(hipStream_t st, const BFArgs &A, unsigned nblocks, int ny, int mask, int na)
{
    if (ny == 1 && mask == BF_MASK_MASS && na == 1) return launch_bf_k<P, 1, BF_MASK_MASS, 1>(st, A, nblocks);
    if (ny == 4 && mask == BF_MASK_STIFF3 && na == 1) return launch_bf_k<P, 4, BF_MASK_STIFF3, 1>(st, A, nblocks);
    if (ny == 4 && mask == BF_MASK_STIFF3 && na == 2) return launch_bf_k<P, 4, BF_MASK_STIFF3, 2>(st, A, nblocks);
    if (ny == 4 && mask == BF_MASK_STIFF2 && na == 1) return launch_bf_k<P, 4, BF_MASK_STIFF2, 1>(st, A, nblocks);
    return launch_bf_k<P, 4, BF_MASK_ALL, 2>(st, A, nblocks);
}

int fused_rows_per_tile(int P) { return BF_TL / P - (P - 1); }

// slots[y][t1]: input arrays (device pointers) of the sweep with their strides; see BFArgs
int launch_bf(hipStream_t st, const igx_patch *pt, const BFInputs &in, double *d_data)
{
    const Axis &AM = *in.mid, &AL = *in.last;
    BFArgs A{};
    int ymax = 0, mask = 0, na = 1;
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1) {
            const int n = in.slot_n[y][t1];
            if (n > 2) { set_error("fused stage: more than two arrays per slot"); return IGX_ERR_UNSUPPORTED; }
            if (n > 0) { mask |= 1 << (4 * y + t1); ymax = std::max(ymax, y); na = std::max(na, n); }
            for (int i = 0; i < 2; ++i) {
                if (i < n) { A.sp[y][t1][i] = in.slot_ptr[y][t1][i]; A.ss[y][t1][i] = in.slice_stride; A.rs[y][t1][i] = AL.G; }
                else { A.sp[y][t1][i] = in.zeros; A.ss[y][t1][i] = 0; A.rs[y][t1][i] = 0; }
            }
        }
    const int ny = ymax == 0 ? 1 : 4;
    A.gmid_lo = in.gmid_lo; A.G2 = AL.G;
    A.V1 = AM.d_V; A.V2 = AL.d_V;
    A.n1 = AM.n; A.N1 = AM.N; A.n2 = AL.n; A.N2 = AL.N;
    A.rp1 = AM.dev.rp; A.rp2 = AL.dev.rp;
    A.pl0 = in.pl0; A.rp0 = in.rp0; A.jlo0 = in.jlo0; A.jhi0 = in.jhi0;
    A.S1 = AM.S; A.S2 = AL.S; A.nnz_off = pt->nnz_off;
    A.data = d_data; A.sym = in.sym;
    const int P = AL.P;
    const int rmax = fused_rows_per_tile(P);
    A.ntiles = (AL.N + rmax - 1) / rmax;
    A.R2 = (AL.N + A.ntiles - 1) / A.ntiles;
    A.mid_lo = in.mid_lo; A.mid_hi = in.mid_hi; A.span_hi = in.span_hi;
    // chunks of the mid axis: enough blocks to fill the chip (each chunk re-sweeps p warm-up spans)
    const int mid_rows = in.mid_hi - in.mid_lo;
    long long blocks = (long long)in.npairs * A.ntiles;
    int nmch = 1;
    if (blocks < 1024) nmch = (int)std::min<long long>((1024 + blocks - 1) / blocks, std::max(1, mid_rows / (2 * P)));
    if (const char *e = getenv("IGX_BF_MCHUNKS")) nmch = std::max(1, std::min(mid_rows, atoi(e)));
    A.mrows = (mid_rows + nmch - 1) / nmch;
    A.nmchunks = (mid_rows + A.mrows - 1) / A.mrows;
    A.npairs = in.npairs;
    blocks = (long long)in.npairs * A.ntiles * A.nmchunks;
    if (blocks > 0x7fffffffLL) { set_error("fused stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
    if (blocks == 0) return IGX_OK;
    const unsigned nb = (unsigned)blocks;
    switch (P) {
    case 2: return launch_bf_p<2>(st, A, nb, ny, mask, na);
    case 3: return launch_bf_p<3>(st, A, nb, ny, mask, na);
    case 4: return launch_bf_p<4>(st, A, nb, ny, mask, na);
    case 5: return launch_bf_p<5>(st, A, nb, ny, mask, na);
    case 6: return launch_bf_p<6>(st, A, nb, ny, mask, na);
    default: set_error("fused stage: degree %d unsupported", P - 1); return IGX_ERR_UNSUPPORTED;
    }
}

int launch_mirror(hipStream_t st, const igx_patch *pt, const MirrorInputs &in, double *d_data)
{
    if (in.ntp == 0) return IGX_OK;
    const Axis &AM = *in.mid, &AL = *in.last;
    MirrorArgs M{};
    M.data = d_data; M.nnz_off = pt->nnz_off; M.S1 = AM.S; M.S2 = AL.S;
    M.rp0 = in.rp0; M.jlo0 = in.jlo0; M.jhi0 = in.jhi0;
    M.rp1 = AM.dev.rp; M.jlo1 = AM.dev.jlo; M.jhi1 = AM.dev.jhi;
    M.rp2 = AL.dev.rp; M.jlo2 = AL.dev.jlo; M.jhi2 = AL.dev.jhi;
    M.N1 = AM.N; M.N2 = AL.N; M.i1_lo = in.i1_lo; M.i1_hi = in.i1_hi;
    M.tpairs = in.tpairs; M.ntp = in.ntp;
    int w1 = 0, w2 = 0;
    for (int i = 0; i < AM.N; ++i) w1 = std::max(w1, AM.jhi[i] - AM.jlo[i]);
    for (int i = 0; i < AL.N; ++i) w2 = std::max(w2, AL.jhi[i] - AL.jlo[i]);
    // rows of the last axis per block: the whole axis when its tile fits ~48 KB of LDS, else equal chunks
    int nch = 1, RC, nj2;
    for (;; ++nch) {
        RC = (AL.N + nch - 1) / nch;
        nj2 = 0;
        for (int cl = 0; cl < AL.N; cl += RC) nj2 = std::max(nj2, AL.jhi[std::min(cl + RC, AL.N) - 1] - AL.jlo[cl]);
        if ((size_t)w1 * nj2 * w2 * sizeof(double) <= 48 * 1024 || RC <= 8) break;
    }
    // a launch wants >= ~1024 blocks
    while ((long long)in.ntp * nch < 1024 && RC > 16) {
        ++nch;
        RC = (AL.N + nch - 1) / nch;
        nj2 = 0;
        for (int cl = 0; cl < AL.N; cl += RC) nj2 = std::max(nj2, AL.jhi[std::min(cl + RC, AL.N) - 1] - AL.jlo[cl]);
    }
    M.RC = RC; M.nchunks = (AL.N + RC - 1) / RC; M.W2max = w2; M.NJ2max = nj2;
    const size_t lds = ((size_t)(3 * AL.N + 1) / 2 + 1 + (size_t)w1 * nj2 * w2) * sizeof(double);
    if (lds > 160 * 1024) { set_error("mirror pass: tile does not fit LDS"); return IGX_ERR_UNSUPPORTED; }
    const long long blocks = (long long)in.ntp * M.nchunks;
    if (blocks > 0x7fffffffLL) { set_error("mirror pass: too many blocks"); return IGX_ERR_UNSUPPORTED; }
    IGX_HIP(hipFuncSetAttribute((const void *)k_mirror, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_mirror<<<dim3((unsigned)blocks), dim3(256), lds, st>>>(M);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

} // namespace igx

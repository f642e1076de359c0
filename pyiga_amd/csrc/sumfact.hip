// Global sum factorisation of the tensor-product quadrature (the fast path).
//
// The reference sums every matrix entry over the full tensor Gauss grid of its support
// (combine, pyiga/assemblers.pyx:1455-1494): cost ~ ((p+1) q)^d per entry.  Because basis
// functions, Gauss grid and sparsity pattern are all tensor products, the same sum factorises
// over the WHOLE PATCH:
//
//   A[(i0,i1,i2),(j0,j1,j2)] = sum_terms sum_g2 PI2[t2][i2,j2,g2]
//                                        sum_g1 PI1[t1][i1,j1,g1]
//                                        sum_g0 PI0[t0][i0,j0,g0] * field_f[g0,g1,g2]
//
// with PIk[t][i,j,g] = (d^tu phi_j)(g) * (d^tv phi_i)(g) and one term per (derivative of u,
// derivative of v) pair of the bilinear form (mass: 1 term, stiffness: d*d terms).  Each stage
// contracts one grid axis against a banded 1D table:
//
//   stage A  (axis 0):  K1[x][r0][g1,g2]   = sum_g0 PI0 * field        one sweep over g0
//   stage B  (axis 1):  K2[y][r0][r1][g2]  = sum_terms sum_g1 PI1 * K1 one sweep over g1   (3D only)
//   final    (last):    A[r0][r1][i,j]     = sum_y sum_g PI_last * K   -> CSR values (+ mirror)
//
// r_k enumerates the 1D dof pairs (i_k, j_k) with overlapping support.  Only the lower triangle
// is formed (pairs j0 <= i0 on axis 0) and the strict lower part is mirrored, exactly like
// assemble_entries(symmetric=True) (pyiga/assemble.py:742-752).  Work drops from
// O(N p^{2d} q^d)... to O(N p^{d+2}); every stage is a streaming kernel bound by HBM.
//
// The sweeps keep the (p+1)x(p+1) active dof pairs of the current span in registers and shift
// the window when the active set changes (by the knot multiplicity), so repeated interior knots
// are handled; each K1/K2 value is written exactly once, no atomics.
#include "igx_internal.h"
#include <algorithm>
#include <cstdio>

namespace igx {

struct Term { int f; int t[3]; };               // field index, type per axis (t = tu + 2*tv)

static int sym_index(int d, int r, int c)       // row-major upper triangle (pyiga/vform.py:28-34)
{
    if (r > c) std::swap(r, c);
    int idx = 0;
    for (int rr = 0; rr < r; ++rr) idx += d - rr;
    return idx + (c - r);
}

static std::vector<Term> form_terms(int dim, int kind)
{
    std::vector<Term> T;
    if (kind == IGX_MASS) {
        T.push_back(Term{0, {0, 0, 0}});
        return T;
    }
    // stiffness: du^T B dv; gradient component c (x,y,z order) differentiates grid axis dim-1-c
    for (int a = 0; a < dim; ++a)          // axis carrying the derivative of u (trial, column j)
        for (int b = 0; b < dim; ++b) {    // axis carrying the derivative of v (test, row i)
            Term t{};
            t.f = sym_index(dim, dim - 1 - a, dim - 1 - b);
            for (int k = 0; k < 3; ++k) t.t[k] = (k < dim) ? ((k == a) ? 1 : 0) + 2 * ((k == b) ? 1 : 0) : 0;
            T.push_back(t);
        }
    return T;
}

// ---------------------------------------------------------------------------------------------
// Stage A: sweep axis 0.  One thread per point of the remaining grid (g1[,g2]), NT types of one field.
struct StageAArgs {
    const double *field;        // [G0_loc][NPL]
    double *out[2];             // K1 arrays [npairs0][NPL]
    int t[2];
    const double *PI0;          // [G0][4][P][P]
    const int *fa0, *rp0, *jlo0, *rl0_of;
    int s_lo, s_hi, n0, N0, q, g0_lo;
    long long NPL;
};

template <int P, int NT>
__global__ void __launch_bounds__(256) k_stageA(StageAArgs A)
{
    long long pt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pt < A.NPL;
    if (!live) pt = A.NPL - 1;
    double acc[NT][P][P];
#pragma unroll
    for (int ty = 0; ty < NT; ++ty)
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) acc[ty][a][b] = 0.0;

    for (int s = A.s_lo; s < A.s_hi; ++s) {
        for (int l = 0; l < A.q; ++l) {
            const int g = s * A.q + l;
            const double bv = A.field[(long long)(g - A.g0_lo) * A.NPL + pt];
            const double *pi = A.PI0 + (size_t)g * 4 * P * P;
#pragma unroll
            for (int ty = 0; ty < NT; ++ty) {
                const double *pt_ = pi + A.t[ty] * P * P;
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[ty][a][b] = fma(pt_[a * P + b], bv, acc[ty][a][b]);
            }
        }
        const int base = A.fa0[s];
        const int m = (s + 1 < A.s_hi && s + 1 < A.n0) ? (A.fa0[s + 1] - base) : P;
        for (int k = 0; k < m; ++k) {
            const int j0 = base + k;
            // dof j0 leaves the active set: its pairs (i0 = j0 + a, j0) are complete
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const int i0 = j0 + a;
                if (a <= P - 1 - k && i0 < A.N0) {
                    const int r = A.rl0_of[A.rp0[i0] + (j0 - A.jlo0[i0])];
                    if (r >= 0 && live) {
#pragma unroll
                        for (int ty = 0; ty < NT; ++ty) A.out[ty][(long long)r * A.NPL + pt] = acc[ty][a][0];
                    }
                }
            }
#pragma unroll
            for (int ty = 0; ty < NT; ++ty) {
#pragma unroll
                for (int a = 0; a < P - 1; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[ty][a][b] = acc[ty][a + 1][b + 1];
#pragma unroll
                for (int b = 0; b < P; ++b) acc[ty][P - 1][b] = 0.0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stage B (3D): sweep axis 1.  Block = (chunk of g2, processed pair r0, output group y).
struct StageBGroup {
    int nterm;
    int x[4];                   // K1 array index of each term
    int t1[4];                  // axis-1 type of each term
};
struct StageBArgs {
    const double *K1;           // [nX][npairs0][G1][G2]
    double *K2;                 // [nY][npairs0][S1][G2]
    StageBGroup grp[4];
    const double *PI1;          // [G1][4][P][P]
    const int *fa1, *rp1, *jlo1;
    const int *pl0;             // [npairs0][2]
    int n1, N1, q, G1, G2, S1, npairs0;
};

template <int P>
__global__ void __launch_bounds__(256) k_stageB(StageBArgs B)
{
    int g2 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g2 < B.G2;
    if (!live) g2 = B.G2 - 1;
    const int r0 = blockIdx.y;
    const int y = blockIdx.z;
    const StageBGroup &G = B.grp[y];
    const bool diag0 = B.pl0[2 * r0] == B.pl0[2 * r0 + 1];
    const long long plane = (long long)B.G1 * B.G2;
    const double *k1base[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        k1base[t] = B.K1 + ((long long)(t < G.nterm ? G.x[t] : 0) * B.npairs0 + r0) * plane + g2;
    double *out = B.K2 + ((long long)y * B.npairs0 + r0) * B.S1 * B.G2 + g2;

    double acc[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) acc[a][b] = 0.0;

    for (int s = 0; s < B.n1; ++s) {
        for (int l = 0; l < B.q; ++l) {
            const int g1 = s * B.q + l;
            const double *pi = B.PI1 + (size_t)g1 * 4 * P * P;
            for (int t = 0; t < G.nterm; ++t) {
                const double kv = k1base[t][(long long)g1 * B.G2];
                const double *pt_ = pi + G.t1[t] * P * P;
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b < P; ++b) acc[a][b] = fma(pt_[a * P + b], kv, acc[a][b]);
            }
        }
        const int base = B.fa1[s];
        const int m = (s + 1 < B.n1) ? (B.fa1[s + 1] - base) : P;
        for (int k = 0; k < m; ++k) {
            const int d = base + k;              // dof leaving the active set
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const int o = d + a;             // partner dof
                if (a <= P - 1 - k && o < B.N1 && live) {
                    // pair (i1 = o, j1 = d): lower or diagonal
                    out[(long long)(B.rp1[o] + (d - B.jlo1[o])) * B.G2] = acc[a][0];
                    // pair (i1 = d, j1 = o): strictly upper; not needed when (i0,j0) is diagonal
                    if (a > 0 && !diag0) out[(long long)(B.rp1[d] + (o - B.jlo1[d])) * B.G2] = acc[0][a];
                }
            }
#pragma unroll
            for (int a = 0; a < P - 1; ++a)
#pragma unroll
                for (int b = 0; b < P - 1; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
            for (int a = 0; a < P; ++a) { acc[a][P - 1] = 0.0; acc[P - 1][a] = 0.0; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Final stage: contract the last (contiguous) grid axis and write CSR values + mirror.
// Block = (row tile of the last axis, line).  A "line" is one K vector of length G_last:
//   3D: line = (r0, r1) -> K2[y][r0][r1][:],   2D: line = r0 -> K1[x][r0][:].
struct FinalArgs {
    const double *K;            // [NY][nlines][G]
    double *data;               // CSR values of the owned rows
    const double *V;            // last axis [G][P][2]
    const int *fa, *mslo, *mshi, *jlo, *jhi, *rp;   // last axis tables
    int N, q, G, TR;            // last axis dofs, q, Gauss count, rows per tile
    long long nlines;
    // leading axes
    int dim;
    const int *pl0;             // [npairs0][2]
    const int *rp0, *jlo0, *jhi0;
    const int *pair1_i, *pair1_j, *rp1, *jlo1, *jhi1;   // axis 1 (3D)
    int S1;                     // pairs of axis 1 (3D), 1 in 2D
    long long Smid, Slast;      // 3D: S1, S2 ; 2D: unused, S1
    int r0_lo, r0_hi;
    long long nnz_off;
    int seg_max;                // LDS segment capacity (Gauss points)
    int ntiles;                 // row tiles per line
};

template <int P, int D>
__device__ inline void acc_add(double (&acc)[2 * P - 1], const double *vs, double cu0, double cu1)
{
#pragma unroll
    for (int b = 0; b < P; ++b)
        if (D + b < 2 * P - 1) acc[D + b] = fma(vs[2 * b], cu0, fma(vs[2 * b + 1], cu1, acc[D + b]));
}

template <int P, int D>
struct AccSwitch {
    __device__ static inline void run(int d, double (&acc)[2 * P - 1], const double *vs, double cu0, double cu1)
    {
        if (d == D) acc_add<P, D>(acc, vs, cu0, cu1);
        else AccSwitch<P, D + 1>::run(d, acc, vs, cu0, cu1);
    }
};
template <int P>
struct AccSwitch<P, 2 * P - 1> {
    __device__ static inline void run(int, double (&)[2 * P - 1], const double *, double, double) {}
};

// contribution of the K-th span of the support of row i (compile-time K so that the accumulator
// index d + b is static in the common case d == K)
template <int P, int NY, int K>
__device__ inline void final_span(const FinalArgs &F, const double *Ks, const double *Vs, int seg_lo,
                                  int i, int slo, int jl, double (&acc)[2 * P - 1])
{
    const int s = slo + K;
    const int fa_s = F.fa[s];
    const int a = i - fa_s;                       // local index of the test function
    const int d = fa_s - jl;                      // output offset of local trial function 0
    for (int l = 0; l < F.q; ++l) {
        const int gl = s * F.q + l - seg_lo;
        const double *vs = Vs + (size_t)gl * P * 2;
        const double v0 = vs[2 * a], v1 = vs[2 * a + 1];
        double cu0, cu1;
        if (NY == 1) { cu0 = v0 * Ks[gl]; cu1 = 0.0; }
        else {
            // K arrays are ordered by type t = tu + 2*tv of the last axis
            cu0 = fma(v1, Ks[2 * F.seg_max + gl], v0 * Ks[gl]);
            cu1 = fma(v1, Ks[3 * F.seg_max + gl], v0 * Ks[F.seg_max + gl]);
        }
        if (d == K) acc_add<P, K>(acc, vs, cu0, cu1);
        else AccSwitch<P, 0>::run(d, acc, vs, cu0, cu1);
    }
}

template <int P, int NY, int K>
struct SpanLoop {
    __device__ static inline void run(const FinalArgs &F, const double *Ks, const double *Vs, int seg_lo,
                                      int i, int slo, int nsp, int jl, double (&acc)[2 * P - 1])
    {
        if (K < nsp) final_span<P, NY, K>(F, Ks, Vs, seg_lo, i, slo, jl, acc);
        SpanLoop<P, NY, K + 1>::run(F, Ks, Vs, seg_lo, i, slo, nsp, jl, acc);
    }
};
template <int P, int NY>
struct SpanLoop<P, NY, P> {
    __device__ static inline void run(const FinalArgs &, const double *, const double *, int, int, int, int, int,
                                      double (&)[2 * P - 1]) {}
};

template <int P, int NY>
__global__ void __launch_bounds__(256) k_final(FinalArgs F)
{
    extern __shared__ double lds[];
    double *Ks = lds;                                   // [NY][seg_max]
    double *Vs = lds + (size_t)NY * F.seg_max;          // [seg_max][P][2]

    const long long line = blockIdx.x / F.ntiles;
    const int tile_lo = (int)(blockIdx.x % F.ntiles) * F.TR;
    const int tile_hi = min(tile_lo + F.TR, F.N);

    // ---- which (i0,j0[,i1,j1]) is this line?
    int r0, i1 = 0, j1 = 0;
    if (F.dim == 3) { r0 = (int)(line / F.S1); const int r1 = (int)(line % F.S1); i1 = F.pair1_i[r1]; j1 = F.pair1_j[r1]; }
    else r0 = (int)line;
    const int i0 = F.pl0[2 * r0], j0 = F.pl0[2 * r0 + 1];
    bool diag_lead = (i0 == j0);
    if (F.dim == 3) {
        if (diag_lead && j1 > i1) return;               // upper part of a diagonal block: mirrored, not computed
        diag_lead = diag_lead && (i1 == j1);
    }
    const bool own_row = i0 >= F.r0_lo && i0 < F.r0_hi;
    const bool own_col = j0 >= F.r0_lo && j0 < F.r0_hi;

    // position coefficients: pos = A + B*rp[i] + C*c[i] + o   
    long long A_d, B_d, C_d, A_m, B_m, C_m;
    {
        const int c0i = F.jhi0[i0] - F.jlo0[i0], c0j = F.jhi0[j0] - F.jlo0[j0];
        if (F.dim == 3) {
            const int c1i = F.jhi1[i1] - F.jlo1[i1], c1j = F.jhi1[j1] - F.jlo1[j1];
            A_d = (long long)F.rp0[i0] * F.Smid * F.Slast + (long long)c0i * F.rp1[i1] * F.Slast - F.nnz_off;
            B_d = (long long)c0i * c1i;
            C_d = (long long)(j0 - F.jlo0[i0]) * c1i + (j1 - F.jlo1[i1]);
            A_m = (long long)F.rp0[j0] * F.Smid * F.Slast + (long long)c0j * F.rp1[j1] * F.Slast - F.nnz_off;
            B_m = (long long)c0j * c1j;
            C_m = (long long)(i0 - F.jlo0[j0]) * c1j + (i1 - F.jlo1[j1]);
        } else {
            A_d = (long long)F.rp0[i0] * F.Slast - F.nnz_off;  B_d = c0i;  C_d = j0 - F.jlo0[i0];
            A_m = (long long)F.rp0[j0] * F.Slast - F.nnz_off;  B_m = c0j;  C_m = i0 - F.jlo0[j0];
        }
    }

    // ---- stage the needed segment of the K line(s) and of the basis table in LDS
    const int seg_lo = F.mslo[tile_lo] * F.q;
    const int seg_hi = F.mshi[tile_hi - 1] * F.q;
    const int seglen = seg_hi - seg_lo;
    for (int idx = threadIdx.x; idx < seglen; idx += blockDim.x) {
#pragma unroll
        for (int y = 0; y < NY; ++y) Ks[y * F.seg_max + idx] = F.K[((long long)y * F.nlines + line) * F.G + seg_lo + idx];
    }
    {
        const double *vsrc = F.V + (size_t)seg_lo * P * 2;
        for (int idx = threadIdx.x; idx < seglen * P * 2; idx += blockDim.x) Vs[idx] = vsrc[idx];
    }
    __syncthreads();

    const int i = tile_lo + threadIdx.x;
    if (i >= tile_hi) return;
    double acc[2 * P - 1];
#pragma unroll
    for (int o = 0; o < 2 * P - 1; ++o) acc[o] = 0.0;
    const int slo = F.mslo[i], nsp = F.mshi[i] - slo, jl = F.jlo[i];
    SpanLoop<P, NY, 0>::run(F, Ks, Vs, seg_lo, i, slo, nsp, jl, acc);

    // ---- write: direct entries (row I, cols j) and mirrored entries (row J, col I)
    const int ci = F.jhi[i] - jl;
    const long long pd = A_d + B_d * F.rp[i] + C_d * ci;
#pragma unroll
    for (int o = 0; o < 2 * P - 1; ++o) {
        if (o < ci) {
            const int j = jl + o;
            if (diag_lead && j > i) continue;             // computed as the mirror of (j, i)
            const double v = acc[o];
            if (own_row) F.data[pd + o] = v;
            if (own_col && !(diag_lead && j == i)) {
                const int cj = F.jhi[j] - F.jlo[j];
                F.data[A_m + B_m * F.rp[j] + C_m * cj + (i - F.jlo[j])] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
int sumfact_supported(const igx_patch *pt)
{
    for (int k = 0; k < pt->dim; ++k)
        if (pt->ax[k].p < 1 || pt->ax[k].p > IGX_MAX_SF_DEGREE) return 0;
    return 1;
}

int sumfact_prepare(igx_patch *pt)
{
    // processed lower pairs of axis 0: j0 <= i0 with the row or the column owned
    const Axis &A0 = pt->ax[0];
    std::vector<int> pl, rl(A0.S, -1);
    for (int i0 = 0; i0 < A0.N; ++i0)
        for (int j0 = A0.jlo[i0]; j0 <= i0; ++j0) {
            const bool own_r = i0 >= pt->r0_lo && i0 < pt->r0_hi;
            const bool own_c = j0 >= pt->r0_lo && j0 < pt->r0_hi;
            if (!own_r && !own_c) continue;
            rl[A0.rp[i0] + (j0 - A0.jlo[i0])] = (int)(pl.size() / 2);
            pl.push_back(i0);
            pl.push_back(j0);
        }
    pt->npairs0 = (int)(pl.size() / 2);
    IGX_HIP(hipMalloc(&pt->d_pl0, std::max<size_t>(1, pl.size()) * sizeof(int)));
    IGX_HIP(hipMalloc(&pt->d_rl0_of, std::max<size_t>(1, rl.size()) * sizeof(int)));
    IGX_HIP(hipMemcpyAsync(pt->d_pl0, pl.data(), pl.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipMemcpyAsync(pt->d_rl0_of, rl.data(), rl.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    return IGX_OK;
}

static int ensure(double **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return IGX_OK;
    if (*buf) { hipFree(*buf); *buf = nullptr; *cap = 0; }
    hipError_t e = hipMalloc(buf, need * sizeof(double));
    if (e != hipSuccess) {
        set_error("hipMalloc of %.2f GB sum-factorisation workspace failed: %s", need * 8.0 / 1e9, hipGetErrorString(e));
        return IGX_ERR_NOMEM;
    }
    *cap = need;
    return IGX_OK;
}

template <int P>
static void launch_stageA(hipStream_t st, const StageAArgs &A, int nt, dim3 grid, dim3 block)
{
    if (nt == 1) k_stageA<P, 1><<<grid, block, 0, st>>>(A);
    else k_stageA<P, 2><<<grid, block, 0, st>>>(A);
}

template <int P>
static int launch_final(hipStream_t st, const FinalArgs &F, int ny, dim3 grid, dim3 block, size_t lds)
{
    if (ny == 1) {
        IGX_HIP(hipFuncSetAttribute((const void *)k_final<P, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_final<P, 1><<<grid, block, lds, st>>>(F);
    } else {
        IGX_HIP(hipFuncSetAttribute((const void *)k_final<P, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_final<P, 4><<<grid, block, lds, st>>>(F);
    }
    return IGX_OK;
}

template <int P>
static void launch_stageB(hipStream_t st, const StageBArgs &B, dim3 grid, dim3 block)
{
    k_stageB<P><<<grid, block, 0, st>>>(B);
}

#define DISPATCH_P(Pv, CALL)                                   \
    switch (Pv) {                                              \
    case 2: { constexpr int PP = 2; CALL; } break;             \
    case 3: { constexpr int PP = 3; CALL; } break;             \
    case 4: { constexpr int PP = 4; CALL; } break;             \
    case 5: { constexpr int PP = 5; CALL; } break;             \
    case 6: { constexpr int PP = 6; CALL; } break;             \
    default: set_error("sum factorisation: degree %d unsupported", (Pv) - 1); return IGX_ERR_UNSUPPORTED; }

int sumfact_assemble(igx_patch *pt, int kind, double *d_data)
{
    hipStream_t st = pt->ctx->stream;
    const int dim = pt->dim;
    const PatchDev &pd = pt->dev;
    std::vector<Term> terms = form_terms(dim, kind);
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1], &A2 = pt->ax[2];
    const long long NPL = (long long)A1.G * (dim == 3 ? A2.G : 1);
    const int np0 = pt->npairs0;
    if (np0 == 0) return IGX_OK;

    // ---- stage-A arrays X = unique (t0, f).  In 2D the final stage wants the arrays ordered by
    // the last-axis type of their (single) consuming term; in 3D any order works.
    struct XA { int t0, f, slot; };
    std::vector<XA> X;
    std::vector<int> term_x(terms.size());
    for (size_t i = 0; i < terms.size(); ++i) {
        int found = -1;
        if (dim == 3)
            for (size_t x = 0; x < X.size(); ++x)
                if (X[x].t0 == terms[i].t[0] && X[x].f == terms[i].f) found = (int)x;
        if (found < 0) {
            found = (int)X.size();
            X.push_back(XA{terms[i].t[0], terms[i].f, dim == 2 ? (kind == IGX_MASS ? 0 : terms[i].t[1]) : found});
        }
        term_x[i] = found;
    }
    const int nX = (int)X.size();
    if (ensure(&pt->d_K1, &pt->K1_cap, (size_t)nX * np0 * NPL)) return IGX_ERR_NOMEM;

    const int nF = (kind == IGX_MASS) ? 1 : dim * (dim + 1) / 2;
    hipEventRecord(pt->ctx->ev[1], st);
    // one launch per field: the types of that field share the field load
    for (int f = 0; f < nF; ++f) {
        StageAArgs A{};
        int nt = 0;
        for (int x = 0; x < nX; ++x)
            if (X[x].f == f) {
                if (nt == 2) { set_error("internal: more than two stage-A types per field"); return IGX_ERR_UNSUPPORTED; }
                A.t[nt] = X[x].t0;
                A.out[nt] = pt->d_K1 + (size_t)X[x].slot * np0 * NPL;
                ++nt;
            }
        if (nt == 0) continue;
        A.field = pt->d_fields + (size_t)f * pd.npts_loc;
        A.PI0 = A0.d_PI; A.fa0 = A0.dev.fa; A.rp0 = A0.dev.rp; A.jlo0 = A0.dev.jlo; A.rl0_of = pt->d_rl0_of;
        A.s_lo = pt->s0_lo; A.s_hi = pt->s0_hi; A.n0 = A0.n; A.N0 = A0.N; A.q = A0.q; A.g0_lo = pd.g0_lo;
        A.NPL = NPL;
        dim3 block(256), grid((unsigned)((NPL + 255) / 256));
        DISPATCH_P(A0.P, launch_stageA<PP>(st, A, nt, grid, block));
        IGX_HIP(hipGetLastError());
        pt->timing.n_launches++;
    }
    hipEventRecord(pt->ctx->ev[2], st);

    // ---- final-stage input
    FinalArgs F{};
    const Axis &AL = (dim == 3) ? A2 : A1;
    int NY;
    if (dim == 3) {
        // stage B groups by the last-axis type y = t2
        StageBArgs B{};
        int ymax = 0;
        for (auto &g : B.grp) g.nterm = 0;
        for (size_t i = 0; i < terms.size(); ++i) {
            const int y = (kind == IGX_MASS) ? 0 : terms[i].t[2];
            StageBGroup &g = B.grp[y];
            g.x[g.nterm] = X[term_x[i]].slot;
            g.t1[g.nterm] = terms[i].t[1];
            g.nterm++;
            ymax = std::max(ymax, y);
        }
        NY = ymax + 1;
        if (ensure(&pt->d_K2, &pt->K2_cap, (size_t)NY * np0 * A1.S * A2.G)) return IGX_ERR_NOMEM;
        B.K1 = pt->d_K1; B.K2 = pt->d_K2; B.PI1 = A1.d_PI;
        B.fa1 = A1.dev.fa; B.rp1 = A1.dev.rp; B.jlo1 = A1.dev.jlo; B.pl0 = pt->d_pl0;
        B.n1 = A1.n; B.N1 = A1.N; B.q = A1.q; B.G1 = A1.G; B.G2 = A2.G; B.S1 = A1.S; B.npairs0 = np0;
        const int bs = A2.G >= 256 ? 128 : 64;
        dim3 block(bs), grid((A2.G + bs - 1) / bs, np0, NY);
        if (np0 > 65535) { set_error("stage B: more than 65535 axis-0 pairs"); return IGX_ERR_UNSUPPORTED; }
        DISPATCH_P(A1.P, launch_stageB<PP>(st, B, grid, block));
        IGX_HIP(hipGetLastError());
        pt->timing.n_launches++;
        F.K = pt->d_K2;
        F.nlines = (long long)np0 * A1.S;
        F.S1 = A1.S; F.Smid = A1.S; F.Slast = A2.S;
    } else {
        NY = (kind == IGX_MASS) ? 1 : 4;
        F.K = pt->d_K1;
        F.nlines = np0;
        F.S1 = 1; F.Smid = 1; F.Slast = A1.S;
    }
    hipEventRecord(pt->ctx->ev[3], st);

    F.data = d_data;
    F.V = AL.d_V; F.fa = AL.dev.fa; F.mslo = AL.dev.mslo; F.mshi = AL.dev.mshi;
    F.jlo = AL.dev.jlo; F.jhi = AL.dev.jhi; F.rp = AL.dev.rp;
    F.N = AL.N; F.q = AL.q; F.G = AL.G;
    F.dim = dim; F.pl0 = pt->d_pl0;
    F.rp0 = A0.dev.rp; F.jlo0 = A0.dev.jlo; F.jhi0 = A0.dev.jhi;
    F.rp1 = A1.dev.rp; F.jlo1 = A1.dev.jlo; F.jhi1 = A1.dev.jhi;
    F.pair1_i = A1.dev.pair_i; F.pair1_j = A1.dev.pair_j;
    F.r0_lo = pt->r0_lo; F.r0_hi = pt->r0_hi; F.nnz_off = pt->nnz_off;
    // row tiles of the last axis
    int TR = std::min(256, ((AL.N + 63) / 64) * 64);
    if (TR > 128 && AL.N > 256) TR = 128;
    F.TR = TR;
    int seg_max = 0;
    for (int lo = 0; lo < AL.N; lo += TR) {
        const int hi = std::min(lo + TR, AL.N);
        seg_max = std::max(seg_max, (AL.mshi[hi - 1] - AL.mslo[lo]) * AL.q);
    }
    F.seg_max = seg_max;
    const size_t lds = ((size_t)NY * seg_max + (size_t)seg_max * AL.P * 2) * sizeof(double);
    if (lds > 160 * 1024) { set_error("final stage needs %zu B of LDS", lds); return IGX_ERR_UNSUPPORTED; }
    {
        F.ntiles = (AL.N + TR - 1) / TR;
        const long long nblocks = F.nlines * F.ntiles;
        if (nblocks > 0x7fffffffLL) { set_error("final stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
        dim3 block(TR), grid((unsigned)nblocks);
        int rc = IGX_OK;
        DISPATCH_P(AL.P, rc = launch_final<PP>(st, F, NY, grid, block, lds));
        if (rc) return rc;
        IGX_HIP(hipGetLastError());
        pt->timing.n_launches++;
    }
    hipEventRecord(pt->ctx->ev[4], st);
    return IGX_OK;
}

} // namespace igx

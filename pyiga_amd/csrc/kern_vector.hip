// Load vector (arity-1 forms): out[i0,i1,i2] = sum_g prod_k B_{i_k}(g_k) * W(g) * f(g)
//
// Replaces inner_products() (pyiga/assemble.py:288-340) and the assemble_vector() of the
// L2Functional assemblers (pyiga/assemblers.pyx:883-1156,2204-2500; genericasm.pxi:438-456,762-778).
// W = gw0*gw1*gw2*|det J| is the mass field of the patch; the sum over the tensor Gauss grid is
// factorised axis by axis (last axis first: every stage shrinks its axis from G to N), which is what
// the reference does with transposed collocation matrices (tensor.apply_tprod).  All three stages
// stream their input once: HBM-bound, 8 B read per Gauss point in the first one.
#include "igx_internal.h"
#include <algorithm>

namespace igx {

// one axis: out[a][i][b] = sum_{g in supp(i)} V[g][i - fa(span g)][0] * in[a][g][b] (* w[a][g][b] in the first stage)
// thread = (a, i, b) with b fastest when B > 1, else i fastest.
template <bool WEIGHT>
__global__ void __launch_bounds__(256) k_contract_axis(const double *__restrict__ in, const double *__restrict__ wfield,
                                                       double *__restrict__ out, const AxisDev ax,
                                                       long long A, long long B, int i_lo, int i_hi, int g_off,
                                                       int deriv, int accumulate)
{
    const int Nout = i_hi - i_lo;
    const long long total = A * Nout * B;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    long long a, b;
    int i;
    if (B > 1) { b = t % B; const long long r = t / B; i = (int)(r % Nout); a = r / Nout; }
    else { i = (int)(t % Nout); a = t / Nout; b = 0; }
    i += i_lo;
    const int q = ax.q, P = ax.P;
    const int s_lo = ax.mslo[i], s_hi = ax.mshi[i];
    const long long Gin = ax.G;                      // extent of the contracted axis in `in` (its first entry is Gauss index g_off)
    double r = 0.0;
    for (int s = s_lo; s < s_hi; ++s) {
        const int aloc = i - ax.fa[s];
        for (int l = 0; l < q; ++l) {
            const int g = s * q + l;
            const long long idx = (a * Gin + (g - g_off)) * B + b;
            double v = in[idx];
            if (WEIGHT) v *= wfield[idx];
            r = fma(ax.V[((size_t)g * P + aloc) * 2 + deriv], v, r);      // deriv = 1: derivative of the basis function
        }
    }
    double *dst = out + (a * Nout + (i - i_lo)) * B + b;
    *dst = accumulate ? *dst + r : r;
}

// Last (contiguous) axis: a block walks LPB consecutive grid lines.  Thread i keeps the basis values of dof i on its
// support in registers (they do not depend on the line); each line (times its weights) is staged in LDS with
// coalesced loads (double-buffered), then thread i sums over its support.
//   out[a][i] = sum_g V[g][i - fa][deriv] * in[a][g] (* w[a][g])
constexpr int VEC_MAXSUP = 36;                           // (p+1) * q for p <= 5
template <bool WEIGHT>
__global__ void __launch_bounds__(256) k_contract_last(const double *__restrict__ in, const double *__restrict__ wfield,
                                                       double *__restrict__ out, const AxisDev ax, long long A, int LPB, int deriv)
{
    extern __shared__ double line[];                     // [2][G]
    const int G = ax.G, q = ax.q, P = ax.P;
    const long long a_lo = (long long)blockIdx.x * LPB, a_hi = min(a_lo + LPB, A);
    // per-thread basis values (threads beyond N idle in the compute phase, all threads help loading)
    const int i = threadIdx.x;
    double vreg[VEC_MAXSUP];
    int g_first = 0, nsup = 0;
    if (i < ax.N) {
        const int s_lo = ax.mslo[i], s_hi = ax.mshi[i];
        g_first = s_lo * q;
        nsup = (s_hi - s_lo) * q;
#pragma unroll
        for (int k = 0; k < VEC_MAXSUP; ++k) {
            double v = 0.0;
            if (k < nsup) {
                const int g = g_first + k, s = g / q;
                v = ax.V[((size_t)g * P + (i - ax.fa[s])) * 2 + deriv];
            }
            vreg[k] = v;
        }
    }
    auto stage = [&](const long long a, double *dst) {
        const double *src = in + a * G;
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            double v = src[g];
            if (WEIGHT) v *= wfield[a * G + g];
            dst[g] = v;
        }
    };
    if (a_lo < a_hi) stage(a_lo, line);
    __syncthreads();
    for (long long a = a_lo; a < a_hi; ++a) {
        double *cur = line + ((a - a_lo) & 1) * G, *nxt = line + ((a - a_lo + 1) & 1) * G;
        if (a + 1 < a_hi) stage(a + 1, nxt);
        if (i < ax.N) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < VEC_MAXSUP; ++k)
                if (k < nsup) r = fma(vreg[k], cur[g_first + k], r);
            out[a * ax.N + i] = r;
        }
        __syncthreads();
    }
}

// 3D, first two contractions in ONE kernel (round 4): a WAVE owns (Gauss plane g0, chunk of spans of the mid axis) and
// walks its grid lines g1 in sequence.  Per line: f and W arrive with 16-byte loads (the loads of the next line are in flight
// under the arithmetic of this one), their products go to a wave-private LDS buffer, lane i2 (+ 64, + 128 ..) sums over its
// support with the dense table Vt[k][i2] of the last axis' basis values (block-shared LDS, read without bank conflicts), and
// the line's N2 values enter the sliding window of the mid axis in registers -- acc[a] += V1[g1][a] t -- exactly like the
// sweeps of the matrix path.  The intermediate [G0][G1][N2] (0.43 GB written and read back at C4) never exists and no
// barrier is needed after the table is staged.  A dof of the mid axis whose support lies inside the chunk is stored; the
// (at most p) dofs shared with the neighbouring chunk are ADDED onto zeros: two addends commute, the sum does not
// depend on which chunk comes first (chunks are at least P spans long: never three addends).
constexpr int LV_MAXPASS = 4;                            // N2 <= 256
constexpr int LV_WAVES = 4;                              // waves of a block: they share the table of the last axis (three blocks per CU at
                                                         // C4: twelve waves, each with a line of loads in flight; measured: 6 waves per
                                                         // block and a shorter unrolled support loop 1.25 ms against 1.01)
typedef const double __attribute__((address_space(4))) *lv_cdp;
typedef const int __attribute__((address_space(4))) *lv_cip;
template <bool WEIGHT, int P, int MAXPC, int NPASS>    // MAXPC: 128-point pieces of a line held in registers (G2 <= 128 MAXPC); NPASS: N2 <= 64 NPASS
__global__ void __launch_bounds__(LV_WAVES * 64) k_lv12(const double *__restrict__ f, const double *__restrict__ wfield, double *__restrict__ t2,
                                              const AxisDev a1, const AxisDev a2, int G0, int chunk_spans, int nchunks, int deriv1, int deriv2)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int q = a2.q, PQ = P * q, N2 = a2.N, G2 = a2.G;
    double *Vt = lds;                                    // [PQ][N2]: basis value of dof i2 at point k of its support (0 past its end)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *buf = lds + ((PQ * N2 + 1) & ~1) + wave * ((G2 + 1) & ~1);    // this wave's line of products
    for (int e = threadIdx.x; e < PQ * N2; e += blockDim.x) {
        const int k = e / N2, i = e - k * N2;
        const int s_lo = a2.mslo[i], nsup = (a2.mshi[i] - s_lo) * q;
        double v = 0.0;
        if (k < nsup) { const int g = s_lo * q + k; v = a2.V[((size_t)g * P + (i - a2.fa[g / q])) * 2 + deriv2]; }
        Vt[e] = v;
    }
    __syncthreads();
    const long long unit = (long long)blockIdx.x * LV_WAVES + wave;
    if (unit >= (long long)G0 * nchunks) return;
    const int g0 = (int)(unit / nchunks), ch = (int)(unit - (long long)g0 * nchunks);
    const int s_a = ch * chunk_spans, s_b = ch == nchunks - 1 ? a1.n : s_a + chunk_spans;     // (the last chunk takes a short remainder along)
    const int q1 = a1.q, G1 = a1.G, N1 = a1.N;
    int gfirst[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) { const int i2 = min(lane + 64 * k, N2 - 1); gfirst[k] = a2.mslo[i2] * q; }
    constexpr int npass = NPASS;
    double acc[P][NPASS];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int k = 0; k < NPASS; ++k) acc[a][k] = 0.0;
    // a line as 16-byte pieces: piece e of the line = doubles 2 e, 2 e + 1 (G2 even: checked on the host)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int npc = (G2 / 2 + 63) >> 6;
    d2 vf[MAXPC], vw[MAXPC];
    auto request = [&](const int g1) {
        const d2 *pf = (const d2 *)(f + ((long long)g0 * G1 + g1) * G2);
        const d2 *pw = WEIGHT ? (const d2 *)(wfield + ((long long)g0 * G1 + g1) * G2) : pf;
#pragma unroll
        for (int c = 0; c < MAXPC; ++c)
            if (c < npc) {
                const int e = min(lane + 64 * c, G2 / 2 - 1);
                vf[c] = pf[e];
                if (WEIGHT) vw[c] = pw[e];
            }
    };
    const int g_a = s_a * q1, g_b = s_b * q1;
    request(g_a);
    lv_cdp V1 = (lv_cdp)a1.V;
    lv_cip fa1 = (lv_cip)a1.fa, mslo1 = (lv_cip)a1.mslo;
    int l = 0, sp = s_a;
    for (int g1 = g_a; g1 < g_b; ++g1) {
        // products of this line -> LDS; then the loads of the next line (their registers are free again)
#pragma unroll
        for (int c = 0; c < MAXPC; ++c)
            if (c < npc) {
                const int e = lane + 64 * c;
                d2 x = vf[c];
                if (WEIGHT) { x.x *= vw[c].x; x.y *= vw[c].y; }
                if (e < G2 / 2) ((d2 *)buf)[e] = x;
            }
        if (g1 + 1 < g_b) request(g1 + 1);
        // last-axis contraction of the line, then the mid-axis window
        double v1[P];
#pragma unroll
        for (int a = 0; a < P; ++a) v1[a] = V1[((size_t)g1 * P + a) * 2 + deriv1];
#pragma unroll
        for (int k = 0; k < NPASS; ++k)
            if (k < npass) {
                const int i2 = min(lane + 64 * k, N2 - 1);
                const double *bl = buf + gfirst[k];
                double r = 0.0;
                for (int m = 0; m < PQ; ++m) r = fma(Vt[m * N2 + i2], bl[min(m, G2 - 1 - gfirst[k])], r);
#pragma unroll
                for (int a = 0; a < P; ++a) acc[a][k] = fma(v1[a], r, acc[a][k]);
            }
        if (++l < q1) continue;
        // end of span sp: the dofs that leave the active set; at the end of the chunk every dof that is still active
        const int base = fa1[sp];
        const int nleave = sp + 1 < a1.n ? (sp + 1 < s_b ? fa1[sp + 1] - base : P) : P;
        for (int j = 0; j < nleave; ++j) {
            const int i1 = base + j;
            if (i1 < N1) {
                const bool whole = mslo1[i1] >= s_a && (sp + 1 < s_b || sp + 1 == a1.n || j < fa1[min(sp + 1, a1.n - 1)] - base);
                double *dst = t2 + ((long long)g0 * N1 + i1) * N2;
#pragma unroll
                for (int k = 0; k < NPASS; ++k)
                    if (k < npass && lane + 64 * k < N2) {
                        if (whole) dst[lane + 64 * k] = acc[0][k];
                        else (void)__hip_atomic_fetch_add(dst + lane + 64 * k, acc[0][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            }
#pragma unroll
            for (int a = 0; a < P - 1; ++a)
#pragma unroll
                for (int k = 0; k < NPASS; ++k) acc[a][k] = acc[a + 1][k];
#pragma unroll
            for (int k = 0; k < NPASS; ++k) acc[P - 1][k] = 0.0;
        }
        l = 0; ++sp;
    }
}

// Can k_lv12 serve the patch, and with which chunks of the mid axis?  (also the launch shape of the generated variant with the
// function inside: rtc.hip, igx_lv12_expr)
bool lv12_shape(const igx_patch *pt, int *clen_out, int *nch_out, size_t *lds_out)
{
    const PatchDev &pd = pt->dev;
    if (pd.dim != 3) return false;
    const AxisDev &a1 = pd.ax[1], &a2 = pd.ax[2];
    const long long G0 = pd.G0_loc;
    const int PQ = a2.P * a2.q;
    const size_t lds12 = ((size_t)((PQ * a2.N + 1) & ~1) + LV_WAVES * (size_t)((a2.G + 1) & ~1)) * sizeof(double);
    if (!(a2.P >= 2 && a2.P <= 6 && a1.P == a2.P && PQ <= VEC_MAXSUP && a2.N <= 64 * LV_MAXPASS && a2.G % 2 == 0 && a2.G <= 640 && lds12 <= 64 * 1024
          && a1.n >= a1.P)) return false;
    // chunks of the mid axis: enough waves for the chip, never shorter than P spans (a shared dof gets two addends)
    int nch = (int)std::min<long long>(std::max<long long>(1, (8192 + G0 - 1) / std::max<long long>(G0, 1)), std::max(1, a1.n / std::max(a1.P, 8)));
    int clen = (a1.n + nch - 1) / nch;
    clen = std::max(clen, a1.P);
    nch = (a1.n + clen - 1) / clen;
    if (nch > 1 && a1.n - (nch - 1) * clen < a1.P) { --nch; }     // a short last chunk joins its neighbour: the kernel lets
    // the last chunk run to the end of the axis (found by tools/fuzz_rhs.py: 65 spans at p = 5 lost their last two)
    *clen_out = clen; *nch_out = nch; *lds_out = lds12;
    return true;
}

// the last contraction of the 3D load vector: [G0][N1][N2] -> the owned dof planes of axis 0
int launch_lv_axis0(hipStream_t st, const igx_patch *pt, const double *d_t2, double *d_out, int deriv0, int accumulate)
{
    const PatchDev &pd = pt->dev;
    const AxisDev &a0 = pd.ax[0], &a1 = pd.ax[1], &a2 = pd.ax[2];
    const int n0 = pd.r0_hi - pd.r0_lo;
    const long long B = (long long)a1.N * a2.N, n = (long long)n0 * B;
    AxisDev ax0 = a0;
    ax0.G = (int)pd.G0_loc;
    k_contract_axis<false><<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(d_t2, nullptr, d_out, ax0, 1, B, pd.r0_lo, pd.r0_hi, pd.g0_lo, deriv0, accumulate);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

// d_f: function values on the RESIDENT Gauss slab (G0_loc x G1 [x G2]); d_W: mass field on the same slab;
// d_out: (r0_hi - r0_lo) x N1 [x N2]; tmp1/tmp2: workspaces (sizes below)
// deriv_axis: grid axis whose basis functions are differentiated (-1: none); accumulate: add to d_out;
// d_W == nullptr: d_f already contains the weights
int launch_load_vector(hipStream_t st, const igx_patch *pt, const double *d_f, const double *d_W, double *d_out,
                       double *d_t1, double *d_t2, int deriv_axis, int accumulate, int *n_launches)
{
    const PatchDev &pd = pt->dev;
    if (n_launches) *n_launches = pd.dim;
    const int dim = pd.dim;
    const int bs = 256;
    // the register/LDS kernel needs one thread per dof of the line, the support in VEC_MAXSUP registers and the
    // double-buffered line in LDS; anything else takes the generic kernel
    auto last_axis_fast = [](const AxisDev &ax) { return ax.N <= 256 && ax.P * ax.q <= VEC_MAXSUP && (size_t)2 * ax.G * sizeof(double) <= 48 * 1024; };
    auto blocks = [&](long long n) { return dim3((unsigned)((n + bs - 1) / bs)); };
    const long long G0 = pd.G0_loc;
    const AxisDev &a0 = pd.ax[0], &a1 = pd.ax[1], &a2 = pd.ax[2];
    const int n0 = pd.r0_hi - pd.r0_lo;
    if (dim == 3) {
        // fused first two contractions (k_lv12): single-digit degrees, even G2 <= 640 (five 16-byte pieces per lane: a longer line
        // would spill the prefetched pieces), lines and table in LDS; anything else takes the three separate contractions below
        int nch, clen;
        size_t lds12;
        if (lv12_shape(pt, &clen, &nch, &lds12)) {
            const long long units = G0 * nch;
            IGX_HIP(hipMemsetAsync(d_t2, 0, (size_t)G0 * a1.N * a2.N * sizeof(double), st));
            const dim3 grid((unsigned)((units + LV_WAVES - 1) / LV_WAVES));
            const int npass = (a2.N + 63) / 64;
#define LV12K(W_, PP, PC, NP) k_lv12<W_, PP, PC, NP><<<grid, LV_WAVES * 64, lds12, st>>>(d_f, d_W, d_t2, a1, a2, (int)G0, clen, nch, deriv_axis == 1, deriv_axis == 2)
#define LV12P(W_, PP, PC) do { if (npass <= 2) LV12K(W_, PP, PC, 2); else if (npass == 3) LV12K(W_, PP, PC, 3); else LV12K(W_, PP, PC, 4); } while (0)
#define LV12(PP) case PP: \
                if (d_W) LV12P(true, PP, 5); else LV12P(false, PP, 5); \
                break
            switch (a2.P) { LV12(2); LV12(3); LV12(4); LV12(5); LV12(6); }
#undef LV12
#undef LV12K
#undef LV12P
            if (int rc = launch_lv_axis0(st, pt, d_t2, d_out, deriv_axis == 0, accumulate)) return rc;
            if (n_launches) *n_launches = 2;
            return IGX_OK;
        }
        // [G0,G1,G2] -> [G0,G1,N2]
        {
            const long long A = G0 * a1.G, n = A * a2.N;
            if (last_axis_fast(a2)) {
                const int tb = a2.N > 128 ? 256 : 128, LPB = 16;
                const dim3 grid((unsigned)((A + LPB - 1) / LPB));
                if (d_W) k_contract_last<true><<<grid, tb, (size_t)2 * a2.G * sizeof(double), st>>>(d_f, d_W, d_t1, a2, A, LPB, deriv_axis == 2);
                else k_contract_last<false><<<grid, tb, (size_t)2 * a2.G * sizeof(double), st>>>(d_f, nullptr, d_t1, a2, A, LPB, deriv_axis == 2);
            } else if (d_W) k_contract_axis<true><<<blocks(n), bs, 0, st>>>(d_f, d_W, d_t1, a2, A, 1, 0, a2.N, 0, deriv_axis == 2, 0);
            else k_contract_axis<false><<<blocks(n), bs, 0, st>>>(d_f, nullptr, d_t1, a2, A, 1, 0, a2.N, 0, deriv_axis == 2, 0);
        }
        // [G0,G1,N2] -> [G0,N1,N2]
        {
            const long long n = G0 * a1.N * a2.N;
            k_contract_axis<false><<<blocks(n), bs, 0, st>>>(d_t1, nullptr, d_t2, a1, G0, a2.N, 0, a1.N, 0, deriv_axis == 1, 0);
        }
        // [G0,N1,N2] -> [n0,N1,N2]   (axis 0: only the owned dof planes; the slab starts at Gauss index g0_lo)
        {
            const long long B = (long long)a1.N * a2.N, n = (long long)n0 * B;
            AxisDev ax0 = a0;
            ax0.G = (int)G0;
            k_contract_axis<false><<<blocks(n), bs, 0, st>>>(d_t2, nullptr, d_out, ax0, 1, B, pd.r0_lo, pd.r0_hi, pd.g0_lo, deriv_axis == 0, accumulate);
        }
    } else {
        {
            const long long n = G0 * a1.N;
            if (last_axis_fast(a1)) {
                const int tb = a1.N > 128 ? 256 : 128, LPB = 4;
                const dim3 grid((unsigned)((G0 + LPB - 1) / LPB));
                if (d_W) k_contract_last<true><<<grid, tb, (size_t)2 * a1.G * sizeof(double), st>>>(d_f, d_W, d_t1, a1, G0, LPB, deriv_axis == 1);
                else k_contract_last<false><<<grid, tb, (size_t)2 * a1.G * sizeof(double), st>>>(d_f, nullptr, d_t1, a1, G0, LPB, deriv_axis == 1);
            } else if (d_W) k_contract_axis<true><<<blocks(n), bs, 0, st>>>(d_f, d_W, d_t1, a1, G0, 1, 0, a1.N, 0, deriv_axis == 1, 0);
            else k_contract_axis<false><<<blocks(n), bs, 0, st>>>(d_f, nullptr, d_t1, a1, G0, 1, 0, a1.N, 0, deriv_axis == 1, 0);
        }
        {
            const long long B = a1.N, n = (long long)n0 * B;
            AxisDev ax0 = a0;
            ax0.G = (int)G0;
            k_contract_axis<false><<<blocks(n), bs, 0, st>>>(d_t1, nullptr, d_out, ax0, 1, B, pd.r0_lo, pd.r0_hi, pd.g0_lo, deriv_axis == 0, accumulate);
        }
    }
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

} // namespace igx

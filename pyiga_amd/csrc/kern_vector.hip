// Load vector (arity-1 forms): out[i0,i1,i2] = sum_g prod_k B_{i_k}(g_k) * W(g) * f(g)
//
// Replaces inner_products() (pyiga/assemble.py:288-340) and the assemble_vector() of the
// L2Functional assemblers (pyiga/assemblers.pyx:883-1156,2204-2500; genericasm.pxi:438-456,762-778).
// W = gw0*gw1*gw2*|det J| is the mass field of the patch; the sum over the tensor Gauss grid is
// factorised axis by axis (last axis first: every stage shrinks its axis from G to N), which is what
// the reference does with transposed collocation matrices (tensor.apply_tprod).  All three stages
// stream their input once: HBM-bound, 8 B read per Gauss point in the first one.
#include "igx_internal.h"

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

// d_f: function values on the RESIDENT Gauss slab (G0_loc x G1 [x G2]); d_W: mass field on the same slab;
// d_out: (r0_hi - r0_lo) x N1 [x N2]; tmp1/tmp2: workspaces (sizes below)
// deriv_axis: grid axis whose basis functions are differentiated (-1: none); accumulate: add to d_out;
// d_W == nullptr: d_f already contains the weights
int launch_load_vector(hipStream_t st, const igx_patch *pt, const double *d_f, const double *d_W, double *d_out,
                       double *d_t1, double *d_t2, int deriv_axis, int accumulate)
{
    const PatchDev &pd = pt->dev;
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

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
    auto blocks = [&](long long n) { return dim3((unsigned)((n + bs - 1) / bs)); };
    const long long G0 = pd.G0_loc;
    const AxisDev &a0 = pd.ax[0], &a1 = pd.ax[1], &a2 = pd.ax[2];
    const int n0 = pd.r0_hi - pd.r0_lo;
    if (dim == 3) {
        // [G0,G1,G2] -> [G0,G1,N2]
        {
            const long long A = G0 * a1.G, n = A * a2.N;
            if (d_W) k_contract_axis<true><<<blocks(n), bs, 0, st>>>(d_f, d_W, d_t1, a2, A, 1, 0, a2.N, 0, deriv_axis == 2, 0);
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
            if (d_W) k_contract_axis<true><<<blocks(n), bs, 0, st>>>(d_f, d_W, d_t1, a1, G0, 1, 0, a1.N, 0, deriv_axis == 1, 0);
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

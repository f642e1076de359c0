// Separable geometry along axis 0 (an extruded 2D map: G(xi0, xi1, xi2) = (g(xi1, xi2), z(xi0)), e.g. the quarter-annulus
// CYLINDER of the BASELINE configs).  Its Jacobian is block diagonal, every quadrature field of the mass and stiffness forms
// is a product f0(xi0) * F(xi1, xi2), and the matrix is a sum of Kronecker products of a 1D and a 2D matrix:
//     M = M0[|z'|] (x) M2D                          K = M0[|z'|] (x) K2D  +  K0[1/|z'|] (x) M2D
// (M0[w]_{ij} = int phi_i phi_j w, K0[w]_{ij} = int phi_i' phi_j' w on axis 0; M2D, K2D the 2D matrices of the map g).  This is
// what the reference does for geo = None (pyiga/assemble.py:125-190,236-282: Kronecker products of 1D matrices); for a
// separable GEOMETRY the same structure holds with weighted 1D matrices and the 2D matrices of the cross-section.
//
// The 2D matrices are assembled by the regular 2D path of the library on a 2D patch (a few microseconds); k_kron3 expands them
// into the canonical CSR values of the 3D patch: row (i0, i1, i2) = for each of its c0 axis-0 columns j0 a copy of the 2D row
// (i1, i2) scaled by the two 1D entries -- c0 * L2 contiguous doubles per row, written once, coalesced: the kernel is a pure
// store stream (12.75 GB at C4).  The tensor Gauss rule factorises exactly like the integrand, so the result equals the
// entry-wise sums of the reference to rounding; the pattern, the row slabs and the exact symmetry are those of the general path.
#include "igx_internal.h"
#include <algorithm>
#include <vector>

namespace igx {

struct KronArgs {
    double *out;                 // CSR values of the owned rows of the 3D patch
    long long nnz_off, S12;
    const double *a0, *b0;       // [N0][C0]: 1D entries (row i0, column jlo0[i0] + k) multiplying A2 resp. B2
    const double *A2, *B2;       // 2D CSR values (canonical layout of the 2D patch); B2 unused when !TWO
    const int *rp0, *jlo0, *jhi0, *rp1, *jlo1, *jhi1, *rp2, *jlo2, *jhi2;
    int C0, N1, N2, i0_lo, i0_hi;
    long long S2;
};

template <bool TWO>
__global__ void __launch_bounds__(256) k_kron3(const KronArgs K)
{
    constexpr int MAXL = 128;                             // (2p+1)^2 <= 121 for p <= 5
    __shared__ double sa[4][MAXL], sb[4][MAXL], s0a[4][16], s0b[4][16];    // 2D row; 1D entries of row i0 (2 p0 + 1 <= 11)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r2 = (long long)blockIdx.x * 4 + wave;
    const int i0 = K.i0_lo + blockIdx.y;
    if (r2 >= (long long)K.N1 * K.N2) return;
    const int i1 = (int)(r2 / K.N2), i2 = (int)(r2 - (long long)i1 * K.N2);
    const int c1 = K.jhi1[i1] - K.jlo1[i1], c2 = K.jhi2[i2] - K.jlo2[i2], L2 = c1 * c2;
    const long long ip2 = (long long)K.rp1[i1] * K.S2 + (long long)c1 * K.rp2[i2];      // start of the 2D row
    for (int e = lane; e < L2; e += 64) {
        sa[wave][e] = K.A2[ip2 + e];
        if (TWO) sb[wave][e] = K.B2[ip2 + e];
    }
    // (wave-private LDS rows: in-order within the wave)
    const int c0 = K.jhi0[i0] - K.jlo0[i0];
    double *dst = K.out + ((long long)K.rp0[i0] * K.S12 + (long long)c0 * ip2 - K.nnz_off);
    if (lane < c0) {
        s0a[wave][lane] = K.a0[(size_t)i0 * K.C0 + lane];
        if (TWO) s0b[wave][lane] = K.b0[(size_t)i0 * K.C0 + lane];
    }
    const int total = c0 * L2;
    int k = 0, e = lane;                                   // flat index f = k L2 + e walks the row 64 entries at a time
    while (e >= L2) { e -= L2; ++k; }
    for (int f = lane; f < total; f += 64) {
        const double va = s0a[wave][k] * sa[wave][e];
        dst[f] = TWO ? fma(s0b[wave][k], sb[wave][e], va) : va;
        e += 64;
        while (e >= L2) { e -= L2; ++k; }
    }
}

int launch_kron3(hipStream_t st, igx_patch *p3, const double *d_a0, const double *d_b0, int C0, const double *d_A2, const double *d_B2)
{
    const PatchDev &pd = p3->dev;
    KronArgs K{};
    K.out = p3->d_data; K.nnz_off = p3->nnz_off;
    K.S2 = p3->ax[2].S; K.S12 = (long long)p3->ax[1].S * p3->ax[2].S;
    K.a0 = d_a0; K.b0 = d_b0 ? d_b0 : d_a0; K.A2 = d_A2; K.B2 = d_B2 ? d_B2 : d_A2;
    K.rp0 = pd.ax[0].rp; K.jlo0 = pd.ax[0].jlo; K.jhi0 = pd.ax[0].jhi;
    K.rp1 = pd.ax[1].rp; K.jlo1 = pd.ax[1].jlo; K.jhi1 = pd.ax[1].jhi;
    K.rp2 = pd.ax[2].rp; K.jlo2 = pd.ax[2].jlo; K.jhi2 = pd.ax[2].jhi;
    K.C0 = C0; K.N1 = p3->ax[1].N; K.N2 = p3->ax[2].N; K.i0_lo = p3->r0_lo; K.i0_hi = p3->r0_hi;
    const long long rows2 = (long long)K.N1 * K.N2;
    const int n0 = p3->r0_hi - p3->r0_lo;
    if (n0 <= 0 || rows2 == 0) return IGX_OK;
    if (n0 > 65535) { set_error("Kronecker expansion: more than 65535 owned dof planes"); return IGX_ERR_UNSUPPORTED; }
    dim3 grid((unsigned)((rows2 + 3) / 4), (unsigned)n0), block(256);
    if (d_B2) k_kron3<true><<<grid, block, 0, st>>>(K);
    else k_kron3<false><<<grid, block, 0, st>>>(K);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

} // namespace igx

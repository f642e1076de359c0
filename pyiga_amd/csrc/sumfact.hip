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
#include "sumfact_stages.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace igx {

// ---------------------------------------------------------------------------------------------
// Final stage, quadrature-lane form (single interior knots, q == P; the default for such axes).
//
// lane = (row i of the last axis, quadrature point l of a span): a wave owns R = 64/P consecutive rows
// for its whole life and keeps the basis values it needs in registers --
//     Vr[k][b] = (value, derivative) of trial function b of the k-th span of row i at point l,
//     va[k]    = the same for the test function i itself --
// so the contraction reads no table.  Work is a host-built list of 32-byte line descriptors (scalar
// loads); a block = FINALQ_WAVES adjacent row chunks going through the same lines in lockstep.
//   * K windows of a line travel HBM -> LDS by DMA (global_load_lds, no registers), FINALQ_DEPTH lines
//     ahead; the wave waits with counted vmcnt (loads and stores retire in issue order, so every store
//     is issued unconditionally -- masked lanes write a dump slot -- to keep the count exact).
//   * lane (i,l) reads its P x NY K values (consecutive lanes, consecutive LDS words) and accumulates
//     the contribution of point l to the 2p+1 entries of row i; the P partial sums are added through
//     a wave-private LDS region into block-shared sums (double-buffered, one s_barrier per line).
//   * direct CSR runs (2p+1 doubles per row) come from the sums in registers, mirrored runs are
//     gathered from the shared sums by the wave that owns the TARGET row, so both are whole runs except
//     next to the block's first and last row.  Store coordinates are precomputed once per wave.
// Measured at C4 (profiles/): VALU/LDS work 3.6 ms, + K reads 4.7 ms, + direct stores 6.2 ms, all 9.3 ms:
// the kernel is bound by the HBM write path of 72-byte runs (WRITE_SIZE 1.35 x the CSR bytes).
struct LineDesc {               // 32 bytes, built by build_qdesc()
    long long A_d, A_m;         // CSR base of the direct / mirrored run family (relative to the slab)
    int bc;                     // B_d | C_d<<8 | B_m<<16 | C_m<<24
    int flags;                  // bit0 row owned, bit1 column owned (mirror), bit2 leading diagonal line
    long long line;             // K line index
};
struct FinalQArgs {
    const double *V;            // last axis [G][P][2]
    const int *fa, *mslo, *mshi, *jlo, *jhi, *rp;
    int N, G;
    long long nlines;           // lines per K array (stride between types)
    const LineDesc *desc;
    long long ndesc;
    int lpw;                    // lines per block
    int nchunks;                // row chunks of the last axis (R rows each)
    int nsuper;                 // blocks per range of lines: ceil(nchunks / FINALQ_WAVES)
    long long dump;             // element index of a scratch slot behind the CSR values: target of masked-off stores
};

#ifndef IGX_Q_WAVES
#define IGX_Q_WAVES 4
#endif
#ifndef IGX_Q_DEPTH
#define IGX_Q_DEPTH 3
#endif
constexpr int FINALQ_WAVES = IGX_Q_WAVES; // adjacent row chunks per block
constexpr int FINALQ_DEPTH = IGX_Q_DEPTH; // K lines in flight per wave (LDS-DMA)
constexpr int FINALQ_KWIN = 96;         // doubles per staged K window: >= (64/P + P - 1) * P for P = 2..6, 3 DMA pieces
#ifndef IGX_Q_NT
#define IGX_Q_NT 0        // aux bits of the K-window DMA (2 = nt: streaming, do not keep in L2)
#endif
#ifndef IGX_Q_DBG
#define IGX_Q_DBG 0      // compile-time ablation mask (1: no stores, 2: no K loads after the first, 4: no mirror)
#endif
typedef const void __attribute__((address_space(1))) *gmem_ptr;
typedef void __attribute__((address_space(3))) *lds_ptr;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int P, int NY>
__global__ void __launch_bounds__(64 * FINALQ_WAVES) __attribute__((amdgpu_waves_per_eu(2))) k_final_q(const double *__restrict__ K, double *__restrict__ data, const FinalQArgs F)
{
    constexpr int W = 2 * P - 1, R = 64 / P, NJ = R + 2 * (P - 1);
    constexpr int NIT_D = (R * W + 63) / 64, NIT_M = (NJ * W + 63) / 64;
    constexpr int NC = (NY == 1) ? 1 : 2;
    constexpr int D = FINALQ_DEPTH, NSLOT = D + 1, KWIN = FINALQ_KWIN;
    constexpr int NGL = NY * 3;                         // LDS-DMA instructions per line (64 lanes x 4 B each)
    constexpr int NST = (IGX_Q_DBG & 1) ? 0 : NIT_D + (IGX_Q_DBG & 4 ? 0 : NIT_M);   // store instructions per line (all unconditional)
    static_assert((R + P - 1) * P <= KWIN, "K window");
    static_assert(D * (NGL + NST) < 64, "vmcnt range");
    constexpr int WAVE_LDS = NSLOT * NY * KWIN + 64 * W;
    constexpr int SUMS = (FINALQ_WAVES * R + 1) * W;              // sums of one line for all rows of the block (+ a junk row)
    __shared__ double lds_all[FINALQ_WAVES * WAVE_LDS + 2 * SUMS];   // ONE shared object (K slots, partial sums, sums x 2)
    typedef const LineDesc __attribute__((address_space(4))) *cdesc;

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // XCD-aware order: consecutive logical blocks (the row chunks of one range of lines, which read the
    // same K lines and complete each other's mirrored runs) run on one XCD and share its L2
    unsigned bid = blockIdx.x;
    {
        const unsigned per = gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    // block = FINALQ_WAVES adjacent row chunks working through the same lines in lockstep (one barrier per
    // line): the sums of a line are shared, so mirrored runs are written whole by the wave that owns the
    // target row and only the rows next to the block's ends are completed by the neighbouring block
    const int super = (int)(bid % F.nsuper);
    const long long wid = (long long)bid * FINALQ_WAVES + wave;
    const int chunk = super * FINALQ_WAVES + wave;
    const long long d_lo = (long long)(bid / F.nsuper) * F.lpw;
    const long long d_hi = min(d_lo + F.lpw, F.ndesc);
    if (d_lo >= d_hi || chunk >= F.nchunks) return;     // before any barrier: finished waves do not count
    double *kslot = lds_all + wave * WAVE_LDS;          // [NSLOT][NY][KWIN]
    double *out_q = kslot + NSLOT * NY * KWIN;
    double *sums = lds_all + FINALQ_WAVES * WAVE_LDS;   // [2][block rows + 1][W]
    const int blk_lo = super * FINALQ_WAVES * R, blk_hi = min(blk_lo + FINALQ_WAVES * R, F.N);
    // masked-off stores go to a scratch slot behind the CSR values (a line of its own per wave id)
    double *dump = data + F.dump + (wid & 1023) * 16;

    // ---- per-wave setup: rows, basis registers, store coordinates
    const int row_lo = chunk * R, row_hi = min(row_lo + R, F.N);
    const int ntr = row_hi - row_lo;
    const int r = lane / P, l = lane - r * P;
    const bool act = r < ntr;
    const int i = act ? row_lo + r : row_lo;
    const int slo = F.mslo[i], nsp = act ? F.mshi[i] - slo : 0;
    const int win0 = __builtin_amdgcn_readfirstlane(F.mslo[row_lo]) * P;   // first element of the chunk's K window
    const int koff = slo * P + l - win0;                // this lane's element (span 0) inside the window
    double Vr[P][P][NC], va[P][NC];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const bool live = k < nsp;
        const int s = live ? slo + k : slo;
        const int a = min(max(i - F.fa[s], 0), P - 1);
        const double *vp = F.V + ((size_t)s * P + l) * P * 2;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int b = 0; b < P; ++b) Vr[k][b][c] = live ? vp[2 * b + c] : 0.0;
            va[k][c] = live ? vp[2 * a + c] : 0.0;
        }
    }
    // direct stores: entry f = (row r_, offset o) of the chunk; element offset = B_d*rp + C_d*c + o
    int d_rp[NIT_D], d_src[NIT_D], d_dst[NIT_D];        // d_src: source index | row length << 12 | offset << 16
    bool d_ok[NIT_D], d_up[NIT_D];
#pragma unroll
    for (int it = 0; it < NIT_D; ++it) {
        const int f = lane + 64 * it;
        const int r_ = f / W, o = f - r_ * W;
        const bool v = r_ < ntr;
        const int ii = v ? row_lo + r_ : row_lo;
        const int jli = F.jlo[ii], ci = F.jhi[ii] - jli;
        d_ok[it] = v && o < ci;
        d_up[it] = jli + o > ii;
        d_rp[it] = F.rp[ii];
        d_src[it] = (v ? (r_ * P) * W + o : 0) | (ci << 12) | (o << 16);
        d_dst[it] = v ? (ii - blk_lo) * W + o : FINALQ_WAVES * R * W;    // junk slot for lanes past the chunk
    }
    // mirrored stores: row j, column i, value from the sums of row i.  This wave writes (j, i) when it owns
    // row j and row i is in the block, or when row j lies outside the block and it owns row i.
    const int jmin = F.jlo[row_lo], nj = F.jhi[row_hi - 1] - jmin;
    int m_rp[NIT_M], m_src[NIT_M];                      // m_src: source index | row length << 12 | offset << 16
    bool m_ok[NIT_M], m_ge[NIT_M];
#pragma unroll
    for (int it = 0; it < NIT_M; ++it) {
        const int f = lane + 64 * it;
        const int rr = f / W, o = f - rr * W;
        const bool v = rr < nj;
        const int j = v ? jmin + rr : jmin;
        const int jlj = F.jlo[j], cj = F.jhi[j] - jlj;
        const int ii = jlj + o;
        const bool own_j = j >= row_lo && j < row_hi, blk_j = j >= blk_lo && j < blk_hi;
        const bool own_i = ii >= row_lo && ii < row_hi, blk_i = ii >= blk_lo && ii < blk_hi;
        const bool in = v && o < cj && ((own_j && blk_i) || (!blk_j && own_i));
        const int iic = in ? ii : row_lo;
        m_ok[it] = in;
        m_ge[it] = j >= ii;
        m_src[it] = ((iic - blk_lo) * W + (in ? j - F.jlo[iic] : 0)) | (cj << 12) | (o << 16);
        m_rp[it] = F.rp[j];
    }

    // ---- K lines: HBM -> LDS by DMA (no registers), FINALQ_DEPTH lines ahead of the arithmetic.
    // A window is KWIN doubles from element win0 of the line; bytes past the support (or the line) are
    // finite and meet zero basis values.
    const long long ystride = F.nlines * (long long)F.G;
    auto issue_line = [&](const long long line, const int slot) {
        const char *src = (const char *)(K + line * F.G + win0) + lane * 4;
        double *dst = kslot + slot * (NY * KWIN);
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                __builtin_amdgcn_global_load_lds((gmem_ptr)(src + y * ystride * 8 + c * 256), (lds_ptr)(dst + y * KWIN + c * 32), 4, 0, IGX_Q_NT);
    };
    auto do_line = [&](const LineDesc &D_, const int slot, double *out_s) {
        const double *ks = kslot + slot * (NY * KWIN) + koff;
        double acc[W];
#pragma unroll
        for (int o = 0; o < W; ++o) acc[o] = 0.0;
        if constexpr (P >= 6 && NY == 4) {
            // p = 5: the basis registers (Vr, va: 168 VGPRs) leave no room for all NY * P window values at once; with them
            // the kernel spilled, and a scratch reload waits for vmcnt(0) -- for every K window in flight and every store
            // of the line -- inside this loop.  The values are read span by span, one span ahead of their use.
            double kc[NY], kn[NY];
#pragma unroll
            for (int y = 0; y < NY; ++y) kc[y] = ks[y * KWIN];
#pragma unroll
            for (int k = 0; k < P; ++k) {
                if (k + 1 < P) {
#pragma unroll
                    for (int y = 0; y < NY; ++y) kn[y] = ks[y * KWIN + (k + 1) * P];
                }
                const double cu0 = fma(va[k][1], kc[2], va[k][0] * kc[0]);     // types 0, 2
                const double cu1 = fma(va[k][1], kc[3], va[k][0] * kc[1]);     // types 1, 3
#pragma unroll
                for (int b = 0; b < P; ++b) acc[k + b] = fma(Vr[k][b][0], cu0, fma(Vr[k][b][1], cu1, acc[k + b]));
#pragma unroll
                for (int o = 0; o < W; ++o) asm volatile("" : "+v"(acc[o]));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int y = 0; y < NY; ++y) kc[y] = kn[y];
            }
        } else {
        double kv[NY][P];
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int k = 0; k < P; ++k) kv[y][k] = ks[y * KWIN + k * P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            double cu0, cu1 = 0.0;
            if constexpr (NY == 1) cu0 = va[k][0] * kv[0][k];
            else {
                cu0 = fma(va[k][1], kv[2][k], va[k][0] * kv[0][k]);     // types 0, 2
                cu1 = fma(va[k][1], kv[3][k], va[k][0] * kv[1][k]);     // types 1, 3
            }
#pragma unroll
            for (int b = 0; b < P; ++b) {
                if constexpr (NY == 1) acc[k + b] = fma(Vr[k][b][0], cu0, acc[k + b]);
                else acc[k + b] = fma(Vr[k][b][0], cu0, fma(Vr[k][b][1], cu1, acc[k + b]));
            }
        }
        }
#pragma unroll
        for (int o = 0; o < W; ++o) out_q[lane * W + o] = acc[o];
        __builtin_amdgcn_wave_barrier();

        const int B_d = D_.bc & 255, C_d = (D_.bc >> 8) & 255, B_m = (D_.bc >> 16) & 255, C_m = (D_.bc >> 24) & 255;
        const bool own_row = D_.flags & 1, own_col = D_.flags & 2, diag_lead = D_.flags & 4;
        double *dst_d = data + D_.A_d, *dst_m = data + D_.A_m;
        // add the P partial sums of every entry; direct stores.  Masked-off lanes store to the dump slot:
        // with no branch around a store the number of outstanding memory operations per line is fixed,
        // which the counted waits on the K windows below rely on.
#pragma unroll
        for (int it = 0; it < NIT_D; ++it) {
            const double *q = out_q + (d_src[it] & 4095);   // entries past the chunk: source 0, masked below
            double v = q[0];
#pragma unroll
            for (int p_ = 1; p_ < P; ++p_) v += q[p_ * W];
            out_s[d_dst[it]] = v;
            const bool st = own_row && d_ok[it] && !(diag_lead && d_up[it]);
            double *p = st ? dst_d + (B_d * d_rp[it] + C_d * ((d_src[it] >> 12) & 15) + (d_src[it] >> 16)) : dump;
            if (!(IGX_Q_DBG & 1)) *p = v;
        }
        // every wave's sums of this line are in LDS before any of them is read; the DMA queue stays untouched
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(IGX_Q_DBG & 4)) {
#pragma unroll
            for (int it = 0; it < NIT_M; ++it) {
                const bool st = own_col && m_ok[it] && !(diag_lead && m_ge[it]);
                double *p = st ? dst_m + (B_m * m_rp[it] + C_m * ((m_src[it] >> 12) & 15) + (m_src[it] >> 16)) : dump;
                if (!(IGX_Q_DBG & 1)) *p = out_s[m_src[it] & 4095];
            }
        }
        __builtin_amdgcn_wave_barrier();                // out_q reads above precede the next line's writes
    };

    cdesc desc = (cdesc)F.desc;
    auto fetch = [&](const long long idx) {             // scalar loads (constant address space)
        cdesc p = desc + min(idx, d_hi - 1);
        LineDesc D_;
        D_.A_d = p->A_d; D_.A_m = p->A_m; D_.bc = p->bc; D_.flags = p->flags; D_.line = p->line;
        return D_;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue_line(desc[min(d_lo + d, d_hi - 1)].line, d);
    LineDesc cur = fetch(d_lo);
    int slot = 0;
    for (long long idx = d_lo; idx < d_hi; ++idx) {
        // line idx + D into the slot that line idx - 1 has finished with (clamped at the end: harmless re-read)
        int nslot = slot + D; if (nslot >= NSLOT) nslot -= NSLOT;
        if (!(IGX_Q_DBG & 2)) issue_line(desc[min(idx + D, d_hi - 1)].line, nslot);
        const LineDesc nxt = fetch(idx + 1);
        // the window of line idx is followed in the queue by the D younger windows and the stores of the
        // (at most D) lines in between
        const int older = (int)min((long long)D, idx - d_lo);
        static_assert(D <= 4, "wait ladder");
        if (older == 0) wait_vm<D * NGL>();
        else if (older == 1) wait_vm<D * NGL + NST>();
        else if (older == 2) wait_vm<D * NGL + (D < 2 ? D : 2) * NST>();
        else if (older == 3) wait_vm<D * NGL + (D < 3 ? D : 3) * NST>();
        else wait_vm<D * NGL + D * NST>();
        // the sums ping-pong: a wave can be one line ahead of the slowest one, never two (the barrier)
        do_line(cur, slot, sums + ((idx - d_lo) & 1) * SUMS);
        cur = nxt;
        if (++slot == NSLOT) slot = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Final stage as a banded FP64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64).
//
//   D[line][(i,o)] = sum_t sum_g  K[t][line][g] * PI_last[t][(i,o)][g]
//
// M = 16 K lines, N = 16 consecutive output slots (i,o) of the last axis (slot = i*W + o, W = 2p+1),
// K dimension = the Gauss window of those slots (<= NCH*8 points) x NY types.  The PI products are
// the stationary B operand: a wave builds them once from the basis table and keeps them in
// registers while it streams line tiles; the A operand comes straight from global memory in
// fragment layout (lane (m,kk) reads two consecutive doubles of line m), reloaded right after the
// MFMA that consumed it, so the loads of the next tile are in flight under the current tile's
// MFMAs.  No LDS, no barriers.  MI355X measured: 65 cycles/MFMA/SIMD = 2048 flop -> 77 TFLOP/s, twice
// what v_fma_f64 reaches at the 1-2 waves/SIMD this kernel runs at (profiles/r01_ubench_fp64_rates.txt).
// Epilogue: lane (n, rows r) holds D[line r][slot n]; CSR positions from a 16-byte line descriptor
// and per-lane slot constants; direct and mirrored stores in runs of up to W doubles per line.
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));

struct FinalMArgs {
    const int4 *desc;           // [nl] {A_d, A_m, B_d | C_d<<8 | B_m<<16 | C_m<<24, line | flags<<28}
    int nl;                     // valid lines
    long long nlines;           // lines per K array (stride between types)
    const double *V;            // last axis [G][P][2]
    const int *fa, *mslo, *mshi, *jlo, *jhi, *rp;
    int N, P, q, G, W, ntile;   // last axis: dofs, p+1, q, Gauss points, 2p+1, N-tiles
    int LC;                     // lines per wave (multiple of 16)
    int nchunk_blocks;          // line chunks
    int debug;
};

template <int NY, int NCH>
__global__ void __launch_bounds__(256) k_final_mfma(const double *__restrict__ K, double *__restrict__ data, const FinalMArgs F)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order: consecutive logical blocks (the N-tiles of one line chunk, which re-read the
    // same K lines) run on one XCD and share its L2
    unsigned bid = blockIdx.x;
    {
        const unsigned nb = gridDim.x, per = nb / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    const int tgroups = (F.ntile + 3) / 4;
    const int tile = (bid % tgroups) * 4 + wave;
    const int lchunk = bid / tgroups;
    if (tile >= F.ntile) return;
    const int n = lane & 15, kk = lane >> 4;

    // ---- per-lane output slot constants
    const int slot = tile * 16 + n;
    const int i = slot / F.W, o = slot - i * F.W;
    bool valid = i < F.N;
    int j = 0, RP = 0, CI = 0, RPm = 0, CJ = 0, om = 0;
    if (valid) {
        const int jl = F.jlo[i];
        CI = F.jhi[i] - jl;
        valid = o < CI;
        if (valid) { j = jl + o; RP = F.rp[i]; RPm = F.rp[j]; CJ = F.jhi[j] - F.jlo[j]; om = i - F.jlo[j]; }
    }
    const bool upper = j > i, ondiag = (j == i);

    // ---- Gauss window of the tile (wave-uniform)
    const int i_first = min(F.N - 1, (tile * 16) / F.W);
    const int glo = F.mslo[i_first] * F.q;
    const bool overrun = glo + NCH * 8 > F.G;       // window padding reaches past the end of the line

    // ---- stationary operand: PI products of this lane's slot at its k positions
    double bf[NY][NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            const int g = glo + 8 * c + 2 * kk + j2;
            double u0 = 0.0, u1 = 0.0, v0 = 0.0, v1 = 0.0;
            if (valid && g < F.G) {
                const int s = g / F.q;
                if (s >= F.mslo[i] && s < F.mshi[i] && s >= F.mslo[j] && s < F.mshi[j]) {
                    const int f = F.fa[s];
                    const double *vu = F.V + ((size_t)g * F.P + (j - f)) * 2;
                    const double *vv = F.V + ((size_t)g * F.P + (i - f)) * 2;
                    u0 = vu[0]; u1 = vu[1]; v0 = vv[0]; v1 = vv[1];
                }
            }
            bf[0][c][j2] = u0 * v0;
            if constexpr (NY == 4) { bf[1][c][j2] = u1 * v0; bf[2][c][j2] = u0 * v1; bf[3][c][j2] = u1 * v1; }
        }

    // ---- stream line tiles
    const int l_lo = lchunk * F.LC, l_hi = min(l_lo + F.LC, F.nl);
    if (l_lo >= l_hi) return;
    const long long tstride = F.nlines * (long long)F.G;
    int goff[NCH];                                  // offset of this lane's pair inside the line
#pragma unroll
    for (int c = 0; c < NCH; ++c) goff[c] = min(glo + 8 * c + 2 * kk, F.G - 2);
    double2_u a[NY][NCH];
    auto load_tile = [&](const int mt) {
        const int mi = min(mt + (lane & 15), l_hi - 1);
        const int line = F.desc[mi].w & 0x0fffffff;
        const double *base = K + (long long)line * F.G;
#pragma unroll
        for (int t = 0; t < NY; ++t)
#pragma unroll
            for (int c = 0; c < NCH; ++c) a[t][c] = *(const double2_u *)(base + t * tstride + goff[c]);
    };
    load_tile(l_lo);
    // line index of the tile after next is requested two tiles ahead, so the address of a reload never
    // waits on a descriptor load
    int line_n1 = F.desc[min(l_lo + 16 + (lane & 15), l_hi - 1)].w & 0x0fffffff;
    int4 dsc_n[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dsc_n[r] = F.desc[min(l_lo + kk + 4 * r, l_hi - 1)];
    for (int mt = l_lo; mt < l_hi; mt += 16) {
        // descriptors of the 4 lines whose results this lane holds (loaded one tile ahead)
        int4 dsc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { dsc[r] = dsc_n[r]; dsc_n[r] = F.desc[min(mt + 16 + kk + 4 * r, l_hi - 1)]; }
        const bool more = mt + 16 < l_hi;
        const double *nbase = K + (long long)line_n1 * F.G;
        line_n1 = F.desc[min(mt + 32 + (lane & 15), l_hi - 1)].w & 0x0fffffff;

        double4_t d0 = {0.0, 0.0, 0.0, 0.0}, d1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < NY; ++t)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                double ax = a[t][c].x, ay = a[t][c].y;
                if (overrun) {                       // never let data of the next line in (0 * inf = nan)
                    const int g = glo + 8 * c + 2 * kk;
                    if (g >= F.G - 1) { ax = (g == F.G - 1) ? ay : 0.0; ay = 0.0; }   // pair was clamped to [G-2, G-1]
                }
                d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ax, bf[t][c][0], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ay, bf[t][c][1], d1, 0, 0, 0);
                if (more) a[t][c] = *(const double2_u *)(nbase + t * tstride + goff[c]);
            }
        // ---- epilogue: D[row = kk + 4r][col = n]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (mt + kk + 4 * r >= l_hi || !valid) continue;
            const double v = d0[r] + d1[r];
            const int4 ds = dsc[r];
            const int flags = (unsigned)ds.w >> 28;     // bit0 own_row, bit1 own_col, bit2 diag_lead
            const bool dl = flags & 4;
            const int Bd = ds.z & 255, Cd = (ds.z >> 8) & 255, Bm = (ds.z >> 16) & 255, Cm = (ds.z >> 24) & 255;
            if ((flags & 1) && !(dl && upper) && !(F.debug & 1)) data[(long long)ds.x + Bd * RP + Cd * CI + o] = v;
            if ((flags & 2) && !(dl && (upper || ondiag)) && !(F.debug & 2)) data[(long long)ds.y + Bm * RPm + Cm * CJ + om] = v;
        }
    }
}

// sum of up to 16 arrays (2D general forms: the stage-A arrays of one last-axis type); n == 0 gives zeros
struct CombineArgs { const double *src[16]; double *dst; long long len; int n; };
__global__ void __launch_bounds__(256) k_combine(const CombineArgs C)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < C.len; i += stride) {
        double v = 0.0;
        for (int k = 0; k < C.n; ++k) v += C.src[k][i];
        C.dst[i] = v;
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

int sumfact_supports_kind(const igx_patch *pt, int kind)
{
    return kind == IGX_MASS || kind == IGX_STIFFNESS || kind == IGX_FORM || (kind == IGX_CONVDIFF && pt->dim == 3);
}

// Line descriptors of the quadrature-lane final kernel: one 32-byte record per K line that is contracted,
// in (r0, i1, j1) order so that consecutive lines of a wave write adjacent CSR runs.
//   {A_d (int64), A_m (int64), B_d | C_d<<8 | B_m<<16 | C_m<<24, flags, line (int64)}
// flags: bit0 row owned, bit1 column owned (mirror), bit2 leading diagonal line (i0==j0 [and i1==j1])
static int build_qdesc(igx_patch *pt, const std::vector<int> &pl, bool sym, int **d_out, long long *n_out)
{
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1], &A2 = pt->ax[2];
    const int dim = pt->dim;
    const long long S1 = A1.S, S2 = A2.S;
    std::vector<int> ld;
    auto push = [&](long long Ad, long long Am, int Bd, int Cd, int Bm, int Cm, int flags, long long line) {
        ld.push_back((int)(Ad & 0xffffffffLL)); ld.push_back((int)(Ad >> 32));
        ld.push_back((int)(Am & 0xffffffffLL)); ld.push_back((int)(Am >> 32));
        ld.push_back(Bd | (Cd << 8) | (Bm << 16) | (Cm << 24));
        ld.push_back(flags);
        ld.push_back((int)(line & 0xffffffffLL)); ld.push_back((int)(line >> 32));
    };
    const int np = (int)(pl.size() / 2);
    int T = 1;          // measured at C4: 1, 2, 4, 16 within noise of each other (the final stage is bound by its HBM writes)
#ifdef IGX_ABLATE
    if (const char *e = getenv("IGX_FINALQ_TILE")) T = std::max(1, atoi(e));
#endif
    if (dim == 3) {
        // group lines by (r0, i1): rows i1 in order, their columns j1 in order
        for (int r0 = 0; r0 < np; ++r0) {
            const int i0 = pl[2 * r0], j0 = pl[2 * r0 + 1];
            const int c0i = A0.jhi[i0] - A0.jlo[i0], c0j = A0.jhi[j0] - A0.jlo[j0];
            int flags = 0;
            if (i0 >= pt->r0_lo && i0 < pt->r0_hi) flags |= 1;
            if (sym && j0 >= pt->r0_lo && j0 < pt->r0_hi) flags |= 2;
            const bool diag0 = sym && i0 == j0;
            // rows i1 in blocks of T, columns j1 outermost inside a block.  T = 1: a row's lines follow each
            // other, so consecutive lines write adjacent direct runs; T > 1 also brings the mirrored runs of
            // neighbouring rows together in time
            for (int a1 = 0; a1 < A1.N; a1 += T)
              for (int j1 = A1.jlo[a1]; j1 < A1.jhi[std::min(a1 + T, A1.N) - 1]; ++j1)
                for (int i1 = a1; i1 < std::min(a1 + T, A1.N); ++i1) {
                    if (j1 < A1.jlo[i1] || j1 >= (diag0 ? i1 + 1 : A1.jhi[i1])) continue;
                    const int r1 = A1.rp[i1] + (j1 - A1.jlo[i1]);
                    const int c1i = A1.jhi[i1] - A1.jlo[i1], c1j = A1.jhi[j1] - A1.jlo[j1];
                    const long long Ad = (long long)A0.rp[i0] * S1 * S2 + (long long)c0i * A1.rp[i1] * S2 - pt->nnz_off;
                    const long long Am = (long long)A0.rp[j0] * S1 * S2 + (long long)c0j * A1.rp[j1] * S2 - pt->nnz_off;
                    push(Ad, Am, c0i * c1i, (j0 - A0.jlo[i0]) * c1i + (j1 - A1.jlo[i1]),
                         c0j * c1j, (i0 - A0.jlo[j0]) * c1j + (i1 - A1.jlo[j1]),
                         flags | ((diag0 && i1 == j1) ? 4 : 0), (long long)r0 * A1.S + r1);
                }
        }
    } else {
        for (int r0 = 0; r0 < np; ++r0) {
            const int i0 = pl[2 * r0], j0 = pl[2 * r0 + 1];
            const int c0i = A0.jhi[i0] - A0.jlo[i0], c0j = A0.jhi[j0] - A0.jlo[j0];
            int flags = 0;
            if (i0 >= pt->r0_lo && i0 < pt->r0_hi) flags |= 1;
            if (sym && j0 >= pt->r0_lo && j0 < pt->r0_hi) flags |= 2;
            if (sym && i0 == j0) flags |= 4;
            push((long long)A0.rp[i0] * S1 - pt->nnz_off, (long long)A0.rp[j0] * S1 - pt->nnz_off,
                 c0i, j0 - A0.jlo[i0], c0j, i0 - A0.jlo[j0], flags, r0);
        }
    }
    *n_out = (long long)(ld.size() / 8);
    IGX_HIP(hipMalloc(d_out, std::max<size_t>(1, ld.size()) * sizeof(int)));
    IGX_HIP(hipMemcpyAsync(*d_out, ld.data(), ld.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    return IGX_OK;
}

// Line descriptors of the stage-kernel final (k_final_q / k_final_mfma): ~0.8 M descriptors (25 MB) at C4, built on
// the host.  The default 3D chain (k_geoA, k_bf, k_mirror) never reads them, so they are made on first use.
static int ensure_line_descriptors(igx_patch *pt)
{
    if (pt->desc_built) return IGX_OK;
    const std::vector<int> &pl = pt->h_pl0;
    const Axis &A0 = pt->ax[0];
    // MFMA kernel: one descriptor per K line that is actually contracted
    {
        const Axis &A1 = pt->ax[1], &A2 = pt->ax[2];
        const int dim = pt->dim;
        std::vector<int> ld;
        const long long S1 = A1.S, S2 = A2.S;
        for (int r0 = 0; r0 < pt->npairs0; ++r0) {
            const int i0 = pl[2 * r0], j0 = pl[2 * r0 + 1];
            const int c0i = A0.jhi[i0] - A0.jlo[i0], c0j = A0.jhi[j0] - A0.jlo[j0];
            int flags = 0;
            if (i0 >= pt->r0_lo && i0 < pt->r0_hi) flags |= 1;
            if (j0 >= pt->r0_lo && j0 < pt->r0_hi) flags |= 2;
            if (dim == 2) {
                const long long Ad = (long long)A0.rp[i0] * S1 - pt->nnz_off, Am = (long long)A0.rp[j0] * S1 - pt->nnz_off;
                const int fl = flags | ((i0 == j0) ? 4 : 0);
                ld.push_back((int)Ad); ld.push_back((int)Am);
                ld.push_back(c0i | ((j0 - A0.jlo[i0]) << 8) | (c0j << 16) | ((i0 - A0.jlo[j0]) << 24));
                ld.push_back(r0 | (fl << 28));
            } else {
                for (int r1 = 0; r1 < A1.S; ++r1) {
                    const int i1 = A1.pair_i[r1], j1 = A1.pair_j[r1];
                    if (i0 == j0 && j1 > i1) continue;          // mirrored, not computed
                    const int c1i = A1.jhi[i1] - A1.jlo[i1], c1j = A1.jhi[j1] - A1.jlo[j1];
                    const long long Ad = (long long)A0.rp[i0] * S1 * S2 + (long long)c0i * A1.rp[i1] * S2 - pt->nnz_off;
                    const long long Am = (long long)A0.rp[j0] * S1 * S2 + (long long)c0j * A1.rp[j1] * S2 - pt->nnz_off;
                    const int Bd = c0i * c1i, Cd = (j0 - A0.jlo[i0]) * c1i + (j1 - A1.jlo[i1]);
                    const int Bm = c0j * c1j, Cm = (i0 - A0.jlo[j0]) * c1j + (i1 - A1.jlo[j1]);
                    const int fl = flags | ((i0 == j0 && i1 == j1) ? 4 : 0);
                    ld.push_back((int)Ad); ld.push_back((int)Am);
                    ld.push_back(Bd | (Cd << 8) | (Bm << 16) | (Cm << 24));
                    ld.push_back((int)((long long)r0 * A1.S + r1) | (fl << 28));
                }
            }
        }
        pt->n_ldesc = (int)(ld.size() / 4);
        pt->ldesc_ok = (long long)pt->npairs0 * (dim == 3 ? A1.S : 1) < (1LL << 28);
        if (pt->d_ldesc) { (void)hipFree(pt->d_ldesc); pt->d_ldesc = nullptr; }     // (a retry after a failed build: no leak)
        IGX_HIP(hipMalloc(&pt->d_ldesc, std::max<size_t>(1, ld.size()) * sizeof(int)));
        IGX_HIP(hipMemcpyAsync(pt->d_ldesc, ld.data(), ld.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
        IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    }
    if (int rc = build_qdesc(pt, pl, true, &pt->d_qdesc, &pt->n_qdesc)) return rc;
    pt->desc_built = true;
    return IGX_OK;
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
    // flush-step tables of the two sweeps: after span s the dofs fa[s] .. fa[s+1]-1 (all remaining
    // ones after the last span) leave the active set, one step each, in increasing order
    auto build_steps = [&](const Axis &A, int width, std::vector<int> &ptr, std::vector<int> &rec, bool stageA) {
        ptr.assign(A.n + 1, 0);
        rec.clear();
        for (int s = 0; s < A.n; ++s) {
            const int base = A.fa[s];
            const int m = (s + 1 < A.n) ? (A.fa[s + 1] - base) : A.P;
            for (int k = 0; k < m; ++k) {
                const int d = base + k;
                std::vector<int> r(width, -1);
                for (int a = 0; a < A.P; ++a) {
                    const int o = d + a;
                    if (a > A.P - 1 - k || o >= A.N) continue;       // slot holds no accumulated pair
                    if (stageA) r[a] = rl[A.rp[o] + (d - A.jlo[o])];
                    else {
                        r[a] = A.rp[o] + (d - A.jlo[o]);
                        if (a > 0) r[8 + a] = A.rp[d] + (o - A.jlo[d]);
                    }
                }
                rec.insert(rec.end(), r.begin(), r.end());
            }
            ptr[s + 1] = (int)(rec.size() / width);
        }
    };
    std::vector<int> ptrA, recA, ptrB, recB;
    build_steps(A0, 8, ptrA, recA, true);
    if (pt->dim == 3) build_steps(pt->ax[1], 16, ptrB, recB, false);
    std::vector<int> tab;
    const size_t oA0 = 0, oA1 = ptrA.size(), oB0 = oA1 + recA.size(), oB1 = oB0 + ptrB.size();
    tab.insert(tab.end(), ptrA.begin(), ptrA.end());
    tab.insert(tab.end(), recA.begin(), recA.end());
    tab.insert(tab.end(), ptrB.begin(), ptrB.end());
    tab.insert(tab.end(), recB.begin(), recB.end());
    IGX_HIP(hipMalloc(&pt->d_steps, std::max<size_t>(1, tab.size()) * sizeof(int)));
    IGX_HIP(hipMemcpyAsync(pt->d_steps, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipStreamSynchronize(pt->ctx->stream));    // `tab` is a local: no copy in flight when an early return below destroys it
    pt->stepA_ptr = pt->d_steps + oA0; pt->stepA_rec = pt->d_steps + oA1;
    pt->stepB_ptr = pt->d_steps + oB0; pt->stepB_rec = pt->d_steps + oB1;
    pt->h_pl0 = pl;                                  // the line descriptors of the stage-kernel final are built on first use
    // fused stage (fused.hip): a row of zeros, the one-dof outer axis of the 2D case, the mirror targets
    {
        const Axis &AL = pt->ax[pt->dim - 1];
        IGX_HIP(hipMalloc(&pt->d_zeros, ((size_t)AL.G + 256) * sizeof(double)));
        IGX_HIP(hipMemsetAsync(pt->d_zeros, 0, ((size_t)AL.G + 256) * sizeof(double), pt->ctx->stream));
        const int triv[6] = {0, 0, 0, 1, 0, 1};        // pl0 {0,0} | rp0 {0,1} | jlo0 {0} | jhi0 {1}
        IGX_HIP(hipMalloc(&pt->d_triv, sizeof(triv)));
        IGX_HIP(hipMemcpyAsync(pt->d_triv, triv, sizeof(triv), hipMemcpyHostToDevice, pt->ctx->stream));
        std::vector<int> tp;
        if (pt->dim == 3) {
            for (int i0 = pt->r0_lo; i0 < pt->r0_hi; ++i0)
                for (int j0 = i0; j0 < A0.jhi[i0]; ++j0) { tp.push_back(i0); tp.push_back(j0); }
        } else { tp.push_back(0); tp.push_back(0); }
        pt->ntp = (int)(tp.size() / 2);
        IGX_HIP(hipMalloc(&pt->d_tpairs, std::max<size_t>(1, tp.size()) * sizeof(int)));
        IGX_HIP(hipMemcpyAsync(pt->d_tpairs, tp.data(), tp.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
        IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    }
    IGX_HIP(hipMalloc(&pt->d_pl0, std::max<size_t>(1, pl.size()) * sizeof(int)));
    IGX_HIP(hipMalloc(&pt->d_rl0_of, std::max<size_t>(1, rl.size()) * sizeof(int)));
    IGX_HIP(hipMemcpyAsync(pt->d_pl0, pl.data(), pl.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipMemcpyAsync(pt->d_rl0_of, rl.data(), rl.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    return IGX_OK;
}

// Pair list and stage-A flush records of a non-symmetric form: every (i0, j0) of the owned rows i0.
static int prepare_nonsym(igx_patch *pt)
{
    if (pt->npairs0n >= 0) return IGX_OK;
    const Axis &A0 = pt->ax[0];
    std::vector<int> pl, rl(A0.S, -1);
    for (int i0 = std::max(0, pt->r0_lo); i0 < std::min(A0.N, pt->r0_hi); ++i0)
        for (int j0 = A0.jlo[i0]; j0 < A0.jhi[i0]; ++j0) {
            rl[A0.rp[i0] + (j0 - A0.jlo[i0])] = (int)(pl.size() / 2);
            pl.push_back(i0);
            pl.push_back(j0);
        }
    std::vector<int> rec;
    for (int s = 0; s < A0.n; ++s) {
        const int base = A0.fa[s];
        const int m = (s + 1 < A0.n) ? (A0.fa[s + 1] - base) : A0.P;
        for (int k = 0; k < m; ++k) {
            const int d = base + k;
            int r[16];
            for (int &v : r) v = -1;
            for (int a = 0; a < A0.P; ++a) {
                const int o = d + a;
                if (a > A0.P - 1 - k || o >= A0.N) continue;
                r[a] = rl[A0.rp[o] + (d - A0.jlo[o])];
                if (a > 0) r[8 + a] = rl[A0.rp[d] + (o - A0.jlo[d])];
            }
            rec.insert(rec.end(), r, r + 16);
        }
    }
    IGX_HIP(hipMalloc(&pt->d_pl0n, std::max<size_t>(1, pl.size()) * sizeof(int)));
    IGX_HIP(hipMalloc(&pt->d_stepsn, std::max<size_t>(1, rec.size()) * sizeof(int)));
    IGX_HIP(hipMemcpyAsync(pt->d_pl0n, pl.data(), pl.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipMemcpyAsync(pt->d_stepsn, rec.data(), rec.size() * sizeof(int), hipMemcpyHostToDevice, pt->ctx->stream));
    IGX_HIP(hipStreamSynchronize(pt->ctx->stream));
    if (int rc = build_qdesc(pt, pl, false, &pt->d_qdescn, &pt->n_qdescn)) return rc;
    pt->npairs0n = (int)(pl.size() / 2);
    return IGX_OK;
}

static int ensure(hipStream_t st, double **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return IGX_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    constexpr size_t PAD = 256;     // >= FINALQ_KWIN doubles: the last K window of k_final_q and the tile window of k_bf may read past the last line
    hipError_t e = hipMalloc(buf, (need + PAD) * sizeof(double));
    if (e != hipSuccess) {
        set_error("hipMalloc of %.2f GB sum-factorisation workspace failed: %s", need * 8.0 / 1e9, hipGetErrorString(e));
        return IGX_ERR_NOMEM;
    }
    // lines that are never produced (upper part of diagonal blocks) must stay finite: the final
    // stage multiplies window padding by exact zeros
    if (hipMemsetAsync(*buf, 0, (need + PAD) * sizeof(double), st) != hipSuccess) return IGX_ERR_HIP;   // same stream as the kernels
    *cap = need;
    return IGX_OK;
}

// split a sweep of `nspans` into chunks so that the launch has enough blocks to fill the chip;
// each chunk re-walks P-1 warm-up spans, so chunks are kept at least min_len = 4*P spans long (3D).  A 2D sweep has so few
// columns that the walk itself -- one dependent step per Gauss plane -- is the kernel's duration: there the chunks go
// down to P spans (twice the steps in total, a quarter of them in sequence).
static SweepChunks sweep_chunks(long long blocks_without, int nspans, int P, int min_len = 0)
{
    SweepChunks c{1, nspans};
    const long long want = 2048;
    if (min_len <= 0) min_len = 4 * P;
    if (blocks_without >= want || nspans < 2 * min_len) return c;
    int n = (int)std::min<long long>((want + blocks_without - 1) / blocks_without, nspans / min_len);
    n = std::max(n, 1);
    c.len = (nspans + n - 1) / n;
    c.nchunks = (nspans + c.len - 1) / c.len;
    return c;
}

// 2D k_geoA: a line of Gauss points gives few blocks (16 at C2), the walk along axis 0 is what takes the time -- chunks of a few
// spans each (every chunk re-walks P - 1 warm-up spans)
static int geoa2d_min_chunk(int P)
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("IGX_GEOA2D_CHUNK"); v = e ? atoi(e) : 0; }
    return v > 0 ? v : 2 * P;
}

#define DISPATCH_P(Pv, CALL)                                   \
    switch (Pv) {                                              \
    case 2: { constexpr int PP = 2; CALL; } break;             \
    case 3: { constexpr int PP = 3; CALL; } break;             \
    case 4: { constexpr int PP = 4; CALL; } break;             \
    case 5: { constexpr int PP = 5; CALL; } break;             \
    case 6: { constexpr int PP = 6; CALL; } break;             \
    default: set_error("sum factorisation: degree %d unsupported", (Pv) - 1); return IGX_ERR_UNSUPPORTED; }

// The fused sweep + final stage (fused.hip) needs single interior knots, equal degrees and q = p + 1 on the swept and
// the last axis.  It is the default in 3D (C4: 12.1 + 3.7 ms against 6.9 + 9.4..10.7 ms for stage B + final, K2 never in
// HBM); in 2D the unfused kernels are faster (C2: 0.105 against 0.141 ms) and stay the default.  IGX_PATH=fused / unfused
// forces a path (a choice of IGX_FINAL implies unfused); the unfused kernels also serve every other case.
// geometry + stage A in one kernel (IGX_GEOA=0: separate field and sweep kernels)
static bool geoA_wanted(const igx_patch *pt, int kind, int nslots)
{
    if (!pt->knobs.geoa) return false;
    return (igx_kind_symmetric(kind) || kind == IGX_CONVDIFF) && geoA_supported(pt, kind, nslots);
}
static bool fused_applicable(const igx_patch *pt);
static bool fused3_axes(const igx_patch *pt, bool sym);

// The single launch beats the stage-kernel chain while its grid is one resident round of tiles of at most 6 x 6 rows
// (a block's duration grows with its tile: 14 us at 2 x 4 rows, 27 us at 6 x 6, 39 us at 8 x 8 against 28-34 us of
// the chain -- profiles/r03_single2d.txt)
constexpr long long SINGLE2D_MAX_BLOCKS = 256;
constexpr int SINGLE2D_MAX_TILE = 36;
static bool single2d_wanted(const igx_patch *pt, int kind)
{
    // 2D: the single-launch kernel where the stage kernels are launch-bound (small patches), or on request (IGX_PATH=single)
    if (pt->knobs.final_sel || !single2d_supported(pt, kind)) return false;
    if (pt->knobs.path == 3) return true;
    if (pt->knobs.path != 0) return false;
    // decided on the WHOLE patch, not on the resident row slab: every slab of a patch takes the same path (the two paths sum
    // in different orders; the slabs of a patch reproduce its rows bit for bit)
    int rows = 0;
    const long long nb = single2d_blocks(pt, kind, &rows, true);
    return nb >= 0 && nb <= SINGLE2D_MAX_BLOCKS && rows <= SINGLE2D_MAX_TILE;
}

bool form_on_fast_chain(const igx_patch *pt);
bool sumfact_single_launch(const igx_patch *pt, int kind) { return single2d_wanted(pt, kind); }
// Repeated knots on the last axis only: the patch is assembled through its axis-exchanged twin (igx_internal.h, igx_patch::twin),
// whose fast chain serves the kinds of twin_kinds (sumfact_twin_kinds, asked once when the twin is created).
static bool twin_route(const igx_patch *pt, int kind)
{
    if (!pt->twin || pt->is_twin) return false;
    // a form given as a table of expressions (igx_patch_set_form_expr hands it to the twin as well): where the twin's fast chain takes it
    if (kind == IGX_FORM) return pt->twin->ftab.valid && pt->ftab.valid && !pt->dev.form_par && form_on_fast_chain(pt->twin);
    return kind >= 0 && kind < 31 && ((pt->twin_kinds >> kind) & 1);
}
int sumfact_twin_kinds(const igx_patch *tw)
{
    if (tw->dim != 3 || !tw->sumfact_ok || tw->knobs.path == 2 || tw->knobs.final_sel || tw->knobs.bf == 2) return 0;
    if (getenv("IGX_NO_TWIN")) return 0;                 // (experiments: the stage kernels for such patches, as before round 6)
    const Axis &AM = tw->ax[1], &AL = tw->ax[2];
    if (!fused3_axes(tw, true) || !fused3_tr_fits(tw->ax[0].p, AM.p, AL.p, AM.S, AL.S)) return 0;
    int kinds = 0;
    if (geoA_wanted(tw, IGX_MASS, 1)) kinds |= 1 << IGX_MASS;
    if (geoA_wanted(tw, IGX_STIFFNESS, 8)) kinds |= 1 << IGX_STIFFNESS;
    // the convection-diffusion form: its coefficient follows the patch's (igx_api.hip, twin_coeff)
    if (fused3_axes(tw, false) && geoA_wanted(tw, IGX_CONVDIFF, 8)) kinds |= 1 << IGX_CONVDIFF;
    return kinds;
}

bool sumfact_needs_fields(const igx_patch *pt, int kind)
{
    if (twin_route(pt, kind)) return false;
    if (single2d_wanted(pt, kind)) return false;
    if (kind == IGX_FORM && pt->dim == 3 && form_on_fast_chain(pt)) return false;
    if (pt->dim == 2) return !(igx_kind_symmetric(kind) && pt->knobs.path != 1 && geoA_wanted(pt, kind, kind == IGX_MASS ? 1 : 4));
    if (pt->dim != 3) return true;
    // the convection-diffusion form: its eight merged slots exist where the fused stage runs (sumfact_assemble)
    if (!igx_kind_symmetric(kind)) return !(kind == IGX_CONVDIFF && (fused_applicable(pt) || fused3_axes(pt, false)) && geoA_wanted(pt, kind, 8));
    return !geoA_wanted(pt, kind, kind == IGX_MASS ? 1 : 8);
}

// k_bf3 (fused3.hip): single knots on the LAST axis only; the swept axis may have repeated knots; the two degrees may differ
// (symmetric 3D forms) and the Gauss points per span are the patch's nqp.  Decided on the whole axes: all slabs of a patch agree.
static bool fused3_axes(const igx_patch *pt, bool sym)
{
    if (pt->knobs.bf == 2 || pt->knobs.path == 2 || pt->knobs.final_sel) return false;
    const int dim = pt->dim;
    if (dim == 2 && (pt->knobs.path != 1 || !sym)) return false;
    const Axis &AM = pt->ax[dim - 2], &AL = pt->ax[dim - 1];
    if (!AL.simple || AM.q != AL.q || !fused3_degrees(AM.P, AL.P, AL.q, sym && dim == 3, AM.simple)) return false;
    if (dim == 2 && !AM.simple) return false;
    return fused_offsets_fit(dim == 3 ? 2 * pt->ax[0].p + 1 : 1, AM.S, AL.S, AM.G, AL.G) &&
           fused3_offsets_fit(dim, dim == 3 ? pt->ax[0].p : 0, AM.p, AL.p, AM.S, AL.S, AL.N);
}

static bool fused_applicable(const igx_patch *pt)
{
    if (pt->knobs.path == 2 || pt->knobs.final_sel) return false;
    const int dim = pt->dim;
    if (dim == 2 && pt->knobs.path != 1) return false;
    const Axis &AM = pt->ax[dim - 2], &AL = pt->ax[dim - 1];
    if (!AM.simple || !AL.simple || AM.q != AM.P || AL.q != AL.P || AM.P != AL.P || AL.P < 2 || AL.P > 6) return false;
    // 32-bit offsets inside a row block of an outer row and inside a K1 slice (k_bf2, k_mirror2): larger patches take the
    // stage kernels.  Decided on the whole axes, like every path choice: all slabs of a patch agree.
    return fused_offsets_fit(dim == 3 ? 2 * pt->ax[0].p + 1 : 1, AM.S, AL.S, AM.G, AL.G);
}

// slot table of the fused stage: input array `ptr` enters the sweep with mid-axis type t1 and last-axis type y
static bool bf_add_slot(BFInputs &in, int y, int t1, const double *ptr)
{
    int &n = in.slot_n[y][t1];
    if (n >= 2) return false;
    in.slot_ptr[y][t1][n++] = ptr;
    return true;
}

static int run_fused(igx_patch *pt, BFInputs &in, bool sym, double *d_data, bool use3)
{
    hipStream_t st = pt->ctx->stream;
    const int dim = pt->dim;
    const Axis &A0 = pt->ax[0];
    in.mid = &pt->ax[dim - 2]; in.last = &pt->ax[dim - 1];
    in.zeros = pt->d_zeros; in.sym = sym ? 1 : 0;
    if (dim == 3) { in.rp0 = A0.dev.rp; in.jlo0 = A0.dev.jlo; in.jhi0 = A0.dev.jhi; }
    else { in.pl0 = pt->d_triv; in.npairs = 1; in.rp0 = pt->d_triv + 2; in.jlo0 = pt->d_triv + 4; in.jhi0 = pt->d_triv + 5; }
    int i1_lo = 0, i1_hi = in.mid->N;
    if (dim == 3) { in.mid_lo = 0; in.mid_hi = in.mid->N; in.span_hi = in.mid->n; }
    else {
        // 2D: the swept axis carries the row slab; symmetric forms also produce the lower entries of the p halo rows
        // above it (mirror sources), as far as the resident spans reach
        i1_lo = pt->r0_lo; i1_hi = pt->r0_hi;
        in.mid_lo = pt->r0_lo; in.mid_hi = sym ? std::min(pt->r0_hi + A0.p, A0.N) : pt->r0_hi; in.span_hi = pt->s0_hi;
    }
    // k_bf3 (fused3.hip): symmetric forms get their upper triangle from the same registers as the lower one -- no mirror pass
    in.tr = pt->is_twin ? 1 : 0;
    if (pt->is_twin && !use3) { set_error("internal: the axis-exchanged twin of a patch left k_bf3"); return IGX_ERR_UNSUPPORTED; }
    if (use3) {
        if (dim == 2 && sym) in.mid_hi = pt->r0_hi;      // (no mirror sources above the slab)
        if (int rc = launch_bf3(st, pt, in, d_data)) return rc;
        pt->last_path |= IGX_PATH_FUSED | IGX_PATH_BF3 | (sym ? IGX_PATH_BOTH : 0);
        pt->timing.n_launches++;
        stage_event(pt, 3, st);
        stage_event(pt, 4, st);
        return IGX_OK;
    }
    if (int rc = launch_bf(st, pt, in, d_data)) return rc;
    pt->last_path |= IGX_PATH_FUSED;
    pt->timing.n_launches++;
    stage_event(pt, 3, st);
#ifdef IGX_ABLATE
    if (sym && !getenv("IGX_NO_MIRROR")) {
#else
    if (sym) {
#endif
        MirrorInputs mi{};
        mi.mid = in.mid; mi.last = in.last; mi.rp0 = in.rp0; mi.jlo0 = in.jlo0; mi.jhi0 = in.jhi0;
        mi.tpairs = pt->d_tpairs; mi.ntp = pt->ntp; mi.i1_lo = i1_lo; mi.i1_hi = i1_hi;
        if (int rc = launch_mirror(st, pt, mi, d_data)) return rc;
        pt->last_path |= IGX_PATH_MIRROR;
        pt->timing.n_launches++;
    }
    stage_event(pt, 4, st);
    return IGX_OK;
}

// The mirror pass gathers 72-byte runs all over the CSR values and takes 3.3 - 4.2 ms at C4 depending on where the driver put
// the 12.75 GB physically -- a property of the BUFFER, stable over its life.  A caller that opts in
// (IGX_PLACEMENT_TRIES) lets the first assembly time the pass on a few candidate buffers and keep the fastest.
float sumfact_probe_mirror(igx_patch *pt, double *buf)
{
    if (pt->dim != 3 || !fused_applicable(pt) || pt->ntp == 0) return -1.0f;
    hipStream_t st = pt->ctx->stream;
    const Axis &A0 = pt->ax[0];
    MirrorInputs mi{};
    mi.mid = &pt->ax[1]; mi.last = &pt->ax[2]; mi.rp0 = A0.dev.rp; mi.jlo0 = A0.dev.jlo; mi.jhi0 = A0.dev.jhi;
    mi.tpairs = pt->d_tpairs; mi.ntp = pt->ntp; mi.i1_lo = 0; mi.i1_hi = pt->ax[1].N;
    hipEvent_t *ev = pt->ctx->ev;
    float best = -1.0f;
    for (int rep = 0; rep < 3; ++rep) {                  // (the first launch also pages the buffer in)
        if (hipEventRecord(ev[6], st) != hipSuccess) return -1.0f;
        if (launch_mirror(st, pt, mi, buf)) return -1.0f;
        if (hipEventRecord(ev[7], st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1.0f;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev[6], ev[7]) != hipSuccess) return -1.0f;
        if (rep > 0 && (best < 0.0f || ms < best)) best = ms;
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// General first-order forms on the fast chain (round 6).  A form given as a coefficient table (igx_patch_set_form_expr: constants
// and expressions in the physical coordinates) needs no field arrays: k_geoA<FORM = 2 | 3> evaluates F = W T P T^T at every point
// inside the axis-0 sweep -- the constants ride in the kernel arguments, an expression is sampled once into an array of its own
// (the generated coefficient kernel of rtc.hip) -- and sums the up to four (axis-0 type, field) sources of every (mid, last)-type
// slot into ONE K1 array, which is what k_bf3 wants.  A symmetric table takes the symmetric chain: lower pairs of axis 0, both
// triangles from k_bf3<SYM = 2>, exactly symmetric result -- half the work of the reference's default for such a form.
// Reference: pyiga/vform.py:705-731 (finalize), pyiga/codegen/cython.py:325-387 (the generated combine), assemble.py:837-897.
struct FormPlan {
    bool ok = false, sym = false;
    GeoAForm g;
    int narr = 0;
    int slot_arr[4][4];                                  // (last-axis type y, mid-axis type t1) -> K1 array, or -1
    bool mass_only = false;
};

static FormPlan form_table_plan(const igx_patch *pt)
{
    FormPlan fp;
    const igx_patch::FormTable &T = pt->ftab;
    if (!T.valid || pt->dim != 3 || pt->dev.form_par || !geoA_form_supported(pt) || pt->knobs.path == 2 || pt->knobs.final_sel) return fp;
    fp.sym = T.sym;
    if (!fused3_axes(pt, fp.sym)) return fp;
    memset(&fp.g, 0, sizeof(fp.g));
    fp.g.sym = fp.sym ? 1 : 0;
    for (int y = 0; y < 4; ++y) for (int t1 = 0; t1 < 4; ++t1) fp.slot_arr[y][t1] = -1;
    for (int k = 0; k < 16; ++k) { fp.g.fslot[k] = -1; fp.g.pc[k] = ((T.is_const >> k) & 1) ? T.cval[k] : 0.0; fp.g.pa[k] = nullptr; }
    fp.g.pmask = T.present;
    // parametric entries F_ab that do not vanish identically, their canonical representative and its place among the fields
    auto has = [&](int r, int c) { return ((T.present >> (4 * r + c)) & 1) != 0; };
    bool any00 = has(0, 0), any0s = false, anyr0 = false, anyrs = false;
    for (int k = 1; k < 4; ++k) {
        any0s = any0s || has(0, k); anyr0 = anyr0 || has(k, 0);
        for (int c = 1; c < 4; ++c) anyrs = anyrs || has(k, c);
    }
    int nf = 0;
    struct Src { int t0, f; bool operator<(const Src &o) const { return t0 != o.t0 ? t0 < o.t0 : f < o.f; } bool operator==(const Src &o) const { return t0 == o.t0 && f == o.f; } };
    std::vector<Src> slot_src[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const bool nz = a == 0 ? (b == 0 ? any00 : any0s) : (b == 0 ? anyr0 : anyrs);
            if (!nz) continue;
            int ca = a, cb = b;                              // canonical entry: F_ab = F_ba where the table says so
            if (a > b && (T.sym || (T.blocksym && b >= 1))) { ca = b; cb = a; }
            if (fp.g.fslot[4 * ca + cb] < 0) {
                if (nf >= (fp.sym ? 10 : 13)) return fp;     // (fields of a point in LDS: GA_NFT, geoa.hip)
                fp.g.fslot[4 * ca + cb] = nf++;
            }
            // a: test function v, b: trial function u; jet index c >= 1 differentiates grid axis 3 - c (form_terms)
            int t[3];
            for (int ax = 0; ax < 3; ++ax) t[ax] = ((b >= 1 && ax == 3 - b) ? 1 : 0) + 2 * ((a >= 1 && ax == 3 - a) ? 1 : 0);
            slot_src[t[2]][t[1]].push_back(Src{t[0], fp.g.fslot[4 * ca + cb]});
        }
    // (the entries that are only representatives of others must still be evaluated: fslot is set for canonical ones only)
    std::vector<std::vector<Src>> arrs;
    int mask = 0;
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1) {
            std::vector<Src> &v = slot_src[y][t1];
            if (v.empty()) continue;
            if (v.size() > 4) return fp;
            std::sort(v.begin(), v.end());
            mask |= 1 << (4 * y + t1);
            int found = -1;
            for (size_t x = 0; x < arrs.size(); ++x)
                if (arrs[x] == v) found = (int)x;
            if (found < 0) { found = (int)arrs.size(); arrs.push_back(v); }
            fp.slot_arr[y][t1] = found;
        }
    if (arrs.empty() || arrs.size() > 8) return fp;
    // arrays with more sources first: array x is swept by wave x, the waves of a workgroup go to the four SIMDs cyclically
    std::vector<int> order(arrs.size());
    for (size_t x = 0; x < arrs.size(); ++x) order[x] = (int)x;
    std::stable_sort(order.begin(), order.end(), [&](int i, int j) { return arrs[i].size() > arrs[j].size(); });
    std::vector<int> newidx(arrs.size());
    for (size_t x = 0; x < arrs.size(); ++x) newidx[order[x]] = (int)x;
    for (int y = 0; y < 4; ++y) for (int t1 = 0; t1 < 4; ++t1) if (fp.slot_arr[y][t1] >= 0) fp.slot_arr[y][t1] = newidx[fp.slot_arr[y][t1]];
    for (size_t x = 0; x < arrs.size(); ++x) {
        const std::vector<Src> &v = arrs[order[x]];
        fp.g.nsrc[x] = (int)v.size();
        for (size_t k = 0; k < v.size(); ++k) { fp.g.stype[x][k] = v[k].t0; fp.g.sfield[x][k] = v[k].f; }
    }
    fp.narr = (int)arrs.size();
    fp.mass_only = mask == 1;
    if (!fp.sym && fp.mass_only) return fp;                  // (u v alone is symmetric; a non-symmetric table cannot get here)
    fp.ok = true;
    return fp;
}

bool form_on_fast_chain(const igx_patch *pt) { return form_table_plan(pt).ok; }

static int run_fused(igx_patch *pt, BFInputs &in, bool sym, double *d_data, bool use3);
static int assemble_form_table(igx_patch *pt, FormPlan &fp, double *d_data)
{
    hipStream_t st = pt->ctx->stream;
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1], &A2 = pt->ax[2];
    const long long NPL = (long long)A1.G * A2.G;
    const bool sym = fp.sym;
    if (!sym && prepare_nonsym(pt)) return IGX_ERR_HIP;
    const int np0 = sym ? pt->npairs0 : pt->npairs0n;
    const int *d_pl0 = sym ? pt->d_pl0 : pt->d_pl0n;
    if (np0 == 0) return IGX_OK;
    // the entries of the table that are functions: sampled once per patch by the generated coefficient kernel
    igx_patch::FormTable &T = pt->ftab;
    const size_t npts = (size_t)pt->dev.npts_loc;
    if (T.narr > 0 && !pt->ftab_ready) {
        std::vector<const char *> list((size_t)T.narr, nullptr);
        for (int k = 0; k < 16; ++k)
            if (T.arr_of[k] >= 0 && !list[(size_t)T.arr_of[k]]) list[(size_t)T.arr_of[k]] = T.expr[k].c_str();
        if (!pt->ftab_arr && hipMalloc((void **)&pt->ftab_arr, std::max<size_t>(1, (size_t)T.narr * npts) * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError(); pt->ftab_arr = nullptr;
            set_error("hipMalloc of %.2f GB for the sampled coefficients of the form failed", T.narr * npts * 8.0 / 1e9);
            return IGX_ERR_NOMEM;
        }
        int hit = 0;
        if (int rc = launch_form_exprs(st, pt, T.narr, list.data(), pt->ftab_arr, &hit)) return rc;
        pt->ftab_ready = true;
    }
    for (int k = 0; k < 16; ++k) fp.g.pa[k] = T.arr_of[k] >= 0 ? pt->ftab_arr + (size_t)T.arr_of[k] * npts : nullptr;
    if (ensure(st, &pt->d_K1, &pt->K1_cap, (size_t)fp.narr * np0 * NPL)) return IGX_ERR_NOMEM;
    stage_event(pt, 1, st);
    double *so[8];
    for (int x = 0; x < fp.narr; ++x) so[x] = pt->d_K1 + (size_t)x * np0 * NPL;
    const SweepChunks ch = sweep_chunks((NPL + 63) / 64, pt->s0_hi - pt->s0_lo, A0.P);
    if (int rc = launch_geoA(st, pt, IGX_FORM, fp.narr, nullptr, nullptr, so, NPL, ch.len, ch.nchunks, nullptr, nullptr, &fp.g)) return rc;
    pt->last_path |= IGX_PATH_GEOA;
    pt->timing.n_launches++;
    stage_event(pt, 2, st);
    BFInputs in{};
    bool ok = true;
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1)
            if (fp.slot_arr[y][t1] >= 0) ok = ok && bf_add_slot(in, y, t1, pt->d_K1 + (size_t)fp.slot_arr[y][t1] * np0 * NPL);
    in.sym = sym ? 1 : 0;
    in.pad_stiff3 = fp.mass_only ? 0 : 1;
    if (!ok || !fused3_supported(in)) { set_error("internal: the slots of the form do not fit the fused stage"); return IGX_ERR_UNSUPPORTED; }
    in.slice_stride = NPL; in.gmid_lo = 0; in.pl0 = d_pl0; in.npairs = np0;
    return run_fused(pt, in, sym, d_data, true);
}

int sumfact_assemble(igx_patch *pt, int kind, double *d_data)
{
    hipStream_t st = pt->ctx->stream;
    const int dim = pt->dim;
    const PatchDev &pd = pt->dev;
    if (twin_route(pt, kind)) {                           // repeated knots on the last axis: the twin's chain, values to THIS layout
        igx_patch *tw = pt->twin;
        memset(&tw->timing, 0, sizeof(tw->timing));
        tw->last_path = 0;
        const int rc = sumfact_assemble(tw, kind, d_data);
        pt->last_path |= tw->last_path | IGX_PATH_TWIN;
        pt->timing.n_launches += tw->timing.n_launches;
        return rc;
    }
    if (kind == IGX_FORM && dim == 3) {                   // a coefficient table on the fast chain: no field arrays
        FormPlan fp = form_table_plan(pt);
        if (fp.ok) return assemble_form_table(pt, fp, d_data);
    }
    std::vector<Term> terms = form_terms(dim, kind, &pd);
    const Axis &A0 = pt->ax[0], &A1 = pt->ax[1], &A2 = pt->ax[2];
    const long long NPL = (long long)A1.G * (dim == 3 ? A2.G : 1);
    const bool sym = igx_kind_symmetric(kind);
    if (!sym && prepare_nonsym(pt)) return IGX_ERR_HIP;
    const int np0 = sym ? pt->npairs0 : pt->npairs0n;
    const int *d_pl0 = sym ? pt->d_pl0 : pt->d_pl0n;
    if (np0 == 0) return IGX_OK;
    if (single2d_wanted(pt, kind)) {
        // one launch between the caller's first and last event: no stage events (a marker costs a few microseconds of
        // stream time, as much as this kernel on a small patch); igx_assemble reports the whole interval as stage 1
        if (int rc = launch_single2d(st, pt, kind, d_data)) return rc;
        pt->last_path |= IGX_PATH_SINGLE;
        return IGX_OK;
    }
    const bool axes3 = fused3_axes(pt, sym);
    const bool fused = fused_applicable(pt) || axes3;
    if (fused && dim == 2) {
        // 2D: the fields ARE the sweep input (axis 0 swept, axis 1 contracted by the contractors): one kernel + mirror
        BFInputs in{};
        bool ok = true;
        for (const Term &t : terms)
            ok = ok && bf_add_slot(in, kind == IGX_MASS ? 0 : t.t[1], t.t[0], pt->d_fields + (size_t)t.f * pd.npts_loc);
        in.sym = sym ? 1 : 0;
        const bool can3 = ok && axes3 && fused3_supported(in), can2 = ok && fused_applicable(pt) && fused_supported(in);
        if (can3 || can2) {
            in.slice_stride = 0; in.gmid_lo = pd.g0_lo;
            stage_event(pt, 1, st);
            stage_event(pt, 2, st);
            return run_fused(pt, in, sym, d_data, can3);
        }
    }

    // ---- stage-A arrays X = unique (t0, f).  In 2D the final stage wants the arrays ordered by
    // the last-axis type of their (single) consuming term; in 3D any order works.
    struct XA { int t0, f, slot, key, xt0, xf, alias; };
    std::vector<XA> X;
    std::vector<int> term_x(terms.size());
    // Fused stage with the stage-A kernel (non-symmetric 3D forms): one K1 array per SLOT (last-axis type, mid-axis type) of the
    // sweep, holding the sum of the (at most two) terms of the slot -- the later stages cannot tell them apart.  Fewer arrays
    // than unique (t0, field) pairs when terms collide (convection-diffusion: 9 instead of 11), and one input per slot.
    bool merged = fused && dim == 3 && !sym;
    if (merged) {                                        // only where the fused stage has a kernel for the set of slots
        BFInputs probe{};
        for (const Term &t : terms) probe.slot_n[kind == IGX_MASS ? 0 : t.t[2]][t.t[1]] = 1;
        merged = fused_supported(probe) != 0;
    }
    if (merged) {
        for (size_t i = 0; i < terms.size() && merged; ++i) {
            const int key = 4 * (kind == IGX_MASS ? 0 : terms[i].t[2]) + terms[i].t[1];
            int found = -1;
            for (size_t x = 0; x < X.size(); ++x)
                if (X[x].key == key) found = (int)x;
            if (found < 0) { found = (int)X.size(); X.push_back(XA{terms[i].t[0], terms[i].f, found, key, -1, -1, 0}); }
            else if (X[found].xf < 0) { X[found].xt0 = terms[i].t[0]; X[found].xf = terms[i].f; }
            else merged = false;                         // three terms in one slot: the general path
            term_x[i] = found;
        }
        if (merged) {                                    // every field may lead at most two arrays (k_stageA groups)
            for (int f = 0; f < 16 && merged; ++f) {
                int n = 0;
                for (const XA &x : X) n += x.f == f;
                if (n > 2) merged = false;
            }
        }
        if (!merged) X.clear();
        else {
            // slots with identical sources share ONE array (stiffness: B12 with axis-0 type 0 feeds two slots); the arrays
            // are numbered over the distinct ones, key < 0 marks an alias that stage A does not produce again
            int na = 0;
            for (size_t a = 0; a < X.size(); ++a) {
                int same = -1;
                for (size_t b = 0; b < a; ++b)
                    if (X[b].t0 == X[a].t0 && X[b].f == X[a].f && X[b].xt0 == X[a].xt0 && X[b].xf == X[a].xf) { same = (int)b; break; }
                if (same >= 0) { X[a].slot = X[same].slot; X[a].alias = 1; }
                else { X[a].slot = na++; X[a].alias = 0; }
            }
            // arrays with two sources first: in k_geoA (non-symmetric) array x is swept by wave x, a workgroup's waves go to the
            // four SIMDs cyclically, and a two-source array takes twice the sweep -- at most one of them per SIMD
            std::vector<int> perm(na, -1);
            int nxt = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (const XA &x : X)
                    if (!x.alias && (x.xf >= 0) == (pass == 0)) perm[x.slot] = nxt++;
            for (XA &x : X) x.slot = perm[x.slot];
        }
    }
    for (size_t i = 0; i < terms.size() && !merged; ++i) {
        int found = -1;
        if (dim == 3)
            for (size_t x = 0; x < X.size(); ++x)
                if (X[x].t0 == terms[i].t[0] && X[x].f == terms[i].f) found = (int)x;
        if (found < 0) {
            found = (int)X.size();
            X.push_back(XA{terms[i].t[0], terms[i].f, (dim == 2 && kind != IGX_FORM) ? (kind == IGX_MASS ? 0 : terms[i].t[1]) : found, -1, -1, -1, 0});
        }
        term_x[i] = found;
    }
    int nX = 0;                                          // K1 arrays (aliases of the merged slots do not count)
    for (const XA &x : X) nX = std::max(nX, x.slot + 1);
    if (!merged || dim == 2) nX = (int)X.size();
    // K1 slice stride: padded when both producer (geoA) and consumer (k_bf) take a stride (experiment: IGX_K1PAD doubles)
    const bool use_geoA = (sym || merged) && geoA_wanted(pt, kind, nX);
    long long NPLs = NPL;
#ifdef IGX_ABLATE
    if (use_geoA && fused && dim == 3) { const char *e = getenv("IGX_K1PAD"); NPLs = NPL + (e ? atoi(e) : 0); }
#endif
    if (ensure(st, &pt->d_K1, &pt->K1_cap, (size_t)nX * np0 * NPLs)) return IGX_ERR_NOMEM;

    const int nF = igx_num_fields(dim, kind, pd.form_n);
    stage_event(pt, 1, st);
    if (use_geoA) {
        // geometry evaluated inside the sweep: no field arrays (geoa.hip)
        int sf[8], stp[8], sxf[8], sxt[8];
        double *so[8];
        for (const XA &xa : X) {
            if (xa.alias) continue;                      // (merged slots with identical sources share one array)
            const int x = merged ? xa.slot : (int)(&xa - X.data());
            sf[x] = xa.f; stp[x] = xa.t0; sxf[x] = xa.xf; sxt[x] = xa.xt0 < 0 ? 0 : xa.xt0;
            so[x] = pt->d_K1 + (size_t)xa.slot * np0 * NPLs;
        }
        const SweepChunks ch = sweep_chunks((NPL + 63) / 64, pt->s0_hi - pt->s0_lo, A0.P, dim == 2 ? geoa2d_min_chunk(A0.P) : 0);
        int rc = launch_geoA(st, pt, kind, nX, sf, stp, so, NPLs, ch.len, ch.nchunks, sxf, sxt);
        if (rc) return rc;
        pt->last_path |= IGX_PATH_GEOA;
        pt->timing.n_launches++;
    } else
    // one launch for all fields (blockIdx.y); the types of a field share the field load
    {
        StageAArgs A{};
        int ng = 0;
        const bool one_type = !sym && A0.P >= 5 && A0.q == A0.P;      // register budget of the full pair window (k_stageA)
        for (int f = 0; f < nF; ++f) {
            StageAGroup g{};
            for (size_t x = 0; x < X.size(); ++x)
                if (X[x].f == f && !X[x].alias) {
                    if (g.nt == 2) { set_error("internal: more than two stage-A types per field"); return IGX_ERR_UNSUPPORTED; }
                    double *o = pt->d_K1 + (size_t)X[x].slot * np0 * NPL;
                    if (g.nt == 0) { g.t0 = X[x].t0; g.out0 = o; } else { g.t1 = X[x].t0; g.out1 = o; }
                    if (X[x].xf >= 0) { g.xfield[g.nt] = pt->d_fields + (size_t)X[x].xf * pd.npts_loc; g.xt[g.nt] = X[x].xt0; }
                    g.nt++;
                }
            if (g.nt == 0) continue;
            g.field = pt->d_fields + (size_t)f * pd.npts_loc;
            if (one_type && g.nt == 2) {                 // two groups of one type each (see k_stageA)
                StageAGroup g1 = g;
                g1.t0 = g.t1; g1.out0 = g.out1; g1.xfield[0] = g.xfield[1]; g1.xt[0] = g.xt[1];
                g1.nt = g.nt = 1; g1.xfield[1] = g.xfield[1] = nullptr;
                if (ng >= 15) { set_error("internal: too many stage-A groups"); return IGX_ERR_UNSUPPORTED; }
                A.grp[ng++] = g1;
            }
            if (ng >= 16) { set_error("internal: too many stage-A groups"); return IGX_ERR_UNSUPPORTED; }
            A.grp[ng++] = g;
        }
        A.PI0 = A0.d_PI; A.step_ptr = pt->stepA_ptr; A.steps = sym ? pt->stepA_rec : pt->d_stepsn;
        A.s_lo = pt->s0_lo; A.s_hi = pt->s0_hi; A.n0 = A0.n; A.N0 = A0.N; A.q = A0.q; A.g0_lo = pd.g0_lo;
        A.NPL = NPL;
        const int bsA = 256;
        const long long bx = (NPL + bsA - 1) / bsA;
        const SweepChunks ch = sweep_chunks(bx * ng, pt->s0_hi - pt->s0_lo, A0.P, dim == 2 ? A0.P : 0);
        A.chunk_len = ch.len;
        const size_t ldsA = (size_t)2 * A0.q * 4 * ((A0.P * A0.P + 1) & ~1) * sizeof(double);
        if ((size_t)A0.q * 4 * A0.P * A0.P > (size_t)SWEEP_MAX_STAGE * bsA) { set_error("stage A: coefficient slice too large"); return IGX_ERR_UNSUPPORTED; }
        dim3 block(bsA), grid((unsigned)bx, ng, ch.nchunks);
        if (A0.P >= 7) stageA_hi(A0.P, st, A, A0.q == A0.P, sym, one_type, dim == 2, grid, block, ldsA);
        else DISPATCH_P(A0.P, launch_stageA<PP>(st, A, A0.q == A0.P, sym, one_type, dim == 2, grid, block, ldsA));
        IGX_HIP(hipGetLastError());
        pt->timing.n_launches++;
    }
    stage_event(pt, 2, st);

    if (fused && dim == 3) {
        BFInputs in{};
        bool ok = true;
        if (merged) {
            for (const XA &x : X) ok = ok && bf_add_slot(in, x.key >> 2, x.key & 3, pt->d_K1 + (size_t)x.slot * np0 * NPLs);
        } else
        for (size_t i = 0; i < terms.size(); ++i)
            ok = ok && bf_add_slot(in, kind == IGX_MASS ? 0 : terms[i].t[2], terms[i].t[1],
                                   pt->d_K1 + (size_t)X[term_x[i]].slot * np0 * NPLs);
        in.sym = sym ? 1 : 0;
        const bool can3 = ok && axes3 && fused3_supported(in), can2 = ok && fused_applicable(pt) && fused_supported(in);
        if (can3 || can2) {
            in.slice_stride = NPLs; in.gmid_lo = 0; in.pl0 = d_pl0; in.npairs = np0;
            return run_fused(pt, in, sym, d_data, can3);
        }
    }
    if (pt->is_twin) { set_error("internal: the axis-exchanged twin of a patch left the fused chain"); return IGX_ERR_UNSUPPORTED; }

    if (int rc = ensure_line_descriptors(pt)) return rc;
    // ---- final-stage input
    FinalArgs F{};
    const Axis &AL = (dim == 3) ? A2 : A1;
    int NY;
    const double *Kfinal = nullptr;
    long long ngroups = 0;
    if (dim == 3) {
        // stage B groups by the last-axis type y = t2
        StageBArgs B{};
        int ymax = 0;
        for (auto &g : B.grp) g.nterm = 0;
        for (size_t i = 0; i < terms.size(); ++i) {
            const int y = (kind == IGX_MASS) ? 0 : terms[i].t[2];
            StageBGroup &g = B.grp[y];
            if (g.nterm >= 9) { set_error("internal: too many stage-B terms"); return IGX_ERR_UNSUPPORTED; }
            g.x[g.nterm] = X[term_x[i]].slot;
            g.t1[g.nterm] = terms[i].t[1];
            g.nterm++;
            ymax = std::max(ymax, y);
        }
        NY = (ymax == 0) ? 1 : 4;                       // the final kernels exist for 1 and 4 K arrays
        if (ensure(st, &pt->d_K2, &pt->K2_cap, (size_t)NY * np0 * A1.S * A2.G)) return IGX_ERR_NOMEM;
        for (int y = 0; y < NY; ++y)                    // a type of the last axis without terms (general forms): zeros
            if (B.grp[y].nterm == 0)
                IGX_HIP(hipMemsetAsync(pt->d_K2 + (size_t)y * np0 * A1.S * A2.G, 0, (size_t)np0 * A1.S * A2.G * sizeof(double), st));
        B.PI1 = A1.d_PI;
        B.step_ptr = pt->stepB_ptr; B.steps = pt->stepB_rec; B.pl0 = d_pl0; B.symmetric = sym;
        B.n1 = A1.n; B.N1 = A1.N; B.q = A1.q; B.G1 = A1.G; B.G2 = A2.G; B.S1 = A1.S; B.npairs0 = np0;
        const int bs = A1.P >= 7 ? 256 : 128;            // (a span's coefficient slice is staged by the block: q * 4 * P * P values)
        const long long bxB = (A2.G + bs - 1) / bs;
        const SweepChunks ch = sweep_chunks(bxB * np0 * NY, A1.n, A1.P);
        B.ngroups = NY; B.chunk_len = ch.len;
        const size_t ldsB = (size_t)2 * A1.q * 4 * ((A1.P * A1.P + 1) & ~1) * sizeof(double);
        if ((size_t)A1.q * 4 * A1.P * A1.P > (size_t)SWEEP_MAX_STAGE * bs) { set_error("stage B: coefficient slice too large"); return IGX_ERR_UNSUPPORTED; }
        dim3 block(bs), grid((unsigned)bxB, np0, NY * ch.nchunks);
        if (np0 > 65535 || NY * ch.nchunks > 65535) { set_error("stage B: grid too large"); return IGX_ERR_UNSUPPORTED; }
        if (A1.P >= 7) stageB_hi(A1.P, st, pt->d_K1, pt->d_K2, B, A1.q == A1.P, grid, block, ldsB);
        else DISPATCH_P(A1.P, launch_stageB<PP>(st, pt->d_K1, pt->d_K2, B, A1.q == A1.P, grid, block, ldsB));
        IGX_HIP(hipGetLastError());
        pt->timing.n_launches++;
        Kfinal = pt->d_K2;
        F.nlines = (long long)np0 * A1.S;
        F.N1 = A1.N; F.S1 = A1.S; F.Smid = A1.S; F.Slast = A2.S;
        ngroups = (long long)np0 * A1.N;
    } else {
        NY = (kind == IGX_MASS) ? 1 : 4;
        Kfinal = pt->d_K1;
        if (kind == IGX_FORM) {
            // general 2D form: several terms share a last-axis type; their stage-A arrays are summed into the
            // array of that type (2D intermediates are small: one extra pass over them)
            int ymax = 0;
            for (const Term &t : terms) ymax = std::max(ymax, t.t[1]);
            NY = (ymax == 0) ? 1 : 4;
            const size_t per = (size_t)np0 * NPL;
            if (ensure(st, &pt->d_K2, &pt->K2_cap, (size_t)NY * per)) return IGX_ERR_NOMEM;
            for (int y = 0; y < NY; ++y) {
                CombineArgs C{};
                for (size_t i = 0; i < terms.size(); ++i)
                    if (terms[i].t[1] == y) C.src[C.n++] = pt->d_K1 + (size_t)X[term_x[i]].slot * per;
                C.dst = pt->d_K2 + (size_t)y * per;
                C.len = (long long)per;
                k_combine<<<dim3((unsigned)std::min<size_t>((per + 255) / 256, 65535u * 16u)), 256, 0, st>>>(C);
            }
            IGX_HIP(hipGetLastError());
            pt->timing.n_launches += NY;
            Kfinal = pt->d_K2;
        }
        F.nlines = np0;
        F.N1 = 1; F.S1 = 1; F.Smid = 1; F.Slast = A1.S;
        ngroups = np0;
    }
    stage_event(pt, 3, st);

    F.V = AL.d_V; F.fa = AL.dev.fa; F.mslo = AL.dev.mslo; F.mshi = AL.dev.mshi;
    F.jlo = AL.dev.jlo; F.jhi = AL.dev.jhi; F.rp = AL.dev.rp;
    F.N = AL.N; F.q = AL.q; F.G = AL.G;
    F.dim = dim; F.pl0 = d_pl0; F.symmetric = sym;
    F.rp0 = A0.dev.rp; F.jlo0 = A0.dev.jlo; F.jhi0 = A0.dev.jhi;
    F.rp1 = A1.dev.rp; F.jlo1 = A1.dev.jlo; F.jhi1 = A1.dev.jhi;
    F.r0_lo = pt->r0_lo; F.r0_hi = pt->r0_hi; F.nnz_off = pt->nnz_off;
    {
        // ---- matrix-core path (opt-in, IGX_FINAL=mfma): banded FP64 GEMM, see k_final_mfma.  Correct and
        // tested, but at 2 waves/SIMD its load->MFMA loop is latency-bound (MFMA pipe 25 % busy,
        // 13.7 ms vs 11.0 ms for the VALU kernel at C4); it needs an LDS-DMA staged A operand to pay off.
        {
            const int W = 2 * AL.P - 1;
            const int ntile = (AL.N * W + 15) / 16;
            int wmax = 0;
            for (int t = 0; t < ntile; ++t) {
                const int i_first = std::min(AL.N - 1, (t * 16) / W), i_last = std::min(AL.N - 1, (t * 16 + 15) / W);
                wmax = std::max(wmax, (AL.mshi[i_last] - AL.mslo[i_first]) * AL.q);
            }
            const int nch = (wmax + 7) / 8;
            const bool want_mfma = pt->knobs.final_sel == 3;
            if (want_mfma && sym && nch >= 1 && nch <= 6 && pt->ldesc_ok && AL.G >= 2) {
                FinalMArgs M{};
                M.desc = (const int4 *)pt->d_ldesc; M.nl = pt->n_ldesc; M.nlines = F.nlines;
                M.V = AL.d_V; M.fa = AL.dev.fa; M.mslo = AL.dev.mslo; M.mshi = AL.dev.mshi;
                M.jlo = AL.dev.jlo; M.jhi = AL.dev.jhi; M.rp = AL.dev.rp;
                M.N = AL.N; M.P = AL.P; M.q = AL.q; M.G = AL.G; M.W = W; M.ntile = ntile;
                int LC = 256;
#ifdef IGX_ABLATE
                if (const char *e = getenv("IGX_FINAL_LC")) LC = std::max(16, (atoi(e) / 16) * 16);
#endif
                M.LC = LC; M.debug = 0;
                M.nchunk_blocks = (M.nl + LC - 1) / LC;
                const long long nblocks = (long long)((ntile + 3) / 4) * M.nchunk_blocks;
                if (nblocks > 0x7fffffffLL) { set_error("final stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
                dim3 block(256), grid((unsigned)nblocks);
#define LAUNCH_M(NYV, NCHV) k_final_mfma<NYV, NCHV><<<grid, block, 0, st>>>(Kfinal, d_data, M)
                if (NY == 1) {
                    switch (nch) { case 1: LAUNCH_M(1, 1); break; case 2: LAUNCH_M(1, 2); break; case 3: LAUNCH_M(1, 3); break;
                                   case 4: LAUNCH_M(1, 4); break; case 5: LAUNCH_M(1, 5); break; default: LAUNCH_M(1, 6); break; }
                } else {
                    switch (nch) { case 1: LAUNCH_M(4, 1); break; case 2: LAUNCH_M(4, 2); break; case 3: LAUNCH_M(4, 3); break;
                                   case 4: LAUNCH_M(4, 4); break; case 5: LAUNCH_M(4, 5); break; default: LAUNCH_M(4, 6); break; }
                }
#undef LAUNCH_M
                IGX_HIP(hipGetLastError());
                pt->timing.n_launches++;
                stage_event(pt, 4, st);
                return IGX_OK;
            }
        }
        // ---- quadrature-lane kernel: single interior knots and q == P on the last axis
        {
            const bool want_q = pt->knobs.final_sel != 2;
            if (want_q && AL.q == AL.P && AL.simple && AL.P <= 6) {       // (k_final_q is instantiated for P <= 6: higher degrees take k_final)
                FinalQArgs Q{};
                Q.V = AL.d_V; Q.fa = AL.dev.fa; Q.mslo = AL.dev.mslo; Q.mshi = AL.dev.mshi;
                Q.jlo = AL.dev.jlo; Q.jhi = AL.dev.jhi; Q.rp = AL.dev.rp;
                Q.N = AL.N; Q.G = AL.G; Q.nlines = F.nlines;
                Q.desc = (const LineDesc *)(sym ? pt->d_qdesc : pt->d_qdescn);
                Q.ndesc = sym ? pt->n_qdesc : pt->n_qdescn;
                Q.dump = pt->nnz;
                const int R = 64 / AL.P;
                Q.nchunks = (AL.N + R - 1) / R;
                // ~8 waves per CU and a few rounds; at least 16 lines per wave to amortise its set-up
                long long target_waves = 8192;
#ifdef IGX_ABLATE
                if (const char *e = getenv("IGX_FINALQ_WAVES")) target_waves = std::max(1, atoi(e));
#endif
                Q.nsuper = (Q.nchunks + FINALQ_WAVES - 1) / FINALQ_WAVES;
                Q.lpw = (int)std::max<long long>(16, (Q.ndesc * Q.nsuper * FINALQ_WAVES + target_waves - 1) / target_waves);
                // a launch of less than one resident round (2 blocks per CU: a 2D patch): as few lines per wave as one
                // round allows -- the lines of a wave are worked through in sequence
                const long long slots = 2 * 256, per_super = slots / Q.nsuper;
                if (per_super > 0 && ((Q.ndesc + 15) / 16) * Q.nsuper <= slots)
                    Q.lpw = (int)std::max<long long>(4, (Q.ndesc + per_super - 1) / per_super);
                const long long nblocks = ((Q.ndesc + Q.lpw - 1) / Q.lpw) * Q.nsuper;
                if (nblocks > 0x7fffffffLL) { set_error("final stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
                dim3 block(64 * FINALQ_WAVES), grid((unsigned)nblocks);
#define LAUNCH_Q(PV) { if (NY == 1) k_final_q<PV, 1><<<grid, block, 0, st>>>(Kfinal, d_data, Q); \
                       else k_final_q<PV, 4><<<grid, block, 0, st>>>(Kfinal, d_data, Q); }
                switch (AL.P) {
                case 2: LAUNCH_Q(2); break;
                case 3: LAUNCH_Q(3); break;
                case 4: LAUNCH_Q(4); break;
                case 5: LAUNCH_Q(5); break;
                default: LAUNCH_Q(6); break;
                }
#undef LAUNCH_Q
                IGX_HIP(hipGetLastError());
                pt->timing.n_launches++;
                stage_event(pt, 4, st);
                return IGX_OK;
            }
        }
        // wave tasks: chunks of CR <= 64 consecutive rows; NW waves share the staged basis-table
        // segment of a row tile (the whole axis when it fits in ~64 KB of LDS)
        const int W = 2 * AL.P - 1;
        F.SSTR = (AL.q * AL.P * 2) | 1;
        F.KSTR = AL.q | 1;
        int ntiles = 1, tile_rows, CR, tsp_max, trow_max, nsp_max;
        // rows per wave task: at most 64 (one lane per row), and few enough that the K window of the task -- the Gauss points
        // under its rows: (rows + p) spans with single knots -- fits the 8 x 64 prefetch slots (degree 7 with q = 8: 56 rows)
        const int crmax = std::max(1, std::min(64, 512 / std::max(AL.q, 1) - AL.p));
        for (;; ++ntiles) {
            const int rows_per_tile = (AL.N + ntiles - 1) / ntiles;
            const int nch = (rows_per_tile + crmax - 1) / crmax;
            CR = (rows_per_tile + nch - 1) / nch;
            tile_rows = CR * nch;
            tsp_max = trow_max = nsp_max = 0;
            for (int lo = 0; lo < AL.N; lo += tile_rows) {
                const int hi = std::min(lo + tile_rows, AL.N);
                tsp_max = std::max(tsp_max, AL.mshi[hi - 1] - AL.mslo[lo]);
                trow_max = std::max(trow_max, AL.jhi[hi - 1] - AL.jlo[lo]);
                for (int cl = lo; cl < hi; cl += CR)
                    nsp_max = std::max(nsp_max, AL.mshi[std::min(cl + CR, hi) - 1] - AL.mslo[cl]);
            }
            if ((size_t)tsp_max * F.SSTR * sizeof(double) <= 64 * 1024 || tile_rows <= crmax) break;
        }
        ntiles = (AL.N + tile_rows - 1) / tile_rows;
        const int kpy = (nsp_max * AL.q + 63) / 64;
        if (kpy > 8) { set_error("final stage: K window does not fit the prefetch registers"); return IGX_ERR_UNSUPPORTED; }
        F.CR = CR; F.tile_rows = tile_rows; F.ntiles = ntiles; F.tsp_max = tsp_max; F.trow_max = trow_max;
        F.nsp_max = nsp_max; F.ngroups = ngroups;
        const size_t vbytes = (size_t)tsp_max * F.SSTR * sizeof(double);
        const size_t kslot = std::max((size_t)NY * nsp_max * F.KSTR, (size_t)CR * W) * sizeof(double);
        const size_t tbytes = ((size_t)5 * trow_max + tsp_max) * sizeof(int);
        if (vbytes + kslot + tbytes > 160 * 1024) { set_error("final stage: basis table segment (%zu B) does not fit LDS", vbytes); return IGX_ERR_UNSUPPORTED; }
        const bool fast = AL.q == AL.P && AL.simple && AL.P < 7;      // (P >= 7: generic instantiation only, launch_final)
        const int max_waves = fast ? 12 : 6;
        int NW = (int)std::min<size_t>(max_waves, (160 * 1024 - vbytes - tbytes) / kslot);
#ifdef IGX_ABLATE
        if (const char *e = getenv("IGX_FINAL_NW")) NW = std::max(1, std::min(NW, atoi(e)));
#endif
        int max_lines = 1;
        if (dim == 3)
            for (int i = 0; i < A1.N; ++i) max_lines = std::max(max_lines, A1.jhi[i] - A1.jlo[i]);
        // row groups per block: enough tasks for ~8 rounds per wave
        int GPB = std::max(1, (8 * NW) / std::max(1, max_lines * (tile_rows / CR)));
#ifdef IGX_ABLATE
        if (const char *e = getenv("IGX_FINAL_GPB")) GPB = std::max(1, atoi(e));
#endif
        F.NW = NW; F.GPB = GPB;
        const size_t lds = vbytes + (size_t)NW * kslot + tbytes;
        const long long nblocks = ((ngroups + GPB - 1) / GPB) * ntiles;
        if (nblocks > 0x7fffffffLL) { set_error("final stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
        dim3 block(64 * NW), grid((unsigned)nblocks);
        int rc = IGX_OK;
        if (AL.P >= 7) rc = final_hi(AL.P, st, Kfinal, d_data, F, NY, fast, kpy, grid, block, lds);
        else DISPATCH_P(AL.P, rc = launch_final<PP>(st, Kfinal, d_data, F, NY, fast, kpy, grid, block, lds));
        if (rc) return rc;
        IGX_HIP(hipGetLastError());
        pt->timing.n_launches++;
    }
    stage_event(pt, 4, st);
    return IGX_OK;
}

} // namespace igx

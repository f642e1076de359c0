// k_bf3: the fused "sweep + final" stage -- second contraction (swept "mid" axis) and third contraction (last axis) of the
// sum-factorised assembly in one kernel, CSR values out -- for every symmetric 3D patch the reference accepts with single knots
// on the last axis, and for the convection-diffusion form.  Successor of k_bf2 + k_mirror2 (fused.hip, rounds 2-4).
//
// (1) BOTH TRIANGLES, NO MIRROR PASS.  k_bf2 forms the lower triangle only; k_mirror2 then read it back (11.9 GB at C4, 72-byte
//     gathers) and wrote the upper one (6.6 GB): 3.5 ms of a 14.7 ms chain that existed only to copy.  The transposed entries are
//     in the registers of the contractor waves when the direct ones are, and they are complete AT THE SAME STEP:
//       a block owns an outer pair (i0, j0 <= i0) and walks the mid axis dof by dof.  When dof d leaves the sweep window, the K2
//       lines of the pairs (d + a, d) and (d, d + a), a = 0..p1, are flushed.  Direct row (i0, i1, .) needs the pairs (i1, j1) for
//       all j1: those with j1 < i1 arrived when j1 left, the others arrive when i1 leaves.  The TRANSPOSED row -- row (j0, j1, .),
//       columns (i0, i1, .) -- needs the pairs (i1, j1) for all i1: those with i1 < j1 arrived when i1 left, the others when j1
//       leaves.  Same schedule with the two families of lines swapped.
//     So the contractors keep TWO sets of entry rings in LDS -- direct rows and transposed rows -- and every row of both leaves as
//     whole (2 p1 + 1) (2 p2 + 1) segments.  Inside a line the transposition on the last axis needs no halo: the entries of target
//     row j2 are sums over the element matrices of the spans of supp(j2), all inside the tile's window -- gathered with
//     ds_bpermute from the same registers as the direct entries, in the same order of addends, so A[I, J] and A[J, I] are the
//     same bits (exact symmetry, as assemble_entries(symmetric=True): pyiga/assemble.py:742-752).  On a diagonal outer block the
//     "transposed" entries are the upper part of the same rows (one set of rings).
// (2) WHAT PAYS FOR THE SECOND SET (160 KB of LDS were full): rings are allocated per LINE, not per row -- the line of distance
//     delta = i1 - j1 lives delta + 1 steps and gets delta + 1 rotating slots; the lines of the current row need one slot (read
//     between the barrier that ends the contraction and the one that starts the next); no store plan in LDS.
// (3) THE STORE DUTY IS ON THE SWEEPER WAVES of the roles 1.. (they wait at the barriers most of a step and have registers to
//     spare), staged: read behind B2, stored behind the next B1 under the sweep, 512 contiguous bytes per instruction
//     (BF3StoreDense).  On the contractor waves the same stores sat in the B2 -> B1 window of every step: 9.7 -> 8.4 ms at C4.
// (4) EVERY PATCH THE REFERENCE ACCEPTS ON THE SWEPT AXIS (round 5): the degrees of the mid and the last axis are separate
//     template parameters (P1, P2), the Gauss points per span a third (Q = nqp = max degree + 1 over all axes:
//     pyiga/assemblers.pyx:1338), and a step is a LEAVING DOF, not a span: a span is swept when its first active dof is due
//     (fa1[]), so a repeated knot on the mid axis just shifts the window by its multiplicity; which lines of a row exist comes
//     from the column range of the row (jlo1 / jhi1).
//
// Semantics follow combine()/entry_impl (pyiga/assemblers.pyx:1455-1540); for equal degrees and single knots the values are
// those of k_bf2 + k_mirror2 bit for bit (tests/test_gpu_parity.py::test_both_triangles_from_the_fused_stage).
#include "igx_internal.h"
#include "fused_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef BF3_FIXED_PASS
#define BF3_FIXED_PASS 1      // the contractor waves keep their passes from step to step (no rotation)
#endif
#ifndef BF3_MASS_AXSYM
#define BF3_MASS_AXSYM 1      // the 3D mass form through its per-axis symmetry (SYM = 3)
#endif
#ifndef BF3_MASS_STW
#define BF3_MASS_STW 1        // ... with the store duty on its sweeper waves (dense slots) at low degree; 0: always on the contractors
#endif
#ifndef BF3_MASS_STW_QMAX
#define BF3_MASS_STW_QMAX 3   // ... "low degree": up to this many Gauss points per span
#endif

namespace igx {

// SYM: 0 non-symmetric form (one set of rings, every pair direct); 1 symmetric, every outer pair diagonal (2D: one set);
//      2 symmetric with off-diagonal outer pairs (3D: direct + transposed set);
//      3 (round 6) symmetric PER AXIS -- the 3D mass form: A[(i0,i1,i2),(j0,j1,j2)] = sum_g W N_i0 N_j0 N_i1 N_j1 N_i2 N_j2 does not
//        change when the indices of ONE axis are exchanged, so the block of ANY outer pair (i0, j0) is symmetric under
//        (i1, i2) <-> (j1, j2): every block contracts like a diagonal one (lower lines only, the upper part of a row from the same
//        element matrices, one set of rings) and a finished row is STORED TWICE -- to row block i0 and, with the same values, to
//        row block j0 (A[(j0,i1,i2),(i0,j1,j2)] is the same number)
template <int P1, int P2, int Q, int NLG, int NRO, int NCW, int SYM> struct BF3Geom {
    static constexpr int P1_ = P1, P2_ = P2, Q_ = Q;
    static constexpr int p1 = P1 - 1, p2 = P2 - 1, W1 = 2 * P1 - 1, W2 = 2 * P2 - 1, TL = 64 * NLG;
    static constexpr int NSET = SYM == 2 ? 2 : 1;
    // bank conflicts of the contractors' reads (lane = span): a line's points lie QS doubles apart per span -- an odd number, so
    // that the 32 lanes of a ds_read_b64 group hit 32 different bank pairs (Q even: 2-way at 6, 4-way at 4) -- and the basis
    // values VS doubles apart per span, VS / 2 odd, so that the 16 lanes of a ds_read_b128 group hit 16 different slots
    // (Q P2 2 = 72 at degree 5: 4-way)
    static constexpr int QS = Q + (Q % 2 == 0 ? 1 : 0), NSP = (TL + Q - 1) / Q;
    static constexpr int TLP = TL + NSP * (QS - Q);         // doubles of one role's part of a line
    static constexpr int VS = Q * P2 * 2 + ((Q * P2 * 2) % 4 == 2 ? 0 : 2), NV2 = NSP * VS;
    static constexpr int LS = NRO * TLP + 2;                // doubles per line (all roles), padded against bank conflicts
    static constexpr int NRING = p1 * (p1 + 1) / 2 + p1;    // row-lines of the ring part: line delta = 1..p1 has delta + 1 slots
    static constexpr int NRL = NRING + P1;                  // ... + the P1 lines that complete with the row itself
    static constexpr int T0 = p2 * (p2 + 1) / 2;            // rp2[i2] = W2 i2 - T0 on interior rows of the last axis
    static constexpr int NEL = 2 * p2 * W2;                 // edge-row table: (row, entry) elements of the <= 2 p2 edge rows
    // LDS image (doubles): lines [W1][LS] | sets [NSET][NRL][R][W2] | basis values [NSP][VS] | edge table (int4) [NEL]
    static constexpr int off_sets() { return (W1 * LS + 1) & ~1; }
    static constexpr int off_v2(int R) { return (off_sets() + NSET * (NRL * R * W2 + 2) + 1) & ~1; }     // (+ 2: the pad of a set)
    static constexpr int off_etab(int R) { return off_v2(R) + NV2; }
    static constexpr int lds_doubles(int R) { return off_etab(R) + NEL * 2; }
    static constexpr int rmax()
    {
        int R = TL / Q - p2;
        while (R > 1 && lds_doubles(R) * 8 > 160 * 1024) --R;
        return R;
    }
    static constexpr int RMAX = rmax();
    static constexpr int RW = RMAX * W2;                    // doubles of one row-line block [row][entry]
    static constexpr int SETSZ = NRL * RW + 2;              // + a pad behind the row-lines: where lanes without an element read and clear
    static constexpr int OFF_PAD = NRL * RW;
    static constexpr int roff(int d) { return RW * ((d - 1) * (d + 2) / 2); }   // first slot of ring line delta = d
    static constexpr int OFF_CUR = NRING * RW;              // lines of the current row: pair (d, d + a) resp. (d + a, d) at a
    static constexpr int OFF_SETS = off_sets(), OFF_V2 = off_v2(RMAX), OFF_ETAB = off_etab(RMAX);
    static constexpr int LDS_BYTES = lds_doubles(RMAX) * 8;
    static constexpr int NSUB = (RW + 63) / 64;             // store chunks of a line block
    // contractor passes (as k_bf2): pieces of PL consecutive spans of a line, 64 / PL pieces per pass, pieces overlap by p2 spans
    static constexpr int npc(int pl) { return (RMAX + pl - p2 - 1) / (pl - p2); }
    static constexpr int npass(int pl) { return (W1 * npc(pl) + 64 / pl - 1) / (64 / pl); }
    static constexpr int pick()
    {
        int best = 64;
        const int cand[4] = {64, 32, 21, 16};
        for (int i = 1; i < 4; ++i)
            if (cand[i] > 2 * p2 && npass(cand[i]) < npass(best)) best = cand[i];
        return best;
    }
    static constexpr int PL = pick(), PPP = 64 / PL, NPC = npc(PL), RP = PL - p2;
};

struct BF3Blk {
    int i0, j0, diag0, c0i, c0j, cj0, ci0, rlo, rhi, row_lo, nrows, stD, stT, ne;      // (rlo, rhi: rows of the mid axis)
    long long S12;
};

typedef int bf3_v4i __attribute__((ext_vector_type(4)));

// scalars of row d of the mid axis (the row that completes with step d + 1): which lines of its segments exist, its place in
// the row blocks, the ring slots it sits in
template <class Gm> struct BF3Row {
    int on, l0, c1, rp1d, sub1, sub2, sub3, sub4, sub5;
    __device__ __forceinline__ static BF3Row of(const BFArgs &A, const BF3Blk &B, const int t)
    {
        cip rp1 = (cip)A.rp1, jlo1 = (cip)A.jlo1, jhi1 = (cip)A.jhi1;
        constexpr int p1 = Gm::p1, RW = Gm::RW;
        BF3Row r;
        const int d = t - 1;
        r.on = d >= B.rlo && d < B.rhi;
        const int dc = min(max(d, 0), A.N1 - 1);
        const int jl1 = jlo1[dc];
        r.c1 = jhi1[dc] - jl1; r.l0 = p1 - (dc - jl1);          // line l of the (2 p1 + 1) possible ones is column dc + l - p1
        r.rp1d = rp1[dc];
        // slot of row d in ring line delta (named scalars: a select by a lane's or a wave's line stays a chain of selects)
        r.sub1 = Gm::roff(1) + (int)((unsigned)dc % 2u) * RW; r.sub2 = Gm::roff(2) + (int)((unsigned)dc % 3u) * RW;
        r.sub3 = Gm::roff(3) + (int)((unsigned)dc % 4u) * RW; r.sub4 = Gm::roff(4) + (int)((unsigned)dc % 5u) * RW;
        r.sub5 = Gm::roff(5) + (int)((unsigned)dc % 6u) * RW;
        return r;
    }
    // line block of line l of the row (doubles from the start of a set); l is a scalar
    __device__ __forceinline__ int line_off(const int l) const
    {
        constexpr int p1 = Gm::p1;
        int o = Gm::OFF_CUR + (l - p1) * Gm::RW;
        o = l == p1 - 1 ? sub1 : o;
        if (p1 >= 2) o = l == p1 - 2 ? sub2 : o;
        if (p1 >= 3) o = l == p1 - 3 ? sub3 : o;
        if (p1 >= 4) o = l == p1 - 4 ? sub4 : o;
        if (p1 >= 5) o = l == p1 - 5 ? sub5 : o;
        return o;
    }
    __device__ __forceinline__ bool line_ok(const int l) const { return l < Gm::W1 && l >= l0 && l - l0 < c1; }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf3_rs(double *ptr, const int len) { return __builtin_amdgcn_make_buffer_rsrc((void *)ptr, (short)0, len, 0x00020000); }

// The store duty of k_bf3.  Behind the barrier B2 of step t the rows d = t - 1 of both sets are complete; their slots are
// reused in step t + 1, so they are read (and cleared where halves add) before the next B1.
//
// BF3Store: per-LINE slots -- (set, chunk c of 64 consecutive doubles of a line block, line l): c is a compile-time constant (row
// constants in registers, immediate offsets), l a scalar of the wave.  NS waves share the duty: wave sw takes, of every chunk c,
// the lines (sw + ROT c) % NS + NS j.  Used where the form has ONE sweeper role (mass): the contractors carry it, at once (read ->
// store inside the B2 -> B1 window).  Rows next to the ends of the last axis (segments of fewer than 2 p2 + 1 columns) are not
// part of the slots (their lanes carry an out-of-range row constant and are not cleared): bf3_edge_rows().
//
// TR (round 6, every store path of the kernel): the patch is one whose MID AND LAST AXIS WERE EXCHANGED by the host (repeated knots on
// the last axis of the caller's patch: sumfact.hip, swap_route) and the values go to the CSR layout of the CALLER's patch -- row
// (i0, i2, d), and inside the segment of an outer column the entry of (line l, entry e) at (cx c2 + e) c1 + (l - l0) instead of
// (cx c1 + l - l0) c2 + e: with rp2 / c2 the row table of the last axis here (the caller's mid axis) and rp1d / c1 those of row d
//     offset = c0 (rp2[i2] S1 + c2 rp1d) + (cx c2 + e) c1 + (l - l0)        (non-TR: c0 (rp1d S2 + c1 rp2[i2]) + (cx c1 + l - l0) c2 + e)
// Only the offsets change: what a lane reads from the rings, and when, is the same.
template <class Gm, int NS, int ROT, int NH, int SYM, bool TR>
struct BF3Store {
    static constexpr int P1 = Gm::P1_, W1 = Gm::W1, W2 = Gm::W2, p2 = Gm::p2, NSUB = Gm::NSUB, RW = Gm::RW;
    static constexpr int JMAX = (W1 + NS - 1) / NS, NST = NSUB * JMAX, NSET = SYM == 2 ? 2 : 1;
    using Row = BF3Row<Gm>;
    int rrv[NSUB];               // W2 i2 - T0 of this lane's row in chunk c, or a value that takes the offset out of range
    int ev[TR ? NSUB : 1];       // TR: 8 x the lane's entry index in chunk c
    int lane8, lanec, inv;       // 8 lane | offset (doubles) of this lane in the last chunk, kept inside the line
    double *pD, *pT;             // descriptor bases (an absent row or line gets length 0: every lane out of range)
    int nD, nT;

    __device__ __forceinline__ void init(const BFArgs &A, const BF3Blk &B, const int /*sw*/, const int lane)
    {
        cip rp0 = (cip)A.rp0;
        lane8 = lane * 8;
        lanec = min((NSUB - 1) * 64 + lane, RW - 1) - (NSUB - 1) * 64;
        if constexpr (TR) {
            // whole row blocks of the two outer rows; a lane without a row gets a row constant beyond both (c0max < 2 c0min)
            pD = A.data + ((long long)rp0[B.i0] * B.S12 - A.nnz_off); nD = B.stD ? (int)((long long)B.c0i * B.S12 * 8) : 0;
            pT = A.data + ((long long)rp0[B.j0] * B.S12 - A.nnz_off); nT = B.stT ? (int)((long long)B.c0j * B.S12 * 8) : 0;
            inv = 2 * (int)A.S2 + 1;
#pragma unroll
            for (int c = 0; c < NSUB; ++c) {
                const int q = c * 64 + lane, rr = q / W2, i2 = B.row_lo + rr;
                const bool ok = q < RW && rr < B.nrows && i2 >= p2 && i2 <= A.N2 - 1 - p2;
                rrv[c] = ok ? W2 * i2 - Gm::T0 : inv;
                ev[c] = 8 * (q - rr * W2);
            }
            return;
        }
        // descriptors: base moved by W2 row_lo - T0 doubles, so that an interior row i2 of the tile sits at (W2 i2 - T0) (c0 c1 - 1)
        // + (chunk element index) + W2 (line terms); length = the row block of the outer row (anything beyond is dropped)
        const long long shift = (long long)W2 * B.row_lo - Gm::T0;
        const long long lenD = ((long long)B.c0i * B.S12 - shift) * 8, lenT = ((long long)B.c0j * B.S12 - shift) * 8;
        pD = A.data + ((long long)rp0[B.i0] * B.S12 - A.nnz_off + shift); nD = B.stD ? (int)lenD : 0;
        pT = A.data + ((long long)rp0[B.j0] * B.S12 - A.nnz_off + shift); nT = B.stT ? (int)lenT : 0;
        const int c0min = min(B.c0i, B.c0j);
        inv = (int)(max(lenD, lenT) / (8 * max(c0min * 2 - 1, 1))) + 1;      // (a row of the mid axis has at least two columns)
#pragma unroll
        for (int c = 0; c < NSUB; ++c) {
            const int q = c * 64 + lane, rr = q / W2, i2 = B.row_lo + rr;
            const bool ok = q < RW && rr < B.nrows && i2 >= p2 && i2 <= A.N2 - 1 - p2;
            rrv[c] = ok ? W2 * i2 - Gm::T0 : inv;
        }
    }
    __device__ __forceinline__ static int line_of(const int sw, const int c, const int j) { return (int)((unsigned)(sw + ROT * c) % (unsigned)NS) + NS * j; }

    // (TR: sK = 8 c0 S1, soff0 = 8 (c0 W2 rp1d + cx W2 c1 - l0), a line adds 8)
    template <int X>
    __device__ __forceinline__ void move_row(double *sets, double *dump, const int sw, const int lane, const Row &r, double *ptr, const int len, const int sK, const int soff0)
    {
        // (addresses: an opaque per-lane base + the scalar line offset, the chunk is the immediate offset of the LDS instruction)
        int lb_ = lane, lc_ = lanec;
        asm volatile("" : "+v"(lb_), "+v"(lc_));
        double *bs = sets + X * Gm::SETSZ + lb_, *bc = sets + X * Gm::SETSZ + (NSUB - 1) * 64 + lc_;
        double v[NST];
#pragma unroll
        for (int c = 0; c < NSUB; ++c)
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int l = line_of(sw, c, j);
                double *src = (c == NSUB - 1 ? bc : bs + c * 64) + r.line_off(min(l, W1 - 1));
                v[c * JMAX + j] = *src;
                if (NH >= 2) {
                    double *cl = (rrv[c] != inv && l < W1) ? src : dump;      // (edge rows belong to bf3_edge_rows)
                    *cl = 0.0;
                }
            }
#ifdef BF3_NOSTORE
        return;
#endif
        int voff[NSUB];
#pragma unroll
        for (int c = 0; c < NSUB; ++c) {
            if constexpr (TR) voff[c] = (int)(__umul24((unsigned)rrv[c], (unsigned)sK) + __umul24((unsigned)ev[c], (unsigned)r.c1));
            else voff[c] = (int)__umul24((unsigned)rrv[c], (unsigned)sK) + lane8;
            asm volatile("" : "+v"(voff[c]));               // (+ c * 512 below is the store's immediate offset, not another register)
        }
#pragma unroll
        for (int c = 0; c < NSUB; ++c)
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const int l = line_of(sw, c, j);
                if constexpr (TR) bf2_buffer_store(bf3_rs(ptr, r.line_ok(l) ? len : 0), voff[c], soff0 + 8 * l, v[c * JMAX + j]);
                else bf2_buffer_store(bf3_rs(ptr, r.line_ok(l) ? len : 0), voff[c] + c * 512, soff0 + 8 * W2 * l, v[c * JMAX + j]);
            }
    }
    // behind B2 of step t: rows d = t - 1 of both sets are complete: out of the rings, to their segments
    __device__ __forceinline__ void fetch(const BFArgs &A, const BF3Blk &B, double *sets, double *dump, const int t, const int sw, const int lane)
    {
        const Row r = Row::of(A, B, t);
        if constexpr (TR) {
            const int sD = 8 * B.c0i * (int)A.S1, sT = 8 * B.c0j * (int)A.S1;
            const int oD = 8 * (W2 * (B.c0i * r.rp1d + B.cj0 * r.c1) - r.l0), oT = 8 * (W2 * (B.c0j * r.rp1d + B.ci0 * r.c1) - r.l0);
            move_row<0>(sets, dump, sw, lane, r, pD, r.on ? nD : 0, sD, oD);
            if (NSET == 2) move_row<1>(sets, dump, sw, lane, r, pT, r.on ? nT : 0, sT, oT);
            if (SYM == 3) move_row<0>(sets, dump, sw, lane, r, pT, r.on ? nT : 0, sT, oT);
            return;
        }
        move_row<0>(sets, dump, sw, lane, r, pD, r.on ? nD : 0, 8 * (B.c0i * r.c1 - 1), 8 * ((int)((long long)B.c0i * A.S2) * r.rp1d + W2 * (B.cj0 * r.c1 - r.l0)));
        if (NSET == 2)
            move_row<1>(sets, dump, sw, lane, r, pT, r.on ? nT : 0, 8 * (B.c0j * r.c1 - 1), 8 * ((int)((long long)B.c0j * A.S2) * r.rp1d + W2 * (B.ci0 * r.c1 - r.l0)));
        if (SYM == 3)       // per-axis symmetry: the same row once more, to the row block of j0 (read again: whole passes only, nothing is cleared)
            move_row<0>(sets, dump, sw, lane, r, pT, r.on ? nT : 0, 8 * (B.c0j * r.c1 - 1), 8 * ((int)((long long)B.c0j * A.S2) * r.rp1d + W2 * (B.ci0 * r.c1 - r.l0)));
    }
};

// BF3StoreDense: the store duty on the sweeper waves of the roles 1.., staged and DENSE: a slot is 64 consecutive doubles of the
// row-major image of a tile's rows -- (row, line, entry), W1 W2 per row -- so a store instruction writes 512 contiguous bytes of
// CSR values and the segments of a row leave whole within one burst.  The rings are [line][row][entry], so a lane's LDS address
// is (block of ITS line) + (row, entry): the line blocks of the step sit in one register (lane j holds the block of line j) and
// come per slot through ds_bpermute.  Per lane and slot one packed constant: line | (row W2 + entry) | row | invalid (edge
// rows, lanes past the tile) -- the same for both sets.  No branch lies around a store (a sweeper also loads: hipcc answers a
// store behind a branch with vmcnt(0) at the next load use): what must not be stored gets an out-of-range offset.
// The waves of one GROUP (sw = 0 .. QSTR-1) take the slots QLO + sw + QSTR k, k < K, below QHI: the sweepers of the last role
// (one input array: registers to spare) take half as many slots again as those of the middle roles.
template <int NR, int NLG, int NQ> struct BF3DenseSplit {
    static constexpr int NA = (NR - 2) * NLG, NB = NLG;                        // waves of the middle roles / of the last role
    static constexpr int ka() { if (NA == 0) return 0; int k = 1; while (NA * k + NB * ((3 * k + 1) / 2) < NQ) ++k; return k; }
    static constexpr int KA = ka(), KB = NA == 0 ? (NQ + NB - 1) / NB : (3 * KA + 1) / 2;
    static constexpr int QB = NA * KA < NQ ? NA * KA : NQ;                     // first slot of the last role's group
};
constexpr int BF3_FAR = 0x7f000000;       // added to the offset of what must not be stored: beyond every descriptor (<= 0.9 GB, fused3_offsets_fit)
template <class Gm, int NH, int SYM, int K, int QLO, int QHI, int QSTR, bool TR>
struct BF3StoreDense {
    static constexpr int p1 = Gm::p1, p2 = Gm::p2, W1 = Gm::W1, W2 = Gm::W2, WW = W1 * W2, RW = Gm::RW, NSET = SYM == 2 ? 2 : 1;
    using Row = BF3Row<Gm>;
    double svD[K], svT[NSET == 2 ? K : 1];
    int pk[K];                   // bits 0-5: 4 line | 6-17: row W2 + entry | 18-25: row | 31: no element (then line 15, row 0, entry 0)
    int lane8;
    double *pD, *pT;
    long long nD, nT;            // bytes of the row blocks of the two outer rows (0: not stored)

    __device__ __forceinline__ void init(const BFArgs &A, const BF3Blk &B, const int sw, const int lane)
    {
        cip rp0 = (cip)A.rp0;
        lane8 = lane * 8;
        pD = A.data + ((long long)rp0[B.i0] * B.S12 - A.nnz_off); nD = B.stD ? (long long)B.c0i * B.S12 * 8 : 0;
        pT = A.data + ((long long)rp0[B.j0] * B.S12 - A.nnz_off); nT = B.stT ? (long long)B.c0j * B.S12 * 8 : 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            svD[k] = 0.0;
            if (NSET == 2) svT[k] = 0.0;
            const int qi = QLO + sw + QSTR * k;
            const int q = qi * 64 + lane, rr = q / WW, rem = q - rr * WW, l = rem / W2, e = rem - l * W2, i2 = B.row_lo + rr;
            const bool ok = qi < QHI && q < Gm::RMAX * WW && rr < B.nrows && i2 >= p2 && i2 <= A.N2 - 1 - p2;
            pk[k] = ok ? (4 * l) | ((rr * W2 + e) << 6) | (rr << 18) : (int)(0x80000000u | 60u);     // no element: "line 15" = the pad
        }
    }

    // behind B2 of step t: rows d = t - 1 of both sets are complete -> registers (cleared where halves add)
    __device__ __forceinline__ void fetch(const BFArgs &A, const BF3Blk &B, double *sets, double *dump, const int t, const int sw, const int lane)
    {
        const Row r = Row::of(A, B, t);
        // block (doubles from the start of a set) of line j of the row, on lane j
        int lv = Gm::OFF_CUR + (lane - p1) * RW;
        lv = lane == p1 - 1 ? r.sub1 : lv;
        if (p1 >= 2) lv = lane == p1 - 2 ? r.sub2 : lv;
        if (p1 >= 3) lv = lane == p1 - 3 ? r.sub3 : lv;
        if (p1 >= 4) lv = lane == p1 - 4 ? r.sub4 : lv;
        if (p1 >= 5) lv = lane == p1 - 5 ? r.sub5 : lv;
        lv = lane == 15 ? Gm::OFF_PAD : lv;                 // "line 15": the pad of the set -- no test, no select per slot
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int pkk = pk[k];
            asm volatile("" : "+v"(pkk));                   // (decoded here, every step: kept decoded the fields of all slots cost 20 registers)
            const int lo = __builtin_amdgcn_ds_bpermute(pkk & 60, lv) + ((pkk >> 6) & 0xfff);
            double *src = sets + lo;
            svD[k] = *src;
#ifndef BF3_NOCLEAR
            if (NH >= 2) *src = 0.0;
#endif
            if (NSET == 2) {
                double *srcT = src + Gm::SETSZ;
                svT[k] = *srcT;
#ifndef BF3_NOCLEAR
                if (NH >= 2) *srcT = 0.0;
#endif
            }
        }
        (void)dump;
    }

    // behind B1 of step t + 1: the rows read behind B2 of step t go out, 512 contiguous bytes per instruction.  The scalars of
    // the row are worked out again here (a handful of scalar instructions) instead of being kept across the sweep.
    __device__ __forceinline__ void issue(const BFArgs &A, const BF3Blk &B, const int t, const int sw)
    {
#ifdef BF3_NOSTORE
        return;
#endif
        const Row r = Row::of(A, B, t);
        // descriptors of the row: base moved by the row's constant part, so that an element sits at 8 (its index in the image)
        // + row * 8 W2 (c0 c1 - W1); what is left of the row block behind the moved base is the length
        if constexpr (TR) {
            // offset of (row rr, line l, entry e) = c0 S1 (W2 (row_lo + rr) - T0) + c0 W2 rp1d + (cx W2 + e) c1 + l - l0
            //                                     = [row constant part: the descriptor's base] + (rr W2 + e) c1 + rr W2 (c0 S1 - c1) + l
            const long long rowc = (long long)W2 * B.row_lo - Gm::T0;
            const long long shD = (long long)B.c0i * A.S1 * rowc + (long long)W2 * (B.c0i * r.rp1d + B.cj0 * r.c1) - r.l0;
            const long long shT = (long long)B.c0j * A.S1 * rowc + (long long)W2 * (B.c0j * r.rp1d + B.ci0 * r.c1) - r.l0;
            const __amdgpu_buffer_rsrc_t dD = bf3_rs(pD + shD, r.on ? (int)max(nD - shD * 8, 0LL) : 0);
            const __amdgpu_buffer_rsrc_t dT = bf3_rs(pT + shT, r.on ? (int)max(nT - shT * 8, 0LL) : 0);
            const int dlD = 8 * W2 * (B.c0i * (int)A.S1 - r.c1), dlT = 8 * W2 * (B.c0j * (int)A.S1 - r.c1), c18 = 8 * r.c1;
            const int lane = lane8 >> 3;
            const int offl = (unsigned)(lane - r.l0) < (unsigned)r.c1 ? lane8 : BF3_FAR;      // lane j: 8 x line j, or far away
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int pkk = pk[k];
                asm volatile("" : "+v"(pkk));
                const int rr = (pkk >> 18) & 0xff;
                const int q8 = __builtin_amdgcn_ds_bpermute(pkk & 60, offl) + (int)__mul24((pkk >> 6) & 0xfff, c18);
                bf2_buffer_store(dD, (int)__mul24(rr, dlD) + q8, 0, svD[k]);
                if (NSET == 2) bf2_buffer_store(dT, (int)__mul24(rr, dlT) + q8, 0, svT[k]);
                if (SYM == 3) bf2_buffer_store(dT, (int)__mul24(rr, dlT) + q8, 0, svD[k]);
            }
            return;
        }
        const int cD = B.c0i * r.c1, cT = B.c0j * r.c1;
        const long long rowc = (long long)W2 * B.row_lo - Gm::T0;
        const long long shD = (long long)B.c0i * A.S2 * r.rp1d + cD * rowc + W2 * (B.cj0 * r.c1 - r.l0);
        const __amdgpu_buffer_rsrc_t dD = bf3_rs(pD + shD, r.on ? (int)max(nD - shD * 8, 0LL) : 0);
        const int dlD = 8 * W2 * (cD - W1);
        const long long shT = (long long)B.c0j * A.S2 * r.rp1d + cT * rowc + W2 * (B.ci0 * r.c1 - r.l0);
        const __amdgpu_buffer_rsrc_t dT = bf3_rs(pT + shT, r.on ? (int)max(nT - shT * 8, 0LL) : 0);
        const int dlT = 8 * W2 * (cT - W1);
        // what must not be stored -- lines outside the column range of the row, lanes without an element ("line 15") -- gets a huge
        // offset through the SAME per-line lookup that finds the line blocks: one ds_bpermute instead of five vector instructions
        const int lane = lane8 >> 3;
        const int offl = (unsigned)(lane - r.l0) < (unsigned)r.c1 ? 0 : BF3_FAR;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int pkk = pk[k];
            asm volatile("" : "+v"(pkk));
            const int rr = (pkk >> 18) & 0xff;
            // (the row term is negative where a segment is shorter than W1 W2 -- 2D, rows of the swept axis with fewer columns --
            // so the slot's own offset is added in the vector register: the sum is what the range check sees)
            const int q8 = __builtin_amdgcn_ds_bpermute(pkk & 60, offl) + ((QLO + sw + QSTR * k) * 512 + lane8);
            bf2_buffer_store(dD, (int)__mul24(rr, dlD) + q8, 0, svD[k]);
            if (NSET == 2) bf2_buffer_store(dT, (int)__mul24(rr, dlT) + q8, 0, svT[k]);
            if (SYM == 3) bf2_buffer_store(dT, (int)__mul24(rr, dlT) + q8, 0, svD[k]);      // per-axis symmetry: the same values to row block j0
        }
    }
};

// Rows next to the ends of the last axis, behind B2 of step t, on the contractor waves: element (row, entry) of the table per
// lane, one (set, line) per round; offset = 8 (c0 c1 rp2[i2] + (cX c1 + m) c2 + o) with the descriptor's shift folded into the
// table.  The elements are read and cleared here and nowhere else.
template <class Gm, int NCW, int NH, int SYM, bool TR>
__device__ __forceinline__ void bf3_edge_rows(const BFArgs &A, const BF3Blk &B, double *sets, const bf3_v4i *etab, const int t, const int cw, const int lane)
{
    constexpr int W1 = Gm::W1, W2 = Gm::W2;
    if (B.ne == 0) return;
    cip rp0 = (cip)A.rp0;
    const BF3Row<Gm> r = BF3Row<Gm>::of(A, B, t);
    const long long shift = (long long)W2 * B.row_lo - Gm::T0;
    static_assert(SYM != 3 || NH == 1, "per-axis symmetry: the second store reads the rows again (no halves, no clears)");
#pragma unroll
    for (int X = 0; X < (SYM == 3 ? 2 : Gm::NSET); ++X) {
        if (X == 0 ? !B.stD : !B.stT) continue;
        const int c0x = X == 0 ? B.c0i : B.c0j, cx = X == 0 ? B.cj0 : B.ci0;
        // TR (table: {ring offset, 8 c2, 8 e, rp2[i2]}): offset = 8 c0 S1 rp2[i2] + 8 c2 (c0 rp1d + cx c1) + 8 e c1 + 8 (l - l0)
        double *px = A.data + ((long long)rp0[X == 0 ? B.i0 : B.j0] * B.S12 - A.nnz_off + (TR ? 0 : shift));
        const int nx = (int)(((long long)c0x * B.S12 - (TR ? 0 : shift)) * 8);
        const unsigned sA = TR ? (unsigned)(c0x * r.rp1d + cx * r.c1) : (unsigned)(c0x * r.c1);
        const unsigned sC = (unsigned)(8 * c0x * (int)A.S1);
        const int soffr = TR ? 0 : 8 * (int)((long long)c0x * A.S2) * r.rp1d;
        for (int l = (int)((unsigned)(cw + NCW - X) % (unsigned)NCW); l < W1; l += NCW) {
            const unsigned sB = TR ? (unsigned)r.c1 : (unsigned)max(cx * r.c1 + l - r.l0, 0);
            const __amdgpu_buffer_rsrc_t dsc = bf3_rs(px, (r.on && r.line_ok(l)) ? nx : 0);
            double *lb = sets + (SYM == 3 ? 0 : X) * Gm::SETSZ + r.line_off(l);
            for (int ch = 0; ch * 64 < B.ne * W2; ++ch) {
                const bf3_v4i e = etab[min(ch * 64 + lane, Gm::NEL - 1)];
                const bool ok = ch * 64 + lane < Gm::NEL && e.w != BF2_OOB;
                double *src = lb + e.x;
                const double v = *src;
                if (NH >= 2 && ok) *src = 0.0;
                int off = (int)(__umul24(sA, (unsigned)e.y) + __umul24(sB, (unsigned)e.z));
                if constexpr (TR) off += (int)__umul24(sC, (unsigned)e.w) + 8 * (l - r.l0);
                else off += e.w;
                bf2_buffer_store(dsc, ok ? off : BF2_OOB, soffr, v);
            }
        }
    }
}

// sweepers: the mid-axis sweep of k_bf2 (fused.hip) with separate degree (P1) and Gauss points per span (Q), driven by LEAVING
// DOFS: step d sweeps the next span when its first active dof is d, then flushes the lines of dof d.  Those of the roles 1..
// carry the store duty (STW).
struct BF3SweepCtx { const BF3Blk *B; double *sets, *dump; int sw, tlp; };
// D: K1 rows in flight per input array (a ring: row l of a span is consumed and its register reloaded with row l + D, which lies
// in the next span for l >= Q - D): D = Q is "one span ahead"; at P = 6 half a span (D = 3) is what 128 registers allow.
template <int P1, int Q, int D, int MASK, int RI, int NA, int NLG, class StoreT, bool STW, bool MULT>
__device__ __forceinline__ void bf3_sweeper(const BFArgs &A, const int r0, const int g2l, const int g2, const int s_begin, const int t_sw, const int d_begin,
                                            const int rhi, double *lines, const int LS, const BF3SweepCtx &sc)
{
    constexpr BFRole R = bf_role(MASK, RI);
    constexpr int p1 = P1 - 1;
    constexpr bool ST = STW && (RI >= 1 || bf_nroles(MASK) == 1);      // (a form with ONE role -- mass -- and the duty on its sweepers)
    const int slane = threadIdx.x & 63;
    StoreT store;
    if constexpr (ST) store.init(A, *sc.B, sc.sw, slane);
    BF_STAMP_DECL
    __builtin_amdgcn_s_setprio(BF2_PRIO_S);
    cdp V1 = (cdp)A.V1;
    cip fa1 = (cip)A.fa1;
    double acc[P1][P1];
#pragma unroll
    for (int a = 0; a < P1; ++a)
#pragma unroll
        for (int b = 0; b < P1; ++b) acc[a][b] = 0.0;
    // input rows through buffer descriptors: (descriptor of the slot's slice: scalar) + (scalar row offset) + (this lane's
    // point): a load costs no vector instruction and no address registers
    __amdgpu_buffer_rsrc_t rsrc[4][NA];
    int urs[4][NA];
    const int voff = g2 * 8;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            urs[t1][i] = A.rs[R.y][t1][i] * 8;
            rsrc[t1][i] = bf2_rsrc(A.sp[R.y][t1][i] + (long long)r0 * A.ss[R.y][t1][i] - (long long)A.gmid_lo * A.rs[R.y][t1][i]);
        }
    auto ld = [&](const int t1, const int i, const int row) {
#ifdef BF3_NOLOAD
        return (double)(row + t1);                           // (timing experiment: no K1 traffic)
#else
        return bf2_buffer_load(rsrc[t1][i], voff, row * urs[t1][i]);
#endif
    };
    static_assert(Q % D == 0, "k_bf3: the prefetch ring must divide the span");
    double kv[D][4][NA];
    {
        const int s = min(s_begin, max(t_sw - 1, 0));
#pragma unroll
        for (int l = 0; l < D; ++l)
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1)
                if (R.has[t1])
#pragma unroll
                    for (int i = 0; i < NA; ++i) kv[l][t1][i] = ld(t1, i, s * Q + l);
    }
    // (the pairs that enter the window need no zero when a sweep follows: its first plane ASSIGNS them)
    auto flush = [&](const bool zero) {
        double *ln = lines + RI * sc.tlp + g2l;             // (g2l: the point's place in the padded line)
#pragma unroll
        for (int a = 0; a < P1; ++a) ln[a * LS] = acc[a][0];
#pragma unroll
        for (int a = 1; a < P1; ++a) ln[(p1 + a) * LS] = acc[0][a];
#pragma unroll
        for (int a = 0; a < P1 - 1; ++a)
#pragma unroll
            for (int b = 0; b < P1 - 1; ++b) acc[a][b] = acc[a + 1][b + 1];
        if (zero) {
#pragma unroll
            for (int a = 0; a < P1; ++a) { acc[a][P1 - 1] = 0.0; acc[P1 - 1][a] = 0.0; }
        }
    };
    // one span of the mid axis: Q Gauss planes into the pair window, the K1 values of the next span requested meanwhile
    auto sweep_span = [&](const int s) __attribute__((always_inline)) {
        const int tn = min(s + 1, t_sw - 1);
        cdp cf = V1 + (size_t)s * Q * P1 * 2;
        double v[P1][2];
#pragma unroll
        for (int b = 0; b < P1; ++b) { v[b][0] = cf[2 * b]; v[b][1] = cf[2 * b + 1]; }
#pragma unroll
        for (int l = 0; l < Q; ++l) {
            double vn[P1][2];
            const int ln_ = l + 1 < Q ? l + 1 : l;
#pragma unroll
            for (int b = 0; b < P1; ++b) { vn[b][0] = cf[(ln_ * P1 + b) * 2]; vn[b][1] = cf[(ln_ * P1 + b) * 2 + 1]; }
            double kt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1)
                if (R.has[t1]) {
                    kt[t1] = kv[l % D][t1][0];
                    if constexpr (NA == 2) kt[t1] += kv[l % D][t1][1];
                    // the register is free: row l + D of this span, or row l + D - Q of the next one
#pragma unroll
                    for (int i = 0; i < NA; ++i) kv[l % D][t1][i] = ld(t1, i, l + D < Q ? s * Q + l + D : tn * Q + l + D - Q);
                }
            if (R.shape == 1) {
#pragma unroll
                for (int b = 0; b < P1; ++b) {
                    double w;
                    if (R.has[2 * R.f] && R.has[2 * R.f + 1]) w = fma(v[b][1], kt[2 * R.f + 1], v[b][0] * kt[2 * R.f]);
                    else if (R.has[2 * R.f]) w = v[b][0] * kt[2 * R.f];
                    else w = v[b][1] * kt[2 * R.f + 1];
#pragma unroll
                    for (int a = 0; a < P1; ++a) {
                        if (l == 0 && (a == P1 - 1 || b == P1 - 1)) acc[a][b] = v[a][R.f] * w;     // a pair new in the window
                        else acc[a][b] = fma(v[a][R.f], w, acc[a][b]);
                    }
                }
            } else {
                constexpr int tu_first = (R.has[0] || R.has[2]) ? 0 : 1;
#pragma unroll
                for (int tu = 0; tu < 2; ++tu) {
                    if (!(R.has[tu] || R.has[tu + 2])) continue;
#pragma unroll
                    for (int a = 0; a < P1; ++a) {
                        double c;
                        if (R.has[tu] && R.has[tu + 2]) c = fma(v[a][1], kt[tu + 2], v[a][0] * kt[tu]);
                        else if (R.has[tu]) c = v[a][0] * kt[tu];
                        else c = v[a][1] * kt[tu + 2];
#pragma unroll
                        for (int b = 0; b < P1; ++b) {
                            if (l == 0 && tu == tu_first && (a == P1 - 1 || b == P1 - 1)) acc[a][b] = v[b][tu] * c;
                            else acc[a][b] = fma(v[b][tu], c, acc[a][b]);
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < P1; ++a)
#pragma unroll
                for (int b = 0; b < P1; ++b) asm volatile("" : "+v"(acc[a][b]));
#pragma unroll
            for (int b = 0; b < P1; ++b) { v[b][0] = vn[b][0]; v[b][1] = vn[b][1]; }
        }
    };
    if constexpr (!MULT) {
        // single knots on the swept axis: dof d is the first active one of span d -- every step below t_sw sweeps its span (no
        // branch in the loop), the steps behind it only drain the window
        int d = d_begin;
        for (; d < t_sw; ++d) {
            bar_lds();                                   // B1
            if constexpr (ST) store.issue(A, *sc.B, d - 1, sc.sw);
            sweep_span(d);
            bar_lds();                                   // B2: the contractors have read the previous lines
            flush(d + 1 >= t_sw);
            if constexpr (ST) store.fetch(A, *sc.B, sc.sets, sc.dump, d, sc.sw, slane);
        }
        for (; d < rhi; ++d) {
            bar_lds();
            if constexpr (ST) store.issue(A, *sc.B, d - 1, sc.sw);
            bar_lds(); flush(true);
            if constexpr (ST) store.fetch(A, *sc.B, sc.sets, sc.dump, d, sc.sw, slane);
        }
    } else {
        // repeated knots: the window moves by the multiplicity of a knot -- a span is swept when its first active dof is the
        // next one to leave, the other steps only flush
        int s = s_begin;                                  // next span to sweep
        for (int d = d_begin; d < rhi; ++d) {
            bar_lds();                                   // B1
            if constexpr (ST) store.issue(A, *sc.B, d - 1, sc.sw);
            if (s < t_sw && fa1[s] == d) { sweep_span(s); ++s; }
            bar_lds();                                   // B2
            flush(true);
            if constexpr (ST) store.fetch(A, *sc.B, sc.sets, sc.dump, d, sc.sw, slane);
        }
    }
    {                                                    // the contractors finish the last row
        bar_lds();
        if constexpr (ST) store.issue(A, *sc.B, rhi - 1, sc.sw);
        bar_lds();
        if constexpr (ST) store.fetch(A, *sc.B, sc.sets, sc.dump, rhi, sc.sw, slane);
    }
    if constexpr (ST) store.issue(A, *sc.B, rhi, sc.sw);  // the last row
    BF_STAMP_END(threadIdx.x >> 6);
}

template <int P1, int Q, int D, int MASK, int NA, int NLG, class StoreA, class StoreB, bool STW, bool MULT, int RI, bool END = (RI >= bf_nroles(MASK))>
struct BF3SweepDispatch {
    __device__ static __forceinline__ void run(const BFArgs &A, int role, int r0, int g2l, int g2, int s_begin, int t_sw, int d_begin, int rhi, double *lines, int LS, const BF3SweepCtx &sc)
    {
        // role 0 carries no stores; the last role the larger share
        if (role == RI) {
            if constexpr (RI == bf_nroles(MASK) - 1) bf3_sweeper<P1, Q, D, MASK, RI, NA, NLG, StoreB, STW, MULT>(A, r0, g2l, g2, s_begin, t_sw, d_begin, rhi, lines, LS, sc);
            else bf3_sweeper<P1, Q, D, MASK, RI, NA, NLG, StoreA, STW, MULT>(A, r0, g2l, g2, s_begin, t_sw, d_begin, rhi, lines, LS, sc);
        } else BF3SweepDispatch<P1, Q, D, MASK, NA, NLG, StoreA, StoreB, STW, MULT, RI + 1>::run(A, role, r0, g2l, g2, s_begin, t_sw, d_begin, rhi, lines, LS, sc);
    }
};
template <int P1, int Q, int D, int MASK, int NA, int NLG, class StoreA, class StoreB, bool STW, bool MULT, int RI>
struct BF3SweepDispatch<P1, Q, D, MASK, NA, NLG, StoreA, StoreB, STW, MULT, RI, true> {
    __device__ static __forceinline__ void run(const BFArgs &, int, int, int, int, int, int, int, int, double *, int, const BF3SweepCtx &) {}
};

// One unit of the contraction: 64 (line, span) items -- the element matrices of a pass (H = 0) or of one half of their rows
// (H = 1: rows 0 .. AH-1, H = 2: rows AH .. p2; the halves ADD their entries into the rings) -- gathered into the entries of the
// direct and of the transposed row of each lane.
struct BF3Unit {
    const double *lines, *V2s;
    double *sets;
    int dd, rlo, rhi, nhi, row_lo, nrows, lane, npieces;   // nhi: pairs (dd, dd + la) exist for la < nhi (column range of row dd)
    int diag0, stD, stT;
    int rbs1, rbs2, rbs3, rbs4, rbs5;     // ring slot (doubles from the start of a set) of the row parked in ring line delta
};
template <class Gm, int NY, int MASK, int SYM, int H>
__device__ __forceinline__ void bf3_unit(const BF3Unit U, const int pass)
{
    constexpr int P2 = Gm::P2_, Q = Gm::Q_, p1 = Gm::p1, p2 = Gm::p2, W1 = Gm::W1, W2 = Gm::W2;
    constexpr int TL = Gm::TL, LS = Gm::LS, RW = Gm::RW, PL = Gm::PL, PPP = Gm::PPP, NPC = Gm::NPC, RP = Gm::RP;
    constexpr int AH = (P2 + 1) / 2;
    constexpr int A0 = H == 2 ? AH : 0, A1 = H == 1 ? AH : P2;
    const int lane = U.lane, dd = U.dd;
    int ln_ = min(lane, PPP * PL - 1);
    asm volatile("" : "+v"(ln_));
    const int ps = ln_ / PL, x = ln_ - ps * PL;
    const int pid = pass * PPP + ps;
    const int k9 = pid / NPC, pj = pid - k9 * NPC;
    const int s1 = pj * RP + x;
    const bool colt = k9 <= p1;                            // pair (dd + la, dd); else (dd, dd + la)
    const int la = colt ? k9 : k9 - p1;
    const int row1 = colt ? dd + la : dd, col1 = colt ? dd : dd + la;
    // EVERY lane forms its element matrix (a span outside the axis has zero basis values in V2s; a lane without a valid item
    // computes something that is never stored); the LDS addresses are kept inside the image
    const double *kl = U.lines + min(k9, W1 - 1) * LS + min(s1, TL / Q - 1) * Gm::QS, *vl = U.V2s + min(s1, TL / Q - 1) * Gm::VS;
    double loc[A1 - A0][P2];
#ifdef BF3_NOELEM
    {                                                      // (timing experiment: no element matrices)
        const double k0 = kl[0], v0 = vl[0];
#pragma unroll
        for (int a = 0; a < A1 - A0; ++a)
#pragma unroll
            for (int b = 0; b < P2; ++b) { loc[a][b] = k0 + v0 * (a + 3 * b); asm volatile("" : "+v"(loc[a][b])); }
    }
#else
    bf_element<P2, NY, MASK, A0, A1, Q>(loc, kl, vl, Gm::TLP);
#endif
    const int r3 = s1 - p2;                                // row of the tile (this lane's span is the first of its support)
    // (the pair exists when the later dof lies in the column range of the earlier one: la < nhi -- single knots: dd + la < N1)
    const bool item = lane < PPP * PL && pid < U.npieces && x >= p2 && r3 < U.nrows && la < U.nhi;
    const int oshv = max(p2 - (U.row_lo + r3), 0);
    int rb = U.rbs1;
    if (p1 >= 2) rb = la == 2 ? U.rbs2 : rb;
    if (p1 >= 3) rb = la == 3 ? U.rbs3 : rb;
    if (p1 >= 4) rb = la == 4 ? U.rbs4 : rb;
    if (p1 >= 5) rb = la == 5 ? U.rbs5 : rb;
    const int cb = Gm::OFF_CUR + la * RW;
    // ---- direct entries of row i2 = this lane's row: entry o = b - a + p2 from the element matrix of span i2 - a
    if (U.stD) {
        double out[W2];
#pragma unroll
        for (int a = A0; a < A1; ++a)
#pragma unroll
            for (int b = 0; b < P2; ++b) {
                if (a == 0) out[b + p2] = loc[0][b];
                else if (a == A0 || b == 0) out[b - a + p2] = bf2_from_lane(((lane - a) & 63) * 4, loc[a - A0][b]);
                else out[b - a + p2] += bf2_from_lane(((lane - a) & 63) * 4, loc[a - A0][b]);
            }
        if (item && row1 >= U.rlo && row1 < U.rhi) {
            double *dste = U.sets + ((colt && la > 0) ? rb : cb) + r3 * W2 - oshv;
            constexpr int OLO = H == 0 ? 0 : p2 - (A1 - 1), OHI = H == 0 ? 2 * p2 : 2 * p2 - A0;
            if (SYM != 0 && U.diag0) {
                // the diagonal line of a diagonal block: entries j2 <= i2 are direct, the others come transposed (below)
                const int omax = la == 0 ? p2 : 2 * p2;
#pragma unroll
                for (int o = OLO; o <= OHI; ++o)
                    if (o >= oshv && o <= omax) {
                        if (H == 0) dste[o] = out[o];
                        else (void)__hip_atomic_fetch_add(dste + o, out[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            } else if (H == 0) {
                if (U.row_lo >= p2) {
#pragma unroll
                    for (int o = 0; o < W2; ++o) dste[o] = out[o];
                } else {
#pragma unroll
                    for (int o = 0; o < W2; ++o)
                        if (o >= oshv) dste[o] = out[o];
                }
            } else {
                // entries that only this half contributes to are written, the shared ones added (onto zeros, or onto the other half);
                // (a tile without low edge rows: no test per entry)
                if (U.row_lo >= p2) {
#pragma unroll
                    for (int o = OLO; o <= OHI; ++o) {
                        if ((H == 1 && o > 2 * p2 - AH) || (H == 2 && o <= p2 - AH)) dste[o] = out[o];
                        else (void)__hip_atomic_fetch_add(dste + o, out[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
#pragma unroll
                    for (int o = OLO; o <= OHI; ++o)
                        if (o >= oshv) {
                            if ((H == 1 && o > 2 * p2 - AH) || (H == 2 && o <= p2 - AH)) dste[o] = out[o];
                            else (void)__hip_atomic_fetch_add(dste + o, out[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                }
            }
        }
    }
    // (the transposed gather starts when the direct entries are on their way to LDS: both sets of entries live at once do not
    // fit the 128 registers of a contractor wave)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < A1 - A0; ++a)
#pragma unroll
        for (int b = 0; b < P2; ++b) asm volatile("" : "+v"(loc[a][b]));
    __builtin_amdgcn_sched_barrier(0);
    // ---- transposed entries: target row j2 = this lane's row, entry e <-> source row i2 = j2 + e - p2; the addend of span j2 - b
    //      is loc[e - p2 + b][b] of the lane b places below -- the same addends in the same order as the direct entry o = 2 p2 - e
    //      of row i2, hence the same bits
#ifdef BF3_NOT
    if (SYM != 0 && U.diag0 && U.stD) {                     // (timing experiment: no transposed rows of off-diagonal blocks)
#else
    if (SYM != 0 && (U.diag0 ? U.stD : U.stT)) {
#endif
        double outT[W2];
#pragma unroll
        for (int b = 0; b < P2; ++b)
#pragma unroll
            for (int a = A0; a < A1; ++a) {
                if (b == 0) outT[a + p2] = loc[a - A0][0];
                else if (a == A0) outT[a + p2 - b] = bf2_from_lane(((lane - b) & 63) * 4, loc[0][b]);
                else outT[a + p2 - b] += bf2_from_lane(((lane - b) & 63) * 4, loc[a - A0][b]);
            }
        if (item && col1 >= U.rlo && col1 < U.rhi) {
            constexpr int ELO = H == 0 ? 0 : A0, EHI = H == 0 ? 2 * p2 : A1 - 1 + p2;
            if (U.diag0) {
                // upper part of row dd of the same block: the line of column dd + la; on the diagonal line only j2 > i2
                double *dste = U.sets + cb + r3 * W2 - oshv;
                const int emin = la == 0 ? p2 + 1 : oshv;
#pragma unroll
                for (int e = ELO; e <= EHI; ++e)
                    if (e >= emin) {
                        if (H == 0) dste[e] = outT[e];
                        else (void)__hip_atomic_fetch_add(dste + e, outT[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            } else {
                double *dste = U.sets + Gm::SETSZ + (colt ? cb : rb) + r3 * W2 - oshv;
                if (H == 0) {
                    if (U.row_lo >= p2) {
#pragma unroll
                        for (int e = 0; e < W2; ++e) dste[e] = outT[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < W2; ++e)
                            if (e >= oshv) dste[e] = outT[e];
                    }
                } else if (U.row_lo >= p2) {
#pragma unroll
                    for (int e = ELO; e <= EHI; ++e) {
                        if ((H == 1 && e < AH) || (H == 2 && e >= p2 + AH)) dste[e] = outT[e];
                        else (void)__hip_atomic_fetch_add(dste + e, outT[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
#pragma unroll
                    for (int e = ELO; e <= EHI; ++e)
                        if (e >= oshv) {
                            if ((H == 1 && e < AH) || (H == 2 && e >= p2 + AH)) dste[e] = outT[e];
                            else (void)__hip_atomic_fetch_add(dste + e, outT[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                }
            }
        }
    }
}

template <int P1, int P2, int Q, int NY, int MASK, int NA, int NLG, int NCW, int NH, int SYM, bool MULT, bool TR = false>
__global__ void __launch_bounds__((bf_nroles(MASK) * NLG + NCW) * 64) k_bf3(const BFArgs A)
{
    using Gm = BF3Geom<P1, P2, Q, NLG, bf_nroles(MASK), NCW, SYM>;
    constexpr int p2 = P2 - 1, W1 = 2 * P1 - 1, W2 = 2 * P2 - 1, TL = Gm::TL, NR = bf_nroles(MASK), NSW = NR * NLG;
    constexpr int RW = Gm::RW, PPP = Gm::PPP, NPC = Gm::NPC;
    constexpr int LS = Gm::LS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lines = lds;                 // [W1][LS]
    double *sets = lds + Gm::OFF_SETS;   // [NSET][NRL][RMAX][W2]
    double *V2s = lds + Gm::OFF_V2;      // [TL][P2][2]
    bf3_v4i *etab = (bf3_v4i *)(lds + Gm::OFF_ETAB);   // [NEL] {ring offset of the element, 8 rp2[i2], 8 c2, 8 (entry - shift) | out of range}

    cip pl0 = (cip)A.pl0, jlo0 = (cip)A.jlo0, jhi0 = (cip)A.jhi0, fa1 = (cip)A.fa1, mslo1 = (cip)A.mslo1, jhi1 = (cip)A.jhi1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned bid = blockIdx.x;
    const bool tail = A.tail_k > 0 && bid >= A.main_blocks;
    {
        const unsigned per = (A.tail_k > 0 ? A.main_blocks : gridDim.x) / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    int mch, mrows;
    if (tail) {
        const unsigned t = bid - A.main_blocks;
        mch = (int)(t % (unsigned)A.tail_k); mrows = A.tail_mrows;
        bid = A.main_blocks + t / (unsigned)A.tail_k;
    } else { mch = (int)((bid / A.ntiles) % A.nmchunks); mrows = A.mrows; }
    const int tile = (int)(bid % A.ntiles);
    const int r0 = (int)(bid / ((unsigned)A.ntiles * (tail ? 1 : A.nmchunks)));
    const int i0 = pl0[2 * r0], j0 = pl0[2 * r0 + 1];
    const bool diag0 = SYM != 0 && i0 == j0;
    const int row_lo = tile * A.R2, row_hi = min(row_lo + A.R2, A.N2), nrows = row_hi - row_lo;
    const int sp_lo = row_lo - p2;
    const int win0 = sp_lo * Q;
    const int rlo = A.mid_lo + mch * mrows, rhi = min(rlo + mrows, A.mid_hi);
    // the first span that matters is the first of supp(rlo): every pair with a dof >= rlo lies on spans from there on; the
    // steps start with the first dof to leave after it; spans are swept while their first active dof is below rhi
    const int s_begin = mslo1[min(rlo, A.N1 - 1)];
    const int d_begin = fa1[s_begin];
    int t_sw = s_begin;
    {
        const int n_sw = min(A.n1, A.span_hi);
        while (t_sw < n_sw && fa1[t_sw] < rhi) ++t_sw;
    }

    for (int idx = threadIdx.x; idx < TL * P2 * 2; idx += blockDim.x) {
        const int gpt = win0 + idx / (2 * P2);
        const int pt = idx / (2 * P2), sp = pt / Q;
        V2s[sp * Gm::VS + (idx - sp * (Q * P2 * 2))] = (gpt >= 0 && gpt < A.G2) ? A.V2[(long long)win0 * P2 * 2 + idx] : 0.0;
    }
    if (NH >= 2)
        for (int idx = threadIdx.x; idx < Gm::OFF_V2 - Gm::OFF_SETS; idx += blockDim.x) sets[idx] = 0.0;     // the halves of a pass add onto zeros
    // edge rows of the tile: i2 < p2 or i2 > N2 - 1 - p2
    const int lo_n = max(0, min(row_hi, min(p2, A.N2)) - row_lo);          // low edge rows start at row_lo (tile 0 only)
    const int hi_s = max(max(p2, A.N2 - p2), row_lo), hi_n = max(0, row_hi - hi_s);
    const int ne = lo_n + hi_n;
    if (threadIdx.x < Gm::NEL) {
        const int k = threadIdx.x / W2, e = threadIdx.x - k * W2;
        bf3_v4i v; v.x = 0; v.y = 0; v.z = 0; v.w = BF2_OOB;
        if (k < ne) {
            const int i2 = k < lo_n ? row_lo + k : hi_s + (k - lo_n);
            const int jl2 = max(i2 - p2, 0), c2 = min(i2 + p2, A.N2 - 1) + 1 - jl2;
            if (e < c2) {
                const int shift = W2 * row_lo - Gm::T0;
                v.x = (i2 - row_lo) * W2 + e; v.y = 8 * A.rp2[i2]; v.z = 8 * c2; v.w = 8 * (e - shift);
                if constexpr (TR) { v.y = 8 * c2; v.z = 8 * e; v.w = A.rp2[i2]; }
            }
        }
        etab[threadIdx.x] = v;
    }
    // (the barrier B1 of the first iteration orders these writes before their first use)

    int task = wave;
    if (NR == 4 && NLG == 2 && NCW == 8) {
        constexpr int tmap[16] = {0, 1, 2, 3, 6, 7, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15};
        task = tmap[wave & 15];
    } else if (NR == 4 && NLG == 3 && NCW == 4) {
        // (waves w, w + 4, w + 8, w + 12 share a SIMD; tasks 0-2 role 0, 3-5 role 1, 6-8 role 2, 9-11 role 3, 12-15 contractors)
        constexpr int tmap[16] = {12, 13, 14, 15, 9, 3, 0, 2, 10, 4, 1, 7, 11, 6, 5, 8};
        task = tmap[wave & 15];
    } else if (NR == 4 && NLG == 2 && NCW == 4) {
        const int sd = wave & 3, k = wave >> 2;
        if (sd < 3) task = k == 0 ? NSW + sd : 2 * (sd + 1) + (k - 1);
        else task = k < 2 ? k : NSW + 3;
    }
    BF3Blk B;
    B.i0 = i0; B.j0 = j0; B.diag0 = diag0; B.c0i = jhi0[i0] - jlo0[i0]; B.c0j = jhi0[j0] - jlo0[j0];
    B.cj0 = j0 - jlo0[i0]; B.ci0 = i0 - jlo0[j0]; B.rlo = rlo; B.rhi = rhi;
    B.row_lo = row_lo; B.nrows = nrows; B.S12 = A.S1 * A.S2; B.ne = ne;
    B.stD = (i0 >= A.own_lo && i0 < A.own_hi) ? 1 : 0;
    B.stT = ((SYM == 2 || SYM == 3) && !diag0 && j0 >= A.own_lo && j0 < A.own_hi) ? 1 : 0;
    // store duty: on the sweepers of the roles 1.. where the form has them (staged, dense), else on the contractors (at once)
    // (mass: on the sweepers at degree <= 2 -- 0.157 against 0.178 ms at BASELINE config 3 --, on the contractors above: the fifteen
    // staged slots of a sweeper at degree 4 cost 154 registers and 6.5-7.1 against 6.3 ms at C4's size)
    constexpr bool STW = NR >= 2 || (SYM == 3 && BF3_MASS_STW && Q <= BF3_MASS_STW_QMAX);
    constexpr int NQ = (Gm::RMAX * W1 * W2 + 63) / 64;
    using Split = BF3DenseSplit<NR < 2 ? 2 : NR, NLG, NQ>;
    using StoreA = BF3StoreDense<Gm, NH, SYM, Split::KA < 1 ? 1 : Split::KA, 0, Split::QB, Split::NA < 1 ? 1 : Split::NA, TR>;   // middle roles
    using StoreB = BF3StoreDense<Gm, NH, SYM, Split::KB, Split::QB, NQ, Split::NB, TR>;                                       // last role
    using StoreC = BF3Store<Gm, NCW, 1, NH, SYM, TR>;                                                                          // contractors (one-role forms)
    double *dump = lines + NR * Gm::TLP;                  // (the padding of line 0: target of the clears that must not happen)
    if (task < NSW) {
        const int role = task / NLG, lg = task % NLG;
        const int g2l = lg * 64 + lane;
        const int g2 = min(max(win0 + g2l, 0), A.G2 - 1);
        const BF3SweepCtx sc{&B, sets, dump, role == NR - 1 ? lg : (role - 1) * NLG + lg, Gm::TLP};
        BF3SweepDispatch<P1, Q, (NH == 3 ? Q / 2 : Q), MASK, NA, NLG, StoreA, StoreB, STW, MULT, 0>::run(A, role, r0, g2l + (g2l / Q) * (Gm::QS - Q), g2, s_begin, t_sw, d_begin, rhi, lines, LS, sc);
        return;
    }

    // ---------------- contractors
    const int cw = task - NSW;
    BF_STAMP_DECL
    __builtin_amdgcn_s_setprio(BF2_PRIO_C);
    StoreC store;
    if constexpr (!STW) store.init(A, B, cw, lane);
    const bool cdiag = diag0 || SYM == 3;                 // the block is contracted like a diagonal one
    const int nlines = cdiag ? P1 : W1;                   // a diagonal outer block: the pairs (d + a, d) give both halves of its rows
    const int npieces = nlines * NPC;
    for (int t = d_begin; t < rhi + 1; ++t) {
        bar_lds();                                        // B1: the lines of flush t-1 are in LDS
        const int dd = t - 1;
        BF_SEG_BEGIN();
        if (dd >= d_begin && dd < rhi) {
            BF3Unit U;
            U.lines = lines; U.V2s = V2s; U.sets = sets; U.dd = dd; U.rlo = rlo; U.rhi = rhi; U.row_lo = row_lo; U.nrows = nrows;
            U.nhi = jhi1[dd] - dd;
            U.lane = lane; U.npieces = npieces; U.diag0 = cdiag; U.stD = SYM == 3 ? (B.stD | B.stT) : B.stD; U.stT = B.stT;
            // ring slot of the row a line is parked for: row dd + delta of ring line delta
            // (named scalars, not an array: the select by the lane's line must stay a chain of v_cndmask)
            U.rbs1 = Gm::roff(1) + (int)((unsigned)(dd + 1) % 2u) * RW; U.rbs2 = Gm::roff(2) + (int)((unsigned)(dd + 2) % 3u) * RW;
            U.rbs3 = Gm::roff(3) + (int)((unsigned)(dd + 3) % 4u) * RW; U.rbs4 = Gm::roff(4) + (int)((unsigned)(dd + 4) % 5u) * RW;
            U.rbs5 = Gm::roff(5) + (int)((unsigned)(dd + 5) % 6u) * RW;
            const int npass = (npieces + PPP - 1) / PPP;
            const int wslot = (int)((unsigned)(cw + t) % (unsigned)NCW);
            if constexpr (NH == 3) {
                // every pass as two half-units (rows 0 .. AH-1 / AH .. p2 of the element matrices): a whole unit at P2 = 6 does not
                // fit the 128 registers that sixteen waves per CU leave a wave
                for (int u = wslot; u < 2 * npass; u += NCW) {
                    if (u < npass) bf3_unit<Gm, NY, MASK, SYM, 2>(U, u);
                    else bf3_unit<Gm, NY, MASK, SYM, 1>(U, u - npass);
                }
            } else if (NH == 1 || npass <= NCW) {
                for (int pass = BF3_FIXED_PASS ? cw : wslot; pass < npass; pass += NCW) bf3_unit<Gm, NY, MASK, SYM, 0>(U, pass);
            } else if (BF3_FIXED_PASS && 2 * (npass - NCW) == NCW) {
                // as many halves as waves: every wave keeps ITS passes for the whole walk -- the whole pass cw and one half of pass
                // NCW + cw / 2 -- so that what a lane derives from (lane, pass) does not change from step to step; the two waves that
                // share a pass swap halves every step (the halves cost 3 : 2)
                bf3_unit<Gm, NY, MASK, SYM, 0>(U, cw);
                if ((cw ^ t) & 1) bf3_unit<Gm, NY, MASK, SYM, 1>(U, NCW + (cw >> 1));
                else bf3_unit<Gm, NY, MASK, SYM, 2>(U, NCW + (cw >> 1));
            } else {
                bf3_unit<Gm, NY, MASK, SYM, 0>(U, wslot);
                const int nsp = npass - NCW;
                for (int u = wslot; u < 2 * nsp; u += NCW) {
                    if (u < nsp) bf3_unit<Gm, NY, MASK, SYM, 2>(U, NCW + u);
                    else bf3_unit<Gm, NY, MASK, SYM, 1>(U, NCW + u - nsp);
                }
            }
        }
        BF_SEG_END(1);
        bar_lds();                                        // B2: lines may be overwritten, entries are visible
        BF_SEG_BEGIN();
        bf3_edge_rows<Gm, NCW, NH, SYM, TR>(A, B, sets, etab, t, cw, lane);
        if constexpr (!STW) store.fetch(A, B, sets, dump, t, cw, lane);
        BF_SEG_END(0);
    }
    BF_SEG_DUMP(cw & 3);
    BF_STAMP_END(wave);
}

// ---------------------------------------------------------------------------------------------
// host side

// k_bf3 relies on the range check of the buffer descriptors to drop what must not be stored, with 32-bit offsets inside the row
// block of an outer row and 24 x 24-bit multiplications of row constants: the patch must stay inside these.
bool fused3_offsets_fit(int dim, int p0, int p1, int p2, long long S_mid, long long S_last, long long N_last)
{
    const long long c0max = dim == 3 ? 2 * p0 + 1 : 1, c0min = dim == 3 ? p0 + 1 : 1;
    const long long W1 = 2 * p1 + 1, W2 = 2 * p2 + 1;
    const long long len = (c0max * S_mid * S_last + W1 * W2 * 64) * 8;
    if (len > 900000000LL) return false;
    const long long inv = len / (8 * std::max<long long>(c0min * 2 - 1, 1)) + 1;
    if (inv >= (1LL << 24) || W2 * N_last >= (1LL << 24)) return false;
    if (inv * 8 * (c0max * W1 - 1) + len + 65536 >= (1LL << 32)) return false;
    return true;
}

// ... and the exchanged-axes store (TR) multiplies a row of the tile by 8 W2 c0 S_mid and a row constant by 8 c0 S_mid (24-bit)
bool fused3_tr_fits(int p0, int p1, int p2, long long S_mid, long long S_last)
{
    const long long c0max = 2 * p0 + 1, W2 = 2 * p2 + 1;
    (void)p1;
    return 8 * W2 * c0max * S_mid < (1LL << 23) && 2 * S_last + 1 < (1LL << 24) && (2 * S_last + 1) * 8 * c0max * S_mid + 8 * W2 * 16 < (1LL << 32);
}

template <int P1, int P2, int Q, int NY, int MASK, int NA, int NLG, int NCW, int NH, int SYM, bool MULT, bool TR = false>
static int launch_bf3_k(hipStream_t st, const BFArgs &A0, int ncu_ctx)
{
    using Gm = BF3Geom<P1, P2, Q, NLG, bf_nroles(MASK), NCW, SYM>;
    constexpr size_t lds = (size_t)Gm::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "k_bf3: LDS");
    static_assert(Gm::RMAX >= 1 && Gm::RMAX <= 255 && Gm::RW < 4096, "k_bf3: packed slot constants");
    static_assert((bf_nroles(MASK) * NLG + NCW) * 64 <= 1024, "k_bf3: block size");
    static_assert(NCW == 4 || NCW == 8, "k_bf3: contractor waves");
    static_assert(Gm::NSUB * 512 <= 4096, "k_bf3: immediate offsets of the stores");
    static_assert(Gm::NEL <= 1024 && Gm::W1 <= 15, "k_bf3: edge table / line index");
    constexpr int nthreads = (bf_nroles(MASK) * NLG + NCW) * 64;
    const void *fn = (const void *)k_bf3<P1, P2, Q, NY, MASK, NA, NLG, NCW, NH, SYM, MULT, TR>;
    IGX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 1, ncu = ncu_ctx;
    {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, nthreads, lds) == hipSuccess && occ >= 1) per_cu = occ;
        if (ncu < 1) ncu = 256;
    }
    BFArgs A = A0;
    A.ntiles = (A.N2 + Gm::RMAX - 1) / Gm::RMAX;
    A.R2 = (A.N2 + A.ntiles - 1) / A.ntiles;
    bf2_choose_chunks(A, (long long)per_cu * ncu, P1);
    long long nblocks = (long long)A.npairs * A.ntiles * A.nmchunks;
    if (A.tail_k > 0) nblocks = A.main_blocks + (nblocks - A.main_blocks) * A.tail_k;
    if (nblocks > 0x7fffffffLL) { set_error("fused stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
    if (nblocks == 0) return IGX_OK;
    k_bf3<P1, P2, Q, NY, MASK, NA, NLG, NCW, NH, SYM, MULT, TR><<<dim3((unsigned)nblocks), dim3(nthreads), lds, st>>>(A);
    IGX_HIP(hipGetLastError());
#ifdef IGX_BF_STAMP
    {
        static std::vector<unsigned long long> h(64 * 1024);
        IGX_HIP(hipStreamSynchronize(st));
        IGX_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_bf_stamp), h.size() * sizeof(unsigned long long)));
        const int nw = bf_nroles(MASK) * NLG + NCW, nb = (int)std::min<long long>(nblocks, 1024);
        for (int w = 0; w < nw; ++w) {
            double wt = 0, tot = 0;
            for (int b = 0; b < nb; ++b) { wt += h[(b * 16 + w) * 2]; tot += h[(b * 16 + w) * 2 + 1]; }
            fprintf(stderr, "k_bf3 stamp: wave %2d  wait %.0f  total %.0f x10ns/block  (busy %.1f %%)\n", w, wt / nb, tot / nb, 100.0 * (1.0 - wt / tot));
        }
    }
#endif
    return IGX_OK;
}

constexpr int BF_MASK_MASS = 0x0001, BF_MASK_STIFF3 = 0x135F, BF_MASK_STIFF2 = 0x1248;

// shapes (lane groups per role, contractor waves, halved passes) by the larger of the two degrees: those of k_bf2 (fused.hip,
// BF2Cfg: chosen per degree and form from measurements)
#ifndef BF3_NH
#define BF3_NH 2
#endif
#ifndef BF3_FIXED_PASS
#define BF3_FIXED_PASS 1      // the contractor waves keep their passes from step to step (no rotation)
#endif
template <int PM, int MASK> struct BF3Cfg { static constexpr int NLG = 2, NCW = 4, NH = 1; };
// Shape of the degree-4 mass kernel (C4's size).  With the per-axis symmetry (SYM = 3) every block is contracted like a diagonal
// one: 5 lines x 2 pieces = 4 passes per step, so FOUR contractor waves all have a pass (of the eight of round 5 -- chosen when an
// off-diagonal block still had 9 lines -- four only shared the store duty), and a block of 3 + 4 waves leaves room for a second
// one on the CU: k_bf3 5.79 -> 4.95 ms at C4's size (6.33 -> 5.56 on a slower box; profiles/r06_g_c4mass_shapes.txt).  Two lane
// groups (smaller tiles) or the store duty on the sweepers are worse there (5.8 / 7.1 ms).
#ifndef BF3_MASS_P5_NCW
#define BF3_MASS_P5_NLG 3
#define BF3_MASS_P5_NCW 4
#endif
template <int PM> struct BF3Cfg<PM, BF_MASK_MASS> { static constexpr int NLG = PM == 5 ? BF3_MASS_P5_NLG : 2, NCW = PM == 5 ? BF3_MASS_P5_NCW : 4, NH = 1; };
// Degree 5 (PM = 6) of the NON-SYMMETRIC form (SYM = 0, BASELINE config 5): sixteen waves per CU -- three lane groups, 27-row
// tiles, half-units only, half a span of K1 prefetch to fit 128 registers -- against twelve waves of 150 registers: 13.25 against
// 14.1-14.4 ms at C5 (profiles/r06_a_c5_geoa_variants.txt).  The SYMMETRIC form at the same size keeps the twelve-wave shape:
// 8.9 against 12.3 ms (profiles/r06_a_c5s_p6_ab.txt) -- its contractors also gather the transposed rows.
#ifndef BF3_P6
#define BF3_P6 1
#endif
#ifndef BF3_PM3_NLG
#define BF3_PM3_NLG 2         // shape of the degree-2 stiffness kernel (BASELINE config 3): lane groups, contractor waves, halved passes
#define BF3_PM3_NCW 4
#define BF3_PM3_NH 1
#endif
template <int PM, int SYMK = 2> struct BF3CfgS3 {
    static constexpr bool P6 = PM == 6 && BF3_P6 && SYMK == 0;
    static constexpr int NLG = PM == 3 ? BF3_PM3_NLG : PM == 5 || P6 ? 3 : 2, NCW = PM == 3 ? BF3_PM3_NCW : PM == 5 ? 4 : PM == 4 ? 8 : 4,
                         NH = PM == 3 ? BF3_PM3_NH : PM == 5 ? BF3_NH : P6 ? 3 : 1;
};
template <int PM> struct BF3Cfg<PM, BF_MASK_STIFF3> : BF3CfgS3<PM, 2> {};
template <int PM> struct BF3Cfg<PM, BF_MASK_STIFF2> { static constexpr int NLG = 2, NCW = PM <= 5 ? 8 : 4, NH = 1; };

template <int P1, int P2, int Q, int NY, int MASK>
static int launch_bf3_c(hipStream_t st, const BFArgs &A, int ncu, int symk, bool mult_in, bool tr)
{
    const bool mult = mult_in;
    if (tr) {
        // values to the layout of the caller's patch, whose mid and last axis the host exchanged (BF3Store, TR): equal degrees,
        // repeated knots on the swept axis (that is why they were exchanged), the three 3D forms
        if constexpr (P1 == P2 && P1 == Q) {
            if constexpr (MASK == BF_MASK_MASS)
                if (symk == 3 && mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, BF3Cfg<P1, MASK>::NLG, BF3Cfg<P1, MASK>::NCW, BF3Cfg<P1, MASK>::NH, 3, true, true>(st, A, ncu);
            if constexpr (MASK == BF_MASK_STIFF3) {
                using C2 = BF3Cfg<P1, MASK>;
                using C0 = BF3CfgS3<P1, 0>;
                if (symk == 2 && mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C2::NLG, C2::NCW, C2::NH, 2, true, true>(st, A, ncu);
                if (symk == 0 && mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C0::NLG, C0::NCW, C0::NH, 0, true, true>(st, A, ncu);
            }
        }
        set_error("fused stage: no kernel with exchanged axes for this form at these degrees");
        return IGX_ERR_UNSUPPORTED;
    }
    constexpr int PM = Q > (P1 > P2 ? P1 : P2) ? Q : (P1 > P2 ? P1 : P2);   // (the registers of a sweeper follow P1 and Q)
    using C = BF3Cfg<PM, MASK>;
    // (degree 5 at sixteen waves per CU, the non-symmetric form: the general loop -- a span swept under a branch -- happens to be
    // the one hipcc fits into 128 registers without spills, so it also serves single knots there)
    using C0 = BF3CfgS3<PM, 0>;
    if constexpr (MASK == BF_MASK_MASS) {
        if constexpr (P1 == P2 && P1 == Q) {
            if (symk == 3 && mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C::NLG, C::NCW, C::NH, 3, true>(st, A, ncu);
        }
        if (symk == 3 && !mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C::NLG, C::NCW, C::NH, 3, false>(st, A, ncu);
    }
    constexpr bool S2 = !(MASK == BF_MASK_MASS && BF3_MASS_AXSYM);     // (the 3D mass form never takes SYM = 2: not compiled)
    if constexpr (P1 == P2 && P1 == Q) {                   // equal degrees: every form, 2D, repeated knots on the swept axis
        if constexpr (S2)
            if (symk == 2) return mult ? launch_bf3_k<P1, P2, Q, NY, MASK, 1, C::NLG, C::NCW, C::NH, 2, true>(st, A, ncu)
                                       : launch_bf3_k<P1, P2, Q, NY, MASK, 1, C::NLG, C::NCW, C::NH, 2, false>(st, A, ncu);
        if (symk == 1 && !mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C::NLG, C::NCW, C::NH, 1, false>(st, A, ncu);
        if constexpr (MASK == BF_MASK_STIFF3)
            if (symk == 0) {
                if constexpr (C0::P6) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C0::NLG, C0::NCW, C0::NH, 0, true>(st, A, ncu);
                else return mult ? launch_bf3_k<P1, P2, Q, NY, MASK, 1, C0::NLG, C0::NCW, C0::NH, 0, true>(st, A, ncu)
                                 : launch_bf3_k<P1, P2, Q, NY, MASK, 1, C0::NLG, C0::NCW, C0::NH, 0, false>(st, A, ncu);
            }
    } else {
        if constexpr (S2)
            if (symk == 2 && !mult) return launch_bf3_k<P1, P2, Q, NY, MASK, 1, C::NLG, C::NCW, C::NH, 2, false>(st, A, ncu);
    }
    set_error("fused stage: no kernel for this form at these degrees");
    return IGX_ERR_UNSUPPORTED;
}
template <int P1, int P2, int Q>
static int launch_bf3_p(hipStream_t st, const BFArgs &A, int ny, int mask, int ncu, int symk, bool mult, bool tr)
{
    if (ny == 1 && mask == BF_MASK_MASS) return launch_bf3_c<P1, P2, Q, 1, BF_MASK_MASS>(st, A, ncu, symk, mult, tr);
    if (ny == 4 && mask == BF_MASK_STIFF3) return launch_bf3_c<P1, P2, Q, 4, BF_MASK_STIFF3>(st, A, ncu, symk, mult, tr);
    if constexpr (P1 == P2 && P1 == Q)
        if (ny == 4 && mask == BF_MASK_STIFF2) return launch_bf3_c<P1, P2, Q, 4, BF_MASK_STIFF2>(st, A, ncu, symk, mult, tr);
    set_error("fused stage: no kernel for this set of types");
    return IGX_ERR_UNSUPPORTED;
}

// Degrees the kernel is compiled for: equal degrees 1 .. 5 with Q = p + 1 (every form), and -- 3D symmetric forms -- one or
// both of the two axes one or two degrees below Q (the reference's nqp = max degree + 1 rule, pyiga/assemblers.pyx:1338: an
// axis of lower degree is integrated with the points of the highest one).  Anything else takes the stage kernels.
bool fused3_degrees(int P1, int P2, int Q, bool sym3d, bool mid_simple)
{
    if (!mid_simple && !(P1 == P2 && P1 == Q)) return false;     // (repeated knots on the swept axis: equal degrees)
    if (P1 < 2 || P2 < 2 || Q > 6) return false;
    if (P1 == P2 && P1 == Q) return true;
    if (!sym3d || Q < 3) return false;
    const int gap = Q <= 5 ? 2 : 1;                            // two degrees below nqp up to nqp = 5 (round 6), one at nqp = 6
    return P1 >= Q - gap && P2 >= Q - gap;
}

// the symmetric forms (mass, stiffness; 2D and 3D) and the 3D convection-diffusion form (its slots merged by k_geoA), one input
// array per slot
bool fused3_supported(const BFInputs &in)
{
    int mask = 0, ymax = 0;
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1) {
            if (in.slot_n[y][t1] > 1) return false;
            if (in.slot_n[y][t1] > 0) { mask |= 1 << (4 * y + t1); ymax = std::max(ymax, y); }
        }
    if (in.pad_stiff3) {                                   // general first-order forms: a subset of the stiffness slots, the rest zero rows
        if (mask & ~BF_MASK_STIFF3) return false;
        mask = BF_MASK_STIFF3; ymax = 3;
    }
    if (!in.sym) return mask == BF_MASK_STIFF3;
    return (ymax == 0 && mask == BF_MASK_MASS) || mask == BF_MASK_STIFF3 || mask == BF_MASK_STIFF2;
}

int launch_bf3(hipStream_t st, const igx_patch *pt, const BFInputs &in, double *d_data)
{
    const Axis &AM = *in.mid, &AL = *in.last;
    BFArgs A{};
    int mask = 0, ymax = 0;
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1) {
            const int n = in.slot_n[y][t1];
            if (n > 0) { mask |= 1 << (4 * y + t1); ymax = std::max(ymax, y); }
            if (in.pad_stiff3 && ((BF_MASK_STIFF3 >> (4 * y + t1)) & 1)) { mask |= 1 << (4 * y + t1); ymax = std::max(ymax, y); }   // (absent: the zero row)
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
    A.fa1 = AM.dev.fa; A.mslo1 = AM.dev.mslo; A.jlo1 = AM.dev.jlo; A.jhi1 = AM.dev.jhi;
    A.pl0 = in.pl0; A.rp0 = in.rp0; A.jlo0 = in.jlo0; A.jhi0 = in.jhi0;
    A.S1 = AM.S; A.S2 = AL.S; A.nnz_off = pt->nnz_off;
    A.data = d_data; A.sym = in.sym;
    A.mid_lo = in.mid_lo; A.mid_hi = in.mid_hi; A.span_hi = in.span_hi;
    A.npairs = in.npairs;
    if (pt->dim == 3) { A.own_lo = pt->r0_lo; A.own_hi = pt->r0_hi; } else { A.own_lo = 0; A.own_hi = 1; }
    // (the 3D mass form is symmetric per axis: SYM = 3 -- unless the rows of the mid axis are cut by repeated knots with unequal
    // degrees, which k_bf3 does not serve at all)
    const int symk = !in.sym ? 0 : pt->dim == 3 ? ((BF3_MASS_AXSYM && mask == BF_MASK_MASS && ny == 1) ? 3 : 2) : 1;
    const int P1 = AM.P, P2 = AL.P, Q = AL.q;
    if (AM.q != AL.q || !fused3_degrees(P1, P2, Q, symk >= 2, AM.simple) || (!AM.simple && symk == 1)) { set_error("fused stage: degrees (%d, %d) with %d Gauss points per span", P1 - 1, P2 - 1, Q); return IGX_ERR_UNSUPPORTED; }
    if (in.tr && !fused3_tr_fits(pt->ax[0].p, AM.p, AL.p, AM.S, AL.S)) { set_error("fused stage: patch too large for the exchanged-axes store"); return IGX_ERR_UNSUPPORTED; }
#define BF3_CASE(p1, p2, q) if (P1 == p1 && P2 == p2 && Q == q) return launch_bf3_p<p1, p2, q>(st, A, ny, mask, pt->ctx->ncu, symk, !AM.simple, in.tr != 0);
    BF3_CASE(2, 2, 2) BF3_CASE(3, 3, 3) BF3_CASE(4, 4, 4) BF3_CASE(5, 5, 5) BF3_CASE(6, 6, 6)
    BF3_CASE(2, 3, 3) BF3_CASE(3, 2, 3) BF3_CASE(2, 2, 3)
    BF3_CASE(3, 4, 4) BF3_CASE(4, 3, 4) BF3_CASE(3, 3, 4)
    BF3_CASE(4, 5, 5) BF3_CASE(5, 4, 5) BF3_CASE(4, 4, 5)
    BF3_CASE(5, 6, 6) BF3_CASE(6, 5, 6) BF3_CASE(5, 5, 6)
    BF3_CASE(2, 4, 4) BF3_CASE(4, 2, 4) BF3_CASE(2, 3, 4) BF3_CASE(3, 2, 4) BF3_CASE(2, 2, 4)
    BF3_CASE(3, 5, 5) BF3_CASE(5, 3, 5) BF3_CASE(3, 4, 5) BF3_CASE(4, 3, 5) BF3_CASE(3, 3, 5)
#undef BF3_CASE
    set_error("fused stage: degrees (%d, %d) with %d Gauss points per span", P1 - 1, P2 - 1, Q);
    return IGX_ERR_UNSUPPORTED;
}

} // namespace igx

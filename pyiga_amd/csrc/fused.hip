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
#include "fused_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace igx {


// =============================================================================================
// k_bf2: the fused stage for 16 waves per CU (P <= 5).  Same data flow as k_bf -- sweepers flush the completed K2 lines to
// LDS, contractors turn them into CSR row segments -- re-tiled around what bounds the stage, the vector ALU (an FP64
// instruction occupies it for 5 cycles, any other vector instruction for 4; scalar, LDS-read and memory instructions of
// other waves issue under it: tools/ubench/valu_f64.hip):
//   * tile = 64 * NLG Gauss points of the last axis (NLG = 3: 38 spans, 34 rows at p = 4; k_bf: 21 spans, 17 rows on 128
//     lanes), NLG waves per role + 4 contractor waves = 16 waves, four per SIMD (<= 128 VGPRs each);
//   * contractor items are (line, span) as before, but lanes hold PIECES of a line (PL consecutive spans, pieces overlap by
//     p spans, 64 / PL pieces per pass): the entries of row i2 are the sum over the p + 1 spans of its support, i.e. over the
//     element matrices of the p + 1 lanes below its own -- fetched with ds_bpermute, no exchange buffer in LDS, no barrier;
//   * entry rings are row-major ([row][line][entry]): a store slot is one LDS read at a compile-time offset and one store at
//     (scalar row base of the step) + (per-lane offset worked out once): no address arithmetic per element on interior rows
//     of the mid axis; the first and last p rows take a general path;
//   * K1 loads are (scalar base) + (per-lane offset): the row advance is scalar arithmetic.
template <int P, int NLG, int NRO, int NCW, int NH, int NSTW = 0> struct BF2Geom {
    static constexpr int NSW_ = NSTW > 0 ? NSTW : NCW;      // waves that store the finished rows (NSTW > 0: sweepers of the last role)
    static constexpr int p = P - 1, W = 2 * P - 1, TL = 64 * NLG;
    static constexpr int ROWR = p * W, ROWC = P * W;        // doubles per row in the ring / cur parts
    static constexpr int LS = NRO * TL + 2;                 // doubles per line (all roles), padded against bank conflicts
    static constexpr int nslots(int R, int per) { return (R * per * W + 64 * NSW_ - 1) / (64 * NSW_); }
    // LDS image (doubles): lines [W][LS] | ring [P+1][R][p][W] | cur [2][R][P][W] | basis values [TL][P][2] | store plan
    // (ints) [NSR + NSC][NCW * 64];  R = rows of the last axis per tile: as many as the window and 160 KB allow
    static constexpr int off_ring() { return (W * LS + 1) & ~1; }
    static constexpr int off_cur(int R) { return (off_ring() + (P + 1) * R * ROWR + 1) & ~1; }
    static constexpr int off_v2(int R) { return (off_cur(R) + 2 * R * ROWC + 1) & ~1; }
    static constexpr int off_plan(int R) { return off_v2(R) + TL * P * 2; }
    static constexpr int lds_doubles(int R) { return off_plan(R) + (nslots(R, p) + nslots(R, P)) * NSW_ * 32; }
    static constexpr int rmax()
    {
        int R = TL / P - p;
        while (R > 1 && lds_doubles(R) * 8 > 160 * 1024) --R;
        return R;
    }
    static constexpr int RMAX = rmax();
    static constexpr int WS = RMAX + p;                     // spans of the tile window
    static constexpr int OFF_RING = off_ring(), OFF_CUR = off_cur(RMAX), OFF_V2 = off_v2(RMAX), OFF_PLAN = off_plan(RMAX);
    static constexpr int LDS_BYTES = lds_doubles(RMAX) * 8;
    static constexpr int NSR = nslots(RMAX, p), NSC = nslots(RMAX, P);     // store slots per contractor wave: ring / cur part
    // contractor passes: pieces of PL consecutive spans of a line, 64 / PL pieces per pass, consecutive pieces overlap by p spans
    static constexpr int npc(int pl) { return (RMAX + pl - p - 1) / (pl - p); }              // pieces per line
    static constexpr int npass(int pl) { return (W * npc(pl) + 64 / pl - 1) / (64 / pl); }   // passes per step
    static constexpr int pick()
    {
        int best = 64;
        const int cand[4] = {64, 32, 21, 16};
        for (int i = 1; i < 4; ++i)
            if (cand[i] > 2 * p && npass(cand[i]) < npass(best)) best = cand[i];
        return best;
    }
    static constexpr int PL = pick();                       // spans (lanes) of a piece
    static constexpr int PPP = 64 / PL;                     // pieces per pass
    static constexpr int NPC = npc(PL);                     // pieces per line
    static constexpr int RP = PL - p;                       // rows a piece completes
};


// offset (doubles, relative to the row block of mid-axis row dd2) of store element q of a part, or -1.
//   ring part: q = (rr * p + kx) * W + o      entries of the pairs (dd2, j1 < dd2), kx = p - (dd2 - j1)
//   cur part:  q = (rr * P + a) * W + o       entries of the pairs (dd2, dd2 + a)
// o counts the existing columns of row i2 from its first one (packed run, as the rings hold it).
template <int P>
__device__ __forceinline__ int bf2_goff(const BFArgs &A, const BF2Blk &B, const bool cur_part, const int q, const int dd2)
{
    constexpr int p = P - 1, W = 2 * P - 1;
    const int per = cur_part ? P : p;
    const int rr = q / (per * W), e2 = q - rr * (per * W);
    const int k = e2 / W, o = e2 - k * W;
    if (rr >= B.nrows) return -1;
    const int i2 = B.row_lo + rr;
    const int jl2 = max(i2 - p, 0), c2 = min(i2 + p, A.N2 - 1) + 1 - jl2;
    if (o >= c2) return -1;
    const int jl1 = max(dd2 - p, 0), c1 = min(dd2 + p, A.N1 - 1) + 1 - jl1;
    int m;                                                  // line of the segment: j1 = jl1 + m
    if (cur_part) {
        if (dd2 + k >= A.N1) return -1;
        if (B.diag0 && (k > 0 || jl2 + o > i2)) return -1;  // diagonal block: only j1 <= i1, and on the diagonal line j2 <= i2
        m = dd2 - jl1 + k;
    } else {
        const int koff = p - (dd2 - jl1);
        if (k < koff) return -1;
        m = k - koff;
    }
    return B.c0i * c1 * ((cip)A.rp2)[i2] + (B.cj0 * c1 + m) * c2 + o;
}

// The store duty of the fused stage -- read the finished row of the rings behind B2, store it behind the next B1 -- as an
// object that either a contractor wave (the default) or a sweeper wave of the last role (NSTW > 0) carries through its loop.
template <class Gm, int P, int NSW, int NH>
struct BF2Store {
    static constexpr int p = P - 1, NSL = Gm::NSR + Gm::NSC, RMAX = Gm::RMAX;
    static constexpr int RCLAMP = RMAX * Gm::ROWR - 1, CCLAMP = RMAX * Gm::ROWC - 1;
    double sv[NSL];
    int s_soff = 0, s_ddc = 0;
    bool s_on = false, s_int = true;
    __amdgpu_buffer_rsrc_t drs, drs0;
    long long rstep;
    __device__ __forceinline__ void init(const BFArgs &A, const BF2Blk &B, int *plan, const int sw, const int lane)
    {
        cip rp0 = (cip)A.rp0;
#pragma unroll
        for (int k = 0; k < NSL; ++k) sv[k] = 0.0;
        int *myplan = plan + sw * 64 + lane;
        const int ddi = min(p, max(A.N1 - P, 0));                      // any interior row gives the same offsets
#pragma unroll
        for (int k = 0; k < Gm::NSR; ++k) { const int g = bf2_goff<P>(A, B, false, (k * NSW + sw) * 64 + lane, ddi); myplan[k * NSW * 64] = g < 0 ? BF2_OOB : g * 8; }
#pragma unroll
        for (int k = 0; k < Gm::NSC; ++k) { const int g = bf2_goff<P>(A, B, true, (k * NSW + sw) * 64 + lane, ddi); myplan[(Gm::NSR + k) * NSW * 64] = g < 0 ? BF2_OOB : g * 8; }
        drs = bf2_rsrc(A.data + ((long long)rp0[B.i0] * B.S12 - A.nnz_off));
        drs0 = __builtin_amdgcn_make_buffer_rsrc((void *)A.data, (short)0, 0, 0x00020000);
        rstep = (long long)B.c0i * A.S2 * 8;
    }
    // behind B1: the row read at the end of the last step goes out
    __device__ __forceinline__ void issue(const BFArgs &A, const BF2Blk &B, const int *plan, const int sw, const int lane)
    {
        const __amdgpu_buffer_rsrc_t d = s_on ? drs : drs0;
        int lq_ = lane;
        asm volatile("" : "+v"(lq_));
        const int q0 = sw * 64 + lq_;
        if (s_int) {
            const int *myplan = plan + q0;
            int sg[NSL];
#pragma unroll
            for (int k = 0; k < NSL; ++k) sg[k] = myplan[k * NSW * 64];
#pragma unroll
            for (int k = 0; k < NSL; ++k) bf2_buffer_store(d, sg[k], s_soff, sv[k]);
        } else {
            for (int k = 0; k < Gm::NSR; ++k) { const int g = bf2_goff<P>(A, B, false, q0 + k * NSW * 64, s_ddc); bf2_buffer_store(d, g < 0 || !s_on ? BF2_OOB : g * 8, s_soff, sv[k]); }
            for (int k = 0; k < Gm::NSC; ++k) { const int g = bf2_goff<P>(A, B, true, q0 + k * NSW * 64, s_ddc); bf2_buffer_store(d, g < 0 || !s_on ? BF2_OOB : g * 8, s_soff, sv[Gm::NSR + k]); }
        }
    }
    // behind B2 of step t: the entries of mid-axis row t - 1 are complete -- into registers (and cleared where halves add)
    __device__ __forceinline__ void fetch(const BFArgs &A, const BF2Blk &B, double *ring, double *cur, const int t, const int sw, const int lane)
    {
        cip rp1 = (cip)A.rp1;
        const int dd2 = t - 1;
        s_on = dd2 >= B.rlo && dd2 < B.rhi;
        const int ddc = min(max(dd2, 0), A.N1 - 1);
        s_soff = (int)(rstep * rp1[ddc]);
        double *rg = ring + (size_t)(ddc % (P + 1)) * RMAX * Gm::ROWR;
        double *cu = cur + (size_t)(ddc & 1) * RMAX * Gm::ROWC;
        int lq_ = lane;
        asm volatile("" : "+v"(lq_));
        const int q0 = sw * 64 + lq_;
#pragma unroll
        for (int k = 0; k < Gm::NSR; ++k) {
            const int o_ = (k + 1) * NSW * 64 <= RCLAMP + 1 ? q0 + k * NSW * 64 : min(q0 + k * NSW * 64, RCLAMP);
            sv[k] = rg[o_];
            if (NH == 2 && s_on && q0 + k * NSW * 64 <= RCLAMP) rg[o_] = 0.0;      // (only the element's owner clears it)
        }
#pragma unroll
        for (int k = 0; k < Gm::NSC; ++k) {
            const int o_ = (k + 1) * NSW * 64 <= CCLAMP + 1 ? q0 + k * NSW * 64 : min(q0 + k * NSW * 64, CCLAMP);
            sv[Gm::NSR + k] = cu[o_];
            if (NH == 2 && s_on && q0 + k * NSW * 64 <= CCLAMP) cu[o_] = 0.0;
        }
        s_int = !s_on || (ddc >= p && ddc < A.N1 - p);  // (a row that is not stored takes the plan's offsets: its descriptor is empty)
        s_ddc = ddc;
    }
};

struct BF2StoreCtx { int *plan; double *ring, *cur; const BF2Blk *B; int sw; };

template <int P, int MASK, int RI, int NA, int NLG, class Gm, int NSTW, int NH>
__device__ __forceinline__ void bf2_sweeper(const BFArgs &A, const int r0, const int g2l, const int g2, const int s_begin,
                                            const int rhi, double *lines, const int LS, const BF2StoreCtx &sc)
{
    constexpr BFRole R = bf_role(MASK, RI);
    constexpr int p = P - 1, TL = 64 * NLG;
    // NSTW > 0: the sweepers of the LAST role (the lightest: one input array at the 3D stiffness form) carry the store duty
    constexpr bool ST = NSTW > 0 && RI == bf_nroles(MASK) - 1;
    BF2Store<Gm, P, (NSTW > 0 ? NSTW : 1), NH> store;
    const int slane = threadIdx.x & 63;
    if constexpr (ST) store.init(A, *sc.B, sc.plan, sc.sw, slane);
    BF_STAMP_DECL
    __builtin_amdgcn_s_setprio(BF2_PRIO_S);
    cdp V1 = (cdp)A.V1;
    double acc[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) acc[a][b] = 0.0;
    // input rows through buffer descriptors: (descriptor of the slot's slice: scalar) + (scalar row offset) + (this lane's
    // point): a load costs no vector instruction and no address registers
    __amdgpu_buffer_rsrc_t rsrc[4][NA];
    int urs[4][NA];
    const int voff = g2 * 8;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            urs[t1][i] = A.rs[R.y][t1][i] * 8;               // bytes between rows of the mid axis (a slice is < 2^31 bytes: checked on the host)
            rsrc[t1][i] = bf2_rsrc(A.sp[R.y][t1][i] + (long long)r0 * A.ss[R.y][t1][i] - (long long)A.gmid_lo * A.rs[R.y][t1][i]);
        }
    auto ld = [&](const int t1, const int i, const int row) {
        return bf2_buffer_load(rsrc[t1][i], voff, row * urs[t1][i]);
    };
    const int n_sw = min(A.n1, A.span_hi);
    const int t_sw = min(n_sw, rhi);
    double kv[P][4][NA];
    {
        const int s = min(s_begin, t_sw - 1);
#pragma unroll
        for (int l = 0; l < P; ++l)
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1)
                if (R.has[t1])
#pragma unroll
                    for (int i = 0; i < NA; ++i) kv[l][t1][i] = ld(t1, i, s * P + l);
    }
    auto flush = [&]() {
        double *ln = lines + RI * TL + g2l;
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
    };
    int t = s_begin;
    for (; t < t_sw; ++t) {
        bar_lds();                                       // B1
        if constexpr (ST) store.issue(A, *sc.B, sc.plan, sc.sw, slane);
        const int tn = min(t + 1, t_sw - 1);
        cdp cf = V1 + (size_t)t * P * P * 2;
        double v[P][2];
#pragma unroll
        for (int b = 0; b < P; ++b) { v[b][0] = cf[2 * b]; v[b][1] = cf[2 * b + 1]; }
#pragma unroll
        for (int l = 0; l < P; ++l) {
            double vn[P][2];
            const int ln_ = l + 1 < P ? l + 1 : l;
#pragma unroll
            for (int b = 0; b < P; ++b) { vn[b][0] = cf[(ln_ * P + b) * 2]; vn[b][1] = cf[(ln_ * P + b) * 2 + 1]; }
            double kt[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1)
                if (R.has[t1]) {
                    kt[t1] = kv[l][t1][0];
                    if (NA == 2) kt[t1] += kv[l][t1][1];
#pragma unroll
                    for (int i = 0; i < NA; ++i) kv[l][t1][i] = ld(t1, i, tn * P + l);
                }
            if (R.shape == 1) {
#pragma unroll
                for (int b = 0; b < P; ++b) {
                    double w;
                    if (R.has[2 * R.f] && R.has[2 * R.f + 1]) w = fma(v[b][1], kt[2 * R.f + 1], v[b][0] * kt[2 * R.f]);
                    else if (R.has[2 * R.f]) w = v[b][0] * kt[2 * R.f];
                    else w = v[b][1] * kt[2 * R.f + 1];
#pragma unroll
                    for (int a = 0; a < P; ++a) acc[a][b] = fma(v[a][R.f], w, acc[a][b]);
                }
            } else {
#pragma unroll
                for (int tu = 0; tu < 2; ++tu) {
                    if (!(R.has[tu] || R.has[tu + 2])) continue;
#pragma unroll
                    for (int a = 0; a < P; ++a) {
                        double c;
                        if (R.has[tu] && R.has[tu + 2]) c = fma(v[a][1], kt[tu + 2], v[a][0] * kt[tu]);
                        else if (R.has[tu]) c = v[a][0] * kt[tu];
                        else c = v[a][1] * kt[tu + 2];
#pragma unroll
                        for (int b = 0; b < P; ++b) acc[a][b] = fma(v[b][tu], c, acc[a][b]);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < P; ++a)
#pragma unroll
                for (int b = 0; b < P; ++b) asm volatile("" : "+v"(acc[a][b]));
#pragma unroll
            for (int b = 0; b < P; ++b) { v[b][0] = vn[b][0]; v[b][1] = vn[b][1]; }
        }
        bar_lds();                                       // B2: the contractors have read the previous lines
        flush();
        if constexpr (ST) store.fetch(A, *sc.B, sc.ring, sc.cur, t, sc.sw, slane);
    }
    for (; t < rhi; ++t) {                               // spans past the end of the axis: the window only drains
        bar_lds();
        if constexpr (ST) store.issue(A, *sc.B, sc.plan, sc.sw, slane);
        bar_lds(); flush();
        if constexpr (ST) store.fetch(A, *sc.B, sc.ring, sc.cur, t, sc.sw, slane);
    }
    for (; t < rhi + 1; ++t) {                           // the contractors finish the last row
        bar_lds();
        if constexpr (ST) store.issue(A, *sc.B, sc.plan, sc.sw, slane);
        bar_lds();
        if constexpr (ST) store.fetch(A, *sc.B, sc.ring, sc.cur, t, sc.sw, slane);
    }
    if constexpr (ST) store.issue(A, *sc.B, sc.plan, sc.sw, slane);       // the last row
    BF_STAMP_END(threadIdx.x >> 6);
}

template <int P, int MASK, int NA, int NLG, class Gm, int NSTW, int NH, int RI, bool END = (RI >= bf_nroles(MASK))>
struct BF2SweepDispatch {
    __device__ static __forceinline__ void run(const BFArgs &A, int role, int r0, int g2l, int g2, int s_begin, int rhi, double *lines, int LS, const BF2StoreCtx &sc)
    {
        if (role == RI) bf2_sweeper<P, MASK, RI, NA, NLG, Gm, NSTW, NH>(A, r0, g2l, g2, s_begin, rhi, lines, LS, sc);
        else BF2SweepDispatch<P, MASK, NA, NLG, Gm, NSTW, NH, RI + 1>::run(A, role, r0, g2l, g2, s_begin, rhi, lines, LS, sc);
    }
};
template <int P, int MASK, int NA, int NLG, class Gm, int NSTW, int NH, int RI>
struct BF2SweepDispatch<P, MASK, NA, NLG, Gm, NSTW, NH, RI, true> {
    __device__ static __forceinline__ void run(const BFArgs &, int, int, int, int, int, int, double *, int, const BF2StoreCtx &) {}
};

template <int P, int NY, int MASK, int NA, int NLG, int NCW, int NH, int NSTW = 0>
__global__ void __launch_bounds__((bf_nroles(MASK) * NLG + NCW) * 64) k_bf2(const BFArgs A)
{
    static_assert(NSTW == 0 || NSTW == NLG, "store duty on the sweepers: one store wave per lane group of the last role");
    using Gm = BF2Geom<P, NLG, bf_nroles(MASK), NCW, NH, NSTW>;
    constexpr int p = P - 1, W = 2 * P - 1, TL = Gm::TL, NR = bf_nroles(MASK), NSW = NR * NLG;
    constexpr int RMAX = Gm::RMAX, PL = Gm::PL, PPP = Gm::PPP, NPC = Gm::NPC, RP = Gm::RP;
    constexpr int LS = Gm::LS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *lines = lds;                 // [W][LS]
    double *ring = lds + Gm::OFF_RING;   // [P+1][RMAX][p][W]: entries of the pairs (i1, j1 < i1), slot i1 mod (P+1), line kx = p - (i1 - j1)
    double *cur = lds + Gm::OFF_CUR;     // [2][RMAX][P][W]:   entries of the pairs (d, d + a), slot d & 1
    double *V2s = lds + Gm::OFF_V2;      // [TL][P][2]: last-axis basis values on the tile window
    int *plan = (int *)(lds + Gm::OFF_PLAN);   // [NSR + NSC][NCW][64]: byte offsets of the store elements (interior mid-axis rows)

    cip pl0 = (cip)A.pl0, jlo0 = (cip)A.jlo0, jhi0 = (cip)A.jhi0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned bid = blockIdx.x;
    const bool tail = A.tail_k > 0 && bid >= A.main_blocks;
    {
        const unsigned per = (A.tail_k > 0 ? A.main_blocks : gridDim.x) / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    int mch, mrows;
    if (tail) {                                          // (nmchunks = 1 with a tail split)
        const unsigned t = bid - A.main_blocks;
        mch = (int)(t % (unsigned)A.tail_k); mrows = A.tail_mrows;
        bid = A.main_blocks + t / (unsigned)A.tail_k;
    } else { mch = (int)((bid / A.ntiles) % A.nmchunks); mrows = A.mrows; }
    const int tile = (int)(bid % A.ntiles);
    const int r0 = (int)(bid / ((unsigned)A.ntiles * (tail ? 1 : A.nmchunks)));
    const int i0 = pl0[2 * r0], j0 = pl0[2 * r0 + 1];
    const bool diag0 = A.sym && i0 == j0;
    const int row_lo = tile * A.R2, row_hi = min(row_lo + A.R2, A.N2), nrows = row_hi - row_lo;
    const int sp_lo = row_lo - p;                                      // first span of the window (virtual: may lie before the axis)
    const int win0 = sp_lo * P;
    const int rlo = A.mid_lo + mch * mrows, rhi = min(rlo + mrows, A.mid_hi);
    const int s_begin = max(rlo - p, 0);

    for (int idx = threadIdx.x; idx < TL * P * 2; idx += blockDim.x) {
        const int gpt = win0 + idx / (2 * P);
        V2s[idx] = (gpt >= 0 && gpt < A.G2) ? A.V2[(long long)win0 * P * 2 + idx] : 0.0;
    }
    if (NH == 2)
        for (int idx = threadIdx.x; idx < Gm::OFF_V2 - Gm::OFF_RING; idx += blockDim.x) ring[idx] = 0.0;     // the halves of a pass add onto zeros
    // (the barrier B1 of the first iteration orders these writes before their first use)

    // wave -> task: a workgroup's waves go to the four SIMDs cyclically; FP64 instructions per step and wave at p = 4:
    // sweepers of role 0: 370, roles 1, 2: 200, role 3: 165; a contractor pass: 400 (six passes: contractors 0, 1 take two)
    int task = wave;
    if (NR == 4 && NLG == 2 && NCW == 8) {
        // per SIMD: a heavy and a light sweeper (roles 0 + 3, or 1 + 2) and two contractors
        constexpr int tmap[16] = {0, 1, 2, 3, 6, 7, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15};
        task = tmap[wave & 15];
    } else if (NR == 4 && NLG == 3 && NCW == 4) {
#ifndef BF2_TMAP
#define BF2_TMAP 0
#endif
        // (waves w, w + 4, w + 8, w + 12 share a SIMD; tasks 0-2 role 0, 3-5 role 1, 6-8 role 2, 9-11 role 3, 12-15 contractors)
        constexpr int tmaps[3][16] = {{12, 13, 14, 15, 9, 3, 0, 2, 10, 4, 1, 7, 11, 6, 5, 8},
                                      {12, 13, 14, 15, 0, 1, 2, 5, 9, 11, 4, 7, 10, 3, 6, 8},
                                      {12, 13, 14, 15, 0, 1, 2, 5, 6, 3, 4, 7, 9, 10, 11, 8}};
        task = tmaps[BF2_TMAP][wave & 15];
    } else if (NR == 4 && NLG == 2 && NCW == 4) {
        const int sd = wave & 3, k = wave >> 2;
        if (sd < 3) task = k == 0 ? NSW + sd : 2 * (sd + 1) + (k - 1);
        else task = k < 2 ? k : NSW + 3;
    }
    BF2Blk B;
    B.i0 = i0; B.j0 = j0; B.diag0 = diag0; B.c0i = jhi0[i0] - jlo0[i0]; B.cj0 = j0 - jlo0[i0]; B.rlo = rlo; B.rhi = rhi;
    B.row_lo = row_lo; B.nrows = nrows; B.S12 = A.S1 * A.S2;
    if (task < NSW) {
        const int role = task / NLG, lg = task % NLG;
        const int g2l = lg * 64 + lane;
        const int g2 = min(max(win0 + g2l, 0), A.G2 - 1);              // points outside the axis: any finite value (their spans are skipped)
        const BF2StoreCtx sc{plan, ring, cur, &B, lg};
        BF2SweepDispatch<P, MASK, NA, NLG, Gm, NSTW, NH, 0>::run(A, role, r0, g2l, g2, s_begin, rhi, lines, LS, sc);
        return;
    }

    // ---------------- contractors
    const int cw = task - NSW;
    BF_STAMP_DECL
    __builtin_amdgcn_s_setprio(BF2_PRIO_C);
    // store plan of the interior rows of the mid axis (p <= i1 < N1 - p: the segment has 2p + 1 lines, the first one is
    // j1 = i1 - p): offset of element q inside the row block, or -1.  Element q of slot k: q = (k * NCW + cw) * 64 + lane.
    // (kept in LDS, one int per element, written and read by the same wave: the registers belong to the element matrices)
    // Stores: the entries of mid-axis row t - 1 are complete behind the barrier B2 of step t.  They (and their offsets) are
    // read from LDS there -- while the sweepers flush -- and stored right behind the next B1, from registers: the LDS latency
    // lies in the barrier wait, not in front of the pass.  A row that is not stored goes through a descriptor of length 0
    // (every lane out of range): no branch around the stores.  The VALUES of a row wait in registers from B2 to the next B1;
    // their OFFSETS are read from the plan in LDS only at store time (one ds_read each): eleven more registers across the
    // barrier spill, and a scratch reload in front of a store waits for vmcnt(0), i.e. for the stores just issued.  Rows
    // without a plan (first / last p of the mid axis) work their offsets out at store time.  (BF2Store; with NSTW > 0 the
    // sweepers of the last role carry it instead.)
    BF2Store<Gm, P, NCW, NH> store;
    if constexpr (NSTW == 0) store.init(A, B, plan, cw, lane);
    const int nlines = diag0 ? P : W;
    const int npieces = nlines * NPC;
    for (int t = s_begin; t < rhi + 1; ++t) {
        bar_lds();                                        // B1: the lines of flush t-1 are in LDS
        BF_SEG_BEGIN();
#ifndef BF2_NOSTORE
        if constexpr (NSTW == 0) store.issue(A, B, plan, cw, lane);
#endif
        BF_SEG_END(0);
        // ---- contract the lines of flush dd = t - 1 with the last axis
        const int dd = t - 1;
        BF_SEG_BEGIN();
#ifndef BF2_NOPASS
        if (dd >= s_begin && dd < rhi) {
            // A wave issues one vector instruction per 8 cycles at best (tools/ubench/valu_f64.hip), so a step is bound by its longest
            // wave: the passes (64 lanes of (line, span) items) go one per contractor wave; NH = 2: the passes beyond that are cut
            // into two halves -- rows 0 .. AH-1 resp. AH .. p of the element matrices -- for two different waves.  The halves ADD
            // their entries into the rings (ds_add_f64 onto zeros: the stores clear what they have read; two addends commute, the
            // sum does not depend on which wave comes first).  The units rotate over the contractor waves from step to step.
            const int npass = (npieces + PPP - 1) / PPP;
            constexpr int AH = (P + 1) / 2;
            auto unit = [&](auto h_, const int pass) {
                constexpr int H = decltype(h_)::value;                 // 0: whole pass, 1: rows 0 .. AH-1, 2: rows AH .. p
                constexpr int A0 = H == 2 ? AH : 0, A1 = H == 1 ? AH : P;
                int ln_ = min(lane, PPP * PL - 1);                     // (lanes past the last piece repeat its last item: nobody reads them)
                asm volatile("" : "+v"(ln_));
                const int ps = ln_ / PL, x = ln_ - ps * PL;              // piece slot of the pass, span of the piece (kept across the
                // steps in two registers instead: measured, no difference)
                const int pid = pass * PPP + ps;
                const int k9 = pid / NPC, pj = pid - k9 * NPC;
                const int s1 = pj * RP + x;                            // span of the window
                const int la = k9 <= p ? k9 : k9 - p;
                const int row1 = k9 <= p ? dd + la : dd, col1 = k9 <= p ? dd : dd + la;
                const bool lok = pid < npieces && row1 >= rlo && row1 < rhi && col1 < A.N1;
                // EVERY lane forms its element matrix -- no branch, no zero fill for the lanes without a valid item: a span outside
                // the axis has zero basis values in V2s (the staging loop above) and finite K values (the sweepers clamp their
                // points), so its matrix IS zero; a lane whose line is not part of the step (or whose span lies beyond the rows of
                // the tile) computes something that is never stored -- the stored rows of a piece (x >= p) gather from lanes of the
                // same piece only.  The LDS addresses are kept inside the image.
                // (round 4: the branch and the zero fill of 25 registers cost the contractors 0.15 ms at C4: r04_c_c4_bf2_branch_free_ab.txt)
                const double *kl = lines + min(k9, W - 1) * LS + min(s1, TL / P - 1) * P, *vl = V2s + min(s1, TL / P - 1) * P * P * 2;
                // entries of row i2 = sp (the row whose function index a = 0 sits on this lane's span): entry o = b - a + p comes
                // from the element matrix of span i2 - a, i.e. of the lane a places below
                double out[W];
                {
                    double loc[A1 - A0][P];
                    bf_element<P, NY, MASK, A0, A1>(loc, kl, vl, TL);
#pragma unroll
                    for (int a = A0; a < A1; ++a)
#pragma unroll
                        for (int b = 0; b < P; ++b) {
                            // (the first addend of an entry is assigned: out[] is not cleared; entries no row of this unit
                            // contributes to are neither written nor read)
                            if (a == 0) out[b + p] = loc[0][b];
                            else if (a == A0 || b == 0) out[b - a + p] = bf2_from_lane(((lane - a) & 63) * 4, loc[a - A0][b]);
                            else out[b - a + p] += bf2_from_lane(((lane - a) & 63) * 4, loc[a - A0][b]);
                        }
                }
                // (the per-lane constants of the pass are kept across the element matrix: since the store duty became an object the
                // contractors have registers to spare -- 116 of 128 -- and working them out again cost 0.13 ms: r04_a_c4_bf2_keep_consts_ab.txt)
                const int x2 = x, k9w = k9, s1w = s1;
                const int law = la, row1w = row1;
                const bool lokw = lane < PPP * PL && lok;
                const int r3 = s1w - p;                                // row of the tile
                if (lokw && x2 >= p && r3 < nrows) {
                    const int i2 = row_lo + r3;
                    const int oshv = max(p - i2, 0);
                    double *dste = (k9w <= p && law > 0) ? ring + ((size_t)((row1w % (P + 1)) * RMAX + r3) * p + (p - law)) * W
                                                         : cur + ((size_t)((dd & 1) * RMAX + r3) * P + (k9w <= p ? 0 : law)) * W;
                    dste -= oshv;
                    if constexpr (H == 0) {
                        if (row_lo >= p) {                              // (uniform) no row of the tile lacks columns on the left
#pragma unroll
                            for (int o = 0; o < W; ++o) dste[o] = out[o];
                        } else {
#pragma unroll
                            for (int o = 0; o < W; ++o)
                                if (o >= oshv) dste[o] = out[o];
                        }
                    } else {
                        // entries this half contributes to: rows a in [A0, A1) -> o = b - a + p
                        constexpr int OLO = p - (A1 - 1), OHI = 2 * p - A0;
#pragma unroll
                        for (int o = OLO; o <= OHI; ++o)
                            if (o >= oshv) (void)__hip_atomic_fetch_add(dste + o, out[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            };
            const int wslot = (int)((unsigned)(cw + t) % (unsigned)NCW);
            if (NH == 1 || npass <= NCW) {
                for (int pass = wslot; pass < npass; pass += NCW) unit(std::integral_constant<int, 0>(), pass);
            } else {
                unit(std::integral_constant<int, 0>(), wslot);
                // the halves of the passes NCW .. npass-1: second halves (lighter) first in the rotation
                const int nsp = npass - NCW;
                for (int u = wslot; u < 2 * nsp; u += NCW) {
                    if (u < nsp) unit(std::integral_constant<int, 2>(), NCW + u);
                    else unit(std::integral_constant<int, 1>(), NCW + u - nsp);
                }
            }
        }
#endif
        BF_SEG_END(1);
        bar_lds();                                        // B2: lines may be overwritten, entries are visible
#ifndef BF2_NOSTORE
        if constexpr (NSTW == 0) store.fetch(A, B, ring, cur, t, cw, lane);
#endif
    }
#ifndef BF2_NOSTORE
    if constexpr (NSTW == 0) store.issue(A, B, plan, cw, lane);       // the last row
#endif
    BF_SEG_DUMP(cw & 3);
    BF_STAMP_END(wave);
}

// ---------------------------------------------------------------------------------------------
// Mirror pass: upper-triangle entries from the lower triangle.  Block = (target outer pair (i0, j0 >= i0), chunk of
// rows i2 of the last axis, range of rows i1 of the mid axis); it walks its rows i1.  Per step the source runs
//     row (j0, j1, j2), columns (i0, i1, i2' in chunk)        for the j1 of the segment and j2 around the chunk
// are gathered into an LDS tile (reads: runs of 2p+1 doubles whose neighbours in memory are the runs of the next steps,
// served by the XCD's L2), and the target segments  row (i0, i1, i2), columns (j0, j1, j2)  are written whole.
// Everything about an element that does not depend on i1 -- which (j2, i2') it is, its place in the tile, the row
// terms of its CSR position -- is worked out once per thread; a step costs two multiply-adds per element.
// (Both axes have single interior knots and the same degree here: the pass follows k_bf.)
struct MirrorArgs {
    double *data;
    long long nnz_off, S1, S2;
    const int *rp0, *jlo0, *jhi0, *rp1, *rp2;
    int N1, N2, p;
    int i1_lo, i1_hi;            // target rows of the mid axis
    int i1_rows, ni1;            // rows i1 per block, blocks per (pair, chunk)
    const int *tpairs;           // [ntp][2] target pairs
    int ntp, RC, nchunks;
    int chunk_fastest;           // block order: the chunks of the last axis of one (pair, row range) are neighbours (k_mirror2)
};

template <int WW> struct MirrorGeom {
    static constexpr int RCM = WW == 3 ? 128 : WW == 5 ? 128 : WW == 7 ? 112 : WW == 9 ? 66 : 40;   // rows i2 per block (tile <= 48 KB)
    static constexpr int NJ2 = RCM + WW - 1;                           // source rows j2 around a chunk
    static constexpr int SG = (NJ2 * WW + 255) / 256;                  // gather elements (j2, offset) per thread and j1
    static constexpr int SS = (RCM * WW * WW + 255) / 256;             // target elements per thread
};

template <int WW>
__global__ void __launch_bounds__(256) k_mirror(const MirrorArgs M)
{
    using Gm = MirrorGeom<WW>;
    constexpr int p = (WW - 1) / 2, NJ2 = Gm::NJ2, SG = Gm::SG, SS = Gm::SS;
    extern __shared__ __attribute__((aligned(16))) double T[];           // [WW][NJ2][WW]
    cip rp0 = (cip)M.rp0, jlo0 = (cip)M.jlo0, jhi0 = (cip)M.jhi0, rp1 = (cip)M.rp1;
    unsigned bid = blockIdx.x;
    {
        const unsigned per = gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    const int ib = (int)(bid % M.ni1), chunk = (int)((bid / M.ni1) % M.nchunks), tp = (int)(bid / ((unsigned)M.ni1 * M.nchunks));
    const int i0 = ((cip)M.tpairs)[2 * tp], j0 = ((cip)M.tpairs)[2 * tp + 1];
    const bool diag = i0 == j0;
    const int c0i = jhi0[i0] - jlo0[i0], c0j = jhi0[j0] - jlo0[j0];
    const int cl = chunk * M.RC, ch = min(cl + M.RC, M.N2);
    const int j2lo = max(cl - p, 0), j2hi = min(ch - 1 + p, M.N2 - 1) + 1, nj2 = j2hi - j2lo;
    const long long S12 = M.S1 * M.S2;
    auto jlo = [&](const int i) { return max(i - p, 0); };
    auto cnt = [&](const int i, const int N) { return min(i + p, N - 1) + 1 - max(i - p, 0); };
    // ---- gather plan: element (j2, op) of a source run = column i2' = jlo2[j2] + op
    int g_rp[SG], g_pk[SG];              // rp2[j2] | c2, op, flags (1: valid, 2: j2 <= i2'), tile offset
#pragma unroll
    for (int s_ = 0; s_ < SG; ++s_) {
        const int f = threadIdx.x + 256 * s_;
        const int j2r = min(f / WW, nj2 - 1), op = f - (f / WW) * WW;
        const int j2 = j2lo + j2r, c2 = cnt(j2, M.N2), i2p = jlo(j2) + op;
        const bool ok = f < nj2 * WW && op < c2 && i2p >= cl && i2p < ch;
        g_rp[s_] = M.rp2[j2];
        g_pk[s_] = c2 | (op << 4) | (ok ? 256 : 0) | (j2 <= i2p ? 512 : 0) | ((j2r * WW + op) << 10);
    }
    // ---- target plan: element (i2, j1 index m, offset o) of a target segment
    int t_rp[SS], t_pk[SS], t_off[SS];    // rp2[i2] | c2, m, o, flags (1: valid, 2: j2 <= i2) | tile offset
#pragma unroll
    for (int s_ = 0; s_ < SS; ++s_) {
        const int q = threadIdx.x + 256 * s_;
        const int rr = q / (WW * WW), e2 = q - rr * (WW * WW);
        const int m = e2 / WW, o = e2 - m * WW;
        const int i2 = min(cl + rr, M.N2 - 1), c2 = cnt(i2, M.N2), j2 = min(jlo(i2) + o, M.N2 - 1);
        const bool ok = cl + rr < ch && o < c2;
        t_rp[s_] = M.rp2[i2];
        t_pk[s_] = c2 | (m << 4) | (o << 8) | (ok ? 4096 : 0) | (j2 <= i2 ? 8192 : 0);
        t_off[s_] = (m * NJ2 + (j2 - j2lo)) * WW + (i2 - jlo(j2));
    }
    const int i1b = M.i1_lo + ib * M.i1_rows, i1e = min(i1b + M.i1_rows, M.i1_hi);
    for (int i1 = i1b; i1 < i1e; ++i1) {
        const int jl1i = jlo(i1), c1i = cnt(i1, M.N1);
        const int a1 = diag ? i1 : jl1i, nj1 = jl1i + c1i - a1;
        // ---- gather: all loads of the step are in flight before the first value is used
        double v[WW][SG];
#pragma unroll
        for (int m = 0; m < WW; ++m) {
            const int j1 = min(a1 + m, M.N1 - 1), c1j = cnt(j1, M.N1);
            const long long Rj = (long long)rp0[j0] * S12 + (long long)c0j * rp1[j1] * M.S2 - M.nnz_off;
            const long long A1 = (long long)c0j * c1j;
            const int B1 = (i0 - jlo0[j0]) * c1j + (i1 - jlo(j1));
#pragma unroll
            for (int s_ = 0; s_ < SG; ++s_) {
                const int pk = g_pk[s_], c2 = pk & 15, op = (pk >> 4) & 15;
                const bool ok = m < nj1 && (pk & 256) && !(diag && m == 0 && (pk & 512));
                const long long src = ok ? Rj + A1 * g_rp[s_] + B1 * c2 + op : 0;
                v[m][s_] = M.data[src];
            }
        }
#pragma unroll
        for (int m = 0; m < WW; ++m)
#pragma unroll
            for (int s_ = 0; s_ < SG; ++s_)
                if (threadIdx.x + 256 * s_ < nj2 * WW) T[m * (NJ2 * WW) + (g_pk[s_] >> 10)] = v[m][s_];     // (elements past the chunk alias its last row)
        __syncthreads();
        // ---- whole target segments
        const long long Ri = (long long)rp0[i0] * S12 + (long long)c0i * rp1[i1] * M.S2 - M.nnz_off;
        const long long A = (long long)c0i * c1i;
        const int B = (j0 - jlo0[i0]) * c1i + (a1 - jl1i);
#pragma unroll
        for (int s_ = 0; s_ < SS; ++s_) {
            const int pk = t_pk[s_], c2 = pk & 15, m = (pk >> 4) & 15, o = (pk >> 8) & 15;
            if ((pk & 4096) && m < nj1 && !(diag && m == 0 && (pk & 8192)))
                M.data[Ri + A * t_rp[s_] + (B + m) * c2 + o] = T[t_off[s_]];
        }
        __syncthreads();
    }
}

// k_mirror2: the same pass with 32-bit offsets inside the row blocks of the two outer rows (buffer descriptors: scalar base of
// the step + per-thread offsets: two multiply-adds per element instead of 64-bit address arithmetic) and with the gather of
// the NEXT row i1 in flight under the stores of the current one.  Used when a row block (c0 S1 S2 values) is below 2^31
// bytes; k_mirror otherwise.
#ifndef MIRROR_ST_AUX
#define MIRROR_ST_AUX 0                                  // cache policy bits of the mirror's stores (2: nt)
#endif
#ifndef MIRROR_IB
#define MIRROR_IB 3                                      // target rows gathered together
#endif
#ifndef MIRROR_VAR
#define MIRROR_VAR 1                                     // tile of the p = 4 case: 1: 33 rows, 0: 44 rows
#endif
template <int WW, int VAR> struct Mirror2Geom {
    static constexpr int RCM = WW == 9 ? (VAR == 1 ? 33 : VAR == 2 ? 22 : VAR == 3 ? 16 : VAR == 5 ? 20 : 44) : WW == 7 ? 56 : 64;     // rows i2 per block
    static constexpr int IB = MIRROR_IB;                // target rows i1 gathered together
    static constexpr int NJ2 = RCM + WW - 1;
    static constexpr int SG = (NJ2 * WW + 255) / 256;
    static constexpr int SS = (RCM * WW * WW + 255) / 256;
};

// The source runs of CONSECUTIVE target rows i1 are neighbours in memory (72 bytes apart inside the source segment) and a
// read request is a whole 128-byte line (profiles/r03_fetch_calibration.txt): gathered one row at a time a line is fetched
// again a step later, when the L2 has long dropped it (2.7x the useful bytes).  IB rows are therefore gathered TOGETHER --
// the second touch of a line follows within the same burst -- and written out one after the other through the tile.
template <int WW, int VAR>
__global__ void __launch_bounds__(256) k_mirror2(const MirrorArgs M)
{
    using Gm = Mirror2Geom<WW, VAR>;
    constexpr int p = (WW - 1) / 2, NJ2 = Gm::NJ2, SG = Gm::SG, SS = Gm::SS, IB = Gm::IB;
    extern __shared__ __attribute__((aligned(16))) double T[];           // [WW][NJ2][WW]
    cip rp0 = (cip)M.rp0, jlo0 = (cip)M.jlo0, jhi0 = (cip)M.jhi0, rp1 = (cip)M.rp1;
    unsigned bid = blockIdx.x;
    {
        const unsigned per = gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    int ib, chunk;
    if (M.chunk_fastest) { chunk = (int)(bid % M.nchunks); ib = (int)((bid / M.nchunks) % M.ni1); }
    else { ib = (int)(bid % M.ni1); chunk = (int)((bid / M.ni1) % M.nchunks); }
    const int tp = (int)(bid / ((unsigned)M.ni1 * M.nchunks));
    const int i0 = ((cip)M.tpairs)[2 * tp], j0 = ((cip)M.tpairs)[2 * tp + 1];
    const bool diag = i0 == j0;
    const int c0i = jhi0[i0] - jlo0[i0], c0j = jhi0[j0] - jlo0[j0];
    const int cl = chunk * M.RC, ch = min(cl + M.RC, M.N2);
    const int j2lo = max(cl - p, 0), j2hi = min(ch - 1 + p, M.N2 - 1) + 1, nj2 = j2hi - j2lo;
    const long long S12 = M.S1 * M.S2;
    auto jlo = [&](const int i) { return max(i - p, 0); };
    auto cnt = [&](const int i, const int N) { return min(i + p, N - 1) + 1 - max(i - p, 0); };
    const __amdgpu_buffer_rsrc_t d0 = __builtin_amdgcn_make_buffer_rsrc((void *)M.data, (short)0, 0, 0x00020000);
    // ---- gather plan: element (j2, op) of a source run = column i2' = jlo2[j2] + op.  Offset inside the row block of (j0, j1):
    //      8 (A1 rp2[j2] + B1 c2 + op), A1 = c0j c1j, B1 = (i0 - jlo0[j0]) c1j + i1 - jlo1(j1): scalars of the step
    int g_a[SG], g_c[SG], g_o[SG], g_t[SG];      // 8 rp2[j2] | 8 c2 | 8 op or BF2_OOB | tile offset, bit 30: j2 <= i2', bit 29: no element
#pragma unroll
    for (int s_ = 0; s_ < SG; ++s_) {
        const int f = threadIdx.x + 256 * s_;
        const int j2r = min(f / WW, nj2 - 1), op = f - (f / WW) * WW;
        const int j2 = j2lo + j2r, c2 = cnt(j2, M.N2), i2p = jlo(j2) + op;
        const bool ok = f < nj2 * WW && op < c2 && i2p >= cl && i2p < ch;
        g_a[s_] = M.rp2[j2] * 8; g_c[s_] = c2 * 8; g_o[s_] = ok ? op * 8 : BF2_OOB;
        g_t[s_] = (j2r * WW + op) | (j2 <= i2p ? 1 << 30 : 0) | (f < nj2 * WW ? 0 : 1 << 29);
    }
    // ---- target plan: element (i2, line m, offset o) of a target segment: 8 (A rp2[i2] + (B + m) c2 + o), A = c0i c1i,
    //      B = (j0 - jlo0[i0]) c1i + a1 - jlo1(i1)
    int t_a[SS], t_c[SS], t_m[SS], t_off[SS];    // 8 rp2[i2] | 8 c2 | 8 (m c2 + o) or BF2_OOB | tile offset, m in bits 24.., bit 30: j2 <= i2
#pragma unroll
    for (int s_ = 0; s_ < SS; ++s_) {
        const int q = threadIdx.x + 256 * s_;
        const int rr = q / (WW * WW), e2 = q - rr * (WW * WW);
        const int m = e2 / WW, o = e2 - m * WW;
        const int i2 = min(cl + rr, M.N2 - 1), c2 = cnt(i2, M.N2), j2 = min(jlo(i2) + o, M.N2 - 1);
        const bool ok = cl + rr < ch && o < c2;
        t_a[s_] = M.rp2[i2] * 8; t_c[s_] = c2 * 8; t_m[s_] = ok ? (m * c2 + o) * 8 : BF2_OOB;
        t_off[s_] = ((m * NJ2 + (j2 - j2lo)) * WW + (i2 - jlo(j2))) | (m << 24) | (j2 <= i2 ? 1 << 30 : 0);
    }
    const int i1b = M.i1_lo + ib * M.i1_rows, i1e = min(i1b + M.i1_rows, M.i1_hi);
    double v[IB][WW][SG];
    auto gather = [&](const int d, const int i1r) {
        const int i1 = min(i1r, i1e - 1);                    // (rows past the end repeat the last one; they are not written)
        const int jl1i = jlo(i1), c1i = cnt(i1, M.N1);
        const int a1 = diag ? i1 : jl1i, nj1 = jl1i + c1i - a1;
#pragma unroll
        for (int m = 0; m < WW; ++m) {
            const int j1 = min(a1 + m, M.N1 - 1), c1j = cnt(j1, M.N1);
            const __amdgpu_buffer_rsrc_t dsc = m < nj1 ? bf2_rsrc(M.data + ((long long)rp0[j0] * S12 + (long long)c0j * rp1[j1] * M.S2 - M.nnz_off)) : d0;
            const int A1 = c0j * c1j, B1 = (i0 - jlo0[j0]) * c1j + (i1 - jlo(j1));
#pragma unroll
            for (int s_ = 0; s_ < SG; ++s_) {
                int off = A1 * g_a[s_] + B1 * g_c[s_] + g_o[s_];
                if (g_o[s_] == BF2_OOB || (m == 0 && diag && (g_t[s_] & (1 << 30)))) off = BF2_OOB;
                v[d][m][s_] = bf2_buffer_load(dsc, off, 0);
            }
        }
    };
    for (int i1s = i1b; i1s < i1e; i1s += IB) {
#pragma unroll
        for (int d = 0; d < IB; ++d) gather(d, i1s + d);     // IB rows in flight: neighbouring runs of a source segment together
#pragma unroll
        for (int d = 0; d < IB; ++d) {
            const int i1 = i1s + d;
            const bool on = i1 < i1e;
            const int i1c = min(i1, i1e - 1);
            const int jl1i = jlo(i1c), c1i = cnt(i1c, M.N1);
            const int a1 = diag ? i1c : jl1i, nj1 = jl1i + c1i - a1;
#pragma unroll
            for (int m = 0; m < WW; ++m)
#pragma unroll
                for (int s_ = 0; s_ < SG; ++s_)
                    if (!(g_t[s_] & (1 << 29))) T[m * (NJ2 * WW) + (g_t[s_] & 0xffffff)] = v[d][m][s_];
            __syncthreads();
            // ---- whole target segments
            const __amdgpu_buffer_rsrc_t dt = on ? bf2_rsrc(M.data + ((long long)rp0[i0] * S12 + (long long)c0i * rp1[i1c] * M.S2 - M.nnz_off)) : d0;
            const int A = c0i * c1i, B = (j0 - jlo0[i0]) * c1i + (a1 - jl1i);
#pragma unroll
            for (int s_ = 0; s_ < SS; ++s_) {
                const int m = (t_off[s_] >> 24) & 15;
                int off = A * t_a[s_] + B * t_c[s_] + t_m[s_];
                if (t_m[s_] == BF2_OOB || m >= nj1 || (diag && m == 0 && (t_off[s_] & (1 << 30)))) off = BF2_OOB;
                bf2_buffer_store<MIRROR_ST_AUX>(dt, off, 0, T[t_off[s_] & 0xffffff]);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side

// k_bf2 reaches the K1 rows of a slice and the CSR values of an outer row through buffer descriptors with 32-bit offsets
// (scalar row offset + per-lane offset, range-checked against BF2_NUMREC); k_mirror2 does the same inside the row blocks of
// two outer rows.  A patch beyond these limits would have its stores dropped silently by the range check, so it never gets
// here: sumfact_assemble takes the stage kernels (64-bit addresses) instead.
bool fused_offsets_fit(long long c0max, long long S_mid, long long S_last, long long G_mid, long long G_last)
{
    constexpr long long LIM = 0x7fff0000LL;
    if (c0max < 1 || S_mid < 0 || S_last < 0 || G_mid < 0 || G_last < 0) return false;
    if (S_mid > LIM || S_last > LIM || G_mid > LIM || G_last > LIM) return false;
    if (c0max * S_mid > LIM / 8 || c0max * S_mid * S_last > LIM / 8) return false;
    if (G_mid * G_last > LIM / 8) return false;
    return true;
}

constexpr int BF_MASK_MASS = 0x0001, BF_MASK_STIFF3 = 0x135F, BF_MASK_STIFF2 = 0x1248;

template <int P, int NY, int MASK, int NA, int NLG, int NCW, int NH, int NSTW = 0>
static int launch_bf2_k(hipStream_t st, const BFArgs &A0, unsigned nblocks, int ncu_ctx)
{
    using Gm = BF2Geom<P, NLG, bf_nroles(MASK), NCW, NH, NSTW>;
    constexpr size_t lds = (size_t)Gm::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "k_bf2: LDS");
    static_assert((bf_nroles(MASK) * NLG + NCW) * 64 <= 1024, "k_bf2: block size");
    constexpr int nthreads = (bf_nroles(MASK) * NLG + NCW) * 64;
    IGX_HIP(hipFuncSetAttribute((const void *)k_bf2<P, NY, MASK, NA, NLG, NCW, NH, NSTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // resident blocks per CU: a property of the code object (registers, LDS), asked per launch for the CURRENT device -- the
    // query is host arithmetic over the kernel descriptor (~1 us), so nothing is cached across devices or threads; the CU
    // count comes from the context the patch belongs to
    int per_cu = 1, ncu = ncu_ctx;
    {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_bf2<P, NY, MASK, NA, NLG, NCW, NH, NSTW>, nthreads, lds) == hipSuccess && occ >= 1) per_cu = occ;
        if (ncu < 1) ncu = 256;
    }
    BFArgs A = A0;
#ifdef IGX_ABLATE
    if (!getenv("IGX_BF_MCHUNKS"))
#endif
    {
        bf2_choose_chunks(A, (long long)per_cu * ncu, P);
        nblocks = (unsigned)((long long)A.npairs * A.ntiles * A.nmchunks);
        if (A.tail_k > 0) nblocks = A.main_blocks + (nblocks - A.main_blocks) * (unsigned)A.tail_k;
    }
    k_bf2<P, NY, MASK, NA, NLG, NCW, NH, NSTW><<<dim3(nblocks), dim3(nthreads), lds, st>>>(A);
    IGX_HIP(hipGetLastError());
#ifdef IGX_BF_STAMP
    {
        static std::vector<unsigned long long> h(64 * 1024);
        IGX_HIP(hipStreamSynchronize(st));
        IGX_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_bf_stamp), h.size() * sizeof(unsigned long long)));
        const int nw = bf_nroles(MASK) * NLG + NCW, nb = std::min<unsigned>(nblocks, 2048);
        for (int w = 0; w < nw; ++w) {
            double wt = 0, tot = 0;
            for (int b = 0; b < nb; ++b) { wt += h[(b * 16 + w) * 2]; tot += h[(b * 16 + w) * 2 + 1]; }
            fprintf(stderr, "k_bf2 stamp: wave %2d  wait %.0f  total %.0f x100ns/block  (busy %.1f %%)\n", w, wt / nb, tot / nb, 100.0 * (1.0 - wt / tot));
        }
        for (int w = 0; w < 4; ++w) {
            double sg[3] = {0, 0, 0};
            for (int b = 0; b < nb; ++b) for (int i = 0; i < 3; ++i) sg[i] += h[32768 + (b * 4 + w) * 3 + i];
            fprintf(stderr, "k_bf2 stamp: contractor %d (+4)  stores %.0f  passes %.0f x100ns/block\n", w, sg[0] / nb, sg[1] / nb);
        }
    }
#endif
    return IGX_OK;
}


// Shapes of k_bf2 per set of types: lane groups per role (tile = 64 NLG points), contractor waves, units per pass.  A wave
// issues at most one vector instruction per 8 cycles (tools/ubench/valu_f64.hip: one wave per SIMD), so a step is bound by
// its longest wave as much as by the SIMDs: three lane groups + four contractors (two of them take two passes per step) beat
// two lane groups + eight contractors at the 3D forms because the larger tile needs fewer instructions per row.
// (P = 6: the window of a 192-point tile does not fit LDS next to the rings; two lane groups, 168 registers per wave)
#ifndef BF2_NSTW
#define BF2_NSTW 0                                       // 3: the sweepers of the last role store the finished rows (p = 4 stiffness)
#endif
template <int P, int MASK, int NA> struct BF2Cfg { static constexpr int NLG = 2, NCW = 4, NH = 1, NSTW = 0; };
// mass, measured per degree (k_bf2 ms, (NLG, NCW)): p = 1 n = 96: (3,8) 0.13, (2,4) 0.08; p = 2 n = 64: 0.11 / 0.08;
// p = 3 n = 96: 1.78 / 1.30; p = 4 n = 128: (3,8) 3.89, (2,4) 4.34
template <int P> struct BF2Cfg<P, BF_MASK_MASS, 1> { static constexpr int NLG = P == 5 ? 3 : 2, NCW = P == 5 ? 8 : 4, NH = 1, NSTW = 0; };
// 3D stiffness, measured per degree (tools/shape_try.py; k_bf2 ms at n = 64 / 128, shapes (NLG, NCW, NH)):
//   p = 1: (3,4,2) 0.32 / 0.61, (2,4,1) 0.25 / 0.56        p = 2: (3,4,2) 0.29 / 1.53, (2,4,1) 0.21 / 1.53
//   p = 3: (3,4,2) 0.92 / 6.42, (3,4,1) 0.74 / 4.78, (2,8,1) 0.70 / 4.46   (the halved passes cost more than they balance)
//   p = 4: (3,4,2) 7.1 at n = 128 (C4), (3,4,1) 7.1, (2,8,1) 8.5
template <int P> struct BF2Cfg<P, BF_MASK_STIFF3, 1> {
    static constexpr int NLG = P == 5 ? BF2_NLG : 2, NCW = P == 5 ? BF2_NCW : P == 4 ? 8 : 4, NH = P == 5 ? BF2_NH : 1;
    static constexpr int NSTW = P == 5 && BF2_NSTW ? NLG : 0;
};
template <int P> struct BF2Cfg<P, BF_MASK_STIFF2, 1> { static constexpr int NLG = 2, NCW = P <= 5 ? 8 : 4, NH = 1, NSTW = 0; };
template <int P, int NY, int MASK, int NA>
static int launch_bf2_c(hipStream_t st, const BFArgs &A, unsigned nblocks, int ncu)
{
    using C = BF2Cfg<P, MASK, NA>;
    return launch_bf2_k<P, NY, MASK, NA, C::NLG, C::NCW, C::NH, C::NSTW>(st, A, nblocks, ncu);
}
template <int P, int MASK, int NA> constexpr int bf2_rmax()
{
    using C = BF2Cfg<P, MASK, NA>;
    return BF2Geom<P, C::NLG, bf_nroles(MASK), C::NCW, C::NH, C::NSTW>::RMAX;
}
template <int P>
static int launch_bf2_p(hipStream_t st, const BFArgs &A, unsigned nblocks, int ny, int mask, int na, int ncu)
{
    if (ny == 1 && mask == BF_MASK_MASS && na == 1) return launch_bf2_c<P, 1, BF_MASK_MASS, 1>(st, A, nblocks, ncu);
    if (ny == 4 && mask == BF_MASK_STIFF3 && na == 1) return launch_bf2_c<P, 4, BF_MASK_STIFF3, 1>(st, A, nblocks, ncu);
    if (ny == 4 && mask == BF_MASK_STIFF3 && na == 2) return launch_bf2_c<P, 4, BF_MASK_STIFF3, 2>(st, A, nblocks, ncu);
    if (ny == 4 && mask == BF_MASK_STIFF2 && na == 1) return launch_bf2_c<P, 4, BF_MASK_STIFF2, 1>(st, A, nblocks, ncu);
    set_error("fused stage: no kernel for this set of types");
    return IGX_ERR_UNSUPPORTED;
}
template <int P> static int bf2_rows_p(int mask, int na)
{
    if (mask == BF_MASK_MASS) return bf2_rmax<P, BF_MASK_MASS, 1>();
    if (mask == BF_MASK_STIFF2) return bf2_rmax<P, BF_MASK_STIFF2, 1>();
    return na == 1 ? bf2_rmax<P, BF_MASK_STIFF3, 1>() : bf2_rmax<P, BF_MASK_STIFF3, 2>();
}
int fused2_rows_per_tile(int P, int mask, int na)
{
    switch (P) {
    case 2: return bf2_rows_p<2>(mask, na);
    case 3: return bf2_rows_p<3>(mask, na);
    case 4: return bf2_rows_p<4>(mask, na);
    case 5: return bf2_rows_p<5>(mask, na);
    case 6: return bf2_rows_p<6>(mask, na);
    }
    return 1;
}
static void bf_signature(const BFInputs &in, int &ny, int &mask, int &na)
{
    int ymax = 0;
    mask = 0; na = 1;
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1)
            if (in.slot_n[y][t1] > 0) { mask |= 1 << (4 * y + t1); ymax = std::max(ymax, y); na = std::max(na, in.slot_n[y][t1]); }
    ny = ymax == 0 ? 1 : 4;
}

// the fused kernel is compiled for the sets of types of the BASELINE forms: mass, stiffness (2D, 3D) and the 3D
// convection-diffusion form; other forms run the unfused kernels
int fused_supported(const BFInputs &in)
{
    int ny, mask, na;
    bf_signature(in, ny, mask, na);
    return (ny == 1 && mask == 0x0001 && na == 1) || (ny == 4 && mask == 0x135F && na <= 2) || (ny == 4 && mask == 0x1248 && na == 1);
}

// slots[y][t1]: input arrays (device pointers) of the sweep with their strides; see BFArgs
int launch_bf(hipStream_t st, const igx_patch *pt, const BFInputs &in, double *d_data)
{
    const Axis &AM = *in.mid, &AL = *in.last;
    BFArgs A{};
    int ny, mask, na;
    bf_signature(in, ny, mask, na);
    if (!fused_supported(in)) { set_error("fused stage: no kernel for this set of types"); return IGX_ERR_UNSUPPORTED; }
    if (!fused_offsets_fit(pt->dim == 3 ? 2 * pt->ax[0].p + 1 : 1, AM.S, AL.S, AM.G, AL.G)) {
        set_error("fused stage: patch beyond the 32-bit offsets of a row block / K1 slice");
        return IGX_ERR_UNSUPPORTED;
    }
    for (int y = 0; y < 4; ++y)
        for (int t1 = 0; t1 < 4; ++t1) {
            const int n = in.slot_n[y][t1];
            for (int i = 0; i < 2; ++i) {
                if (i < n) { A.sp[y][t1][i] = in.slot_ptr[y][t1][i]; A.ss[y][t1][i] = in.slice_stride; A.rs[y][t1][i] = AL.G; }
                else { A.sp[y][t1][i] = in.zeros; A.ss[y][t1][i] = 0; A.rs[y][t1][i] = 0; }
            }
        }
    A.gmid_lo = in.gmid_lo; A.G2 = AL.G;
    A.V1 = AM.d_V; A.V2 = AL.d_V;
    A.n1 = AM.n; A.N1 = AM.N; A.n2 = AL.n; A.N2 = AL.N;
    A.rp1 = AM.dev.rp; A.rp2 = AL.dev.rp;
    A.pl0 = in.pl0; A.rp0 = in.rp0; A.jlo0 = in.jlo0; A.jhi0 = in.jhi0;
    A.S1 = AM.S; A.S2 = AL.S; A.nnz_off = pt->nnz_off;
    A.data = d_data; A.sym = in.sym;
    const int P = AL.P;
    const int rmax = fused2_rows_per_tile(P, mask, na);
    A.ntiles = (AL.N + rmax - 1) / rmax;
    A.R2 = (AL.N + A.ntiles - 1) / A.ntiles;
    A.mid_lo = in.mid_lo; A.mid_hi = in.mid_hi; A.span_hi = in.span_hi;
    // chunks of the mid axis: enough blocks to fill the chip (each chunk re-sweeps p warm-up spans)
    const int mid_rows = in.mid_hi - in.mid_lo;
    long long blocks = (long long)in.npairs * A.ntiles;
    int nmch = 1;
    if (blocks < 1024) nmch = (int)std::min<long long>((1024 + blocks - 1) / blocks, std::max(1, mid_rows / (2 * P)));
#ifdef IGX_ABLATE
    if (const char *e = getenv("IGX_BF_MCHUNKS")) nmch = std::max(1, std::min(mid_rows, atoi(e)));
#endif
    A.mrows = (mid_rows + nmch - 1) / nmch;
    A.nmchunks = (mid_rows + A.mrows - 1) / A.mrows;
    A.npairs = in.npairs;
    blocks = (long long)in.npairs * A.ntiles * A.nmchunks;
    if (blocks > 0x7fffffffLL) { set_error("fused stage: too many blocks"); return IGX_ERR_UNSUPPORTED; }
    if (blocks == 0) return IGX_OK;
    const unsigned nb = (unsigned)blocks;
    switch (P) {
    case 2: return launch_bf2_p<2>(st, A, nb, ny, mask, na, pt->ctx->ncu);
    case 3: return launch_bf2_p<3>(st, A, nb, ny, mask, na, pt->ctx->ncu);
    case 4: return launch_bf2_p<4>(st, A, nb, ny, mask, na, pt->ctx->ncu);
    case 5: return launch_bf2_p<5>(st, A, nb, ny, mask, na, pt->ctx->ncu);
    case 6: return launch_bf2_p<6>(st, A, nb, ny, mask, na, pt->ctx->ncu);
    default: set_error("fused stage: degree %d unsupported", P - 1); return IGX_ERR_UNSUPPORTED;
    }
}

template <int WW>
static int launch_mirror_k(hipStream_t st, MirrorArgs &M, int N2)
{
    using Gm = MirrorGeom<WW>;
    M.nchunks = (N2 + Gm::RCM - 1) / Gm::RCM;
    M.RC = (N2 + M.nchunks - 1) / M.nchunks;
    // a launch wants >= ~1536 blocks: split the rows i1 of a (pair, chunk)
    const int rows = M.i1_hi - M.i1_lo;
    long long base = (long long)M.ntp * M.nchunks;
    M.ni1 = (int)std::max<long long>(1, std::min<long long>((1536 + base - 1) / base, std::max(1, rows / 8)));
    M.i1_rows = (rows + M.ni1 - 1) / M.ni1;
    M.ni1 = (rows + M.i1_rows - 1) / M.i1_rows;
    const long long blocks = base * M.ni1;
    if (blocks > 0x7fffffffLL) { set_error("mirror pass: too many blocks"); return IGX_ERR_UNSUPPORTED; }
    if (blocks == 0) return IGX_OK;
    const size_t lds = (size_t)WW * Gm::NJ2 * WW * sizeof(double);
    IGX_HIP(hipFuncSetAttribute((const void *)k_mirror<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_mirror<WW><<<dim3((unsigned)blocks), dim3(256), lds, st>>>(M);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

template <int WW, int VAR>
static int launch_mirror2_k(hipStream_t st, MirrorArgs &M, int N2)
{
    using Gm = Mirror2Geom<WW, VAR>;
    M.nchunks = (N2 + Gm::RCM - 1) / Gm::RCM;
    M.RC = (N2 + M.nchunks - 1) / M.nchunks;
    const int rows = M.i1_hi - M.i1_lo;
    long long base = (long long)M.ntp * M.nchunks;
    M.ni1 = (int)std::max<long long>(1, std::min<long long>((2048 + base - 1) / base, std::max(1, rows / 8)));
    M.i1_rows = (rows + M.ni1 - 1) / M.ni1;
    M.ni1 = (rows + M.i1_rows - 1) / M.i1_rows;
    const long long blocks = base * M.ni1;
    if (blocks > 0x7fffffffLL) { set_error("mirror pass: too many blocks"); return IGX_ERR_UNSUPPORTED; }
    if (blocks == 0) return IGX_OK;
    const size_t lds = (size_t)WW * Gm::NJ2 * WW * sizeof(double);
    IGX_HIP(hipFuncSetAttribute((const void *)k_mirror2<WW, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_mirror2<WW, VAR><<<dim3((unsigned)blocks), dim3(256), lds, st>>>(M);
    IGX_HIP(hipGetLastError());
    return IGX_OK;
}

int launch_mirror(hipStream_t st, const igx_patch *pt, const MirrorInputs &in, double *d_data)
{
    if (in.ntp == 0) return IGX_OK;
    const Axis &AM = *in.mid, &AL = *in.last;
    if (!AM.simple || !AL.simple || AM.P != AL.P) { set_error("mirror pass: axes must have single knots and equal degree"); return IGX_ERR_UNSUPPORTED; }
    MirrorArgs M{};
    M.data = d_data; M.nnz_off = pt->nnz_off; M.S1 = AM.S; M.S2 = AL.S;
    M.rp0 = in.rp0; M.jlo0 = in.jlo0; M.jhi0 = in.jhi0;
    M.rp1 = AM.dev.rp; M.rp2 = AL.dev.rp;
    M.N1 = AM.N; M.N2 = AL.N; M.p = AL.p; M.i1_lo = in.i1_lo; M.i1_hi = in.i1_hi;
    M.tpairs = in.tpairs; M.ntp = in.ntp;
    M.chunk_fastest = 0;
#ifdef IGX_ABLATE
    if (const char *e = getenv("IGX_MIRROR_ORDER")) M.chunk_fastest = atoi(e);
#endif
    // 32-bit offsets inside a row block (k_mirror2): c0 S1 S2 values of 8 bytes below 2^31
    const long long c0max = 2 * pt->ax[0].p + 1;
#ifdef MIRROR_OLD
    if (false) {
#else
    if (pt->dim == 3 && fused_offsets_fit(c0max, AM.S, AL.S, 0, 0)) {
#endif
        switch (2 * AL.p + 1) {
        case 5: return launch_mirror2_k<5, 0>(st, M, AL.N);
        case 7: return launch_mirror2_k<7, 0>(st, M, AL.N);
        case 9: return launch_mirror2_k<9, MIRROR_VAR>(st, M, AL.N);
        }
    }
    switch (2 * AL.p + 1) {
    case 3: return launch_mirror_k<3>(st, M, AL.N);
    case 5: return launch_mirror_k<5>(st, M, AL.N);
    case 7: return launch_mirror_k<7>(st, M, AL.N);
    case 9: return launch_mirror_k<9>(st, M, AL.N);
    case 11: return launch_mirror_k<11>(st, M, AL.N);
    default: set_error("mirror pass: degree %d unsupported", AL.p); return IGX_ERR_UNSUPPORTED;
    }
}

} // namespace igx

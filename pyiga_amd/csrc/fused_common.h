// Pieces shared by the fused sweep + final kernels (fused.hip: k_bf2 + mirror pass; fused3.hip: k_bf3, which writes both
// triangles itself): sweeper roles, the kernel arguments, the element matrices of the contractors, buffer-descriptor helpers.
#pragma once
#include "igx_internal.h"
#include <algorithm>

namespace igx {

typedef const double __attribute__((address_space(4))) *cdp;
typedef const int __attribute__((address_space(4))) *cip;


// Sweeper roles: one per last-axis type y that occurs.  The arrays entering the sweep are grouped by (y, mid-axis type
// t1 = tu + 2 tv); MASK has bit 4 y + t1 set for the groups that exist.  Per Gauss point a role accumulates
//   shape 0 (one tu = f):   acc[a][b] += V[b][f] * (V[a][0] * K[f] + V[a][1] * K[f + 2])                 30..35 FMAs
//   shape 1 (one tv = f):   acc[a][b] += V[a][f] * (V[b][0] * K[2 f] + V[b][1] * K[2 f + 1])             30..35 FMAs
//   shape 2 (all four t1):  acc[a][b] += sum_tu V[b][tu] * (V[a][0] * K[tu] + V[a][1] * K[tu + 2])       70 FMAs
// (a: test function = row, b: trial function; PI[t1][a][b] = V[b][tu] V[a][tv]) instead of 25 FMAs per array with 25
// coefficients each.  has[t1] says which K exist.
struct BFRole { int y, shape, f, has[4]; };
constexpr int bf_roles_of_y(int m) { return m == 0 ? 0 : 1; }
constexpr int bf_nroles(int MASK)
{
    return bf_roles_of_y(MASK & 15) + bf_roles_of_y((MASK >> 4) & 15) + bf_roles_of_y((MASK >> 8) & 15) + bf_roles_of_y((MASK >> 12) & 15);
}
constexpr BFRole bf_role_y(int y, int m)
{
    if ((m & 10) == 0) return BFRole{y, 0, 0, {m & 1, 0, (m >> 2) & 1, 0}};                // only tu = 0
    if ((m & 5) == 0) return BFRole{y, 0, 1, {0, (m >> 1) & 1, 0, (m >> 3) & 1}};          // only tu = 1
    if ((m & 12) == 0) return BFRole{y, 1, 0, {m & 1, (m >> 1) & 1, 0, 0}};                // only tv = 0
    if ((m & 3) == 0) return BFRole{y, 1, 1, {0, 0, (m >> 2) & 1, (m >> 3) & 1}};          // only tv = 1
    return BFRole{y, 2, 0, {m & 1, (m >> 1) & 1, (m >> 2) & 1, (m >> 3) & 1}};
}
constexpr BFRole bf_role(int MASK, int r)
{
    for (int y = 0; y < 4; ++y) {
        const int m = (MASK >> (4 * y)) & 15, n = bf_roles_of_y(m);
        if (r < n) return bf_role_y(y, m);
        r -= n;
    }
    return BFRole{0, 0, 0, {0, 0, 0, 0}};
}
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
    // tail split: the blocks of the last, partly filled round (block ids >= main_blocks) walk a (1 / tail_k)-th of the mid
    // axis each, so that they fill the chip and the launch ends after a fraction of a round (0: every block alike)
    unsigned main_blocks;
    int tail_k, tail_mrows;
    int mid_lo, mid_hi;           // rows of the mid axis to produce
    int span_hi;                  // spans of the mid axis below this one are resident (2D row slabs; else n1)
    int npairs;
    const int *fa1, *mslo1, *jlo1, *jhi1;   // mid axis (k_bf3): first active dof of a span, first span of a dof's support, column range of a row
    int own_lo, own_hi;           // owned rows of the outer axis (k_bf3: a block stores the direct rows of an owned i0, the transposed ones of an owned j0)
};

// Diagnostic build (-DIGX_BF_STAMP, never the shipped library): every wave adds up the shader cycles it spends waiting
// at the two barriers of a step; wave 0 lane 0 of each role group of block 0.. writes {wait, total} per wave at the end.
#ifdef IGX_BF_STAMP
static __device__ unsigned long long g_bf_stamp[64 * 1024];
#define BF_STAMP_DECL unsigned long long st_wait = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_seg[3] = {0, 0, 0}, st_a = 0;
#define BF_SEG_BEGIN() do { __builtin_amdgcn_sched_barrier(0); st_a = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BF_SEG_END(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory"); st_seg[i] += __builtin_amdgcn_s_memtime() - st_a; __builtin_amdgcn_sched_barrier(0); } while (0)
#define BF_SEG_DUMP(w) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) for (int i_ = 0; i_ < 3; ++i_) g_bf_stamp[32768 + (blockIdx.x * 4 + (w)) * 3 + i_] = st_seg[i_]; } while (0)
#define BF_STAMP_END(w) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); \
        g_bf_stamp[(blockIdx.x * 16 + (w)) * 2] = st_wait; g_bf_stamp[(blockIdx.x * 16 + (w)) * 2 + 1] = t1_ - st_t0; } } while (0)
#define bar_lds() do { const unsigned long long a_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        st_wait += __builtin_amdgcn_s_memtime() - a_; } while (0)
#else
#define BF_STAMP_DECL
#define BF_SEG_BEGIN()
#define BF_SEG_END(i)
#define BF_SEG_DUMP(w)
#define BF_STAMP_END(w)
__device__ __forceinline__ void bar_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// Element matrices of one K2 line on one span of the last axis (the contractors' first half): with K[y][l] the line's values
// of type y = tu + 2 tv at the span's Gauss points and V[l][.][.] the basis values there,
//     loc[a][b] = sum_l sum_tu V[l][b][tu] * (sum_tv V[l][a][tv] * K[tu + 2 tv][l])        (a: test function, b: trial function)
// -- every K and V value is read from LDS once per (line, span), not once per row.  The values of point l+1 are requested
// before the arithmetic of point l.  One role per type: role index = position of the type among those that occur.
template <int P, int NY, int MASK, int A0 = 0, int A1 = P, int Q = P>           // Q: Gauss points per span (nqp = max degree + 1 of the patch)
__device__ __forceinline__ void bf_element(double (&loc)[A1 - A0][P], const double *kl, const double *vl, const int TL)
{
    constexpr int NA_ = A1 - A0;             // rows A0 .. A1-1 of the element matrix (test functions)
    // LDS row of type y inside the line image (roles are ordered by type)
    constexpr int ry0 = 0;
    constexpr int ry1 = bf_roles_of_y(MASK & 15);
    constexpr int ry2 = ry1 + bf_roles_of_y((MASK >> 4) & 15);
    constexpr int ry3 = ry2 + bf_roles_of_y((MASK >> 8) & 15);
    constexpr bool h0 = (MASK & 15) != 0, h1 = ((MASK >> 4) & 15) != 0, h2 = ((MASK >> 8) & 15) != 0, h3 = ((MASK >> 12) & 15) != 0;
    double K[4], V[P][2];
    auto load_K = [&](const int l) {
        K[0] = h0 ? kl[ry0 * TL + l] : 0.0;
        if (NY == 4) { K[1] = h1 ? kl[ry1 * TL + l] : 0.0; K[2] = h2 ? kl[ry2 * TL + l] : 0.0; K[3] = h3 ? kl[ry3 * TL + l] : 0.0; }
    };
    auto load_V = [&](const int l, const int b) { V[b][0] = vl[(l * P + b) * 2]; V[b][1] = vl[(l * P + b) * 2 + 1]; };
    load_K(0);
#pragma unroll
    for (int b = 0; b < P; ++b) load_V(0, b);
#pragma unroll
    for (int l = 0; l < Q; ++l) {
        // every value is replaced by the one of the next point right after its last use (one register set)
        double c0[NA_], c1[NA_];
#pragma unroll
        for (int a = 0; a < NA_; ++a) {
            if (NY == 1) { c0[a] = V[A0 + a][0] * K[0]; c1[a] = 0.0; }
            else {
                c0[a] = fma(V[A0 + a][1], K[2], V[A0 + a][0] * K[0]);      // types 0, 2
                c1[a] = fma(V[A0 + a][1], K[3], V[A0 + a][0] * K[1]);      // types 1, 3
            }
        }
        if (l + 1 < Q) {
#pragma unroll
            for (int a = 0; a < NA_; ++a) asm volatile("" : "+v"(c0[a]), "+v"(c1[a]));     // K is dead from here
            load_K(l + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < P; ++b) {
#pragma unroll
            for (int a = 0; a < NA_; ++a) {
                if (l == 0) {                                   // the first point assigns (no zero fill)
                    if (NY == 1) loc[a][b] = V[b][0] * c0[a];
                    else loc[a][b] = fma(V[b][0], c0[a], V[b][1] * c1[a]);
                } else if (NY == 1) loc[a][b] = fma(V[b][0], c0[a], loc[a][b]);
                else loc[a][b] = fma(V[b][0], c0[a], fma(V[b][1], c1[a], loc[a][b]));
            }
            if (l + 1 < Q) {
#pragma unroll
                for (int a = 0; a < NA_; ++a) asm volatile("" : "+v"(loc[a][b]));          // V[b] is dead from here
                load_V(l + 1, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the reads of point l+2 may not be hoisted above this point (register blow-up)
#pragma unroll
        for (int a = 0; a < NA_; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) asm volatile("" : "+v"(loc[a][b]));
        asm volatile("" ::: "memory");
    }
}

__device__ __forceinline__ double bf2_from_lane(const int src4, const double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src4, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

typedef int bf2_v2i __attribute__((ext_vector_type(2)));
#ifndef BF2_PRIO_S
#define BF2_PRIO_S 0                                     // s_setprio of the sweepers / contractors (experiments)
#endif
#ifndef BF2_PRIO_C
#define BF2_PRIO_C 3
#endif
#ifndef BF2_NLG
#define BF2_NLG 3
#endif
#ifndef BF2_NCW
#define BF2_NCW 4
#endif
#ifndef BF2_NH
#define BF2_NH 2                                         // 2: the passes beyond one per contractor wave are cut into halves (1: whole passes)
#endif
constexpr int BF2_NUMREC = 0x7ffffff0;                     // bytes a descriptor covers; per-lane offsets at or above it are out of range
constexpr int BF2_OOB = 0x7ffffff8;                        // per-lane offset of a lane that must not store (dropped by the range check)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf2_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, BF2_NUMREC, 0x00020000);
}
__device__ __forceinline__ double bf2_buffer_load(__amdgpu_buffer_rsrc_t r, const int voff, const int soff)
{
    const bf2_v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double(v.y, v.x);
}
template <int AUX = 0>
__device__ __forceinline__ void bf2_buffer_store(__amdgpu_buffer_rsrc_t r, const int voff, const int soff, const double x)
{
    bf2_v2i v;
    v.x = __double2loint(x); v.y = __double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX);
}

struct BF2Blk {
    int i0, j0, diag0, c0i, cj0, rlo, rhi, row_lo, nrows;
    long long S12;
};

// Chunks of the mid axis.  A block walks the rows of its chunk in sequence (plus p warm-up rows and a fixed set-up), the chip
// holds `slots` blocks at a time, and a launch takes as long as its rounds: ceil(blocks / slots) x the walk of a block.  A
// launch is modelled two ways and the cheaper one taken: every block in m chunks, or whole blocks for some full rounds and only
// the rest of the blocks in chunks (below).  The entries do not depend on the split.  (The mirror pass is bandwidth-bound -- its last, partly filled round is short -- and gains
// nothing from the same model: measured.)
inline void bf2_choose_chunks(BFArgs &A, long long slots, int P)
{
    const int mid_rows = A.mid_hi - A.mid_lo;
    const long long per_chunk = (long long)A.npairs * A.ntiles;
    const int mmax = std::max(1, std::min(16, mid_rows / (2 * P)));
    constexpr int SETUP_ROWS = 4;                        // fixed cost of a block in rows of its walk
    long long best = -1;
    for (int m = 1; m <= mmax; ++m) {
        const int rows = (mid_rows + m - 1) / m, chunks = (mid_rows + rows - 1) / rows;
        const long long rounds = (per_chunk * chunks + slots - 1) / slots;
        const long long cost = rounds * (rows + (chunks > 1 ? P - 1 : 0) + SETUP_ROWS);
        if (best < 0 || cost < best) { best = cost; A.mrows = rows; A.nmchunks = chunks; }
    }
    // Whole blocks for `a` full rounds, the REST of the blocks cut into k chunks of the mid axis each (they follow the whole
    // blocks in launch order): C4, 2600 blocks on 256 CUs = 10 rounds of whole blocks and 40 blocks more -- those 40 become 240
    // blocks of 22 rows and the launch ends a fifth of a round after the tenth; an eighth of C4 (360 blocks): one round of whole
    // blocks + 104 blocks in halves (136 + 74 row steps) instead of every block in halves (3 rounds of 74).
    A.tail_k = 0; A.main_blocks = 0; A.tail_mrows = 0;
#ifndef BF2_NO_TAIL_SPLIT
    if (per_chunk > slots) {
        const long long full = per_chunk / slots;
        for (long long a = full; a >= 1 && a + 2 > full; --a) {
            const long long rest = per_chunk - a * slots;
            if (rest == 0) continue;
            for (int k = 2; k <= mmax; ++k) {
                const int rows = (mid_rows + k - 1) / k, chunks = (mid_rows + rows - 1) / rows;
                const long long rounds = (rest * chunks + slots - 1) / slots;
                const long long cost = a * (mid_rows + SETUP_ROWS) + rounds * (rows + P - 1 + SETUP_ROWS);
                if (cost < best) {
                    best = cost;
                    A.mrows = mid_rows; A.nmchunks = 1;
                    A.main_blocks = (unsigned)(a * slots); A.tail_mrows = rows; A.tail_k = chunks;
                }
            }
        }
    }
#endif
}


} // namespace igx

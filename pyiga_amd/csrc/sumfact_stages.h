// Stage kernels of the global sum factorisation (k_stageA, k_stageB, k_final) and their launchers: shared by sumfact.hip
// (degrees 1..5) and sumfact_hi.hip (degrees 6, 7 -- a translation unit of its own: the instantiations of these templates
// are what the build spends its time on, and the two halves compile side by side).  The algorithm is described at the
// top of sumfact.hip.
#pragma once
#include "igx_internal.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace igx {

struct Term { int f; int t[3]; };               // field index, type per axis (t = tu + 2*tv)

static int sym_index(int d, int r, int c)       // row-major upper triangle (pyiga/vform.py:28-34)
{
    if (r > c) std::swap(r, c);
    int idx = 0;
    for (int rr = 0; rr < r; ++rr) idx += d - rr;
    return idx + (c - r);
}

static std::vector<Term> form_terms(int dim, int kind, const PatchDev *pd = nullptr)
{
    std::vector<Term> T;
    if (kind == IGX_FORM) {
        // one term per stored field; jet index c >= 1 differentiates grid axis dim - c (x is the LAST axis)
        for (int k = 0; k < pd->form_n; ++k) {
            if (pd->form_par) {                  // parametric jet form: masks over the grid axes (igx_patch_set_pform)
                const int mv = pd->form_ab[k] >> 3, mu = pd->form_ab[k] & 7;
                Term t{};
                t.f = k;
                for (int ax = 0; ax < 3; ++ax) t.t[ax] = ax < dim ? ((mu >> ax) & 1) + 2 * ((mv >> ax) & 1) : 0;
                T.push_back(t);
                continue;
            }
            const int a = pd->form_ab[k] >> 2, b = pd->form_ab[k] & 3;     // a: test function v, b: trial function u
            Term t{};
            t.f = k;
            for (int ax = 0; ax < 3; ++ax) t.t[ax] = ax < dim ? ((b >= 1 && ax == dim - b) ? 1 : 0) + 2 * ((a >= 1 && ax == dim - a) ? 1 : 0) : 0;
            T.push_back(t);
        }
        return T;
    }
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
    if (kind == IGX_CONVDIFF)              // + (beta . du) v: fields 6.. = beta in (x,y,z) order, v undifferentiated
        for (int a = 0; a < dim; ++a) {
            Term t{};
            t.f = dim * (dim + 1) / 2 + (dim - 1 - a);
            for (int k = 0; k < 3; ++k) t.t[k] = (k < dim && k == a) ? 1 : 0;
            T.push_back(t);
        }
    return T;
}

// ---------------------------------------------------------------------------------------------
// Tables that every lane reads at the same index are accessed through the constant address space:
// hipcc then emits scalar loads (s_load_dwordx16 ...) and the coefficients feed v_fma_f64 straight
// from SGPRs.  (Through a plain pointer it falls back to per-lane global_load + vmcnt(0) waits.)
typedef const double __attribute__((address_space(4))) *cdp;
typedef const int __attribute__((address_space(4))) *cip;

// ---------------------------------------------------------------------------------------------
// The two sweep kernels share one structure.  A thread owns one point of the axes that are NOT
// contracted and walks along the contracted axis span by span, holding the (p+1)x(p+1) dof pairs
// that are active on the current span in registers.  The coefficients PI[g][t][a][b] are the same
// for every lane; the slice of the current span (q x 4 x P x P doubles, a few KB) is staged in LDS,
// double-buffered (one barrier per span), and read back with broadcast ds_reads.
// (Scalar loads were tried first: 25 coefficients x terms x q per span exceed the ~100 SGPRs of a
// wave and hipcc spilled them through v_writelane/v_readlane -- 4x more VALU than FMAs.)
//
// Parallelism along the sweep: blockIdx.z selects a chunk of spans.  A chunk starts P-1 spans
// early with zero accumulators (warm-up, nothing is written) so that every pair it completes inside
// its own range has seen all of its spans.

#ifndef SA_XPRE_MINP
#define SA_XPRE_MINP 6         // second-source values preloaded with the field values from this many functions per axis on
#endif
constexpr int SWEEP_MAX_STAGE = 8;     // PI slice values staged per thread (q*4*P*P / blockDim)

struct SweepChunks { int nchunks, len; };

// ---------------------------------------------------------------------------------------------
// Stage A: sweep axis 0.  One thread per point of the remaining grid (g1[,g2]); blockIdx.y selects
// the field; the (at most two) types of that field share the field load.
struct StageAGroup {
    const double *field;        // [G0_loc][NPL]
    double *out0, *out1;        // K1 arrays [npairs0][NPL]
    int t0, t1, nt;
    // optional second source of an output (merged slots of the fused stage: two terms that differ only in their axis-0 type
    // and field feed the same later stages): out_k = sum_g0 PI0[t_k] field + PI0[xt[k]] xfield[k]
    const double *xfield[2];
    int xt[2];
};
struct StageAArgs {
    StageAGroup grp[16];
    const double *PI0;          // [G0][4][P][P]
    const int *step_ptr;        // [n0+1] first flush step of each span
    const int *steps;           // symmetric: [nsteps][8] K1 slot of pair (leaving dof + a, leaving dof), or -1;
                                // non-symmetric: [nsteps][16], [a] as before and [8+a] = slot of (leaving dof, leaving dof + a)
    int s_lo, s_hi, n0, N0, q, g0_lo;
    int chunk_len;
    long long NPL;
};

template <int P, int NT, int Q, bool SYM, bool HASX = false, bool PF = false>
__device__ __forceinline__ void stageA_body(const double *__restrict__ field, double *__restrict__ out0,
                                            double *__restrict__ out1, const int t0, const int t1,
                                            const StageAArgs &A, const long long pt, const bool live, double *pis,
                                            const double *xf0 = nullptr, const double *xf1 = nullptr, const int xt0 = 0, const int xt1 = 0)
{
    cip step_ptr = (cip)A.step_ptr, steps = (cip)A.steps;
    const int q = Q ? Q : A.q;
    constexpr int PP = (P * P + 1) & ~1;                  // padded row: 16-byte aligned => ds_read_b128
    const int SL = q * 4 * P * P;                         // coefficient slice of one span (global)
    const int SLP = q * 4 * PP;                           // ... and its padded LDS image
    const int own_lo = A.s_lo + blockIdx.z * A.chunk_len; // spans whose completed pairs this chunk writes
    const int own_hi = min(own_lo + A.chunk_len, A.s_hi);
    const int s_begin = max(A.s_lo, own_lo - (P - 1));
    double acc[NT][P][P];
#pragma unroll
    for (int ty = 0; ty < NT; ++ty)
#pragma unroll
        for (int a = 0; a < P; ++a)
#pragma unroll
            for (int b = 0; b < P; ++b) acc[ty][a][b] = 0.0;

    double pfv[Q && PF ? Q : 1];                          // field values of the next span (PF)
    double stg[SWEEP_MAX_STAGE];
    auto stage_load = [&](const int s) {
        const double *src = A.PI0 + (size_t)s * SL;
#pragma unroll
        for (int c = 0; c < SWEEP_MAX_STAGE; ++c) {
            const int idx = threadIdx.x + c * blockDim.x;
            if (idx < SL) stg[c] = src[idx];
        }
    };
    auto stage_store = [&](const int buf) {
#pragma unroll
        for (int c = 0; c < SWEEP_MAX_STAGE; ++c) {
            const int idx = threadIdx.x + c * blockDim.x;
            if (idx < SL) pis[buf * SLP + (idx / (P * P)) * PP + idx % (P * P)] = stg[c];
        }
    };
    stage_load(s_begin);
    stage_store(0);

    const double *fp = field + (long long)(s_begin * q - A.g0_lo) * A.NPL + pt;
    const double *xp[2] = {HASX && xf0 ? xf0 + (long long)(s_begin * q - A.g0_lo) * A.NPL + pt : nullptr,
                           HASX && xf1 ? xf1 + (long long)(s_begin * q - A.g0_lo) * A.NPL + pt : nullptr};
    for (int s = s_begin; s < own_hi; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < own_hi) stage_load(s + 1);            // in flight during this span
        const double *pi_s = pis + buf * SLP;
        auto accumulate = [&](const int l, const double bv) {
#pragma unroll
            for (int ty = 0; ty < NT; ++ty) {
                const double *pt_ = (const double *)__builtin_assume_aligned(pi_s + (l * 4 + (ty == 0 ? t0 : t1)) * PP, 16);
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b <= (SYM ? a : P - 1); ++b) acc[ty][a][b] = fma(pt_[a * P + b], bv, acc[ty][a][b]);
                // keep hipcc from hoisting the LDS reads of every batch to the top (register blow-up):
                // the next batch's reads may not cross this point, and the FMAs above must precede it
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b <= (SYM ? a : P - 1); ++b) asm volatile("" : "+v"(acc[ty][a][b]));
                asm volatile("" ::: "memory");
            }
        };
        // second source of output ty (uniform per group): acc[ty] += PI0[xt] * value of the other field
        auto accumulate_x = [&](const int ty, const int xt, const int l, const double bv) {
            const double *pt_ = (const double *)__builtin_assume_aligned(pi_s + (l * 4 + xt) * PP, 16);
#pragma unroll
            for (int a = 0; a < P; ++a)
#pragma unroll
                for (int b = 0; b <= (SYM ? a : P - 1); ++b) acc[ty][a][b] = fma(pt_[a * P + b], bv, acc[ty][a][b]);
#pragma unroll
            for (int a = 0; a < P; ++a)
#pragma unroll
                for (int b = 0; b <= (SYM ? a : P - 1); ++b) asm volatile("" : "+v"(acc[ty][a][b]));
            asm volatile("" ::: "memory");
        };
        if (Q && PF) {
            // short chunks (2D): the walk of a thread is a handful of spans and the field loads of a span, awaited where they
            // are issued, are a memory latency each -- the values of the NEXT span are requested before this one is swept
            if (s == s_begin) {
#pragma unroll
                for (int l = 0; l < Q; ++l) pfv[l] = fp[(long long)l * A.NPL];
            }
            double bv[Q ? Q : 1];
#pragma unroll
            for (int l = 0; l < Q; ++l) bv[l] = pfv[l];
            if (s + 1 < own_hi) {
#pragma unroll
                for (int l = 0; l < Q; ++l) pfv[l] = fp[(long long)(q + l) * A.NPL];
            }
            __syncthreads();                              // slice `buf` is complete
#pragma unroll
            for (int l = 0; l < Q; ++l) accumulate(l, bv[l]);
        } else if (Q) {
            double bv[Q ? Q : 1];
#pragma unroll
            for (int l = 0; l < Q; ++l) bv[l] = fp[(long long)l * A.NPL];      // Q loads in flight
            // the values of the second source as well (single-type groups): loaded inside the loop below each of them is
            // awaited on its own, Q memory latencies per span in a row
            double xv[Q ? Q : 1];
            const bool xpre = HASX && NT == 1 && P >= SA_XPRE_MINP && xp[0];
            if (xpre) {
#pragma unroll
                for (int l = 0; l < Q; ++l) xv[l] = xp[0][(long long)l * A.NPL];
            }
            __syncthreads();                              // slice `buf` is complete
#pragma unroll
            for (int l = 0; l < Q; ++l) accumulate(l, bv[l]);
            if (xpre) {
#pragma unroll
                for (int l = 0; l < Q; ++l) accumulate_x(0, xt0, l, xv[l]);
                xp[0] += (long long)q * A.NPL;
            }
        } else {
            __syncthreads();
            for (int l = 0; l < q; ++l) accumulate(l, fp[(long long)l * A.NPL]);
        }
        if (HASX && !(Q && NT == 1 && P >= SA_XPRE_MINP)) {
#pragma unroll
            for (int ty = 0; ty < NT; ++ty)
                if (xp[ty]) {
                    for (int l = 0; l < q; ++l) accumulate_x(ty, ty == 0 ? xt0 : xt1, l, xp[ty][(long long)l * A.NPL]);
                    xp[ty] += (long long)q * A.NPL;
                }
        }
        fp += (long long)q * A.NPL;
        if (s + 1 < own_hi) stage_store(buf ^ 1);

        // dofs that leave the active set after this span: their pairs are complete (one flush step
        // per leaving dof; the K1 slots come from a host-built table, one scalar load per step)
        const bool write = live && s >= own_lo;
        for (int st = step_ptr[s]; st < step_ptr[s + 1]; ++st) {
            cip rec = steps + (size_t)st * (SYM ? 8 : 16);
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const int r = rec[a];
                if (r >= 0 && write) {
                    out0[(long long)r * A.NPL + pt] = acc[0][a][0];
                    if (NT == 2) out1[(long long)r * A.NPL + pt] = acc[NT - 1][a][0];
                }
                if (!SYM && a > 0) {
                    const int ru = rec[8 + a];
                    if (ru >= 0 && write) {
                        out0[(long long)ru * A.NPL + pt] = acc[0][0][a];
                        if (NT == 2) out1[(long long)ru * A.NPL + pt] = acc[NT - 1][0][a];
                    }
                }
            }
#pragma unroll
            for (int ty = 0; ty < NT; ++ty) {
#pragma unroll
                for (int a = 0; a < P - 1; ++a)
#pragma unroll
                    for (int b = 0; b <= (SYM ? a : P - 2); ++b) acc[ty][a][b] = acc[ty][a + 1][b + 1];
#pragma unroll
                for (int b = 0; b < P; ++b) { acc[ty][P - 1][b] = 0.0; if (!SYM) acc[ty][b][P - 1] = 0.0; }
            }
        }
    }
}

// ONE: every group carries a single type (the host splits two-type groups): the non-symmetric sweep of a high degree keeps
// P x P accumulators per type, and with two types (218 registers at P = 6) only two waves fit a SIMD -- too few to keep
// a streaming kernel's loads in flight; the field of a split group is read twice instead
template <int P, int Q, bool SYM, bool ONE = false, bool PF = false>
__global__ void __launch_bounds__(256) k_stageA(const StageAArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double pis[];   // [2][q*4*PP]
    long long pt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pt < A.NPL;
    if (!live) pt = A.NPL - 1;
    const StageAGroup &G = A.grp[blockIdx.y];
    if (!SYM && (G.xfield[0] || G.xfield[1])) {          // (merged slots occur with the non-symmetric forms only)
        if (!ONE && G.nt == 2) stageA_body<P, 2, Q, SYM, true>(G.field, G.out0, G.out1, G.t0, G.t1, A, pt, live, pis, G.xfield[0], G.xfield[1], G.xt[0], G.xt[1]);
        else stageA_body<P, 1, Q, SYM, true>(G.field, G.out0, G.out0, G.t0, G.t0, A, pt, live, pis, G.xfield[0], nullptr, G.xt[0], 0);
        return;
    }
    if (!ONE && G.nt == 2) stageA_body<P, 2, Q, SYM, false, PF>(G.field, G.out0, G.out1, G.t0, G.t1, A, pt, live, pis);
    else stageA_body<P, 1, Q, SYM, false, PF>(G.field, G.out0, G.out0, G.t0, G.t0, A, pt, live, pis);
}

// ---------------------------------------------------------------------------------------------
// Stage B (3D): sweep axis 1.  Block = (chunk of g2, processed pair r0, output group y [x span chunk]).
struct StageBGroup {
    int nterm;
    int x[12];                  // K1 array index of each term
    int t1[12];                 // axis-1 type of each term
};
struct StageBArgs {
    StageBGroup grp[4];
    const double *PI1;          // [G1][4][P][P]
    const int *step_ptr;        // [n1+1]
    const int *steps;           // [nsteps][16]: [a] = pair index of (d+a, d), [8+a] = pair index of (d, d+a); -1 if none
    const int *pl0;             // [npairs0][2]
    int n1, N1, q, G1, G2, S1, npairs0;
    int ngroups, chunk_len;
    int symmetric;
};

template <int P, int NTERM, int Q>
__device__ __forceinline__ void stageB_body(const double *__restrict__ K1, double *__restrict__ K2,
                                            const StageBArgs &B, const StageBGroup &G, const int y, const int chunk,
                                            const int g2, const bool live, double *pis)
{
    cip step_ptr = (cip)B.step_ptr, steps = (cip)B.steps, pl0 = (cip)B.pl0;
    const int r0 = blockIdx.y;
    const bool diag0 = B.symmetric && pl0[2 * r0] == pl0[2 * r0 + 1];
    const long long plane = (long long)B.G1 * B.G2;
    const int q = Q ? Q : B.q;
    constexpr int PP = (P * P + 1) & ~1;
    const int SL = q * 4 * P * P;
    const int SLP = q * 4 * PP;
    const int own_lo = chunk * B.chunk_len;
    const int own_hi = min(own_lo + B.chunk_len, B.n1);
    const int s_begin = max(0, own_lo - (P - 1));
    const double *kp[NTERM];
    int t1[NTERM];
#pragma unroll
    for (int t = 0; t < NTERM; ++t) {
        kp[t] = K1 + ((long long)G.x[t] * B.npairs0 + r0) * plane + (long long)s_begin * q * B.G2 + g2;
        t1[t] = G.t1[t];
    }
    double *out = K2 + ((long long)y * B.npairs0 + r0) * B.S1 * B.G2 + g2;

    double acc[P][P];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) acc[a][b] = 0.0;

    double stg[SWEEP_MAX_STAGE];
    auto stage_load = [&](const int s) {
        const double *src = B.PI1 + (size_t)s * SL;
#pragma unroll
        for (int c = 0; c < SWEEP_MAX_STAGE; ++c) {
            const int idx = threadIdx.x + c * blockDim.x;
            if (idx < SL) stg[c] = src[idx];
        }
    };
    auto stage_store = [&](const int buf) {
#pragma unroll
        for (int c = 0; c < SWEEP_MAX_STAGE; ++c) {
            const int idx = threadIdx.x + c * blockDim.x;
            if (idx < SL) pis[buf * SLP + (idx / (P * P)) * PP + idx % (P * P)] = stg[c];
        }
    };
    stage_load(s_begin);
    stage_store(0);

    for (int s = s_begin; s < own_hi; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < own_hi) stage_load(s + 1);
        const double *pi_s = pis + buf * SLP;
        auto accumulate = [&](const int l, const double (&kv)[NTERM]) {
#pragma unroll
            for (int t = 0; t < NTERM; ++t) {
                const double *pt_ = (const double *)__builtin_assume_aligned(pi_s + (l * 4 + t1[t]) * PP, 16);
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b < P; ++b) acc[a][b] = fma(pt_[a * P + b], kv[t], acc[a][b]);
#pragma unroll
                for (int a = 0; a < P; ++a)
#pragma unroll
                    for (int b = 0; b < P; ++b) asm volatile("" : "+v"(acc[a][b]));
                asm volatile("" ::: "memory");
            }
        };
        if (Q) {
            double kv[Q ? Q : 1][NTERM];
#pragma unroll
            for (int l = 0; l < Q; ++l)
#pragma unroll
                for (int t = 0; t < NTERM; ++t) kv[l][t] = kp[t][(long long)l * B.G2];
            __syncthreads();
#pragma unroll
            for (int l = 0; l < Q; ++l) accumulate(l, kv[l]);
        } else {
            __syncthreads();
            for (int l = 0; l < q; ++l) {
                double kv[NTERM];
#pragma unroll
                for (int t = 0; t < NTERM; ++t) kv[t] = kp[t][(long long)l * B.G2];
                accumulate(l, kv);
            }
        }
#pragma unroll
        for (int t = 0; t < NTERM; ++t) kp[t] += (long long)q * B.G2;
        if (s + 1 < own_hi) stage_store(buf ^ 1);

        const bool write = live && s >= own_lo;
        for (int st = step_ptr[s]; st < step_ptr[s + 1]; ++st) {
            cip rec = steps + (size_t)st * 16;
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const int rl = rec[a];               // pair (i1 = d + a, j1 = d): lower or diagonal
                if (rl >= 0 && write) out[(long long)rl * B.G2] = acc[a][0];
                if (a > 0) {
                    const int ru = rec[8 + a];       // pair (i1 = d, j1 = d + a): strictly upper,
                    if (ru >= 0 && write && !diag0) out[(long long)ru * B.G2] = acc[0][a];   // not needed on a diagonal (i0,j0)
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

template <int P, int Q>
__global__ void __launch_bounds__(256) k_stageB(const double *__restrict__ K1, double *__restrict__ K2, const StageBArgs B)
{
    extern __shared__ __attribute__((aligned(16))) double pis[];   // [2][q*4*PP]
    int g2 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g2 < B.G2;
    if (!live) g2 = B.G2 - 1;
    const int y = blockIdx.z % B.ngroups, chunk = blockIdx.z / B.ngroups;
    const StageBGroup &G = B.grp[y];
    switch (G.nterm) {
    case 0: break;                                      // empty group: its K2 array was zero-filled by the host
    case 1: stageB_body<P, 1, Q>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    case 2: stageB_body<P, 2, Q>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    case 3: stageB_body<P, 3, Q>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    case 4: stageB_body<P, 4, Q>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    case 5: stageB_body<P, 5, 0>(K1, K2, B, G, y, chunk, g2, live, pis); break;     // run-time q: fewer live registers
    case 6: stageB_body<P, 6, 0>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    case 7: stageB_body<P, 7, 0>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    case 8: stageB_body<P, 8, 0>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    default: stageB_body<P, 9, 0>(K1, K2, B, G, y, chunk, g2, live, pis); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Final stage: contract the last (contiguous) grid axis and write CSR values + mirror.
//
// Block = (row tile of the last axis, row group).  A row group is all K lines that end up in the same
// CSR rows:  3D: (r0, i1) with its lines j1 = jlo1[i1] .. (K2[y][r0][r1][:]),   2D: r0 (one line,
// K1[x][r0][:]).  The basis table segment is staged in LDS once per block; the K lines are streamed
// through registers into LDS one after the other (the next line's loads are in flight while the
// current one is contracted); results leave through an LDS transpose so that every store
// instruction writes runs of up to 2p+1 consecutive doubles.
struct FinalArgs {
    const double *V;            // last axis [G][P][2]
    const int *fa, *mslo, *mshi, *jlo, *jhi, *rp;   // last axis tables
    int N, q, G;                // last axis dofs, q, Gauss count
    long long nlines;
    int dim;
    const int *pl0;             // [npairs0][2]
    const int *rp0, *jlo0, *jhi0;
    const int *rp1, *jlo1, *jhi1;   // axis 1 (3D)
    int N1, S1;                 // dofs / pairs of axis 1 (3D); 1 in 2D
    long long Smid, Slast;      // 3D: S1, S2 ; 2D: unused, S1
    int r0_lo, r0_hi;
    long long nnz_off;
    int nsp_max;                // LDS segment capacity in spans
    int SSTR, KSTR;             // padded per-span strides (doubles) of the V / K images in LDS (odd => conflict-free)
    int tile_rows, ntiles;      // rows per block tile (multiple of CR), tiles per line
    int tsp_max, trow_max;      // LDS capacities: spans / table rows of a tile
    int CR;                     // rows per wave task
    int NW, GPB;                // waves per block, row groups per block
    long long ngroups;
    int symmetric;
};

template <int P, int D>
__device__ __forceinline__ void acc_add(double (&acc)[2 * P - 1], const double *vs, double cu0, double cu1)
{
#pragma unroll
    for (int b = 0; b < P; ++b)
        if (D + b < 2 * P - 1) acc[D + b] = fma(vs[2 * b], cu0, fma(vs[2 * b + 1], cu1, acc[D + b]));
}

template <int P, int D>
struct AccSwitch {
    __device__ static __forceinline__ void run(int d, double (&acc)[2 * P - 1], const double *vs, double cu0, double cu1)
    {
        if (d == D) acc_add<P, D>(acc, vs, cu0, cu1);
        else AccSwitch<P, D + 1>::run(d, acc, vs, cu0, cu1);
    }
};
template <int P>
struct AccSwitch<P, 2 * P - 1> {
    __device__ static __forceinline__ void run(int, double (&)[2 * P - 1], const double *, double, double) {}
};

// contribution of the K-th span of the support of row i.  `d` is the output offset of the span's
// local trial function 0; for single interior knots d == K for every row, which makes the
// accumulator index static (the generic path is a compile-time switch over d).
template <int P, int NY, int Q, bool SIMPLE, int K>
__device__ __forceinline__ void final_span(const int q_rt, const int kcap, const int fa_s, const double *Ksp,
                                           const double *Vsp, int i, int jl, double (&acc)[2 * P - 1])
{
    const int a = i - fa_s;                       // local index of the test function
    const int d = fa_s - jl;
    const int q = Q ? Q : q_rt;
    auto point = [&](const int l, double &cu0, double &cu1) -> const double * {
        const double *vs = Vsp + l * P * 2;
        const double v0 = vs[2 * a], v1 = vs[2 * a + 1];
        if (NY == 1) { cu0 = v0 * Ksp[l]; cu1 = 0.0; }
        else {
            // K arrays are ordered by type t = tu + 2*tv of the last axis
            cu0 = fma(v1, Ksp[2 * kcap + l], v0 * Ksp[l]);
            cu1 = fma(v1, Ksp[3 * kcap + l], v0 * Ksp[kcap + l]);
        }
        return vs;
    };
    if (SIMPLE || d == K) {
#pragma unroll
        for (int l = 0; l < q; ++l) { double c0, c1; const double *vs = point(l, c0, c1); acc_add<P, K>(acc, vs, c0, c1); }
    } else {
        for (int l = 0; l < q; ++l) { double c0, c1; const double *vs = point(l, c0, c1); AccSwitch<P, 0>::run(d, acc, vs, c0, c1); }
    }
}

template <int P, int NY, int Q, bool SIMPLE, int K>
struct SpanLoop {
    __device__ static __forceinline__ void run(const int q, const int kcap, const int sstr, const int kstr, const int *fa_sp,
                                               const double *Ksp, const double *Vsp, int i, int nsp, int jl,
                                               double (&acc)[2 * P - 1])
    {
        if (K < nsp) final_span<P, NY, Q, SIMPLE, K>(q, kcap, fa_sp[K], Ksp + K * kstr, Vsp + K * sstr, i, jl, acc);
        SpanLoop<P, NY, Q, SIMPLE, K + 1>::run(q, kcap, sstr, kstr, fa_sp, Ksp, Vsp, i, nsp, jl, acc);
    }
};
template <int P, int NY, int Q, bool SIMPLE>
struct SpanLoop<P, NY, Q, SIMPLE, P> {
    __device__ static __forceinline__ void run(const int, const int, const int, const int, const int *, const double *,
                                               const double *, int, int, int, double (&)[2 * P - 1]) {}
};

// Block = NW independent waves sharing the basis table of the last axis (staged once in LDS).
// A wave task = (K line, chunk of CR <= 64 consecutive rows): the wave stages the K window of its
// chunk in a private LDS region (through registers; the next task's window is in flight while the
// current one is contracted), lane = matrix row, and the results leave through the same private
// region as an LDS transpose so that store instructions write runs of 2p+1 consecutive doubles.
// After the initial staging there is no block-wide barrier: LDS operations of one wave execute in
// order, so a wave's private region needs no synchronisation.
template <int P, int NY, int Q, int KPY, bool SIMPLE>
__global__ void __launch_bounds__(SIMPLE ? 768 : 384) k_final(const double *__restrict__ K, double *__restrict__ data, const FinalArgs F)
{
    constexpr int W = 2 * P - 1;
    extern __shared__ double lds[];
    const int kcap = F.nsp_max * F.KSTR;                // doubles per K array window
    const int kslot = max(NY * kcap, F.CR * W);         // doubles per wave (K windows, reused as out_s[CR][W])
    double *Vs = lds;                                   // [tile spans] x SSTR (q x P x (value, derivative), padded)
    double *Kbase = Vs + (size_t)F.tsp_max * F.SSTR;
    int *row_jl = (int *)(Kbase + (size_t)F.NW * kslot);  // per-row tables of the rows this tile touches
    int *row_c = row_jl + F.trow_max;
    int *row_rp = row_c + F.trow_max;
    int *row_slo = row_rp + F.trow_max;
    int *row_nsp = row_slo + F.trow_max;
    int *fa_s = row_nsp + F.trow_max;                   // [tile spans]

    cip pl0 = (cip)F.pl0, rp0 = (cip)F.rp0, jlo0 = (cip)F.jlo0, jhi0 = (cip)F.jhi0;
    cip rp1 = (cip)F.rp1, jlo1 = (cip)F.jlo1, jhi1 = (cip)F.jhi1;
    cip mslo = (cip)F.mslo, mshi = (cip)F.mshi, jlo = (cip)F.jlo, jhi = (cip)F.jhi;

    // ---- row tile of this block; stage its basis-table segment and row tables (once per block)
    const int tile = blockIdx.x % F.ntiles;
    const int tile_lo = tile * F.tile_rows, tile_hi = min(tile_lo + F.tile_rows, F.N);
    const int spb = mslo[tile_lo];                      // first span of the segment
    const int rb = jlo[tile_lo];                        // first row of the tables
    {
        const int per_span = F.q * P * 2;
        const int total = (mshi[tile_hi - 1] - spb) * per_span;
        const double *vsrc = F.V + (size_t)spb * per_span;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int sp = idx / per_span;
            Vs[sp * F.SSTR + (idx - sp * per_span)] = vsrc[idx];
        }
        const int nrows = jhi[tile_hi - 1] - rb;
        for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
            const int lo = F.jlo[rb + r], sl = F.mslo[rb + r];
            row_jl[r] = lo; row_c[r] = F.jhi[rb + r] - lo; row_rp[r] = F.rp[rb + r];
            row_slo[r] = sl; row_nsp[r] = F.mshi[rb + r] - sl;
        }
        for (int sp = threadIdx.x; sp < mshi[tile_hi - 1] - spb; sp += blockDim.x) fa_s[sp] = F.fa[spb + sp];
    }
    __syncthreads();
    const int nchunks = (tile_hi - tile_lo + F.CR - 1) / F.CR;   // chunks of this tile

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *Ks = Kbase + (size_t)wave * kslot;          // private to this wave

    const long long grp_lo = (long long)(blockIdx.x / F.ntiles) * F.GPB;
    const long long grp_hi = min(grp_lo + F.GPB, F.ngroups);
    int task_base = 0;                                  // tasks of the groups before `grp` in this block
    for (long long grp = grp_lo; grp < grp_hi; ++grp) {
        // ---- row group -> (i0, j0[, i1]) and its K lines
        int r0, i1 = 0, nl = 1, jl1 = 0;
        long long line0;
        if (F.dim == 3) { r0 = (int)(grp / F.N1); i1 = (int)(grp % F.N1); }
        else r0 = (int)grp;
        const int i0 = pl0[2 * r0], j0 = pl0[2 * r0 + 1];
        const bool diag0 = F.symmetric && (i0 == j0);
        if (F.dim == 3) {
            jl1 = jlo1[i1];
            nl = diag0 ? (i1 - jl1 + 1) : (jhi1[i1] - jl1);    // upper part of a diagonal block is mirrored, not computed
            line0 = (long long)r0 * F.S1 + rp1[i1];
        } else line0 = r0;
        const bool own_row = i0 >= F.r0_lo && i0 < F.r0_hi;
        const bool own_col = F.symmetric && j0 >= F.r0_lo && j0 < F.r0_hi;
        const int c0i = jhi0[i0] - jlo0[i0], c0j = jhi0[j0] - jlo0[j0];
        const int ntask = nl * nchunks;
        const int first = (wave - task_base % F.NW + F.NW) % F.NW;
        task_base += ntask;

        double kreg[NY][KPY];
        auto window = [&](const int chunk, int &row_lo, int &row_hi, int &sp_lo, int &seglen) {
            row_lo = tile_lo + chunk * F.CR;
            row_hi = min(row_lo + F.CR, tile_hi);
            sp_lo = mslo[row_lo];
            seglen = (mshi[row_hi - 1] - sp_lo) * F.q;
        };
        auto prefetch = [&](const int task) {
            const int ln = task / nchunks, chunk = task - ln * nchunks;
            int row_lo, row_hi, sp_lo, seglen;
            window(chunk, row_lo, row_hi, sp_lo, seglen);
#pragma unroll
            for (int y = 0; y < NY; ++y) {
                const double *src = K + ((long long)y * F.nlines + (line0 + ln)) * F.G + sp_lo * F.q;
#pragma unroll
                for (int c = 0; c < KPY; ++c) {
                    const int idx = lane + c * 64;
                    if (idx < seglen) kreg[y][c] = src[idx];
                }
            }
        };
        if (first < ntask) prefetch(first);

        for (int task = first; task < ntask; task += F.NW) {
            const int ln = task / nchunks, chunk = task - ln * nchunks;
            const int j1 = jl1 + ln;
            int row_lo, row_hi, sp_lo, seglen;
            window(chunk, row_lo, row_hi, sp_lo, seglen);
            const int ntr = row_hi - row_lo;
            // K window: registers -> private LDS (per-span padded stride KSTR)
#pragma unroll
            for (int y = 0; y < NY; ++y)
#pragma unroll
                for (int c = 0; c < KPY; ++c) {
                    const int idx = lane + c * 64;
                    if (idx < seglen) {
                        const int sp = idx / F.q;
                        Ks[y * kcap + sp * F.KSTR + (idx - sp * F.q)] = kreg[y][c];
                    }
                }
            if (task + F.NW < ntask) prefetch(task + F.NW);    // in flight during the contraction below

            // ---- this lane's row
            const int i = row_lo + lane;
            const bool active = lane < ntr;
            double acc[W];
#pragma unroll
            for (int o = 0; o < W; ++o) acc[o] = 0.0;
            if (active) {
                const int slo = row_slo[i - rb], nsp = row_nsp[i - rb], jl = row_jl[i - rb];
                int fa_sp[P];
#pragma unroll
                for (int k = 0; k < P; ++k) fa_sp[k] = (k < nsp) ? fa_s[slo - spb + k] : 0;
                SpanLoop<P, NY, Q, SIMPLE, 0>::run(F.q, kcap, F.SSTR, F.KSTR, fa_sp, Ks + (slo - sp_lo) * F.KSTR,
                                                   Vs + (slo - spb) * F.SSTR, i, nsp, jl, acc);
            }
            __builtin_amdgcn_wave_barrier();
            double *out_s = Ks;                             // [ntr][W]; LDS ops of a wave execute in order
            if (active) {
#pragma unroll
                for (int o = 0; o < W; ++o) out_s[lane * W + o] = acc[o];
            }
            __builtin_amdgcn_wave_barrier();

            // position coefficients: pos = A + B*rp[row] + C*c[row] + o   
            const bool diag_lead = diag0 && (F.dim == 2 || j1 == i1);
            long long A_d, B_d, C_d, A_m, B_m, C_m;
            if (F.dim == 3) {
                const int c1i = jhi1[i1] - jlo1[i1], c1j = jhi1[j1] - jlo1[j1];
                A_d = (long long)rp0[i0] * F.Smid * F.Slast + (long long)c0i * rp1[i1] * F.Slast - F.nnz_off;
                B_d = (long long)c0i * c1i;
                C_d = (long long)(j0 - jlo0[i0]) * c1i + (j1 - jlo1[i1]);
                A_m = (long long)rp0[j0] * F.Smid * F.Slast + (long long)c0j * rp1[j1] * F.Slast - F.nnz_off;
                B_m = (long long)c0j * c1j;
                C_m = (long long)(i0 - jlo0[j0]) * c1j + (i1 - jlo1[j1]);
            } else {
                A_d = (long long)rp0[i0] * F.Slast - F.nnz_off;  B_d = c0i;  C_d = j0 - jlo0[i0];
                A_m = (long long)rp0[j0] * F.Slast - F.nnz_off;  B_m = c0j;  C_m = i0 - jlo0[j0];
            }
            // direct entries: row (.., i), columns jl + o
            if (own_row) {
                for (int f = lane; f < ntr * W; f += 64) {
                    const int r = f / W, o = f - r * W;
                    const int ii = row_lo + r;
                    const int jli = row_jl[ii - rb], ci = row_c[ii - rb];
                    if (o < ci && !(diag_lead && jli + o > ii))
                        data[A_d + B_d * row_rp[ii - rb] + C_d * ci + o] = out_s[f];
                }
            }
            // mirrored entries: row (.., j), column (.., i) for every computed (i, j) of this chunk
            if (own_col) {
                const int jmin = jlo[row_lo], jmax = jhi[row_hi - 1];
                const int nj = jmax - jmin;
                for (int f = lane; f < nj * W; f += 64) {
                    const int rr = f / W, o = f - rr * W;
                    const int j = jmin + rr;
                    const int cj = row_c[j - rb];
                    const int ii = row_jl[j - rb] + o;
                    if (o < cj && ii >= row_lo && ii < row_hi && !(diag_lead && j >= ii))
                        data[A_m + B_m * row_rp[j - rb] + C_m * cj + o] = out_s[(ii - row_lo) * W + (j - row_jl[ii - rb])];
                }
            }
            __builtin_amdgcn_wave_barrier();            // out_s reads precede the next task's K writes
        }
    }
}


template <int P>
static void launch_stageA(hipStream_t st, const StageAArgs &A, bool qeq, bool sym, bool one, bool short_chunks, dim3 grid, dim3 block, size_t lds)
{
    if constexpr (P >= 7) {
        // degrees 6, 7: one instantiation per symmetry (run-time q, two types per group at most) -- the variants below are
        // tunings for the degrees the benchmarks run at, and every instantiation of these kernels costs build time
        if (!sym && one && qeq) k_stageA<P, P, false, true><<<grid, block, lds, st>>>(A);
        else if (sym) k_stageA<P, 0, true><<<grid, block, lds, st>>>(A);
        else k_stageA<P, 0, false><<<grid, block, lds, st>>>(A);
        return;
    }
    if (!sym && one && qeq) { k_stageA<P, P, false, true><<<grid, block, lds, st>>>(A); return; }
    if (sym) {
        if (qeq && short_chunks) k_stageA<P, P, true, false, true><<<grid, block, lds, st>>>(A);
        else if (qeq) k_stageA<P, P, true><<<grid, block, lds, st>>>(A);
        else k_stageA<P, 0, true><<<grid, block, lds, st>>>(A);
    } else if (qeq) k_stageA<P, P, false><<<grid, block, lds, st>>>(A);   // measured: compile-time q wins for every p (p=5: 16.8 -> 11.0 ms at C5)
    else k_stageA<P, 0, false><<<grid, block, lds, st>>>(A);
}

template <int P>
static void launch_stageB(hipStream_t st, const double *K1, double *K2, const StageBArgs &B, bool qeq, dim3 grid, dim3 block, size_t lds)
{
    if (qeq && P < 7) k_stageB<P, (P < 7 ? P : 0)><<<grid, block, lds, st>>>(K1, K2, B);
    else k_stageB<P, 0><<<grid, block, lds, st>>>(K1, K2, B);
}

template <int P, int NY, int Q, int KPY, bool SIMPLE>
static int launch_final_k(hipStream_t st, const double *K, double *data, const FinalArgs &F, dim3 grid, dim3 block, size_t lds)
{
    IGX_HIP(hipFuncSetAttribute((const void *)k_final<P, NY, Q, KPY, SIMPLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_final<P, NY, Q, KPY, SIMPLE><<<grid, block, lds, st>>>(K, data, F);
    return IGX_OK;
}

// Two instantiations per (P, NY): the fast one fixes q == P at compile time and assumes single
// interior knots on the last axis (output offset of span k is k); the generic one handles any q and
// any knot multiplicities.
template <int P>
static int launch_final(hipStream_t st, const double *K, double *data, const FinalArgs &F, int ny, bool fast, int kpy,
                        dim3 grid, dim3 block, size_t lds)
{
    if constexpr (P >= 7) {                          // (degrees 6, 7: the generic instantiation with the large K window only)
        return ny == 1 ? launch_final_k<P, 1, 0, 8, false>(st, K, data, F, grid, block, lds) : launch_final_k<P, 4, 0, 8, false>(st, K, data, F, grid, block, lds);
    }
    if (ny == 1) {
        if (fast) return kpy <= 4 ? launch_final_k<P, 1, P, 4, true>(st, K, data, F, grid, block, lds) : launch_final_k<P, 1, P, 8, true>(st, K, data, F, grid, block, lds);
        return kpy <= 4 ? launch_final_k<P, 1, 0, 4, false>(st, K, data, F, grid, block, lds) : launch_final_k<P, 1, 0, 8, false>(st, K, data, F, grid, block, lds);
    }
    if (fast) return kpy <= 4 ? launch_final_k<P, 4, P, 4, true>(st, K, data, F, grid, block, lds) : launch_final_k<P, 4, P, 8, true>(st, K, data, F, grid, block, lds);
    return kpy <= 4 ? launch_final_k<P, 4, 0, 4, false>(st, K, data, F, grid, block, lds) : launch_final_k<P, 4, 0, 8, false>(st, K, data, F, grid, block, lds);
}


// degrees 6, 7 (P = 7, 8): instantiated in sumfact_hi.hip
void stageA_hi(int P, hipStream_t st, const StageAArgs &A, bool qeq, bool sym, bool one, bool short_chunks, dim3 grid, dim3 block, size_t lds);
void stageB_hi(int P, hipStream_t st, const double *K1, double *K2, const StageBArgs &B, bool qeq, dim3 grid, dim3 block, size_t lds);
int final_hi(int P, hipStream_t st, const double *K, double *data, const FinalArgs &F, int ny, bool fast, int kpy, dim3 grid, dim3 block, size_t lds);

} // namespace igx

// Microbenchmark: the axis-0 sweep of k_geoA (p = 4: 15 lower pairs, 5 Gauss planes per span) in two forms, everything
// else of the kernel replaced by a stand-in, to price the FP64-MFMA sweep before building it.
// build: hipcc --offload-arch=gfx950 -O3 geoa_sweep.hip -o geoa_sweep
//   block = 8 waves on the same 64 points, wave w = one K1 array; LDS holds field values and basis values (static here);
//   VALU form:  per plane 5 products V_a * f and 15 multiply-adds, window shifted at every span end, 5 x 512-B stores;
//   MFMA form:  per span ONE v_mfma_f64_16x16x4 per 16-point tile for the first four planes (A = products V_b V_a of the
//               16 cyclic pair slots x 4 planes, B = field values of 4 planes x 16 points), the fifth plane with 16 vector
//               multiply-adds in the accumulator layout; completed slots are stored (quarter-wave runs of 128 B) and cleared
//               in place -- no window shift.
//   GEO:        stand-in for the geometry evaluation: per 8 planes every wave issues GEO dependent-chain FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int NPL = 20;                 // planes resident in LDS (ring)
constexpr int PST = 6 * 64 + 16;
constexpr int PST2 = 6 * 64 + 1;        // MODE 2 (16-byte lane stride): odd plane stride puts the two plane groups of a half wave on different banks        // doubles per plane (padded: the four plane groups of an MFMA operand on different banks)

__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, double x)
{
    v2i v; v.x = __double2loint(x); v.y = __double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, 0, 0);
}

template <int MODE, int GEO, int NOST>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_sweep(double *out, long long stride, long long xstride, int nspans, const double *init)
{
    __shared__ double fld[NPL * PST];          // (PST >= PST2)
    __shared__ __attribute__((aligned(16))) double rec[NPL][12];       // basis rows [value | derivative][6]
    __shared__ __attribute__((aligned(16))) double atab[4][5][16];     // products per type, plane of the span, slot
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NPL * PST; i += 512) fld[i] = init[i % 4096];
    for (int i = tid; i < NPL * 12; i += 512) (&rec[0][0])[i] = init[(i * 7) % 4096];
    for (int i = tid; i < 4 * 5 * 16; i += 512) (&atab[0][0][0])[i] = init[(i * 3) % 4096];
    __syncthreads();
    const int fi = w % 6, t = w & 3, tu = t & 1, tv = t >> 1;
    const long long pt = (long long)blockIdx.x * 64;
    double g0 = init[lane], g1 = init[lane + 64], g2 = init[lane + 128], g3 = init[lane + 192];
    const double gm = init[300];
    auto geo = [&]() {
#pragma unroll 8
        for (int i = 0; i < GEO / 4; ++i) { g0 = fma(g0, gm, g1); g1 = fma(g1, gm, g2); g2 = fma(g2, gm, g3); g3 = fma(g3, gm, g0); }
    };
    if (MODE == 0) {
        double acc[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = 0.0;
        double *o = out + (long long)w * xstride + pt + lane;
        int pl = 0, cnt = 0;
        auto row = [&](double (&v)[6], const int p_, const int d) {
            const d2 *r = (const d2 *)&rec[p_][6 * d];
#pragma unroll
            for (int k = 0; k < 2; ++k) { const d2 x = r[k]; v[2 * k] = x.x; v[2 * k + 1] = x.y; }
            v[4] = rec[p_][6 * d + 4];
        };
        double bv = fld[fi * 64 + lane], va[6], vb[6];
        row(va, 0, tv); row(vb, 0, tu);
        for (int s = 0; s < nspans; ++s) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pn = pl + 1 == NPL ? 0 : pl + 1;
                const double bvn = fld[pn * PST + fi * 64 + lane];
                double c[5];
#pragma unroll
                for (int a = 0; a < 5; ++a) c[a] = va[a] * bv;
#pragma unroll
                for (int a = 0; a < 5; ++a) asm volatile("" : "+v"(c[a]));
                asm volatile("" ::: "memory");
                row(va, pn, tv);
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[a][b] = fma(vb[b], c[a], acc[a][b]);
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) asm volatile("" : "+v"(acc[a][b]));
                asm volatile("" ::: "memory");
                row(vb, pn, tu);
                bv = bvn;
                pl = pn;
                if (++cnt == 8) { cnt = 0; geo(); __syncthreads(); }
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) if (!NOST) o[(long long)(5 * (s + a) + 4 - a) * stride] = acc[a][0];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[4][b] = 0.0;
        }
        if (acc[0][0] == 1.234e300) out[0] = acc[1][1] + acc[2][2] + acc[3][3] + acc[4][4] + acc[3][1];
    } else if (MODE == 3) {
        double acc[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[a][b] = 0.0;
        double *o = out + (long long)w * xstride + pt + lane;
        int pl = 0, cnt = 0;
        auto row = [&](double (&v)[6], const int p_, const int d) {
            const d2 *r = (const d2 *)&rec[p_][6 * d];
#pragma unroll
            for (int k = 0; k < 2; ++k) { const d2 x = r[k]; v[2 * k] = x.x; v[2 * k + 1] = x.y; }
            v[4] = rec[p_][6 * d + 4];
        };
        double bv = fld[fi * 64 + lane], va[6], vb[6];
        double park[5] = {0, 0, 0, 0, 0};
        long long poff[5] = {0, 0, 0, 0, 0};
        bool have = false;
        row(va, 0, tv); row(vb, 0, tu);
        for (int s = 0; s < nspans; ++s) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pn = pl + 1 == NPL ? 0 : pl + 1;
                const double bvn = fld[pn * PST + fi * 64 + lane];
                double c[5];
#pragma unroll
                for (int a = 0; a < 5; ++a) c[a] = va[a] * bv;
#pragma unroll
                for (int a = 0; a < 5; ++a) asm volatile("" : "+v"(c[a]));
                asm volatile("" ::: "memory");
                row(va, pn, tv);
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) acc[a][b] = fma(vb[b], c[a], acc[a][b]);
#pragma unroll
                for (int a = 0; a < 5; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) asm volatile("" : "+v"(acc[a][b]));
                asm volatile("" ::: "memory");
                row(vb, pn, tu);
                if (have && !NOST) o[poff[j]] = park[j];      // the completed pair j of the LAST span goes out with plane j of this one
                bv = bvn;
                pl = pn;
                if (++cnt == 8) { cnt = 0; geo(); __syncthreads(); }
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) { park[a] = acc[a][0]; poff[a] = (long long)(5 * (s + a) + 4 - a) * stride; }
            have = true;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) acc[a][b] = acc[a + 1][b + 1];
#pragma unroll
            for (int b = 0; b < 5; ++b) acc[4][b] = 0.0;
        }
        if (acc[0][0] == 1.234e300) out[0] = acc[1][1] + acc[2][2] + acc[3][3] + acc[4][4] + acc[3][1];
    } else if (MODE == 1) {
        double4_t acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = {0, 0, 0, 0};
        const int g = lane >> 4, n = lane & 15;
        // slot classes: rows 0-4 delta 0 (period 5), 5-8 delta 1 (4), 9-11 delta 2 (3), 12-13 delta 3 (2), 14 delta 4 (1), 15 unused
        int dl[4], rr[4], per[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = 4 * g + v;
            dl[v] = m < 5 ? 0 : m < 9 ? 1 : m < 12 ? 2 : m < 14 ? 3 : m < 15 ? 4 : 5;
            rr[v] = m < 5 ? m : m < 9 ? m - 5 : m < 12 ? m - 9 : m < 14 ? m - 12 : 0;
            per[v] = 5 - dl[v];
        }
        int pl = 0, cnt = 0;
        int ph[4] = {0, 0, 0, 0};                    // s mod period of this lane's four rows
        const double *xbase = out + (long long)w * xstride + pt;
        for (int s = 0; s < nspans; ++s) {
            const double am = atab[t][g][n];
            const d2 *cr = (const d2 *)&atab[t][4][4 * g];
            const d2 c01 = cr[0], c23 = cr[1];
            const int pl4 = pl + 4 >= NPL ? pl + 4 - NPL : pl + 4;
            double bm[4], b5[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int pk = pl + g; pk = pk >= NPL ? pk - NPL : pk;
                bm[nt] = fld[pk * PST + fi * 64 + 16 * nt + n];
                b5[nt] = fld[pl4 * PST + fi * 64 + 16 * nt + n];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm[nt], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt][0] = fma(c01.x, b5[nt], acc[nt][0]); acc[nt][1] = fma(c01.y, b5[nt], acc[nt][1]);
                acc[nt][2] = fma(c23.x, b5[nt], acc[nt][2]); acc[nt][3] = fma(c23.y, b5[nt], acc[nt][3]);
            }
            pl = pl + 5 >= NPL ? pl + 5 - NPL : pl + 5;
            cnt += 5;
            // completed slots: stored and cleared in place
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xbase + (long long)(5 * s) * stride), (short)0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool done = dl[v] < 5 && ph[v] == rr[v];
                const unsigned off = done ? (unsigned)((5 * dl[v] + 4 - dl[v]) * stride * 8) + n * 8 : 0x7ffffff8u;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (!NOST) bstore(rs, off + (done ? nt * 128 : 0), acc[nt][v]);
                    acc[nt][v] = done ? 0.0 : acc[nt][v];
                }
                ph[v] = ph[v] + 1 >= per[v] ? 0 : ph[v] + 1;
            }
            if (cnt >= 8) { cnt -= 8; geo(); __syncthreads(); }
        }
        if (acc[0][0] == 1.234e300) out[0] = acc[1][1] + acc[2][2] + acc[3][3];
    } else if (MODE == 2) {
        // lean form: row i of the accumulator tile = lane / 16 + 4 v (measured layout), points interleaved so that a lane holds
        // two neighbouring points per register pair (16-byte stores, 256 B per completed row and store); completed rows are
        // found with one compare per register (phase counters of the five slot classes packed into one scalar), cleared
        // under an exec mask
        double4_t acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = {0, 0, 0, 0};
        const int g = lane >> 4, n = lane & 15;
        int sh[4], rr[4];
        unsigned cst[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = g + 4 * v;
            const int dl = m < 5 ? 0 : m < 9 ? 1 : m < 12 ? 2 : m < 14 ? 3 : 4;
            rr[v] = m < 5 ? m : m < 9 ? m - 5 : m < 12 ? m - 9 : m < 14 ? m - 12 : m < 15 ? 0 : 15;
            sh[v] = 4 * dl;
            cst[v] = (unsigned)((5 * dl + 4 - dl) * stride * 8) + n * 8;
        }
        int pl = 0, cnt = 0;
        unsigned phw = 0;                             // s mod (5 - delta) in nibble delta (scalar)
        int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        const double *xbase = out + (long long)w * xstride + pt;
        // point of (tile nt, column n): 2 n + (nt & 1) + 32 (nt >> 1)
        for (int s = 0; s < nspans; ++s) {
            const double am = atab[t][g][n];
            const d2 *cr = (const d2 *)&atab[t][4][4 * g];
            const d2 c01 = cr[0], c23 = cr[1];
            const int pl4 = pl + 4 >= NPL ? pl + 4 - NPL : pl + 4;
            int pk = pl + g; pk = pk >= NPL ? pk - NPL : pk;
            double bm[4], b5[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                bm[nt] = fld[pk * PST + fi * 64 + 16 * nt + n];
                b5[nt] = fld[pl4 * PST + fi * 64 + 16 * nt + n];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bm[nt], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt][0] = fma(c01.x, b5[nt], acc[nt][0]); acc[nt][1] = fma(c01.y, b5[nt], acc[nt][1]);
                acc[nt][2] = fma(c23.x, b5[nt], acc[nt][2]); acc[nt][3] = fma(c23.y, b5[nt], acc[nt][3]);
            }
            pl = pl + 5 >= NPL ? pl + 5 - NPL : pl + 5;
            cnt += 5;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xbase + (long long)(5 * s) * stride), (short)0, 0x7ffffff0, 0x00020000);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool done = (int)((phw >> sh[v]) & 15u) == rr[v];
                const unsigned off = done ? cst[v] : 0x7ffffff8u;
                if (!NOST) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) bstore(rs, off + nt * 128u, acc[nt][v]);   // (out-of-range lanes stay out of range)
                }
                {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(done);
                    unsigned long long sv;
                    asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tv_mov_b64 %[a], 0\n\tv_mov_b64 %[b], 0\n\tv_mov_b64 %[c], 0\n\tv_mov_b64 %[d], 0\n\ts_mov_b64 exec, %[sv]"
                                 : [a] "+v"(acc[0][v]), [b] "+v"(acc[1][v]), [c] "+v"(acc[2][v]), [d] "+v"(acc[3][v]), [sv] "=&s"(sv) : [m] "s"(m) : "scc");
                }
            }
            p0 = p0 == 4 ? 0 : p0 + 1; p1 = p1 == 3 ? 0 : p1 + 1; p2 = p2 == 2 ? 0 : p2 + 1; p3 = p3 ^ 1;
            phw = (unsigned)(p0 | (p1 << 4) | (p2 << 8) | (p3 << 12));
            if (cnt >= 8) { cnt -= 8; geo(); __syncthreads(); }
        }
        if (acc[0][0] == 1.234e300) out[0] = acc[1][1] + acc[2][2] + acc[3][3];
    }
    if (g0 + g1 + g2 + g3 == 1.234e300) out[1] = g0;
}

template <int MODE, int GEO, int NOST = 0>
static float run(double *out, long long stride, long long xstride, int nspans, const double *init, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_sweep<MODE, GEO, NOST><<<blocks, 512>>>(out, stride, xstride, nspans, init);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        k_sweep<MODE, GEO, NOST><<<blocks, 512>>>(out, stride, xstride, nspans, init);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return best;
}

int main()
{
    const int nspans = 128, blocks = 6400;
    const long long stride = 640LL * 640, npairs = 5LL * (nspans + 5), xstride = npairs * stride;
    double *out, *init;
    if (hipMalloc(&out, (size_t)xstride * 8 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&init, 4096 * 8);
    double h[4096];
    srand(2);
    for (int i = 0; i < 4096; ++i) h[i] = 0.5 + rand() / (double)RAND_MAX * 1e-3;
    hipMemcpy(init, h, sizeof h, hipMemcpyHostToDevice);
    printf("C4-shaped sweep: %d blocks x 8 waves x 64 points, %d spans of 5 planes, K1 %.1f GB\n", blocks, nspans, blocks * 64.0 * 8 * 5 * nspans * 8 / 1e9);
#define ROW(G) printf("geometry stand-in %3d FMAs / 8 planes:  VALU %.2f  VALU, stores spread over the next span %.2f  MFMA(first form) %.2f  MFMA(lean) %.2f ms   | without the K1 stores: %.2f  %.2f  %.2f ms\n", G, \
        run<0, G>(out, stride, xstride, nspans, init, blocks), run<3, G>(out, stride, xstride, nspans, init, blocks), run<1, G>(out, stride, xstride, nspans, init, blocks), run<2, G>(out, stride, xstride, nspans, init, blocks), \
        run<0, G, 1>(out, stride, xstride, nspans, init, blocks), run<1, G, 1>(out, stride, xstride, nspans, init, blocks), run<2, G, 1>(out, stride, xstride, nspans, init, blocks))
    ROW(0); ROW(128); ROW(256);
    return 0;
}

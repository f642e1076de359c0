// Microbenchmark: the K1 store pattern of the axis-0 sweep.  K1[x][r][pt]: NX arrays, NR rows of NPL doubles.
// A block owns NT consecutive 64-point tiles and NXB arrays (one wave per (array, tile)); it walks the rows in
// flushes of 5 rows.  Variants: how many tiles / arrays a block covers, and 16-byte stores (2 points per lane).
// build: hipcc --offload-arch=gfx950 -O3 k1_store.hip -o k1_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W, int NT_>   // doubles per lane per store (1 or 2); NT_: 0 plain stores, 1 nontemporal
__global__ void k_store(double *K1, long long NPL, int NR, int NX, int NXB, int NT, int spin, int do_store, const double *tbl, int use_tbl)
{
    extern __shared__ double occupancy_limiter[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (spin < 0) occupancy_limiter[threadIdx.x] = 1.0;
    const int xl = w / NT, tl = w % NT;                       // array within the block's group, tile within the block
    const int ngroups = NX / NXB;
    const int group = blockIdx.x % ngroups;
    const long long tile = (long long)(blockIdx.x / ngroups) * NT + tl;
    const int x = group * NXB + xl;
    const long long pt = tile * 64 * W + lane * W;
    if (pt >= NPL) return;
    double *out = K1 + (long long)x * NR * NPL + pt;
    double v = lane, u[8];
    for (int k = 0; k < 8; ++k) u[k] = lane + k;
    for (int f = 0; f < NR / 5; ++f) {
        for (int i = 0; i < spin; ++i) {                               // stand-in for the arithmetic between flushes
            double c = 1.0000001;
            if (use_tbl) c = ((const double __attribute__((address_space(4))) *)tbl)[(((f * 64 + i) * 25 + x) * 16) & 65535];   // wave-uniform: scalar load, 512 KB table
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = fma(u[k], c, 1e-9);
        }
        v = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
        if (!do_store) { if (v == 12345.678) out[0] = v; continue; }
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            if (W == 1 && NT_) __builtin_nontemporal_store(v, out + (long long)(f * 5 + a) * NPL);
            else if (W == 1) out[(long long)(f * 5 + a) * NPL] = v;
            else { double2 d; d.x = v; d.y = v; *(double2 *)(out + (long long)(f * 5 + a) * NPL) = d; }
        }
    }
}

int main(int argc, char **argv)
{
    const long long NPL = 640 * 640;
    const int NR = 650, NX = 8;
    double *K1;
    if (hipMalloc(&K1, (size_t)NX * NR * NPL * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    double *tbl; (void)hipMalloc(&tbl, 65536 * 8); (void)hipMemset(tbl, 0, 65536 * 8);
    { double one = 1.0000001; std::vector<double> h(65536, one); (void)hipMemcpy(tbl, h.data(), 65536 * 8, hipMemcpyHostToDevice); }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double gb = (double)NX * (NR / 5 * 5) * NPL * 8 / 1e9;
    printf("K1 store pattern, %.1f GB per launch\n", gb);
    struct V { int nxb, nt, w, spin, st, lds, ntmp, tb; } vs[] = {
        {8, 1, 1, 0, 1, 65536, 0, 0},
        {8, 1, 1, 25, 0, 65536, 0, 0}, {8, 1, 1, 25, 1, 65536, 0, 0},
        {8, 1, 1, 25, 0, 65536, 0, 1}, {8, 1, 1, 25, 1, 65536, 0, 1},
        {8, 1, 1, 12, 0, 65536, 0, 1}, {8, 1, 1, 12, 1, 65536, 0, 1},
    };
    (void)hipFuncSetAttribute((const void *)k_store<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)hipFuncSetAttribute((const void *)k_store<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)hipFuncSetAttribute((const void *)k_store<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (auto &v : vs) {
        const int nw = v.nxb * v.nt;
        const long long tiles = (NPL / (64 * v.w) + v.nt - 1) / v.nt;
        dim3 grid((unsigned)(tiles * (NX / v.nxb))), block(nw * 64);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (v.w == 1) { if (v.ntmp) k_store<1, 1><<<grid, block, v.lds>>>(K1, NPL, NR, NX, v.nxb, v.nt, v.spin, v.st, tbl, v.tb); else k_store<1, 0><<<grid, block, v.lds>>>(K1, NPL, NR, NX, v.nxb, v.nt, v.spin, v.st, tbl, v.tb); }
            else k_store<2, 0><<<grid, block, v.lds>>>(K1, NPL, NR, NX, v.nxb, v.nt, v.spin, v.st, tbl, v.tb);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("tbl %d lds %6d nt %d arrays/block %d, tiles/block %d, %2d B/lane, spin %3d, stores %d: %7.3f ms  %7.1f GB/s  (%d waves/block, %u blocks)\n",
               v.tb, v.lds, v.ntmp, v.nxb, v.nt, 8 * v.w, v.spin, v.st, best, gb / best * 1e3, nw, grid.x);
    }
    return 0;
}

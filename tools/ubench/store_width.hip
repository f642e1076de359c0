// Microbenchmark: write rate of streaming stores by width per lane (8 / 16 bytes) and by how the lanes of a wave are spread
// over independent streams -- the K1 stores of k_geoA are 8 bytes per lane, 512 contiguous bytes per instruction, five
// streams (slices) per wave and span.  build: hipcc --offload-arch=gfx950 -O3 store_width.hip -o store_width
//   A  8 B per lane, one stream:   lane l of wave-instruction k writes p[(k * 64 + l)]
//   B  16 B per lane, one stream
//   C  8 B per lane, 5 streams 3.3 MB apart (K1 slices), consecutive instructions go to different streams
//   D  16 B per lane, half-waves on two streams (lanes 0-31 -> stream s, 32-63 -> stream s + 1): the DPP-paired form
//   E  16 B per lane, 5 streams
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr long long SL = 409600;                          // doubles per slice (C4 plane)

template <int MODE>
__global__ void __launch_bounds__(512) k_st(double *p, int nspan, double v)
{
    // block = 8 waves on 64 points (MODE A-C, E-with-2-points: 32 lanes x 2) of a plane tile; wave w writes array w: arrays 650 slices apart
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *base = p + (long long)w * 650 * SL + (long long)blockIdx.x * 64;
    for (int s = 0; s < nspan; ++s) {
        if (MODE == 0) {
#pragma unroll
            for (int a = 0; a < 5; ++a) base[(long long)(5 * s + a) * SL + lane] = v + a;
        } else if (MODE == 1) {
            // 16 B per lane: even half of the lanes covers the tile for pair a, odd half for pair a + 1
#pragma unroll
            for (int a = 0; a < 6; a += 2) {
                const int half = lane >> 5, l2 = lane & 31;
                if (a + half < 5) { d2 x = {v + a, v + a + 1}; *(d2 *)(base + (long long)(5 * s + a + half) * SL + 2 * l2) = x; }
            }
        }
    }
}
template <int W>
__global__ void __launch_bounds__(256) k_fill(double *p, size_t n, double v)
{
    if (W == 1) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
    else { d2 x = {v, v}; d2 *q = (d2 *)p; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * blockDim.x) q[i] = x; }
}
template <class F> static float ms_of(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
    return best;
}
int main()
{
    const size_t n = (size_t)8 * 650 * SL + 4096;          // K1 of C4: 17 GB
    double *p; if (hipMalloc(&p, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double gb = 8.0 * 650 * SL * 8 / 1e9;
    float a = ms_of([&] { k_fill<1><<<8192, 256>>>(p, (size_t)8 * 650 * SL, 1.0); });
    float b = ms_of([&] { k_fill<2><<<8192, 256>>>(p, (size_t)8 * 650 * SL, 1.0); });
    printf("plain fill of %.1f GB: 8 B per lane %.2f ms (%.2f TB/s), 16 B per lane %.2f ms (%.2f TB/s)\n", gb, a, gb / a, b, gb / b);
    float c = ms_of([&] { k_st<0><<<6400, 512>>>(p, 130, 1.0); });
    float d = ms_of([&] { k_st<1><<<6400, 512>>>(p, 130, 1.0); });
    printf("K1-shaped (6400 blocks x 8 arrays x 130 spans x 5 slices): 8 B per lane %.2f ms (%.2f TB/s), 16 B half-wave pairs %.2f ms (%.2f TB/s)\n", c, gb / c, d, gb / d);
    return 0;
}

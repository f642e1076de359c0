// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(double *out, int iters, double a0, double b0)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k_fma(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float time_ms(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *out; hipMalloc(&out, 1 << 26);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz\n", p.name, cus, ghz);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        dim3 grid(cus), block(256 * waves_per_simd);
        {
            float ms = time_ms([&] { k_mfma<1><<<grid, block>>>(out, iters, 1.0, 2.0); });
            double n = (double)iters * 1 * cus * 4 * waves_per_simd;
            printf("mfma f64 16x16x4, %d wave/SIMD, 1 acc (dependent): %.1f cycles/MFMA/SIMD, %.1f TFLOP/s\n", waves_per_simd,
                   ms * 1e-3 * ghz * 1e9 / (iters * 1.0 * waves_per_simd), n * 2048 / (ms * 1e-3) / 1e12);
        }
        {
            float ms = time_ms([&] { k_mfma<4><<<grid, block>>>(out, iters, 1.0, 2.0); });
            double n = (double)iters * 4 * cus * 4 * waves_per_simd;
            printf("mfma f64 16x16x4, %d wave/SIMD, 4 acc: %.1f cycles/MFMA/SIMD, %.1f TFLOP/s\n", waves_per_simd,
                   ms * 1e-3 * ghz * 1e9 / (iters * 4.0 * waves_per_simd), n * 2048 / (ms * 1e-3) / 1e12);
        }
        {
            float ms = time_ms([&] { k_fma<8><<<grid, block>>>(out, iters, 1.0000001, 1e-9); });
            double n = (double)iters * 8 * cus * 4 * waves_per_simd;
            printf("v_fma_f64, %d wave/SIMD, 8 acc: %.1f cycles/FMA/SIMD, %.1f TFLOP/s\n", waves_per_simd,
                   ms * 1e-3 * ghz * 1e9 / (iters * 8.0 * waves_per_simd), n * 128 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

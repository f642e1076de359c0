// Microbenchmark: do v_mfma_f64_16x16x4_f64 (matrix core) and v_fma_f64 (VALU) of DIFFERENT waves on one SIMD overlap?
// build: hipcc --offload-arch=gfx950 -O3 mix_f64.hip -o mix_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

// waves [0, 4*nm) of a block run MFMAs, waves [4*nm, 4*(nm+nf)) run FMAs (wave % 4 = SIMD)
__global__ void __launch_bounds__(1024) k_mix(double *out, int nm, int it_m, int it_f, double a0, double b0)
{
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4 * nm) {
        double4_t acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = {0, 0, 0, 0};
        double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = i;
        double a = a0 + threadIdx.x * 1e-9, b = b0 * 1e-9;
        for (int it = 0; it < it_f; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
        }
        for (int i = 0; i < 8; ++i) s += acc[i];
    }
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
    printf("device %s, %d CUs, %.2f GHz; per wave: 4000x4 MFMA or 8000x8 FMA\n", p.name, cus, ghz);
    const int it_m = 4000, it_f = 8000;
    for (int nm = 0; nm <= 2; ++nm)
        for (int nf = 0; nf <= 3; ++nf) {
            if (nm + nf == 0) continue;
            dim3 grid(cus), block(256 * (nm + nf));
            float ms = time_ms([&] { k_mix<<<grid, block>>>(out, nm, it_m, it_f, 1.0000001, 2.0); });
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf("%d MFMA wave(s) + %d FMA wave(s) per SIMD: %.3f ms = %.0f cycles", nm, nf, ms, cyc);
            if (nm) printf("  | %.1f cyc/MFMA", cyc / (it_m * 4.0 * nm));
            if (nf) printf("  | %.2f cyc/FMA", cyc / (it_f * 8.0 * nf));
            printf("\n");
        }
    return 0;
}

// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: SIMD_ID = bits 5:4, WAVE_ID = bits 3:0, CU_ID = bits 11:8)
// build: hipcc --offload-arch=gfx950 -O3 wave_simd.hip -o wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);      // HW_ID[15:0]
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main()
{
    int *d; (void)hipMalloc(&d, 4096 * 16 * sizeof(int));
    for (int nw : {4, 7, 8, 12, 16}) {
        (void)hipMemset(d, 0xff, 4096 * 16 * sizeof(int));
        hipLaunchKernelGGL(k, dim3(1024), dim3(nw * 64), 0, 0, d);
        (void)hipDeviceSynchronize();
        static int h[4096 * 16];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%2d waves per workgroup; SIMD of wave 0..%d in blocks 0, 1, 500, 1023:\n", nw, nw - 1);
        for (int b : {0, 1, 500, 1023}) {
            printf("   block %4d (CU %2d):", b, (h[b * 16] >> 8) & 15);
            for (int w = 0; w < nw; ++w) printf(" %d", (h[b * 16 + w] >> 4) & 3);
            printf("\n");
        }
        int hist[16][4] = {};
        for (int b = 0; b < 1024; ++b) for (int w = 0; w < nw; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
        printf("   histogram wave -> SIMD over 1024 blocks:");
        for (int w = 0; w < nw; ++w) printf(" [%d %d %d %d]", hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("\n");
    }
    return 0;
}

// Microbenchmark: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, decoded from the hardware.
// build: hipcc --offload-arch=gfx950 -O3 mfma_layout.hip -o mfma_layout
// Hypothesis for the inputs (AMD's published layout): A[i][k] sits in lane 16 k + i, B[k][j] in lane 16 k + j.  With
// A[i][0] = i + 1, A[i][1] = 1, B[0][j] = 1, B[1][j] = 32 (j + 1) every D[i][j] = (i + 1) + 32 (j + 1) is unique, so the
// (lane, register) that holds it can be read off.  A second launch checks the decoded layout on random matrices.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_probe(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = acc[v];
}

int main()
{
    double ha[64], hb[64], hd[256], *a, *b, *d;
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&d, sizeof hd);
    for (int l = 0; l < 64; ++l) {
        const int k = l / 16, i = l % 16;
        ha[l] = k == 0 ? i + 1 : k == 1 ? 1 : 0;
        hb[l] = k == 0 ? 1 : k == 1 ? 32 * (i + 1) : 0;
    }
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    k_probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    int li[64][4], lj[64][4];
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const int x = (int)hd[l * 4 + v];
            li[l][v] = x % 32 - 1; lj[l][v] = x / 32 - 1;
        }
    printf("lane: (i,j) of registers 0..3\n");
    for (int l = 0; l < 64; ++l)
        printf("%2d: (%2d,%2d) (%2d,%2d) (%2d,%2d) (%2d,%2d)\n", l, li[l][0], lj[l][0], li[l][1], lj[l][1], li[l][2], lj[l][2], li[l][3], lj[l][3]);
    bool h1 = true, h2 = true;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            if (lj[l][v] != l % 16) h1 = h2 = false;
            if (li[l][v] != 4 * (l / 16) + v) h1 = false;
            if (li[l][v] != (l / 16) + 4 * v) h2 = false;
        }
    printf("D layout: %s\n", h1 ? "i = 4 (lane / 16) + v, j = lane % 16" : h2 ? "i = lane / 16 + 4 v, j = lane % 16" : "other (see table)");
    // random check of the decoded layout
    srand(1);
    double A[16][4], B[4][16];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i][k] = rand() / (double)RAND_MAX - 0.5;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k][j] = rand() / (double)RAND_MAX - 0.5;
    for (int l = 0; l < 64; ++l) { ha[l] = A[l % 16][l / 16]; hb[l] = B[l / 16][l % 16]; }
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    k_probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            double r = 0;
            for (int k = 0; k < 4; ++k) r = fma(A[li[l][v]][k], B[k][lj[l][v]], r);
            worst = fmax(worst, fabs(r - hd[l * 4 + v]));
        }
    printf("random matrices through the decoded layout: max |D - A B| = %.3g (ascending-k fma chain on the host)\n", worst);
    return 0;
}

// Microbenchmark: does an intermediate that is written by one kernel and read back by the next one come out of the 256 MiB
// Infinity Cache faster than out of HBM?  (Round-4 question: K1 slices of the assembly chain kept resident.)
// build: hipcc --offload-arch=gfx950 -O3 mall_k1.hip -o mall_k1
//   write kernel: coalesced 8-byte stores (the shape of k_geoA's K1 stores); read kernel: coalesced 8-byte loads, 8 in
//   flight per lane (the shape of k_bf2's K1 loads).  For slice sizes W: (a) read slices nobody touched for > 4 GB of
//   traffic (HBM), (b) write slice s, then read slice s (resident if W fits), (c) the write alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_write(double *p, size_t n, double v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (double)i;
}
__global__ void __launch_bounds__(256) k_read(const double *p, size_t n, double *out)
{
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += p[i + k * stride];
    }
    for (; i < n; i += stride) s[0] += p[i];
    double t = 0;
    for (int k = 0; k < 8; ++k) t += s[k];
    if (t == 1.2345e300) out[0] = t;
}

int main()
{
    const size_t TOTAL = (size_t)6 << 30;                 // 6 GiB buffer
    double *buf, *out;
    if (hipMalloc(&buf, TOTAL) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 64);
    k_write<<<4096, 256>>>(buf, TOTAL / 8, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int sizes_mb[] = {16, 32, 64, 96, 128, 192, 256, 512, 1024};
    printf("slice MB | read cold GB/s | read after write GB/s | ratio | write GB/s | write+read pair GB/s (bytes of both)\n");
    for (int W : sizes_mb) {
        const size_t bytes = (size_t)W << 20, n = bytes / 8;
        const int nsl = (int)(TOTAL / bytes);
        const int use = nsl > 48 ? 48 : nsl;
        const int grid = 2048;
        float ms;
        // (a) cold reads: the whole buffer was last written front to back; read slices from the front
        double t_cold = 0;
        // flush: touch the tail of the buffer so that the front is long gone
        k_read<<<grid, 256>>>(buf + (TOTAL / 8) / 2, (TOTAL / 8) / 2, out);
        for (int s = 0; s < use; ++s) {
            hipEventRecord(e0);
            k_read<<<grid, 256>>>(buf + (size_t)s * n, n, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); t_cold += ms;
        }
        // (c) write alone
        double t_w = 0;
        for (int s = 0; s < use; ++s) {
            hipEventRecord(e0);
            k_write<<<grid, 256>>>(buf + (size_t)s * n, n, 2.0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); t_w += ms;
        }
        // (b) write slice, read slice
        double t_r = 0, t_pair = 0;
        hipEvent_t e2; hipEventCreate(&e2);
        for (int s = 0; s < use; ++s) {
            hipEventRecord(e0);
            k_write<<<grid, 256>>>(buf + (size_t)s * n, n, 3.0);
            hipEventRecord(e2);
            k_read<<<grid, 256>>>(buf + (size_t)s * n, n, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e2, e1); t_r += ms;
            hipEventElapsedTime(&ms, e0, e1); t_pair += ms;
        }
        const double gb = (double)bytes * use / 1e9;
        printf("%5d | %8.0f | %8.0f | %5.2f | %8.0f | %8.0f\n", W, gb / (t_cold * 1e-3), gb / (t_r * 1e-3), t_cold / t_r, gb / (t_w * 1e-3), 2 * gb / (t_pair * 1e-3));
    }
    return 0;
}

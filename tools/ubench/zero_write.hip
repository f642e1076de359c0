// Microbenchmark: does the memory side count (or perform) stores of all-zero lines like any other store?  (Round 4: WRITE_SIZE of the
// non-symmetric k_geoA at C5 reads 17.2 GB where 22.95 GB of K1 are stored; two of its eight arrays are exact zeros for a
// tensor-product geometry.)  Run under: rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- ./zero_write
// build: hipcc --offload-arch=gfx950 -O3 zero_write.hip -o zero_write
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_fill(double *p, size_t n, double v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void __launch_bounds__(256) k_fill_zero_over_zero(double *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0; }
__global__ void __launch_bounds__(256) k_fill_zero_over_ones(double *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0; }
__global__ void __launch_bounds__(256) k_fill_ones_over_ones(double *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
__global__ void __launch_bounds__(256) k_fill_ones_over_zero(double *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
int main()
{
    const size_t n = (size_t)1 << 28;                     // 2 GiB of doubles
    double *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    k_fill<<<4096, 256>>>(b, n, 1.0);
    hipDeviceSynchronize();
    k_fill_zero_over_zero<<<4096, 256>>>(a, n);
    hipDeviceSynchronize();
    k_fill_ones_over_ones<<<4096, 256>>>(b, n);
    hipDeviceSynchronize();
    k_fill_zero_over_ones<<<4096, 256>>>(b, n);
    hipDeviceSynchronize();
    k_fill_ones_over_zero<<<4096, 256>>>(a, n);
    hipDeviceSynchronize();
    printf("four fills of %.2f GB each\n", n * 8 / 1e9);
    return 0;
}

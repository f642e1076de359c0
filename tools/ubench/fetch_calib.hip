// Calibration of the rocprofv3 FETCH_SIZE counter on gfx950 for the read patterns of the assembly chain:
// every kernel reads a KNOWN set of bytes exactly once (no reuse), so FETCH_SIZE * 1024 / bytes gives the factor by which the
// counter has to be corrected for that access width.
//   stream16 : 16 bytes per lane, consecutive lanes consecutive (the guide's case: counted at 1/2)
//   stream8  : 8 bytes per lane, consecutive (the K1 reads of k_bf2: buffer_load_dwordx2)
//   runs72   : runs of 9 doubles, one run per 5832-byte row (the gathers of the mirror pass: 7 runs per wave instruction),
//              every run read once; the 128-byte lines / 64-byte halves / 32-byte sectors it touches are counted on the host
// run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_calib -- ./tools/ubench/fetch_calib
//       (then tools/fetch_calib_summary.py gpurun_out/fetch_calib)
// build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_stream16(const double2 *in, double *out, long long n)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    double s = 0;
    for (long long i = i0; i < n; i += stride) { const double2 v = in[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
__global__ void k_stream8(const double *in, double *out, long long n)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    double s = 0;
    for (long long i = i0; i < n; i += stride) s += in[i];
    if (s == 1.2345e300) out[0] = s;
}
// row r (ROWLEN doubles) holds one run of 9 doubles at offset OFF; a wave instruction reads the runs of 7 consecutive rows
constexpr long long ROWLEN = 729, OFF = 360;
__global__ void k_runs72(const double *in, double *out, long long nrows)
{
    const int lane = threadIdx.x & 63, r = lane / 9, o = lane - r * 9;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    double s = 0;
    for (long long g = wave; g * 7 + 6 < nrows; g += nwaves)
        if (r < 7) s += in[(g * 7 + r) * ROWLEN + OFF + o];
    if (s == 1.2345e300) out[0] = s;
}

int main()
{
    const long long n = 1LL << 29;                     // 4 GiB of doubles
    double *buf, *out;
    CK(hipMalloc(&buf, n * sizeof(double)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, n * sizeof(double)));
    const long long nrows = n / ROWLEN;
    // bytes the patterns touch, by granule
    long long runs = (nrows / 7) * 7, l128 = 0, l64 = 0, l32 = 0;
    for (long long r = 0; r < runs; ++r) {
        const long long b0 = (r * ROWLEN + OFF) * 8, b1 = b0 + 71;
        l128 += b1 / 128 - b0 / 128 + 1; l64 += b1 / 64 - b0 / 64 + 1; l32 += b1 / 32 - b0 / 32 + 1;
    }
    printf("CALIB stream16 bytes %lld\n", n * 8);
    printf("CALIB stream8 bytes %lld\n", n * 8);
    printf("CALIB runs72 bytes_useful %lld bytes_128B_lines %lld bytes_64B_halves %lld bytes_32B_sectors %lld\n", runs * 72, l128 * 128, l64 * 64, l32 * 32);
    for (int rep = 0; rep < 2; ++rep) {
        k_stream16<<<4096, 256>>>((const double2 *)buf, out, n / 2);
        k_stream8<<<4096, 256>>>(buf, out, n);
        k_runs72<<<4096, 256>>>(buf, out, nrows);
        CK(hipDeviceSynchronize());
    }
    return 0;
}

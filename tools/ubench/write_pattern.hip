// Micro-benchmark: what does the HBM write path of MI355X deliver for the store patterns of the final stage?
//   stream   : every wave instruction writes 512 contiguous bytes, waves walk the buffer linearly
//   runs72   : a wave instruction writes 7 runs of 9 doubles (72 B) 5832 B apart (7 CSR rows of one line);
//              the next instruction of the same wave writes the runs right behind them (direct stores, T = 1)
//   runs72far: same runs, but consecutive instructions jump ~757 KB (mirrored stores: another row block)
//   seg648   : runs of 81 doubles (648 B = the 9 lines of a row group batched), 5832 B apart
// Total bytes written are the same (~12.7 GB).  Build: hipcc --offload-arch=gfx950 -O3 write_pattern.hip -o write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr long long ROWLEN = 729;            // doubles per CSR row (3D p=4 interior)
constexpr long long NROWS = 2180000;         // ~1.59e9 / 729

__global__ void k_stream(double *buf, long long n)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = i0; i < n; i += stride) buf[i] = (double)i;
}

// mode 0: runs of RUN doubles, a wave covers rows [r0, r0+ROWS_PER_WAVE) and walks the row's slots in order
// mode 1: same, but the slot visited at step s of a wave is far from the previous one (different row block)
template <int RUN>
__global__ void k_runs(double *buf, long long nrows, int mode)
{
    constexpr int RPW = 64 / RUN > 0 ? 64 / RUN : 1;     // rows per wave instruction (7 for RUN = 9)
    constexpr int SLOTS = (int)(ROWLEN / RUN);          // 81 slots of 9, or 9 slots of 81
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ngroups = nrows / RPW;
    for (long long g = wave; g < ngroups; g += nwaves) {
        for (int s = 0; s < SLOTS; ++s) {
            if (RUN <= 64) {
                const int r = lane / RUN, o = lane - r * RUN;
                if (r < RPW) {
                    long long row = g * RPW + r;
                    int slot = s;
                    if (mode == 1) row = (row + (long long)s * 17387) % nrows;      // jump to another row block per step
                    buf[row * ROWLEN + (long long)slot * RUN + o] = (double)s;
                }
            } else {
                // RUN > 64: one row per pass, lanes cover the run in pieces of 64
                for (int o = lane; o < RUN; o += 64) buf[g * ROWLEN + (long long)s * RUN + o] = (double)s;
            }
        }
    }
}

// K interleaved streams per wave: step s writes to stream s % K (each stream = its own group of rows far away),
// slot s / K -- consecutive stores of ONE stream are adjacent in memory but K - 1 other stores apart in time
__global__ void k_interleave(double *buf, long long nrows, int K)
{
    constexpr int RUN = 9, RPW = 7, SLOTS = 81;
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ngroups = nrows / RPW;
    const int r = lane / RUN, o = lane - r * RUN;
    for (long long g0 = wave * K; g0 + K <= ngroups; g0 += nwaves * K)
        for (int s = 0; s < SLOTS * K; ++s) {
            const long long g = g0 + (s % K);
            if (r < RPW) buf[(g * RPW + r) * ROWLEN + (long long)(s / K) * RUN + o] = (double)s;
        }
}

int main()
{
    const long long n = NROWS * ROWLEN;
    double *buf;
    CK(hipMalloc(&buf, n * sizeof(double)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(a));
            launch();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("%-10s %8.3f ms  %7.1f GB/s\n", name, best, n * 8.0 / best / 1e6);
    };
    time("stream", [&] { k_stream<<<4096, 256>>>(buf, n); });
    time("runs72", [&] { k_runs<9><<<4096, 256>>>(buf, NROWS, 0); });
    time("runs72far", [&] { k_runs<9><<<4096, 256>>>(buf, NROWS, 1); });
    time("seg648", [&] { k_runs<81><<<4096, 256>>>(buf, NROWS, 0); });
    for (int K : {1, 2, 3, 4, 6, 8, 12, 16, 32}) {
        char name[32]; snprintf(name, sizeof name, "ileave%d", K);
        time(name, [&] { k_interleave<<<4096, 256>>>(buf, NROWS, K); });
    }
    CK(hipFree(buf));
    return 0;
}

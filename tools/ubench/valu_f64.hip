// Microbenchmark: what does an FP64 VALU instruction cost on gfx950, by operand kind, and what do the instructions
// around it (integer VALU, SALU, LDS reads, moves) cost next to it?  Every pattern is a hand-written asm loop with fixed
// registers (no compiler scheduling), 32 FP64 instructions per iteration on 16 independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 valu_f64.hip -o valu_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
    "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
    "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20","s21","s22","s23","s24","s25","s26","s27","scc","vcc","memory"

// accumulators v[0:1] .. v[30:31]; multiplicands v[40:41] (banks 0,1), v[42:43] (banks 2,3), v[44:45] (banks 0,1); s[22:23], s[24:25]
#define ACC16(OP, SRC) \
    OP " v[0:1], " SRC "\n" OP " v[2:3], " SRC "\n" OP " v[4:5], " SRC "\n" OP " v[6:7], " SRC "\n" \
    OP " v[8:9], " SRC "\n" OP " v[10:11], " SRC "\n" OP " v[12:13], " SRC "\n" OP " v[14:15], " SRC "\n" \
    OP " v[16:17], " SRC "\n" OP " v[18:19], " SRC "\n" OP " v[20:21], " SRC "\n" OP " v[22:23], " SRC "\n" \
    OP " v[24:25], " SRC "\n" OP " v[26:27], " SRC "\n" OP " v[28:29], " SRC "\n" OP " v[30:31], " SRC "\n"
// the same with one extra instruction X after every FP64 instruction / after every second / fourth one
#define ACC16_X1(OP, SRC, X) \
    OP " v[0:1], " SRC "\n" X OP " v[2:3], " SRC "\n" X OP " v[4:5], " SRC "\n" X OP " v[6:7], " SRC "\n" X \
    OP " v[8:9], " SRC "\n" X OP " v[10:11], " SRC "\n" X OP " v[12:13], " SRC "\n" X OP " v[14:15], " SRC "\n" X \
    OP " v[16:17], " SRC "\n" X OP " v[18:19], " SRC "\n" X OP " v[20:21], " SRC "\n" X OP " v[22:23], " SRC "\n" X \
    OP " v[24:25], " SRC "\n" X OP " v[26:27], " SRC "\n" X OP " v[28:29], " SRC "\n" X OP " v[30:31], " SRC "\n" X
#define ACC16_X2(OP, SRC, X) \
    OP " v[0:1], " SRC "\n" OP " v[2:3], " SRC "\n" X OP " v[4:5], " SRC "\n" OP " v[6:7], " SRC "\n" X \
    OP " v[8:9], " SRC "\n" OP " v[10:11], " SRC "\n" X OP " v[12:13], " SRC "\n" OP " v[14:15], " SRC "\n" X \
    OP " v[16:17], " SRC "\n" OP " v[18:19], " SRC "\n" X OP " v[20:21], " SRC "\n" OP " v[22:23], " SRC "\n" X \
    OP " v[24:25], " SRC "\n" OP " v[26:27], " SRC "\n" X OP " v[28:29], " SRC "\n" OP " v[30:31], " SRC "\n" X
#define ACC16_X4(OP, SRC, X) \
    OP " v[0:1], " SRC "\n" OP " v[2:3], " SRC "\n" OP " v[4:5], " SRC "\n" OP " v[6:7], " SRC "\n" X \
    OP " v[8:9], " SRC "\n" OP " v[10:11], " SRC "\n" OP " v[12:13], " SRC "\n" OP " v[14:15], " SRC "\n" X \
    OP " v[16:17], " SRC "\n" OP " v[18:19], " SRC "\n" OP " v[20:21], " SRC "\n" OP " v[22:23], " SRC "\n" X \
    OP " v[24:25], " SRC "\n" OP " v[26:27], " SRC "\n" OP " v[28:29], " SRC "\n" OP " v[30:31], " SRC "\n" X

#define PROLOG \
    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0x3ff00000\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0x3e100000\n" \
    "v_mov_b32 v44, 0\n v_mov_b32 v45, 0x3e100000\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n" \
    "v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n" \
    "s_mov_b32 s22, 0\n s_mov_b32 s23, 0x3e100000\n s_mov_b32 s24, 0\n s_mov_b32 s25, 0x3ff00000\n s_mov_b32 s26, 0\n" \
    "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n" \
    "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n" \
    "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" \
    "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
#define LOOP(BODY) \
    "s_mov_b32 s20, %1\n" PROLOG "1:\n" BODY BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" \
    "v_add_f64 v[0:1], v[0:1], v[2:3]\n v_add_f64 v[0:1], v[0:1], v[30:31]\n v_add_f64 v[0:1], v[0:1], v[46:47]\n v_mov_b32 %0, v0\n"

#define KERNEL(NAME, BODY) \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters) { \
        __shared__ double sh[2048]; sh[threadIdx.x] = threadIdx.x; __syncthreads(); \
        float r; asm volatile(LOOP(BODY) : "=v"(r) : "s"(iters) : CLOB); out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)sh[(threadIdx.x * 7) & 2047]; }

KERNEL(k_fmac_vv, ACC16("v_fmac_f64", "v[40:41], v[42:43]"))                 // D += A*B, A and B in different banks
KERNEL(k_fmac_vv_same, ACC16("v_fmac_f64", "v[40:41], v[44:45]"))            // A and B in the same banks
KERNEL(k_fmac_sv, ACC16("v_fmac_f64", "s[22:23], v[42:43]"))                 // scalar multiplicand
KERNEL(k_fma_vvv, ACC16("v_fma_f64", "v[40:41], v[42:43], v[50:51]"))        // three-address form, no accumulate dependency
KERNEL(k_mul_vv, ACC16("v_mul_f64", "v[40:41], v[42:43]"))
KERNEL(k_mul_sv, ACC16("v_mul_f64", "s[22:23], v[42:43]"))
KERNEL(k_add_vv, ACC16("v_add_f64", "v[40:41], v[42:43]"))
KERNEL(k_fmac_sv_salu1, ACC16_X1("v_fmac_f64", "s[22:23], v[42:43]", "s_add_u32 s26, s26, 1\n"))
KERNEL(k_fmac_sv_vint1, ACC16_X1("v_fmac_f64", "s[22:23], v[42:43]", "v_add_u32 v46, v46, v48\n"))
KERNEL(k_fmac_sv_vint2, ACC16_X2("v_fmac_f64", "s[22:23], v[42:43]", "v_add_u32 v46, v46, v48\n"))
KERNEL(k_fmac_sv_vint4, ACC16_X4("v_fmac_f64", "s[22:23], v[42:43]", "v_add_u32 v46, v46, v48\n"))
KERNEL(k_fmac_sv_mov64_2, ACC16_X2("v_fmac_f64", "s[22:23], v[42:43]", "v_mov_b64 v[52:53], v[50:51]\n"))
KERNEL(k_fmac_sv_ldsr4, ACC16_X4("v_fmac_f64", "s[22:23], v[42:43]", "ds_read_b64 v[52:53], v48\n"))
KERNEL(k_fmac_sv_ldsr128_4, ACC16_X4("v_fmac_f64", "s[22:23], v[42:43]", "ds_read_b128 v[52:55], v48\n"))
KERNEL(k_fmac_sv_ldsw4, ACC16_X4("v_fmac_f64", "s[22:23], v[42:43]", "ds_write_b64 v48, v[50:51]\n"))

// round 6: the DP-ALU form of DPP (row_newbcast: lane N of each row of 16 lanes as the first multiplicand)
#define DPPSRC "v[40:41], v[42:43] row_newbcast:3 row_mask:0xf bank_mask:0xf"
KERNEL(k_fmac_dpp, ACC16("v_fmac_f64_dpp", DPPSRC))
KERNEL(k_fmac_dpp_ldsr4, ACC16_X4("v_fmac_f64_dpp", DPPSRC, "ds_read_b64 v[52:53], v48\n"))

// integer-only and mixed reference loops written separately (32-bit destinations)
#define INT16(X) X X X X X X X X X X X X X X X X
__global__ void __launch_bounds__(1024) k_vint(float *out, int iters)
{
    float r;
    asm volatile("s_mov_b32 s20, %1\n v_mov_b32 v46, 0\n v_mov_b32 v48, 1\n v_mov_b32 v47, 0\n 1:\n"
                 INT16("v_add_u32 v46, v46, v48\n v_add_u32 v47, v47, v48\n")
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n v_mov_b32 %0, v46\n" : "=v"(r) : "s"(iters) : CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*kern_t)(float *, int);
struct Pat { const char *name; kern_t k; int fp64_per_iter; };

static float time_ms(kern_t k, int grid, int block, float *out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main()
{
    float *out; hipMalloc(&out, 1 << 26);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz nominal; cycles per FP64 instruction and SIMD (32 per iteration, 16 accumulators)\n", p.name, cus, ghz);
    const Pat pats[] = {
        {"v_fmac_f64 v, v, v   (A, B different banks)", k_fmac_vv, 32}, {"v_fmac_f64 v, v, v   (A, B same banks)", k_fmac_vv_same, 32},
        {"v_fmac_f64 v, s, v", k_fmac_sv, 32}, {"v_fma_f64  v, v, v, v (no acc dependency)", k_fma_vvv, 32},
        {"v_mul_f64  v, v, v", k_mul_vv, 32}, {"v_mul_f64  v, s, v", k_mul_sv, 32}, {"v_add_f64  v, v, v", k_add_vv, 32},
        {"fmac s,v + 1 s_add_u32 per fmac", k_fmac_sv_salu1, 32}, {"fmac s,v + 1 v_add_u32 per fmac", k_fmac_sv_vint1, 32},
        {"fmac s,v + 1 v_add_u32 per 2 fmac", k_fmac_sv_vint2, 32}, {"fmac s,v + 1 v_add_u32 per 4 fmac", k_fmac_sv_vint4, 32},
        {"fmac s,v + 1 v_mov_b64 per 2 fmac", k_fmac_sv_mov64_2, 32}, {"fmac s,v + 1 ds_read_b64 per 4 fmac", k_fmac_sv_ldsr4, 32},
        {"fmac s,v + 1 ds_read_b128 per 4 fmac", k_fmac_sv_ldsr128_4, 32}, {"fmac s,v + 1 ds_write_b64 per 4 fmac", k_fmac_sv_ldsw4, 32},
        {"v_fmac_f64_dpp v, v, v row_newbcast", k_fmac_dpp, 32}, {"fmac_dpp + 1 ds_read_b64 per 4 fmac", k_fmac_dpp_ldsr4, 32},
        {"v_add_u32 only (32 per iteration)", k_vint, 32},
    };
    const int iters = 4000;
    printf("%-46s", "pattern \\ waves per SIMD");
    for (int w = 1; w <= 4; ++w) printf("%9d", w);
    printf("\n");
    for (const Pat &pt : pats) {
        printf("%-46s", pt.name);
        for (int w = 1; w <= 4; ++w) {
            const float ms = time_ms(pt.k, cus, 256 * w, out, iters);
            printf("%9.2f", ms * 1e-3 * ghz * 1e9 / ((double)iters * pt.fp64_per_iter * w));
        }
        printf("\n");
    }
    return 0;
}

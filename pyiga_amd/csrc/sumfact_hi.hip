// Degrees 6 and 7 of the sum-factorised stage kernels (P = 7, 8 functions per span): the same templates as sumfact.hip, in a
// translation unit of their own so that the two halves of the instantiations compile side by side.
#include "sumfact_stages.h"

namespace igx {

void stageA_hi(int P, hipStream_t st, const StageAArgs &A, bool qeq, bool sym, bool one, bool short_chunks, dim3 grid, dim3 block, size_t lds)
{
    if (P == 7) launch_stageA<7>(st, A, qeq, sym, one, short_chunks, grid, block, lds);
    else launch_stageA<8>(st, A, qeq, sym, one, short_chunks, grid, block, lds);
}

void stageB_hi(int P, hipStream_t st, const double *K1, double *K2, const StageBArgs &B, bool qeq, dim3 grid, dim3 block, size_t lds)
{
    if (P == 7) launch_stageB<7>(st, K1, K2, B, qeq, grid, block, lds);
    else launch_stageB<8>(st, K1, K2, B, qeq, grid, block, lds);
}

int final_hi(int P, hipStream_t st, const double *K, double *data, const FinalArgs &F, int ny, bool fast, int kpy, dim3 grid, dim3 block, size_t lds)
{
    if (P == 7) return launch_final<7>(st, K, data, F, ny, fast, kpy, grid, block, lds);
    return launch_final<8>(st, K, data, F, ny, fast, kpy, grid, block, lds);
}

} // namespace igx

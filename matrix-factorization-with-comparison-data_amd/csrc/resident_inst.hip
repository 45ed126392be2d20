// resident_inst.hip — instantiations of the resident kernel for ONE factor width (-DMFCD_RES_D=<d>), so the
// widths build in parallel.  Slice sizes: Q in {1, 2, 4, 16} registers per array (64..1024 elements per wave).
#include "resident_kernel.h"

#ifndef MFCD_RES_D
#error "compile with -DMFCD_RES_D=<factor width>"
#endif
#define MFCD_CAT2(a, b) a##b
#define MFCD_CAT(a, b) MFCD_CAT2(a, b)

namespace {

template <int D, int Q>
bool launch_q(const mfcd_detail::ResidentArgs &a, int blocks, hipStream_t st)
{
    if constexpr ((64 * Q) % D == 0) {
        const bool look = a.B <= 64 && a.lookahead > 0;
#define MFCD_LAUNCH(L, F) \
    hipLaunchKernelGGL((mfcd_detail::resident_train_kernel<D, Q, L, F>), dim3(blocks), dim3(256), a.lds_pad, st, a)
        if (look && a.lookahead >= 8 && a.fast_math) MFCD_LAUNCH(8, true);
        else if (look && a.lookahead >= 8) MFCD_LAUNCH(8, false);
        else if (look && a.fast_math) MFCD_LAUNCH(4, true);
        else if (look) MFCD_LAUNCH(4, false);
        else if (a.fast_math) MFCD_LAUNCH(0, true);
        else MFCD_LAUNCH(0, false);
#undef MFCD_LAUNCH
        return true;
    } else {
        return false;
    }
}

// workgroups per CU the runtime admits for the instantiation the launch above would pick (0: query failed)
template <int D, int Q>
int occupancy_q(int look, int fast, int lds_pad)
{
    if constexpr ((64 * Q) % D == 0) {
        int nb = 0;
        hipError_t e;
#define MFCD_OCC(L, F)                                                                                         \
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfcd_detail::resident_train_kernel<D, Q, L, F>, 256, \
                                                     (size_t)lds_pad)
        if (look >= 8 && fast) MFCD_OCC(8, true);
        else if (look >= 8) MFCD_OCC(8, false);
        else if (look > 0 && fast) MFCD_OCC(4, true);
        else if (look > 0) MFCD_OCC(4, false);
        else if (fast) MFCD_OCC(0, true);
        else MFCD_OCC(0, false);
#undef MFCD_OCC
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        return nb;
    } else {
        return 0;
    }
}

}  // namespace

extern "C" int MFCD_CAT(mfcd_resident_occupancy_d, MFCD_RES_D)(int Q, int look, int fast, int lds_pad)
{
    switch (Q) {
        case 1: return occupancy_q<MFCD_RES_D, 1>(look, fast, lds_pad);
        case 2: return occupancy_q<MFCD_RES_D, 2>(look, fast, lds_pad);
        case 4: return occupancy_q<MFCD_RES_D, 4>(look, fast, lds_pad);
        case 16: return occupancy_q<MFCD_RES_D, 16>(look, fast, lds_pad);
        default: return 0;
    }
}

extern "C" int MFCD_CAT(mfcd_resident_launch_d, MFCD_RES_D)(const mfcd_detail::ResidentArgs *a, int Q, int blocks,
                                                           void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    bool ok = false;
    switch (Q) {
        case 1: ok = launch_q<MFCD_RES_D, 1>(*a, blocks, st); break;
        case 2: ok = launch_q<MFCD_RES_D, 2>(*a, blocks, st); break;
        case 4: ok = launch_q<MFCD_RES_D, 4>(*a, blocks, st); break;
        case 16: ok = launch_q<MFCD_RES_D, 16>(*a, blocks, st); break;
        default: break;
    }
    if (!ok) return MFCD_EINVAL;
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

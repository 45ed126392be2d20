// resident_inst.hip — instantiations of the resident kernel for ONE factor width (-DMFCD_RES_D=<d>), so the
// widths build in parallel.  Slice sizes: Q in {1, 2, 4, 16, 32} registers per array (64..2048 elements per wave);
// fp32 and bf16 tables (BASELINE configs[2]), both Adam flavours.
#include "resident_kernel.h"

#ifndef MFCD_RES_D
#error "compile with -DMFCD_RES_D=<factor width>"
#endif
#define MFCD_CAT2(a, b) a##b
#define MFCD_CAT(a, b) MFCD_CAT2(a, b)

namespace {

// dynamic LDS of one workgroup: the four waves' row-gradient accumulators for Q >= 16, plus the tuning pad
template <int Q>
constexpr size_t slice_lds() { return Q >= 16 ? (size_t)4 * 64 * Q * sizeof(float) : 0; }

// applies `fn(kernel pointer)` to the instantiation a launch with these switches uses; false if there is none
template <int D, int Q, typename F>
bool with_kernel(int look, int fast, int bf16, F fn)
{
    if constexpr ((64 * Q) % D == 0) {
        using mfcd_detail::resident_train_kernel;
        // the look-ahead depth is a launch argument (ResidentArgs::lookahead): LOOK only says "look-ahead form"
        if (bf16) {
            if (look > 0 && fast) fn(resident_train_kernel<D, Q, 4, true, true>);
            else if (look > 0) fn(resident_train_kernel<D, Q, 4, false, true>);
            else if (fast) fn(resident_train_kernel<D, Q, 0, true, true>);
            else fn(resident_train_kernel<D, Q, 0, false, true>);
            return true;
        }
        if (look > 0 && fast) fn(resident_train_kernel<D, Q, 4, true, false>);
        else if (look > 0) fn(resident_train_kernel<D, Q, 4, false, false>);
        else if (fast) fn(resident_train_kernel<D, Q, 0, true, false>);
        else fn(resident_train_kernel<D, Q, 0, false, false>);
        return true;
    } else {
        return false;
    }
}

template <int D, int Q>
int launch_q(const mfcd_detail::ResidentArgs &a, int blocks, hipStream_t st)
{
    const int look = (a.B <= 64 && a.lookahead > 0) ? a.lookahead : 0;
    const size_t lds = slice_lds<Q>() + (size_t)a.lds_pad;
    hipError_t err = hipSuccess;
    const bool ok = with_kernel<D, Q>(look, a.fast_math, a.bf16, [&](auto kernel) {
        static size_t allowed = 48 * 1024;   // per instantiation: raise the dynamic-LDS limit only when needed
        if (lds > allowed) {
            err = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (err != hipSuccess) return;
            allowed = lds;
        }
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, st, a);
    });
    if (!ok) return MFCD_EINVAL;
    if (err != hipSuccess) return (int)err;
    return 0;
}

// workgroups per CU the runtime admits for the instantiation the launch above would pick (0: query failed / none)
template <int D, int Q>
int occupancy_q(int look, int fast, int bf16, int lds_pad)
{
    int nb = 0;
    hipError_t e = hipSuccess;
    const bool ok = with_kernel<D, Q>(look, fast, bf16, [&](auto kernel) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, slice_lds<Q>() + (size_t)lds_pad);
    });
    if (!ok) return 0;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}

}  // namespace

extern "C" int MFCD_CAT(mfcd_resident_occupancy_d, MFCD_RES_D)(int Q, int look, int fast, int bf16, int lds_pad)
{
    switch (Q) {
        case 1: return occupancy_q<MFCD_RES_D, 1>(look, fast, bf16, lds_pad);
        case 2: return occupancy_q<MFCD_RES_D, 2>(look, fast, bf16, lds_pad);
        case 4: return occupancy_q<MFCD_RES_D, 4>(look, fast, bf16, lds_pad);
        case 16: return occupancy_q<MFCD_RES_D, 16>(look, fast, bf16, lds_pad);
        case 32: return occupancy_q<MFCD_RES_D, 32>(look, fast, bf16, lds_pad);
        default: return 0;
    }
}

extern "C" int MFCD_CAT(mfcd_resident_launch_d, MFCD_RES_D)(const mfcd_detail::ResidentArgs *a, int Q, int blocks,
                                                           void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    int rc = MFCD_EINVAL;
    switch (Q) {
        case 1: rc = launch_q<MFCD_RES_D, 1>(*a, blocks, st); break;
        case 2: rc = launch_q<MFCD_RES_D, 2>(*a, blocks, st); break;
        case 4: rc = launch_q<MFCD_RES_D, 4>(*a, blocks, st); break;
        case 16: rc = launch_q<MFCD_RES_D, 16>(*a, blocks, st); break;
        case 32: rc = launch_q<MFCD_RES_D, 32>(*a, blocks, st); break;
        default: break;
    }
    if (rc) return rc;
    MFCD_HIP_TRY(hipGetLastError());
    return 0;
}

// api.hip — version and error strings of the C-ABI (include/mfcd.h).
#include "common.h"

extern "C" int mfcd_abi_version(void) { return MFCD_ABI_VERSION; }

extern "C" const char *mfcd_error_string(int code)
{
    switch (code) {
        case 0: return "success";
        case MFCD_EINVAL: return "mfcd: invalid argument";
        case MFCD_EWORKSPACE: return "mfcd: workspace too small";
        case MFCD_EALIGN: return "mfcd: table pointer not 4-byte aligned";
        case MFCD_EINDEX: return "mfcd: sample index out of range";
        case MFCD_ESTATE: return "mfcd: workspace not initialised for these sizes (mfcd_train_workspace_init)";
        case MFCD_ERCCL: return "mfcd: RCCL is not loadable in this process or an RCCL call failed";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "mfcd: unknown error";
}

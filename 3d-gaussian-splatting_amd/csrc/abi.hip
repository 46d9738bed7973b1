// abi.hip -- version and error-text entry points of the C ABI (include/gsplat_amd.h).
#include "gsa_common.h"

extern "C" int gsa_abi_version(void) { return GSA_ABI_VERSION; }

extern "C" const char *gsa_error_string(int code) {
    switch (code) {
        case GSA_OK: return "ok";
        case GSA_ERR_INVALID_ARGUMENT: return "gsplat_amd: invalid argument";
        case GSA_ERR_WORKSPACE_TOO_SMALL: return "gsplat_amd: workspace too small";
        case GSA_ERR_UNSUPPORTED: return "gsplat_amd: unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "gsplat_amd: unknown error";
}

// xfm_capi.hip -- ABI bookkeeping shared by all entry points of libxfm_hip.so.
#include "xfm_common.hpp"

namespace xfm {

static thread_local hipError_t g_last = hipSuccess;

void set_last_hip_error(hipError_t e) { g_last = e; }

int check_launch() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last = e;
        return XFM_ELAUNCH;
    }
    return XFM_OK;
}

}  // namespace xfm

extern "C" {

int xfm_abi_version(void) { return XFM_ABI_VERSION; }

const char *xfm_strerror(int code) {
    switch (code) {
        case XFM_OK: return "ok";
        case XFM_EINVAL: return "invalid argument (null pointer, non-positive size, dim % n_groups != 0, or missing x for a multi-chunk scan)";
        case XFM_EDTYPE: return "unsupported dtype combination (u/delta/B/C must share fp32|fp16|bf16; out is fp32 or the input dtype)";
        case XFM_ELIMIT: return "shape outside kernel limits (dstate > 256, or plane too large for LDS)";
        case XFM_ELAUNCH: return "HIP kernel launch failed (see xfm_last_hip_error)";
    }
    return "unknown error";
}

const char *xfm_last_hip_error(void) { return hipGetErrorString(xfm::g_last); }
}

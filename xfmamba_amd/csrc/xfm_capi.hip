// xfm_capi.hip -- ABI bookkeeping shared by all entry points of libxfm_hip.so.
#include "xfm_common.hpp"

namespace xfm {

static thread_local hipError_t g_last = hipSuccess;

void set_last_hip_error(hipError_t e) { g_last = e; }

// Profiling hook (xfm_prof_main_kernel): a pair of events the next launcher that knows about it records right around its MAIN
// kernel, so that a caller's per-kernel timer does not also span the small kernels an entry point launches behind it.
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

void prof_before_main(hipStream_t s) {
    if (g_prof_start) (void)hipEventRecord(g_prof_start, s);
}
void prof_after_main(hipStream_t s) {
    if (g_prof_stop) (void)hipEventRecord(g_prof_stop, s);
    g_prof_start = g_prof_stop = nullptr;
}

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

int xfm_prof_main_kernel(void *start_event, void *stop_event) {
    const int pending = xfm::g_prof_start != nullptr || xfm::g_prof_stop != nullptr;
    xfm::g_prof_start = (hipEvent_t)start_event;
    xfm::g_prof_stop = (hipEvent_t)stop_event;
    return pending;
}
}

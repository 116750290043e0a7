// ss2d_fused.hip -- placeholder until the fused SS2D core lands.
#include "xfm_common.hpp"
extern "C" {
int xfm_ss2d_fwd(const xfm_ss2d_params_t *, void *) { return XFM_ELIMIT; }
int xfm_ss2d_bwd(const xfm_ss2d_params_t *, void *) { return XFM_ELIMIT; }
}

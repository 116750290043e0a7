// bf16-in / fp32-out instantiations of the fused SS2D kernels (one translation unit per input dtype: parallel builds).
#include "ss2d_kernels.hpp"
namespace xfm {
template <> int ss2d_dispatch<bf16_t, float>(const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s) {
    return ss2d_dispatch_impl<bf16_t, float>(a, pl, bwd, s);
}
}  // namespace xfm

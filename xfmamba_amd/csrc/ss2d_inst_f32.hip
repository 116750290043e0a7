// fp32-I/O instantiations of the fused SS2D kernels (one translation unit per input dtype: parallel builds).
#include "ss2d_kernels.hpp"
namespace xfm {
template <> int ss2d_dispatch<float, float>(const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s) {
    return ss2d_dispatch_impl<float, float>(a, pl, bwd, s);
}
}  // namespace xfm

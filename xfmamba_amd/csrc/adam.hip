// adam.hip -- the optimizer step of the reference training loop (torch.optim.Adam(lr=1e-4, weight_decay=1e-5),
// 1_train_model.py:141; libs/training.py:195) for ALL parameters in one multi-tensor launch, fused with the refresh of the
// bf16 weight shadows the mixed-precision forward reads (xfmamba_amd/amp.py): one pass over p, g, m, v per step instead of
// the library's multi-tensor Adam followed by a multi-tensor cast.
//   g' = g + wd * p;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (exactly torch.optim.Adam's update)
// The step count t lives on the device (graph-capturable); HBM-bound: 4 fp32 reads + 3 fp32 writes (+ 2 B) per element.
// Data-parallel runs hand in the all-reduced SUM of the ranks' gradients straight from the bf16 wire bucket (numel entry
// with bit 62 set: the gradient pointer is bf16) and grad_scale = 1 / world: no widening copy, no separate averaging pass.
#include "xfm_common.hpp"

namespace xfm {

struct AdamArgs {
    const int64_t *p, *g, *m, *v, *shadow;   // device arrays [ntensors] of device pointers (shadow entries may be 0)
    const int64_t *numel;                    // [ntensors]
    const int64_t *chunks;                   // [nchunks]: tensor index | (chunk index inside the tensor << 32)
    float *step;                             // device scalar, incremented by this launch (block 0)
    float lr, b1, b2, eps, wd;
    float gscale;                            // gradient multiplier (1 / world for summed data-parallel gradients)
    int chunk;
};

constexpr int64_t kAdamBf16Grad = (int64_t)1 << 62;   // flag in a numel entry

__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamArgs a) {
    const int64_t ce = a.chunks[blockIdx.x];
    const int ti = (int)(ce & 0xffffffff);
    const int64_t off = (ce >> 32) * a.chunk;
    const int64_t nraw = a.numel[ti];
    const int64_t n = nraw & ~kAdamBf16Grad;
    const bool g16 = (nraw & kAdamBf16Grad) != 0;
    float *p = reinterpret_cast<float *>(a.p[ti]);
    const float *g = reinterpret_cast<const float *>(a.g[ti]);
    const uint16_t *gh = reinterpret_cast<const uint16_t *>(a.g[ti]);
    float *m = reinterpret_cast<float *>(a.m[ti]);
    float *v = reinterpret_cast<float *>(a.v[ti]);
    uint16_t *sh = reinterpret_cast<uint16_t *>(a.shadow[ti]);
    const float t = *a.step + 1.f;                                  // every block reads the pre-increment value
    const float bc1 = 1.f - __builtin_amdgcn_exp2f(t * __builtin_amdgcn_logf(a.b1));
    const float bc2 = 1.f - __builtin_amdgcn_exp2f(t * __builtin_amdgcn_logf(a.b2));
    const float step_size = a.lr / bc1, rs2 = __builtin_amdgcn_rsqf(bc2);
    const int64_t end = min(n, off + a.chunk);
    const bool vec = ((n & 3) == 0) && (((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0 &&
                     ((uintptr_t)g & (g16 ? 7 : 15)) == 0;
    if (vec) {
        for (int64_t i = off + 4 * threadIdx.x; i < end; i += 4 * 256) {
            float4 pv = *reinterpret_cast<const float4 *>(p + i), gv;
            if (g16) {
                const uint2 gq = *reinterpret_cast<const uint2 *>(gh + i);
                gv = make_float4(__uint_as_float(gq.x << 16), __uint_as_float(gq.x & 0xffff0000u), __uint_as_float(gq.y << 16),
                                 __uint_as_float(gq.y & 0xffff0000u));
            } else {
                gv = *reinterpret_cast<const float4 *>(g + i);
            }
            float4 mv = *reinterpret_cast<const float4 *>(m + i), vv = *reinterpret_cast<const float4 *>(v + i);
            float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
            float mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gq = fmaf(a.wd, pp[q], gg[q] * a.gscale);
                mm[q] = fmaf(a.b1, mm[q], (1.f - a.b1) * gq);
                vq[q] = fmaf(a.b2, vq[q], (1.f - a.b2) * gq * gq);
                pp[q] -= step_size * mm[q] / (sqrtf(vq[q]) * rs2 + a.eps);
            }
            *reinterpret_cast<float4 *>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *reinterpret_cast<float4 *>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *reinterpret_cast<float4 *>(v + i) = make_float4(vq[0], vq[1], vq[2], vq[3]);
            if (sh) {
                uint2 s2 = make_uint2(pack_bf16x2(pp[0], pp[1]), pack_bf16x2(pp[2], pp[3]));
                *reinterpret_cast<uint2 *>(sh + i) = s2;
            }
        }
    } else {
        for (int64_t i = off + threadIdx.x; i < end; i += 256) {
            const float gi = g16 ? __uint_as_float((uint32_t)gh[i] << 16) : g[i];
            const float gq = fmaf(a.wd, p[i], gi * a.gscale);
            const float mq = fmaf(a.b1, m[i], (1.f - a.b1) * gq);
            const float vq = fmaf(a.b2, v[i], (1.f - a.b2) * gq * gq);
            const float pq = p[i] - step_size * mq / (sqrtf(vq) * rs2 + a.eps);
            p[i] = pq; m[i] = mq; v[i] = vq;
            if (sh) sh[i] = (uint16_t)(pack_bf16x2(pq, 0.f) & 0xffffu);
        }
    }
}

__global__ void adam_step_inc_kernel(float *step) { *step += 1.f; }

}  // namespace xfm

extern "C" int xfm_adam_multi_scaled(const void *p_ptrs, const void *g_ptrs, const void *m_ptrs, const void *v_ptrs,
                                     const void *shadow_ptrs, const void *numel, const void *chunks, int nchunks, int chunk,
                                     float *step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     float grad_scale, void *stream) {
    using namespace xfm;
    if (!p_ptrs || !g_ptrs || !m_ptrs || !v_ptrs || !shadow_ptrs || !numel || !chunks || !step) return XFM_EINVAL;
    if (nchunks <= 0 || chunk <= 0 || chunk % 1024) return XFM_EINVAL;
    AdamArgs a{};
    a.p = (const int64_t *)p_ptrs; a.g = (const int64_t *)g_ptrs; a.m = (const int64_t *)m_ptrs; a.v = (const int64_t *)v_ptrs;
    a.shadow = (const int64_t *)shadow_ptrs; a.numel = (const int64_t *)numel;
    a.chunks = (const int64_t *)chunks;
    a.step = step; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale; a.chunk = chunk;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, a);
    int rc = check_launch();
    if (rc != XFM_OK) return rc;
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, s, step);   // after every block has read the old count
    return check_launch();
}


extern "C" int xfm_adam_multi(const void *p_ptrs, const void *g_ptrs, const void *m_ptrs, const void *v_ptrs,
                              const void *shadow_ptrs, const void *numel, const void *chunks, int nchunks, int chunk, float *step,
                              float lr, float beta1, float beta2, float eps, float weight_decay, void *stream) {
    return xfm_adam_multi_scaled(p_ptrs, g_ptrs, m_ptrs, v_ptrs, shadow_ptrs, numel, chunks, nchunks, chunk, step, lr, beta1, beta2,
                                 eps, weight_decay, 1.f, stream);
}

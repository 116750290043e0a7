// cross_scan.hip -- CrossScan / CrossMerge (4-route 2-D unrolling) and the two-view channel swap.
//
// Semantics: models/csm_triton.py:22-85 (cross_scan_fwd / cross_merge_fwd, scans=0, channel-first)
// and models/fusion_vmamba.py:189-213 (SwappingScan_multiview.forward).  Pure data movement, so
// the design goal is that every HBM access has lanes along the contiguous axis: a workgroup
// stages a run of whole (b, c) planes in LDS (coalesced), and the column-major routes are read
// out of LDS with an odd row pitch instead of being gathered from HBM with a stride.
#include "xfm_common.hpp"

namespace xfm {

// LDS pitch for an H x W plane so that walking a column (stride = pitch) is bank-conflict free.
__host__ __device__ inline int odd_pitch(int w) { return w | 1; }

// x: (P, H, W) planes -> y: per batch (4, C, L).  One workgroup handles `pp` consecutive planes.
template <typename T>
__global__ void __launch_bounds__(256) cross_scan_kernel(const T *__restrict__ x, T *__restrict__ y, int planes, int C,
                                                         int H, int W, int pp) {
    extern __shared__ float smem[];
    const int L = H * W, pitch = odd_pitch(W), psz = H * pitch;
    const int p0 = blockIdx.x * pp;
    const int np = min(pp, planes - p0);
    const int total = np * L;
    // coalesced load of np contiguous planes into the padded LDS layout
    for (int e = threadIdx.x; e < total; e += 256) {
        const int pl = e / L, l = e - pl * L;
        const int h = l / W, w = l - h * W;
        smem[pl * psz + h * pitch + w] = ldf<T>(x + (int64_t)p0 * L + e);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 256) {
        const int pl = e / L, l = e - pl * L;
        const int plane = p0 + pl;
        const int b = plane / C, c = plane - b * C;
        T *yb = y + ((int64_t)b * 4 * C + c) * L;  // route k adds k*C*L
        const int64_t ks = (int64_t)C * L;
        // row-major value at l, and column-major value at l (= x[l % H][l / H])
        const int h0 = l / W, w0 = l - h0 * W;
        const float vr = smem[pl * psz + h0 * pitch + w0];
        const int w1 = l / H, h1 = l - w1 * H;
        const float vc = smem[pl * psz + h1 * pitch + w1];
        stf<T>(yb + l, vr);
        stf<T>(yb + ks + l, vc);
        stf<T>(yb + 2 * ks + (L - 1 - l), vr);
        stf<T>(yb + 3 * ks + (L - 1 - l), vc);
    }
}

// y: (B, 4, C, L) -> x: (B, C, L);  x[h*W+w] = y0[l] + y2[L-1-l] + y1[m] + y3[L-1-m], m = w*H+h.
template <typename Tin, typename Tout>
__global__ void __launch_bounds__(256) cross_merge_kernel(const Tin *__restrict__ y, Tout *__restrict__ x, int planes,
                                                          int C, int H, int W, int pp) {
    extern __shared__ float smem[];
    const int L = H * W, pitch = odd_pitch(H), psz = W * pitch;  // column-major planes: W rows of H
    const int p0 = blockIdx.x * pp;
    const int np = min(pp, planes - p0);
    const int total = np * L;
    const int64_t ks = (int64_t)C * L;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int pl = e / L, m = e - pl * L;
        const int plane = p0 + pl;
        const int b = plane / C, c = plane - b * C;
        const Tin *yb = y + ((int64_t)b * 4 * C + c) * L;
        const float v = ldf<Tin>(yb + ks + m) + ldf<Tin>(yb + 3 * ks + (L - 1 - m));
        const int w = m / H, h = m - w * H;
        smem[pl * psz + w * pitch + h] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 256) {
        const int pl = e / L, l = e - pl * L;
        const int plane = p0 + pl;
        const int b = plane / C, c = plane - b * C;
        const Tin *yb = y + ((int64_t)b * 4 * C + c) * L;
        const int h = l / W, w = l - h * W;
        const float v = ldf<Tin>(yb + l) + ldf<Tin>(yb + 2 * ks + (L - 1 - l)) + smem[pl * psz + w * pitch + h];
        stf<Tout>(x + (int64_t)plane * L + l, v);
    }
}

// out[b,0,c,:] = (c even ? x2 : x)[b,c,:],  out[b,1,c,:] = (c even ? x : x2)[b,c,:]
template <typename T>
__global__ void __launch_bounds__(256) swap_scan_kernel(const T *__restrict__ x, const T *__restrict__ x2,
                                                        T *__restrict__ out, int64_t total, int C, int L) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / L;  // b*C + c
        const int l = (int)(e - row * L);
        const int64_t b = row / C;
        const int c = (int)(row - b * C);
        const bool even = (c & 1) == 0;
        const T a = x[e], bb = x2[e];
        T *o = out + ((b * 2) * C + c) * (int64_t)L + l;
        o[0] = even ? bb : a;
        o[(int64_t)C * L] = even ? a : bb;
    }
}

static int planes_per_block(int L, int pitch_plane_floats, int planes) {
    // aim for >= 2048 elements per workgroup, stay under 48 KiB of LDS, keep the grid >= 1024 blocks when possible
    int pp = (2048 + L - 1) / L;
    if (pp < 1) pp = 1;
    while (pp > 1 && (size_t)pp * pitch_plane_floats * sizeof(float) > 48 * 1024) --pp;
    while (pp > 1 && (planes + pp - 1) / pp < 1024) --pp;
    return pp;
}

template <typename T>
static int launch_scan(const void *x, void *y, int B, int C, int H, int W, hipStream_t s) {
    const int planes = B * C;
    const int psz = H * odd_pitch(W);
    const int pp = planes_per_block(H * W, psz, planes);
    const size_t lds = (size_t)pp * psz * sizeof(float);
    if (lds > 160 * 1024) return XFM_ELIMIT;
    hipLaunchKernelGGL((cross_scan_kernel<T>), dim3((planes + pp - 1) / pp), dim3(256), lds, s, (const T *)x, (T *)y,
                       planes, C, H, W, pp);
    return check_launch();
}

template <typename Tin, typename Tout>
static int launch_merge(const void *y, void *x, int B, int C, int H, int W, hipStream_t s) {
    const int planes = B * C;
    const int psz = W * odd_pitch(H);
    const int pp = planes_per_block(H * W, psz, planes);
    const size_t lds = (size_t)pp * psz * sizeof(float);
    if (lds > 160 * 1024) return XFM_ELIMIT;
    hipLaunchKernelGGL((cross_merge_kernel<Tin, Tout>), dim3((planes + pp - 1) / pp), dim3(256), lds, s,
                       (const Tin *)y, (Tout *)x, planes, C, H, W, pp);
    return check_launch();
}

// ---- route split / merge of the x_proj output ------------------------------------------------------------------
// x_proj is evaluated once on the map in natural order (the projection commutes with the route permutations); this
// kernel hands its (B, 4, R+2N, H*W) result to the consumers in the layout contract of xfm_ss2d_fwd: three contiguous
// tensors dt-input (B,4,R,L), Bs (B,4,N,L), Cs (B,4,N,L), with the planes of the column routes (k odd) transposed to
// column-major.  One workgroup per (b, k, c) plane; transposition through a padded LDS tile.
template <typename Ts, typename Td>
__device__ __forceinline__ void plane_move(const Ts *src, Td *dst, int R0, int C0, bool transpose, float *tile) {
    const int L = R0 * C0;
    if (!transpose) {
        for (int e = threadIdx.x; e < L; e += blockDim.x) stf<Td>(dst + e, ldf<Ts>(src + e));
        return;
    }
    const int pitch = C0 + 1;
    for (int e = threadIdx.x; e < L; e += blockDim.x) {
        const int r = e / C0, c = e - r * C0;
        tile[r * pitch + c] = ldf<Ts>(src + e);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L; e += blockDim.x) {     // (R0 x C0) row-major -> (C0 x R0) row-major
        const int c = e / R0, r = e - c * R0;
        stf<Td>(dst + e, tile[r * pitch + c]);
    }
}

// Bs32 / Cs32 (optional): the same B / C rows once more as fp32 -- the wide-map scan BACKWARD of ss2d_w.hpp reads them without
// an unpack (its forward keeps the 16-bit rows: with five operand vectors per chunk row it is bound by the vector-memory path)
template <typename T>
__global__ void __launch_bounds__(256) route_split_kernel(const T *xd, T *xr, T *Bs, T *Cs, float *Bs32, float *Cs32, int R, int N,
                                                          int H, int W) {
    extern __shared__ float tile[];
    const int C2 = R + 2 * N, L = H * W;
    const int c = blockIdx.x % C2, bk = blockIdx.x / C2, k = bk & 3;
    const T *src = xd + (int64_t)blockIdx.x * L;
    const bool tr = (k & 1) != 0;
    if (c < R) {
        plane_move<T, T>(src, xr + ((int64_t)bk * R + c) * L, H, W, tr, tile);
        return;
    }
    const int64_t off = c < R + N ? ((int64_t)bk * N + (c - R)) * L : ((int64_t)bk * N + (c - R - N)) * L;
    T *dst = (c < R + N ? Bs : Cs) + off;
    if (!Bs32) {
        plane_move<T, T>(src, dst, H, W, tr, tile);
        return;
    }
    // both copies of the row in one pass over the source
    float *dst32 = (c < R + N ? Bs32 : Cs32) + off;
    if (!tr) {
        for (int e = threadIdx.x; e < L; e += blockDim.x) {
            const float v = ldf<T>(src + e);
            stf<T>(dst + e, v);
            dst32[e] = v;
        }
        return;
    }
    const int pitch = W + 1;
    for (int e = threadIdx.x; e < L; e += blockDim.x) {
        const int r = e / W, cc = e - r * W;
        tile[r * pitch + cc] = ldf<T>(src + e);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L; e += blockDim.x) {     // (H x W) row-major -> (W x H) row-major
        const int cc = e / H, r = e - cc * H;
        const float v = tile[r * pitch + cc];
        stf<T>(dst + e, v);
        dst32[e] = v;
    }
}

// adjoint: gradients arrive as dt-input grad (T), dBs / dCs (fp32 accumulators of the scan backward)
template <typename T>
__global__ void __launch_bounds__(256) route_merge_kernel(const T *dxr, const float *dBs, const float *dCs, T *dxd, int R,
                                                          int N, int H, int W) {
    extern __shared__ float tile[];
    const int C2 = R + 2 * N, L = H * W;
    const int c = blockIdx.x % C2, bk = blockIdx.x / C2, k = bk & 3;
    T *dst = dxd + (int64_t)blockIdx.x * L;
    const bool tr = (k & 1) != 0;
    if (c < R) plane_move<T, T>(dxr + ((int64_t)bk * R + c) * L, dst, W, H, tr, tile);
    else if (c < R + N) plane_move<float, T>(dBs + ((int64_t)bk * N + (c - R)) * L, dst, W, H, tr, tile);
    else plane_move<float, T>(dCs + ((int64_t)bk * N + (c - R - N)) * L, dst, W, H, tr, tile);
}

template <typename T>
static int launch_route(bool merge, const void *a, const void *b, const void *c, void *d, void *e, void *f, int B, int R,
                        int N, int H, int W, hipStream_t s, float *e32 = nullptr, float *f32 = nullptr) {
    const size_t lds = (size_t)(H > W ? H : W) * ((H > W ? W : H) + 1) * sizeof(float) + (size_t)(H + W) * sizeof(float);
    if (lds > 64 * 1024) return XFM_ELIMIT;
    const dim3 grid((unsigned)((int64_t)B * 4 * (R + 2 * N)));
    if (merge)
        hipLaunchKernelGGL((route_merge_kernel<T>), grid, dim3(256), lds, s, (const T *)a, (const float *)b,
                           (const float *)c, (T *)d, R, N, H, W);
    else
        hipLaunchKernelGGL((route_split_kernel<T>), grid, dim3(256), lds, s, (const T *)a, (T *)d, (T *)e, (T *)f, e32, f32, R, N, H, W);
    return check_launch();
}

}  // namespace xfm

extern "C" {

int xfm_ss2d_route_split(const void *xd, void *xr, void *Bs, void *Cs, int B, int R, int N, int H, int W, int dtype,
                         void *stream) {
    using namespace xfm;
    if (!xd || !xr || !Bs || !Cs || B <= 0 || R <= 0 || N <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return launch_route<float>(false, xd, nullptr, nullptr, xr, Bs, Cs, B, R, N, H, W, s);
        case XFM_F16: return launch_route<f16_t>(false, xd, nullptr, nullptr, xr, Bs, Cs, B, R, N, H, W, s);
        case XFM_BF16: return launch_route<bf16_t>(false, xd, nullptr, nullptr, xr, Bs, Cs, B, R, N, H, W, s);
    }
    return XFM_EDTYPE;
}

int xfm_ss2d_route_split_bc32(const void *xd, void *xr, void *Bs, void *Cs, float *Bs32, float *Cs32, int B, int R, int N, int H,
                              int W, int dtype, void *stream) {
    using namespace xfm;
    if (!xd || !xr || !Bs || !Cs || !Bs32 || !Cs32 || B <= 0 || R <= 0 || N <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return launch_route<float>(false, xd, nullptr, nullptr, xr, Bs, Cs, B, R, N, H, W, s, Bs32, Cs32);
        case XFM_F16: return launch_route<f16_t>(false, xd, nullptr, nullptr, xr, Bs, Cs, B, R, N, H, W, s, Bs32, Cs32);
        case XFM_BF16: return launch_route<bf16_t>(false, xd, nullptr, nullptr, xr, Bs, Cs, B, R, N, H, W, s, Bs32, Cs32);
    }
    return XFM_EDTYPE;
}

int xfm_ss2d_route_merge(const void *dxr, const float *dBs, const float *dCs, void *dxd, int B, int R, int N, int H,
                         int W, int dtype, void *stream) {
    using namespace xfm;
    if (!dxr || !dBs || !dCs || !dxd || B <= 0 || R <= 0 || N <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return launch_route<float>(true, dxr, dBs, dCs, dxd, nullptr, nullptr, B, R, N, H, W, s);
        case XFM_F16: return launch_route<f16_t>(true, dxr, dBs, dCs, dxd, nullptr, nullptr, B, R, N, H, W, s);
        case XFM_BF16: return launch_route<bf16_t>(true, dxr, dBs, dCs, dxd, nullptr, nullptr, B, R, N, H, W, s);
    }
    return XFM_EDTYPE;
}

int xfm_cross_scan(const void *x, void *y, int B, int C, int H, int W, int dtype, void *stream) {
    using namespace xfm;
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return launch_scan<float>(x, y, B, C, H, W, s);
        case XFM_F16: return launch_scan<f16_t>(x, y, B, C, H, W, s);
        case XFM_BF16: return launch_scan<bf16_t>(x, y, B, C, H, W, s);
    }
    return XFM_EDTYPE;
}

int xfm_cross_merge(const void *y, void *x, int B, int C, int H, int W, int in_dtype, int out_dtype, void *stream) {
    using namespace xfm;
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == XFM_F32 && out_dtype == XFM_F32) return launch_merge<float, float>(y, x, B, C, H, W, s);
    if (in_dtype == XFM_F32 && out_dtype == XFM_BF16) return launch_merge<float, bf16_t>(y, x, B, C, H, W, s);
    if (in_dtype == XFM_F32 && out_dtype == XFM_F16) return launch_merge<float, f16_t>(y, x, B, C, H, W, s);
    if (in_dtype == XFM_BF16 && out_dtype == XFM_BF16) return launch_merge<bf16_t, bf16_t>(y, x, B, C, H, W, s);
    if (in_dtype == XFM_F16 && out_dtype == XFM_F16) return launch_merge<f16_t, f16_t>(y, x, B, C, H, W, s);
    if (in_dtype == XFM_BF16 && out_dtype == XFM_F32) return launch_merge<bf16_t, float>(y, x, B, C, H, W, s);
    if (in_dtype == XFM_F16 && out_dtype == XFM_F32) return launch_merge<f16_t, float>(y, x, B, C, H, W, s);
    return XFM_EDTYPE;
}

int xfm_swap_scan(const void *x, const void *x2, void *out, int B, int C, int L, int dtype, void *stream) {
    using namespace xfm;
    if (!x || !x2 || !out || B <= 0 || C <= 0 || L <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * C * L;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 8);
    switch (dtype) {
        case XFM_F32:
            hipLaunchKernelGGL((swap_scan_kernel<float>), dim3(grid), dim3(256), 0, s, (const float *)x,
                               (const float *)x2, (float *)out, total, C, L);
            break;
        case XFM_F16:
        case XFM_BF16:
            hipLaunchKernelGGL((swap_scan_kernel<uint16_t>), dim3(grid), dim3(256), 0, s, (const uint16_t *)x,
                               (const uint16_t *)x2, (uint16_t *)out, total, C, L);
            break;
        default: return XFM_EDTYPE;
    }
    return check_launch();
}
}

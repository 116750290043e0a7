// rowscan.hpp -- selective scan for SHORT rows (the 7x7 maps of XFMamba: stage 3 of the trunk and both
// fusion blocks, L = 49, d_state 16 or 1).  Same operator boundary as selective_scan.hip.
//
// For L <= 64 a parallel scan along the row wastes the machine: chunks of a few elements, 2*log2(64)
// cross-lane steps per state, and (in the backward) per-element reductions.  Here ONE LANE OWNS ONE
// ROW: a wavefront takes 64 consecutive rows of one (batch, group) -- they share B and C -- stages
// their operands in LDS with coalesced loads (row pitch L = 49 is odd: conflict-free per-lane walks),
// and every lane runs the recurrence sequentially with the state in registers.  No scan, no shuffles.
//   forward : t outer, n inner, h[n] in registers.
//   backward: n outer; a forward pass keeps h_t (L registers), the reverse pass folds dh into per-t
//             accumulators; dB/dC (a sum over the 64 rows for every (n,t)) go through an LDS transpose
//             so that lane t sums its column and the atomics of a wave are one contiguous run.
#pragma once

#include "scan_core.hpp"

namespace xfm {

struct RowScanArgs {
    xfm_scan_params_t p;
    int dim_per_group;
};

template <int LT> constexpr int rs_pad() { return (LT + 3) & ~3; }

// states processed together by the backward kernel, and the LDS floats its staging / transpose tiles need
template <int NS> struct RowScanNI { static constexpr int value = 1; };   // 2 spills (343 live floats + temporaries > 512 VGPRs)
template <int LT, int NS> constexpr int rs_bwd_tile() {
    return (64 * LT > RowScanNI<NS>::value * LT * 65) ? 64 * LT : RowScanNI<NS>::value * LT * 65;
}

// stage one (64 x LT) operand tile: rows r0..r0+63 of tensor `src` -> flat LDS [row*LT + t]
template <typename T, typename S, int LT>
__device__ __forceinline__ void rs_stage(T *dst, const S *src, int64_t batch_off, int64_t row_stride, int r0, int lane) {
    for (int e = lane; e < 64 * LT; e += 64) {
        const int row = e / LT, t = e - row * LT;
        const float v = ldf<S>(src + batch_off + (int64_t)(r0 + row) * row_stride + t);
        if constexpr (sizeof(T) == 4) dst[e] = v;
        else stf<T>(dst + e, v);
    }
}

template <typename Tin, typename Tout, int LT, int NS>
__global__ void __launch_bounds__(64) rowscan_fwd_kernel(const RowScanArgs a) {
    constexpr int LP = rs_pad<LT>();
    extern __shared__ float smem[];
    const xfm_scan_params_t &p = a.p;
    const int lane = threadIdx.x;
    const int tiles_pb = p.dim / 64;
    const int b = blockIdx.x / tiles_pb, r0 = (blockIdx.x - b * tiles_pb) * 64;
    const int k = r0 / a.dim_per_group, r = r0 + lane;
    float *dy = smem;                                   // [64][LT] delta in, y out (in place, lane-private slots)
    float *Bt = dy + 64 * LT;                           // [NS][LP]
    float *Ct = Bt + NS * LP;
    Tin *ut = reinterpret_cast<Tin *>(Ct + NS * LP);    // [64][LT]
    rs_stage<float, Tin, LT>(dy, (const Tin *)p.delta, (int64_t)b * p.delta_batch_stride, p.delta_d_stride, r0, lane);
    rs_stage<Tin, Tin, LT>(ut, (const Tin *)p.u, (int64_t)b * p.u_batch_stride, p.u_d_stride, r0, lane);
    const Tin *Bg = (const Tin *)p.B + (int64_t)b * p.B_batch_stride + (int64_t)k * p.B_group_stride;
    const Tin *Cg = (const Tin *)p.C + (int64_t)b * p.C_batch_stride + (int64_t)k * p.C_group_stride;
    for (int e = lane; e < NS * LT; e += 64) {
        const int n = e / LT, t = e - n * LT;
        Bt[n * LP + t] = ldf<Tin>(Bg + (int64_t)n * p.B_dstate_stride + t);
        Ct[n * LP + t] = ldf<Tin>(Cg + (int64_t)n * p.C_dstate_stride + t);
    }
    float A2[NS], h[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        A2[n] = p.A[(int64_t)r * p.A_d_stride + n] * kLog2e;
        h[n] = 0.f;
    }
    const float Dr = p.D ? p.D[r] : 0.f, bias = p.delta_bias ? p.delta_bias[r] : 0.f;
    wave_sync();
#pragma unroll 1
    for (int t = 0; t < LT; ++t) {
        float dl = dy[lane * LT + t] + bias;
        if (p.delta_softplus) dl = softplus20(dl);
        const float uu = ldf<Tin>(ut + lane * LT + t);
        const float du = dl * uu;
        float y = Dr * uu;
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const float av = exp2_fast(dl * A2[n]);
            h[n] = fmaf(av, h[n], du * Bt[n * LP + t]);       // B, C reads are wave-uniform: LDS broadcast
            y = fmaf(Ct[n * LP + t], h[n], y);
        }
        dy[lane * LT + t] = y;
    }
    wave_sync();
    Tout *ob = (Tout *)p.out + (int64_t)b * p.out_batch_stride;
    for (int e = lane; e < 64 * LT; e += 64) {
        const int row = e / LT, t = e - row * LT;
        stf<Tout>(ob + (int64_t)(r0 + row) * p.out_d_stride + t, dy[e]);
    }
}

template <typename Tin, typename Tout, int LT, int NS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))   // 1 wave/SIMD: whole 512-register file
rowscan_bwd_kernel(const RowScanArgs a) {
    constexpr int LP = rs_pad<LT>();
    extern __shared__ float smem[];
    const xfm_scan_params_t &p = a.p;
    const int lane = threadIdx.x;
    const int tiles_pb = p.dim / 64;
    const int b = blockIdx.x / tiles_pb, r0 = (blockIdx.x - b * tiles_pb) * 64;
    const int k = r0 / a.dim_per_group, r = r0 + lane;
    const int L = LT;
    // LDS: one (64 x LT) fp32 tile used for staging / transposes / outputs, plus B and C of the group.
    // The row's own operands live in REGISTERS for the whole state loop (a wave is alone on its SIMD anyway:
    // the tile keeps occupancy LDS-bound, so the 512-register file is free to use).
    float *sc = smem;                                   // [64][LT] staging, or NI transposed tiles [LT][65]
    float *Bt = sc + rs_bwd_tile<LT, NS>();             // [NS][LP]
    float *Ct = Bt + NS * LP;
    Tin *ut = reinterpret_cast<Tin *>(Ct + NS * LP);    // [64][LT] u in its storage dtype: read per step, not held
    const Tin *Bg = (const Tin *)p.B + (int64_t)b * p.B_batch_stride + (int64_t)k * p.B_group_stride;
    const Tin *Cg = (const Tin *)p.C + (int64_t)b * p.C_batch_stride + (int64_t)k * p.C_group_stride;
    for (int e = lane; e < NS * LT; e += 64) {
        const int n = e / LT, t = e - n * LT;
        Bt[n * LP + t] = ldf<Tin>(Bg + (int64_t)n * p.B_dstate_stride + t);
        Ct[n * LP + t] = ldf<Tin>(Cg + (int64_t)n * p.C_dstate_stride + t);
    }
    rs_stage<Tin, Tin, LT>(ut, (const Tin *)p.u, (int64_t)b * p.u_batch_stride, p.u_d_stride, r0, lane);
    const Tin *ur = ut + lane * LT;
    const float Dr = p.D ? p.D[r] : 0.f, bias = p.delta_bias ? p.delta_bias[r] : 0.f;
    float dl[LT], g[LT];                                // delta' , dout  of this lane's row
    rs_stage<float, Tin, LT>(sc, (const Tin *)p.delta, (int64_t)b * p.delta_batch_stride, p.delta_d_stride, r0, lane);
    wave_sync();
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        float v = sc[lane * LT + t] + bias;
        if (p.delta_softplus) v = softplus20(v);
        dl[t] = v;
    }
    wave_sync();
    rs_stage<float, Tout, LT>(sc, (const Tout *)p.dout, (int64_t)b * p.dout_batch_stride, p.dout_d_stride, r0, lane);
    wave_sync();
    float dD = 0.f;
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        g[t] = sc[lane * LT + t];
        dD = fmaf(g[t], ldf<Tin>(ur + t), dD);
    }
    wave_sync();
    float s1[LT], s2[LT];
#pragma unroll
    for (int t = 0; t < LT; ++t) s1[t] = s2[t] = 0.f;
    float *dBg = p.dB + ((int64_t)b * p.n_groups + k) * NS * L;
    float *dCg = p.dC + ((int64_t)b * p.n_groups + k) * NS * L;
    // s1 accumulates sum_n dh*B, s2 sum_n dh*A*(a*h_prev)
    // NI states run interleaved: their recurrences are independent dependency chains, which is the only
    // latency hiding a wave that is alone on its SIMD gets.  Tiles T[j] ([LT][TP], TP = 65: conflict-free both for
    // the per-row writes and for the per-column sums) transpose the dB / dC contributions of the 64 rows.
    constexpr int NI = RowScanNI<NS>::value;
    constexpr int TP = 65;
#pragma unroll 1
    for (int n = 0; n < NS; n += NI) {
        float An[NI], A2[NI], hp[NI], E[NI], dA[NI];
        float h[NI][LT];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            An[j] = p.A[(int64_t)r * p.A_d_stride + n + j];
            A2[j] = An[j] * kLog2e;
            hp[j] = E[j] = dA[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            const float du = dl[t] * ldf<Tin>(ur + t);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                hp[j] = fmaf(exp2_fast(dl[t] * A2[j]), hp[j], du * Bt[(n + j) * LP + t]);
                h[j][t] = hp[j];
            }
            if (t % 7 == 6) __builtin_amdgcn_sched_barrier(0);      // stop the scheduler from preloading all of B/C
        }
#pragma unroll
        for (int t = LT - 1; t >= 0; --t) {
            const float du = dl[t] * ldf<Tin>(ur + t);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float Bv = Bt[(n + j) * LP + t];
                float ea = dl[t] * A2[j];
                asm volatile("" : "+v"(ea));                       // recompute a_t: keeping 49*NI of them alive spills
                const float av = exp2_fast(ea);
                const float dh = fmaf(Ct[(n + j) * LP + t], g[t], E[j]);
                E[j] = av * dh;
                const float ah = h[j][t] - du * Bv;                 // a_t * h_{t-1}
                s1[t] = fmaf(dh, Bv, s1[t]);
                s2[t] = fmaf(dh * An[j], ah, s2[t]);
                dA[j] = fmaf(dh * dl[t], ah, dA[j]);
                sc[(j * LT + t) * TP + lane] = dh * du;              // dB contribution of this row, transposed
                h[j][t] = g[t] * h[j][t];                            // dC contribution (h_t is dead after this)
            }
            if (t % 7 == 0) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) atomicAdd(p.dA + (int64_t)r * NS + n + j, dA[j]);   // summed over the batch
        wave_sync();
        if (lane < LT) {                                             // lane t sums column t over the 64 rows
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float *col = sc + (j * LT + lane) * TP;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 1
                for (int q0 = 0; q0 < 64; q0 += 16) {                // 16 reads in flight: keeps the register peak low
#pragma unroll
                    for (int q = q0; q < q0 + 16; q += 4) {
                        a0 += col[q];
                        a1 += col[q + 1];
                        a2 += col[q + 2];
                        a3 += col[q + 3];
                    }
                }
                atomicAdd(dBg + (n + j) * L + lane, (a0 + a1) + (a2 + a3));
            }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int t = 0; t < LT; ++t) sc[(j * LT + t) * TP + lane] = h[j][t];
        wave_sync();
        if (lane < LT) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float *col = sc + (j * LT + lane) * TP;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 1
                for (int q0 = 0; q0 < 64; q0 += 16) {                // 16 reads in flight: keeps the register peak low
#pragma unroll
                    for (int q = q0; q < q0 + 16; q += 4) {
                        a0 += col[q];
                        a1 += col[q + 1];
                        a2 += col[q + 2];
                        a3 += col[q + 3];
                    }
                }
                atomicAdd(dCg + (n + j) * L + lane, (a0 + a1) + (a2 + a3));
            }
        }
        wave_sync();
    }
    // per-row outputs: du, ddelta (through the LDS tile for coalesced stores), dD, ddelta_bias
    float db = 0.f;
    Tin *dub = (Tin *)p.du + ((int64_t)b * p.dim + r0) * L;
    Tin *ddb = (Tin *)p.ddelta + ((int64_t)b * p.dim + r0) * L;
#pragma unroll
    for (int t = 0; t < LT; ++t) sc[lane * LT + t] = fmaf(dl[t], s1[t], Dr * g[t]);
    wave_sync();
    for (int e = lane; e < 64 * LT; e += 64) stf<Tin>(dub + e, sc[e]);
    wave_sync();
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        float dd = fmaf(ldf<Tin>(ur + t), s1[t], s2[t]);
        if (p.delta_softplus && dl[t] <= 20.f) dd *= 1.f - __expf(-dl[t]);
        db += dd;
        sc[lane * LT + t] = dd;
    }
    wave_sync();
    for (int e = lane; e < 64 * LT; e += 64) stf<Tin>(ddb + e, sc[e]);
    if (p.dD) atomicAdd(p.dD + r, dD);
    if (p.ddelta_bias) atomicAdd(p.ddelta_bias + r, db);
}

template <typename Tin, typename Tout, int LT, int NS>
static int rowscan_launch(const xfm_scan_params_t &p, bool bwd, hipStream_t s) {
    constexpr int LP = rs_pad<LT>();
    RowScanArgs a;
    a.p = p;
    a.dim_per_group = p.dim / p.n_groups;
    const unsigned grid = (unsigned)((int64_t)p.batch * (p.dim / 64));
    if (bwd) {
        const size_t lds = (size_t)(rs_bwd_tile<LT, NS>() + 2 * NS * LP) * sizeof(float) + (size_t)64 * LT * sizeof(Tin);
        hipLaunchKernelGGL((rowscan_bwd_kernel<Tin, Tout, LT, NS>), dim3(grid), dim3(64), lds, s, a);
    } else {
        const size_t lds = (size_t)(64 * LT + 2 * NS * LP) * sizeof(float) + (size_t)64 * LT * sizeof(Tin);
        hipLaunchKernelGGL((rowscan_fwd_kernel<Tin, Tout, LT, NS>), dim3(grid), dim3(64), lds, s, a);
    }
    return check_launch();
}

// true if the rowscan kernels cover this call (and then `rc` holds the launch result)
template <typename Tin, typename Tout>
static bool rowscan_try(const xfm_scan_params_t &p, bool bwd, hipStream_t s, int *rc) {
    if (p.seqlen != 49 || (p.dim / p.n_groups) % 64 != 0) return false;
    if (p.dstate == 16) {
        *rc = rowscan_launch<Tin, Tout, 49, 16>(p, bwd, s);
        return true;
    }
    if (p.dstate == 1) {
        *rc = rowscan_launch<Tin, Tout, 49, 1>(p, bwd, s);
        return true;
    }
    return false;
}

}  // namespace xfm

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
constexpr int kRsTP = 68;   // pitch of the transposed [t][row] tiles: rows are 16-byte aligned and both the per-row
                            // 4-byte writes and the per-column 16-byte reads are bank-conflict free
template <int LT, int NS> constexpr int rs_bwd_tile() {
    return (64 * LT > RowScanNI<NS>::value * LT * kRsTP) ? 64 * LT : RowScanNI<NS>::value * LT * kRsTP;
}

// One state's B (or C) row, wave-uniform.  bf16 operands: packed pairs in 25 SGPRs (filled with v_readlane from the
// fp32 LDS table -- exact, the table was widened from bf16), unpacked by the scalar unit where they are used, so the
// recurrence issues no LDS read for them.  fp32 operands: read from the LDS table (broadcast).
template <typename Tin, int LT> struct RsRow {
    static constexpr bool kPacked = sizeof(Tin) == 2;
    static constexpr int NP = (LT + 1) / 2;
    uint32_t pk[kPacked ? NP : 1];
    const float *tab;
    __device__ __forceinline__ void load(const float *row, int lane) {
        tab = row;
        if constexpr (kPacked) {
            const int i2 = 2 * (lane < NP ? lane : NP - 1);
            const uint32_t lo = __float_as_uint(row[i2]), hi = __float_as_uint(row[i2 + 1]);   // (pad slot for odd LT)
            const uint32_t v = (hi & 0xffff0000u) | (lo >> 16);
#pragma unroll
            for (int i = 0; i < NP; ++i) pk[i] = __builtin_amdgcn_readlane(v, i);
        }
    }
    __device__ __forceinline__ float operator()(int t) const {
        if constexpr (kPacked) return __uint_as_float((t & 1) ? (pk[t >> 1] & 0xffff0000u) : (pk[t >> 1] << 16));
        else return tab[t];
    }
};

template <typename S> struct RsVec { static constexpr bool ok = false; };          // 16-byte vector I/O available
template <> struct RsVec<float> { static constexpr bool ok = true; };
template <> struct RsVec<bf16_t> { static constexpr bool ok = true; };

template <typename S> __device__ __forceinline__ bool rs_aligned(const S *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

typedef uint32_t rs_u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte vector (a native vector: stays in registers)

__device__ __forceinline__ void rs_unpack(rs_u32x4 r, float (&v)[4], const float *) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
}
__device__ __forceinline__ void rs_unpack(rs_u32x4 r, float (&v)[8], const bf16_t *) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r[i] << 16);
        v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
}

// A contiguous, 16-byte aligned run of NE elements read by one wave as 16-byte vectors.  Contiguous rows (row stride
// == LT, what every caller in the model produces) make a 64-row operand tile ONE such run.  issue() only issues the
// loads -- unconditionally; the last, partial round re-reads the run's last vector -- so a kernel can put the loads of
// ALL its operands in flight before it waits for the first (2-byte loads waited on four at a time, one operand after
// the other, made the staging and not the recurrence the longest phase of these kernels).
template <typename S, int NE> struct RsRun {
    static constexpr int VE = 16 / (int)sizeof(S), NV = NE / VE, PER = (NV + 63) / 64;
    static_assert(NE % VE == 0, "a whole number of 16-byte vectors");
    rs_u32x4 raw[PER];
    __device__ __forceinline__ void issue(const S *base, int lane) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = lane + 64 * i;
            raw[i] = reinterpret_cast<const rs_u32x4 *>(base)[q < NV ? q : NV - 1];
        }
    }
    // flat copy into LDS: same element type -> raw copy, otherwise widened to fp32
    template <typename T> __device__ __forceinline__ void commit_flat(T *dst, int lane) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = lane + 64 * i;
            if (64 * i + 63 < NV || q < NV) {
                if constexpr (sizeof(T) == sizeof(S)) {
                    reinterpret_cast<rs_u32x4 *>(dst)[q] = raw[i];
                } else {
                    float v[VE];
                    rs_unpack(raw[i], v, static_cast<const S *>(nullptr));
#pragma unroll
                    for (int k = 0; k < VE; k += 4)
                        reinterpret_cast<float4 *>(dst + q * VE)[k / 4] = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
                }
            }
        }
    }
    // (NS x LT) block -> fp32 table tab[n * SN + t * ST]
    template <int LT, int SN, int ST> __device__ __forceinline__ void commit_rows(float *tab, int lane) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = lane + 64 * i;
            if (64 * i + 63 < NV || q < NV) {
                float v[VE];
                rs_unpack(raw[i], v, static_cast<const S *>(nullptr));
#pragma unroll
                for (int k = 0; k < VE; ++k) {
                    const int e = q * VE + k, n = e / LT;
                    tab[n * SN + (e - n * LT) * ST] = v[k];
                }
            }
        }
    }
};

// element-wise staging of one (64 x LT) operand tile (views: any row stride, any alignment) -> flat LDS [row*LT + t]
template <typename T, typename S, int LT>
__device__ __forceinline__ void rs_stage(T *dst, const S *base, int64_t row_stride, int lane) {
    for (int e = lane; e < 64 * LT; e += 64) {
        const int row = e / LT, t = e - row * LT;
        const float v = ldf<S>(base + (int64_t)row * row_stride + t);
        if constexpr (sizeof(T) == 4) dst[e] = v;
        else stf<T>(dst + e, v);
    }
}

// element-wise staging of the (NS x LT) B and C blocks of one (batch, group) into fp32 tables tab[n * SN + t * ST]
template <typename S, int LT, int NS, int SN, int ST>
__device__ __forceinline__ void rs_stage_rows(float *Bt, float *Ct, const S *Bg, const S *Cg, int64_t bstride, int64_t cstride,
                                              int lane) {
    for (int e = lane; e < NS * LT; e += 64) {
        const int n = e / LT, t = e - n * LT;
        Bt[n * SN + t * ST] = ldf<S>(Bg + (int64_t)n * bstride + t);
        Ct[n * SN + t * ST] = ldf<S>(Cg + (int64_t)n * cstride + t);
    }
}

// Operand staging shared by both kernels: u -> ut (storage dtype), delta -> sc (fp32), B / C -> tables.  With
// contiguous, aligned operands every load is issued before the first is waited for; the backward kernel passes a
// run for dout as well and commits it later (it reuses the sc tile).
template <typename Tin, int LT, int NS> struct RsOperands {
    static constexpr bool kVec = RsVec<Tin>::ok;
    static constexpr bool kRowsVec = kVec && (NS * LT) % (16 / (int)sizeof(Tin)) == 0;
    const Tin *ub, *db, *Bg, *Cg;
    __device__ __forceinline__ RsOperands(const xfm_scan_params_t &p, int b, int k, int r0) {
        ub = (const Tin *)p.u + (int64_t)b * p.u_batch_stride + (int64_t)r0 * p.u_d_stride;
        db = (const Tin *)p.delta + (int64_t)b * p.delta_batch_stride + (int64_t)r0 * p.delta_d_stride;
        Bg = (const Tin *)p.B + (int64_t)b * p.B_batch_stride + (int64_t)k * p.B_group_stride;
        Cg = (const Tin *)p.C + (int64_t)b * p.C_batch_stride + (int64_t)k * p.C_group_stride;
    }
    __device__ __forceinline__ bool fast(const xfm_scan_params_t &p) const {           // wave-uniform
        if constexpr (!kVec) return false;
        bool ok = p.u_d_stride == LT && p.delta_d_stride == LT && rs_aligned(ub) && rs_aligned(db);
        if constexpr (kRowsVec) ok = ok && p.B_dstate_stride == LT && p.C_dstate_stride == LT && rs_aligned(Bg) && rs_aligned(Cg);
        return ok;
    }
};

// write a (64 x LT) fp32 LDS tile to 64 rows of a global tensor (same fast path as rs_stage)
template <typename T, int LT>
__device__ __forceinline__ void rs_unstage(T *base, int64_t row_stride, const float *tile, int lane) {
    if constexpr (RsVec<T>::ok) {
        constexpr int VE = Pack<T>::N, NV = 64 * LT / VE, PER = (NV + 63) / 64;
        if (row_stride == LT && (reinterpret_cast<uintptr_t>(base) & 15) == 0) {
#pragma unroll
            for (int i = 0; i < PER; ++i)
                if (64 * i + 63 < NV || lane + 64 * i < NV) {
                    float v[VE];
#pragma unroll
                    for (int k = 0; k < VE; k += 4) {
                        const float4 r = reinterpret_cast<const float4 *>(tile + (lane + 64 * i) * VE)[k / 4];
                        v[k] = r.x; v[k + 1] = r.y; v[k + 2] = r.z; v[k + 3] = r.w;
                    }
                    Pack<T>::st(base + (lane + 64 * i) * VE, v);
                }
            return;
        }
    }
    for (int e = lane; e < 64 * LT; e += 64) {
        const int row = e / LT, t = e - row * LT;
        stf<T>(base + (int64_t)row * row_stride + t, tile[e]);
    }
}

template <typename Tin, typename Tout, int LT, int NS>
__global__ void __launch_bounds__(64) rowscan_fwd_kernel(const RowScanArgs a) {
    constexpr int LP = rs_pad<LT>();
    extern __shared__ __align__(16) float smem[];
    const xfm_scan_params_t &p = a.p;
    const int lane = threadIdx.x;
    const int tiles_pb = p.dim / 64;
    const int b = blockIdx.x / tiles_pb, r0 = (blockIdx.x - b * tiles_pb) * 64;
    const int k = r0 / a.dim_per_group, r = r0 + lane;
    float *dy = smem;                                   // [64][LT] delta in, y out (in place, lane-private slots)
    float *Bt = dy + 64 * LT;                           // [LT][NS]: one step's B (C) of all states is one 16-byte-aligned run
    float *Ct = Bt + NS * LP;
    Tin *ut = reinterpret_cast<Tin *>(Ct + NS * LP);    // [64][LT]
    float A2[NS], h[NS];                                // (per-row parameters first: their loads fly with the tiles')
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        A2[n] = p.A[(int64_t)r * p.A_d_stride + n] * kLog2e;
        h[n] = 0.f;
    }
    const float Dr = p.D ? p.D[r] : 0.f, bias = p.delta_bias ? p.delta_bias[r] : 0.f;
    const RsOperands<Tin, LT, NS> op(p, b, k, r0);
    using Ops = RsOperands<Tin, LT, NS>;
    bool staged = false;
    if constexpr (Ops::kVec) {
        if (op.fast(p)) {
            RsRun<Tin, 64 * LT> rd, ru;
            rd.issue(op.db, lane);
            ru.issue(op.ub, lane);
            if constexpr (Ops::kRowsVec) {
                RsRun<Tin, NS * LT> rb, rc;
                rb.issue(op.Bg, lane);
                rc.issue(op.Cg, lane);
                rd.commit_flat(dy, lane);
                ru.commit_flat(ut, lane);
                rb.template commit_rows<LT, 1, NS>(Bt, lane);
                rc.template commit_rows<LT, 1, NS>(Ct, lane);
            } else {
                rs_stage_rows<Tin, LT, NS, 1, NS>(Bt, Ct, op.Bg, op.Cg, p.B_dstate_stride, p.C_dstate_stride, lane);
                rd.commit_flat(dy, lane);
                ru.commit_flat(ut, lane);
            }
            staged = true;
        }
    }
    if (!staged) {
        rs_stage<float, Tin, LT>(dy, op.db, p.delta_d_stride, lane);
        rs_stage<Tin, Tin, LT>(ut, op.ub, p.u_d_stride, lane);
        rs_stage_rows<Tin, LT, NS, 1, NS>(Bt, Ct, op.Bg, op.Cg, p.B_dstate_stride, p.C_dstate_stride, lane);
    }
    wave_sync();
    // The recurrence, software-pipelined by hand: the operands of step t+1 (B, C of all states: wave-uniform LDS
    // broadcasts; delta and u of this lane's row) are requested before step t computes, so no LDS latency is exposed.
    struct Step { float Bv[NS], Cv[NS], dl, uu; };
    auto fetch = [&](Step &o, int t) {
        if constexpr (NS % 4 == 0) {
#pragma unroll
            for (int n = 0; n < NS; n += 4) {
                const float4 bq = *reinterpret_cast<const float4 *>(Bt + t * NS + n);
                const float4 cq = *reinterpret_cast<const float4 *>(Ct + t * NS + n);
                o.Bv[n] = bq.x; o.Bv[n + 1] = bq.y; o.Bv[n + 2] = bq.z; o.Bv[n + 3] = bq.w;
                o.Cv[n] = cq.x; o.Cv[n + 1] = cq.y; o.Cv[n + 2] = cq.z; o.Cv[n + 3] = cq.w;
            }
        } else {
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                o.Bv[n] = Bt[t * NS + n];
                o.Cv[n] = Ct[t * NS + n];
            }
        }
        o.dl = dy[lane * LT + t];
        o.uu = ldf<Tin>(ut + lane * LT + t);
    };
    auto step = [&](const Step &o, int t) {
        float dl = o.dl + bias;
        if (p.delta_softplus) dl = softplus20(dl);
        const float du = dl * o.uu;
        float y = Dr * o.uu;
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const float av = exp2_fast(dl * A2[n]);
            h[n] = fmaf(av, h[n], du * o.Bv[n]);
            y = fmaf(o.Cv[n], h[n], y);
        }
        dy[lane * LT + t] = y;
    };
    Step s0, s1;
    fetch(s0, 0);
#pragma unroll 1
    for (int t = 0; t + 1 < LT; t += 2) {
        fetch(s1, t + 1);
        step(s0, t);
        fetch(s0, t + 2 < LT ? t + 2 : LT - 1);
        step(s1, t + 1);
    }
    if (LT & 1) step(s0, LT - 1);
    wave_sync();
    rs_unstage<Tout, LT>((Tout *)p.out + (int64_t)b * p.out_batch_stride + (int64_t)r0 * p.out_d_stride, p.out_d_stride, dy, lane);
}

// sum of one 64-float row of a transposed tile: 16-byte reads, eight in flight
__device__ __forceinline__ float rs_colsum(const float *row) {
    const float4 *c = reinterpret_cast<const float4 *>(row);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = c[q0 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc.x += v[q].x;
            acc.y += v[q].y;
            acc.z += v[q].z;
            acc.w += v[q].w;
        }
    }
    return (acc.x + acc.y) + (acc.z + acc.w);
}

template <typename Tin, typename Tout, int LT, int NS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))   // 1 wave/SIMD: whole 512-register file
rowscan_bwd_kernel(const RowScanArgs a) {
    constexpr int LP = rs_pad<LT>();
    extern __shared__ __align__(16) float smem[];
    const xfm_scan_params_t &p = a.p;
    const int lane = threadIdx.x;
    const int tiles_pb = p.dim / 64;
    const int b = blockIdx.x / tiles_pb, r0 = (blockIdx.x - b * tiles_pb) * 64;
    const int k = r0 / a.dim_per_group, r = r0 + lane;
    const int L = LT;
    // LDS: one (64 x LT) fp32 tile used for staging / transposes / outputs, plus B and C of the group.
    // The row's own operands live in REGISTERS for the whole state loop (a wave is alone on its SIMD anyway:
    // the tile keeps occupancy LDS-bound, so the 512-register file is free to use).
    float *sc = smem;                                   // [64][LT] staging, or NI transposed tiles [LT][kRsTP]
    float *Bt = sc + rs_bwd_tile<LT, NS>();             // [NS][LP]
    float *Ct = Bt + NS * LP;
    Tin *ut = reinterpret_cast<Tin *>(Ct + NS * LP);    // [64][LT] u in its storage dtype
    const RsOperands<Tin, LT, NS> op(p, b, k, r0);
    using Ops = RsOperands<Tin, LT, NS>;
    const Tout *gb = (const Tout *)p.dout + (int64_t)b * p.dout_batch_stride + (int64_t)r0 * p.dout_d_stride;
    const Tin *ur = ut + lane * LT;
    // bf16: the row's u as 25 packed registers for the whole kernel (no per-step LDS read); fp32: read from the tile
    constexpr bool kUP = sizeof(Tin) == 2;
    uint32_t up[kUP ? (LT + 1) / 2 : 1];
    auto U = [&](int t) -> float {
        if constexpr (kUP) return __uint_as_float((t & 1) ? (up[t >> 1] & 0xffff0000u) : (up[t >> 1] << 16));
        else return ldf<Tin>(ur + t);
    };
    const float Dr = p.D ? p.D[r] : 0.f, bias = p.delta_bias ? p.delta_bias[r] : 0.f;
    float dl[LT], g[LT];                                // delta' , dout  of this lane's row
    auto take_delta = [&]() {                           // sc holds delta (and ut holds u): this lane's row -> registers
        wave_sync();
        if constexpr (kUP) {
            const uint16_t *u16 = reinterpret_cast<const uint16_t *>(ur);
#pragma unroll
            for (int i = 0; i < (LT + 1) / 2; ++i)
                up[i] = (uint32_t)u16[2 * i] | (2 * i + 1 < LT ? (uint32_t)u16[2 * i + 1] << 16 : 0u);
        }
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            float v = sc[lane * LT + t] + bias;
            if (p.delta_softplus) v = softplus20(v);
            dl[t] = v;
        }
        wave_sync();
    };
    bool staged = false;
    if constexpr (Ops::kVec && RsVec<Tout>::ok) {
        if (op.fast(p) && p.dout_d_stride == LT && rs_aligned(gb)) {
            RsRun<Tin, 64 * LT> rd, ru;
            RsRun<Tout, 64 * LT> rg;
            rd.issue(op.db, lane);
            ru.issue(op.ub, lane);
            if constexpr (Ops::kRowsVec) {
                RsRun<Tin, NS * LT> rb, rc;
                rb.issue(op.Bg, lane);
                rc.issue(op.Cg, lane);
                rg.issue(gb, lane);
                rd.commit_flat(sc, lane);
                ru.commit_flat(ut, lane);
                rb.template commit_rows<LT, LP, 1>(Bt, lane);
                rc.template commit_rows<LT, LP, 1>(Ct, lane);
            } else {
                rg.issue(gb, lane);
                rs_stage_rows<Tin, LT, NS, LP, 1>(Bt, Ct, op.Bg, op.Cg, p.B_dstate_stride, p.C_dstate_stride, lane);
                rd.commit_flat(sc, lane);
                ru.commit_flat(ut, lane);
            }
            take_delta();
            rg.commit_flat(sc, lane);
            staged = true;
        }
    }
    if (!staged) {
        rs_stage_rows<Tin, LT, NS, LP, 1>(Bt, Ct, op.Bg, op.Cg, p.B_dstate_stride, p.C_dstate_stride, lane);
        rs_stage<Tin, Tin, LT>(ut, op.ub, p.u_d_stride, lane);
        rs_stage<float, Tin, LT>(sc, op.db, p.delta_d_stride, lane);
        take_delta();
        rs_stage<float, Tout, LT>(sc, gb, p.dout_d_stride, lane);
    }
    wave_sync();
    float dD = 0.f;
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        g[t] = sc[lane * LT + t];
        dD = fmaf(g[t], U(t), dD);
    }
    wave_sync();
    float s1[LT], s2[LT];
#pragma unroll
    for (int t = 0; t < LT; ++t) s1[t] = s2[t] = 0.f;
    float *dBg = p.dB + ((int64_t)b * p.n_groups + k) * NS * L;
    float *dCg = p.dC + ((int64_t)b * p.n_groups + k) * NS * L;
    // s1 accumulates sum_n dh*B, s2 sum_n dh*A*(a*h_prev)
    // NI states run interleaved: their recurrences are independent dependency chains, which is the only
    // latency hiding a wave that is alone on its SIMD gets.  Tiles T[j] ([LT][TP], TP = 68: conflict-free both for
    // the per-row writes and for the per-column sums) transpose the dB / dC contributions of the 64 rows.
    constexpr int NI = RowScanNI<NS>::value;
    constexpr int TP = kRsTP;
    static_assert(NI == 1, "one state at a time (see RowScanNI)");
    float A_next = p.A[(int64_t)r * p.A_d_stride];      // A[r][n] is fetched one state ahead of its use
#pragma unroll 1
    for (int n = 0; n < NS; n += NI) {
        float An[NI], A2[NI], hp[NI], E[NI], dA[NI];
        float h[NI][LT];
        RsRow<Tin, LT> Bn, Cn;
        Bn.load(Bt + n * LP, lane);
        Cn.load(Ct + n * LP, lane);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            An[j] = A_next;
            A_next = p.A[(int64_t)r * p.A_d_stride + (n + 1 < NS ? n + 1 : n)];
            A2[j] = An[j] * kLog2e;
            hp[j] = E[j] = dA[j] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < LT; ++t) {
            const float du = dl[t] * U(t);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                hp[j] = fmaf(exp2_fast(dl[t] * A2[j]), hp[j], du * Bn(t));
                h[j][t] = hp[j];
            }
            if (t % 7 == 6) __builtin_amdgcn_sched_barrier(0);      // stop the scheduler from preloading all of u
        }
#pragma unroll
        for (int t = LT - 1; t >= 0; --t) {
            const float du = dl[t] * U(t);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float Bv = Bn(t);
                float ea = dl[t] * A2[j];
                asm volatile("" : "+v"(ea));                       // recompute a_t: keeping 49*NI of them alive spills
                const float av = exp2_fast(ea);
                const float dh = fmaf(Cn(t), g[t], E[j]);
                E[j] = av * dh;
                const float ah = h[j][t] - du * Bv;                 // a_t * h_{t-1}
                s1[t] = fmaf(dh, Bv, s1[t]);
                const float q = dh * ah;
                s2[t] = fmaf(An[j], q, s2[t]);
                dA[j] = fmaf(dl[t], q, dA[j]);
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
                atomicAdd(dBg + (n + j) * L + lane, rs_colsum(sc + (j * LT + lane) * TP));
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
                atomicAdd(dCg + (n + j) * L + lane, rs_colsum(sc + (j * LT + lane) * TP));
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
    rs_unstage<Tin, LT>(dub, LT, sc, lane);
    wave_sync();
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        float dd = fmaf(U(t), s1[t], s2[t]);
        if (p.delta_softplus && dl[t] <= 20.f) dd *= 1.f - __expf(-dl[t]);
        db += dd;
        sc[lane * LT + t] = dd;
    }
    wave_sync();
    rs_unstage<Tin, LT>(ddb, LT, sc, lane);
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

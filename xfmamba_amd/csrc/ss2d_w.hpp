// ss2d_w.hpp -- wide-map SS2D kernels, fourth generation (round 6): 56 x 56 / 28 x 28 (and 48 x 48 / 24 x 24), d_state 1, bf16
// I/O, step sizes arriving activated (delta_softplus == 2), fp32 copies of the B / C rows for the backward (bc_f32: the production path).  Included by ss2d_l3.hip, whose tile staging,
// merge, LDS-direct prefetch and workgroup map it shares; same layout contract (include/xfm_hip.h), same algorithm (reference
// models/fusion_vmamba.py:1145-1174; adjoint per selective_scan_bwd_kernel.cuh:141-273).  What changed against ss2d_l3.hip,
// and the measurement behind each change (tools/ubench/valu_rate.hip on MI355X: at two waves per SIMD a plain fp32 FMA costs
// the SIMD 2.5 cycles, an integer / DPP / packed-fp32 instruction 3.3 - 3.5, a transcendental 6.3; a lone wave 5 per
// instruction whatever it is):
//   * CHECKPOINTS PER LANE: the forward stores the state entering every lane's 8-position chunk (one coalesced 256-byte store
//     per chunk row: +0.5 B per route element) instead of one per chunk row, so the backward replays the states straight from
//     them: no fold of the state maps, no ascending wave scan (8 FMA + 22 DPP-class instructions per chunk row less);
//   * the backward's lanes run AGAINST the route (lane l owns route-order chunk 63 - l): the adjoint scan is the ascending one
//     (row_shr + row_bcast: 14 DPP instructions) instead of the descending one with its readlane / EXEC-mask cross-row steps;
//     its wait states are filled with the state replay (8 independent FMAs) instead of s_nop;
//   * chunk rows are unrolled STATICALLY (7 / 2 per plane): tail handling, accumulator row and operand addresses are compile-time,
//     the raw operand vectors are unpacked before the next row's requests are issued (scheduling barrier), so the requests land
//     in the registers just freed: the ~23 v_mov_b64 per chunk row that rotated the prefetch buffers are gone.
//   * the 64-position LAST chunk row of a 56 x 56 (24 x 24) plane runs ONE POSITION PER LANE (WTail1 below): as a row of 8-position
//     chunks it kept 8 of 64 lanes busy for the price of a full row.  56 x 56 backward 159.5 -> 152 us, 24 x 24 233 -> 209 us.
// Roofline: HBM (24 B per (b,d,p) element backward, 14 B forward at this boundary; + 2 B each way for the checkpoints).
#pragma once

namespace xfm {

struct WOps { uint4 d, b0, b1, c0, c1; float h; };      // step sizes (8 bf16), B and C (8 fp32 each), the entering state

// Global operands through BUFFER descriptors: address = descriptor base (4 scalar registers per tensor) + scalar byte offset
// (plane / route / chunk row: scalar adds or the instruction's immediate) + ONE 32-bit lane offset in a vector register for
// every 16-byte operand of the kernel.  Written as pointer + index the compiler builds a 64-bit vector address per access and,
// with the chunk rows unrolled, hoists all of them out of the tile loop (~70 registers held through the sweeps, spilled).
// The range check does the tail row's lane masking: a dead lane's offset is kWDead -- loads return 0, stores are dropped.
typedef unsigned int w_u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kWDead = 0xffffff00u;
struct WBuf { __amdgpu_buffer_rsrc_t dts, ddts, Bs, Cs, chk; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w_rsrc(const void *p, const uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 w_ld16(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
    const w_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float w_ld4(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// (A 16-byte buffer store reads its data registers over several cycles.  The compiler's hazard recognizer covers a vector
//  instruction that overwrites them right behind the store only when the store's scalar-offset field is NOT a register; here
//  it is one, and on gfx950 the hazard is there all the same: with a second workgroup on the CU, four lanes of every row of 16
//  stored the LDS address that re-used the first data register instead of the packed step-size gradients -- seen at 48 x 48
//  with 256 channels and more, found through the XFMamba-B gradient test.  Wait states behind the store, pinned, and the data registers kept allocated past the next few instructions.)
__device__ __forceinline__ void w_st16(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff, const uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(w_u32x4{v.x, v.y, v.z, v.w}, r, voff, soff, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3");
}
__device__ __forceinline__ void w_st4(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff, const float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, voff, soff, 0);
}
__device__ __forceinline__ uint32_t w_ld2(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff) {
    return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
__device__ __forceinline__ void w_st2(const __amdgpu_buffer_rsrc_t r, const uint32_t voff, const int soff, const uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b16((short)v, r, voff, soff, 0);
}
// ONE POSITION PER LANE in the last chunk row where it holds exactly 64 positions (56 x 56: 3136 = 6 x 512 + 64): as a row of
// 8-position chunks it keeps 8 of 64 lanes busy for the price of a full row -- a seventh of the sweeps' instructions.  Per lane
// one step size, one B, one C (2- / 4-byte loads), the forward stores the state entering every POSITION of that row in the row's
// 256 checkpoint bytes (as many as a row of per-chunk checkpoints), so the backward needs no state scan there either: one FMA
// replays the state, one ascending DPP scan of (a, a C g) carries the adjoint.  Needs the row's dB / dC sums in the LDS strip.
template <int HW> struct WTail1 {
    using G = L3Geom<HW>;
    // (56 x 56 and 24 x 24; the row's dB / dC sums are a register pair, so the LDS strip -- if there is one -- must hold nothing else)
    static constexpr bool on = G::HAS_TAIL && G::TAILV == 8 && G::NREG >= G::NSEG - 1;
};
// (forward: that row's operands travel in the first dword of the vectors a full row uses, WFOps d.x / b.x / c.x.  Backward: in
//  registers of their own, W1Ops -- the short row does not cover the latency of the requests a row issues for its successor, so the
//  row BEFORE it requests both its operands and its successor's)
struct W1Ops { uint32_t d; float b, c, h; };
__device__ __forceinline__ WBuf w_bufs(const LeanArgs &a, const int nseg, const bool bwd) {
    const uint32_t planes = (uint32_t)a.batch * 4u * (uint32_t)a.D_, L = (uint32_t)a.L;
    WBuf rs;
    rs.dts = w_rsrc(a.dts, planes * L * 2u);
    rs.ddts = w_rsrc(a.ddts, planes * L * 2u);
    // the backward reads the fp32 copies of the B / C rows (xfm_ss2d_route_split_bc32), the forward the 16-bit rows: with five
    // operand vectors per chunk row at four waves per SIMD the forward is bound by the vector-memory path (56 x 56: 105 vs 83 us)
    rs.Bs = bwd ? w_rsrc(a.Bs32, (uint32_t)a.batch * 4u * L * 4u) : w_rsrc(a.Bs, (uint32_t)a.batch * 4u * L * 2u);
    rs.Cs = bwd ? w_rsrc(a.Cs32, (uint32_t)a.batch * 4u * L * 4u) : w_rsrc(a.Cs, (uint32_t)a.batch * 4u * L * 2u);
    rs.chk = w_rsrc(a.chk, planes * (uint32_t)nseg * 256u);
    return rs;
}

// 8 fp32 of two 16-byte vectors -> four pairs in TRAVERSAL order (register renaming: no instruction)
template <bool REV> __device__ __forceinline__ void w_pairs(const uint4 &v0, const uint4 &v1, l3f2 (&o)[4]) {
    const float p[8] = {__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w),
                        __uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w)};
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = REV ? l3f2{p[7 - 2 * q], p[6 - 2 * q]} : l3f2{p[2 * q], p[2 * q + 1]};
}

// ascending inclusive scan of the affine maps (Q, R) with the state replay h[e] = a[e] h[e-1] + bb[e] of the lane's own chunk
// (independent of the scan) issued in its wait states.  EXEC must be all ones.
__device__ __forceinline__ void w_scan_replay(float &Q, float &R, const l3f2 (&a)[4], const l3f2 (&bb)[4], const float hin,
                                              l3f2 (&h)[4]) {
    float h0, h1, h2, h3, h4, h5, h6, h7;
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %[R], %[R], %[Q] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h0], %[a0], %[hin], %[b0]\n\t"
        "v_mul_f32_dpp %[Q], %[Q], %[Q] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h1], %[a1], %[h0], %[b1]\n\t"
        "v_fmac_f32_dpp %[R], %[R], %[Q] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h2], %[a2], %[h1], %[b2]\n\t"
        "v_mul_f32_dpp %[Q], %[Q], %[Q] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h3], %[a3], %[h2], %[b3]\n\t"
        "v_fmac_f32_dpp %[R], %[R], %[Q] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h4], %[a4], %[h3], %[b4]\n\t"
        "v_mul_f32_dpp %[Q], %[Q], %[Q] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h5], %[a5], %[h4], %[b5]\n\t"
        "v_fmac_f32_dpp %[R], %[R], %[Q] row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h6], %[a6], %[h5], %[b6]\n\t"
        "v_mul_f32_dpp %[Q], %[Q], %[Q] row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fma_f32 %[h7], %[a7], %[h6], %[b7]\n\t"
        "v_fmac_f32_dpp %[R], %[R], %[Q] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %[Q], %[Q], %[Q] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %[R], %[R], %[Q] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %[Q], %[Q], %[Q] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        : [Q] "+v"(Q), [R] "+v"(R), [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [h4] "=&v"(h4),
          [h5] "=&v"(h5), [h6] "=&v"(h6), [h7] "=&v"(h7)
        : [hin] "v"(hin), [a0] "v"(a[0].x), [a1] "v"(a[0].y), [a2] "v"(a[1].x), [a3] "v"(a[1].y), [a4] "v"(a[2].x),
          [a5] "v"(a[2].y), [a6] "v"(a[3].x), [a7] "v"(a[3].y), [b0] "v"(bb[0].x), [b1] "v"(bb[0].y), [b2] "v"(bb[1].x),
          [b3] "v"(bb[1].y), [b4] "v"(bb[2].x), [b5] "v"(bb[2].y), [b6] "v"(bb[3].x), [b7] "v"(bb[3].y));
    h[0] = l3f2{h0, h1};
    h[1] = l3f2{h2, h3};
    h[2] = l3f2{h4, h5};
    h[3] = l3f2{h6, h7};
}

// sums of three values over the wave, left in lane 63 (row_shr steps, then the row totals handed up by row_bcast)
__device__ __forceinline__ void w_wave_sum3(float &x, float &y, float &z) {
#define W_SUM_STEP(CTRL, MASK)                     \
    {                                              \
        const float xs = dpp_mov<CTRL, MASK>(0.f, x); \
        const float ys = dpp_mov<CTRL, MASK>(0.f, y); \
        const float zs = dpp_mov<CTRL, MASK>(0.f, z); \
        x += xs;                                   \
        y += ys;                                   \
        z += zs;                                   \
    }
    W_SUM_STEP(kRowShr1, 0xf)
    W_SUM_STEP(kRowShr2, 0xf)
    W_SUM_STEP(kRowShr4, 0xf)
    W_SUM_STEP(kRowShr8, 0xf)
    W_SUM_STEP(kRowBcast15, 0xa)
    W_SUM_STEP(kRowBcast31, 0xc)
#undef W_SUM_STEP
}

template <int N, typename F> __device__ __forceinline__ void w_static_for(F &&f) {
    if constexpr (N > 0) {
        w_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, one route over one plane.  `op` holds the operands of the plane's first row to process (route-order row NSEG - 1)
// on entry and, on exit, those of the next plane's (when `chain`: more planes in this tile).
// The map's last physical chunk row (the one with dead lanes) is PEELED -- it is the first row a forward route's backward
// processes and the last of a reversed route's -- so its lane masks are compile-time; the full rows run in a rolled loop (fully
// unrolled, the register allocator spilled the dB / dC sums around every row).
// ---------------------------------------------------------------------------------------------------------------------
template <int HW, bool REV, int PFPL>
__device__ __forceinline__ void w_bwd_plane(const WBuf &rs, const int dso, const int bso, const int cso, const bool chain,
                                            const float An, const float Dr,
                                            const bf16_t *xq, const bf16_t *gq, bf16_t *dxq, float *ldsacc,
                                            l3f2 (&rB)[L3Geom<HW>::NACC][4], l3f2 (&rC)[L3Geom<HW>::NACC][4], l3f2 &rT, float &dA_acc,
                                            float &dD_acc, float &dbias_acc, const int lane, WOps &op, W1Ops &op1, const L3Next &nx) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, NSEG = G::NSEG;
    constexpr bool T1 = WTail1<HW>::on;
    const float A2 = An * kLog2e;
    const int ci = REV ? lane : 63 - lane;            // physical chunk of this lane: the lanes run AGAINST the route
    const bool tail_live = !G::HAS_TAIL || ci < G::TAILV;
    const uint32_t lo16 = (uint32_t)ci * 16u;         // its byte offset in a chunk row of bf16 operands
    const uint32_t lo4 = (uint32_t)(63 - lane) * 4u;  // ... and in a row of checkpoints (route-order chunk = the forward's lane)
    const uint32_t lo16t = tail_live ? lo16 : kWDead, lo4t = tail_live ? lo4 : kWDead;     // ... in the tail row
    const uint32_t lo32 = 2u * lo16, lo32t = tail_live ? lo32 : kWDead;                     // ... of the fp32 B / C rows
    const bf16_t *xql = xq + ci * 8, *gql = gq + ci * 8;
    bf16_t *dxql = dxq + ci * 8;
    float Ec = 0.f;                                   // adjoint flowing in from the chunk row processed before
    l3f2 dA2 = {0.f, 0.f}, dD2 = dA2, db2 = dA2;
    uint4 lx, lg;                                     // x / dy of the row to process, read from LDS one row ahead
    {
        constexpr int sp0 = REV ? 0 : NSEG - 1;
        if (T1 && sp0 == NSEG - 1) {
            lx = lg = make_uint4(0, 0, 0, 0);
            lx.x = reinterpret_cast<const uint16_t *>(xq)[sp0 * G::ROW + ci];
            lg.x = reinterpret_cast<const uint16_t *>(gq)[sp0 * G::ROW + ci];
        } else if (G::HAS_TAIL && sp0 == NSEG - 1 && !tail_live) {
            lx = lg = make_uint4(0, 0, 0, 0);
        } else {
            lx = *reinterpret_cast<const uint4 *>(xql + sp0 * G::ROW);
            lg = *reinterpret_cast<const uint4 *>(gql + sp0 * G::ROW);
        }
    }
    // the next tile's planes by LDS-direct loads: sent for once the operands of the tile's first row are here (nothing of
    // this wave is in flight behind the explicit wait, so the asm loads cannot make a compiler-counted wait unsafe)
    auto dma_next_tile = [&](const int i) {
        if constexpr (PFPL > 0) {
            if (nx.go && i == NSEG - 1) {
                int lz;
                asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, 0" : "=v"(lz)::"memory");
                l3_dma_tile<PFPL>(nx.x, nx.g, nx.xdst, nx.gdst, nx.wave, lane + lz);
            }
        }
    };
    // ---- request the next row to process (row i - 1 of this plane, or the first row of the next plane): global operands
    // now, its x / dy from LDS behind the scan (a short round trip: 8 registers less at the row's register peak)
    // (next1: the row requested is the one-position-per-lane tail -- known at compile time at every call site, so that the rolled
    //  loop over the full rows stays free of branches: with a run-time test there the 56 x 56 launch took 182 us instead of 157)
    auto request_full = [&](const int spn, const int dsn, const int csn) {
        const bool tail_n = G::HAS_TAIL && spn == NSEG - 1;
        const uint32_t vo16 = tail_n ? lo16t : lo16, vo4 = tail_n ? lo4t : lo4, vo32 = tail_n ? lo32t : lo32;
        op.d = w_ld16(rs.dts, vo16, dsn);
        op.b0 = w_ld16(rs.Bs, vo32, bso + spn * (G::ROW * 4));
        op.b1 = w_ld16(rs.Bs, vo32, bso + spn * (G::ROW * 4) + 16);
        op.c0 = w_ld16(rs.Cs, vo32, bso + spn * (G::ROW * 4));
        op.c1 = w_ld16(rs.Cs, vo32, bso + spn * (G::ROW * 4) + 16);
        op.h = w_ld4(rs.chk, vo4, csn);
    };
    auto request_next = [&](auto next1_tag, const int i) {
        constexpr bool next1 = decltype(next1_tag)::value;
        if constexpr (next1) {
            // the row after this one is the one-position tail (of this plane: REV, i == 1; of the next plane: !REV, i == 0): its
            // operands, and those of the full row behind it (REV: the next plane's first row; !REV: the next plane's row NSEG - 2)
            constexpr int spt = NSEG - 1;
            const bool same = REV;                    // the tail belongs to this plane
            if (same || chain) {
                const int dst_ = dso + (same ? 0 : L * 2) + spt * (G::ROW * 2);
                const int cst_ = cso + (same ? 0 : (NSEG + NSEG - 1)) * 256;
                op1.d = w_ld2(rs.dts, (uint32_t)ci * 2u, dst_);
                op1.b = w_ld4(rs.Bs, (uint32_t)ci * 4u, bso + spt * (G::ROW * 4));
                op1.c = w_ld4(rs.Cs, (uint32_t)ci * 4u, bso + spt * (G::ROW * 4));
                op1.h = w_ld4(rs.chk, lo4, cst_);
            }
            if (chain) {
                constexpr int spf = REV ? 0 : NSEG - 2;       // physical row of the full row behind the tail, in the NEXT plane
                constexpr int rf = REV ? NSEG - 1 : NSEG - 2; // ... its route-order index
                request_full(spf, dso + L * 2 + spf * (G::ROW * 2), cso + (NSEG + rf) * 256);
            }
        } else {
            const bool has_next = i > 0;
            const int spn = has_next ? (REV ? NSEG - i : i - 1) : (REV ? 0 : NSEG - 1);
            if (has_next || chain)
                request_full(spn, dso + (has_next ? 0 : L * 2) + spn * (G::ROW * 2), cso + (has_next ? i - 1 : NSEG + NSEG - 1) * 256);
        }
    };
    auto lds_next = [&](auto next1_tag, const int i) {
        constexpr bool next1 = decltype(next1_tag)::value;
        const bool has_next = i > 0;
        const int spn = has_next ? (REV ? NSEG - i : i - 1) : (REV ? 0 : NSEG - 1);
        const bool tail_n = G::HAS_TAIL && spn == NSEG - 1;
        if (has_next) {
            if constexpr (next1) {
                lx.x = reinterpret_cast<const uint16_t *>(xq)[spn * G::ROW + ci];
                lg.x = reinterpret_cast<const uint16_t *>(gq)[spn * G::ROW + ci];
            } else if (!tail_n || tail_live) {
                lx = *reinterpret_cast<const uint4 *>(xql + spn * G::ROW);
                lg = *reinterpret_cast<const uint4 *>(gql + spn * G::ROW);
            } else {
                lx = lg = make_uint4(0, 0, 0, 0);
            }
        }
    };
    // ---- the tail row with one position per lane (WTail1)
    auto row1 = [&](const int i) {
        constexpr int sp = NSEG - 1;
        // (a real branch in front of the row: as straight-line code the compiler schedules it into its neighbours, and the
        //  56 x 56 launch takes 168 us instead of 158)
        int skip;
        asm volatile("s_mov_b32 %0, 0" : "=s"(skip));
        if (skip) return;
        const float v = __uint_as_float(op1.d << 16), u = __uint_as_float(lx.x << 16), g = __uint_as_float(lg.x << 16);
        const float Bq = op1.b, Cq = op1.c, hin = op1.h;
        __builtin_amdgcn_sched_barrier(0);
        dma_next_tile(i);
        __builtin_amdgcn_sched_barrier(0);
        const float a = exp2_fast(v * A2);
        const float vu = v * u, bb = vu * Bq, cg = Cq * g, acg = a * cg;
        float Q = a, R = acg;
        l3_scan_up(Q, R);                             // the lanes run against the route: ascending = the adjoint's direction
        const float h = fmaf(a, hin, bb);             // state after this position (the forward stored the one entering it)
        const float tE = fmaf(Q, Ec, R);
        const float E = dpp_mov<kWaveShr1>(Ec, tE);   // adjoint entering this position
        Ec = bcast_lane<63>(tE);
        __builtin_amdgcn_sched_barrier(0);
        lds_next(std::false_type{}, i);
        __builtin_amdgcn_sched_barrier(0);
        const float dh = cg + E;
        const float sg = 1.f - exp2_fast(v * (-kLog2e));
        const float ah = h - bb, s1 = dh * Bq, dhah = dh * ah;
        dA2.x = fmaf(v, dhah, dA2.x);
        const float du = fmaf(v, s1, g * Dr);
        const float dd = fmaf(u, s1, dhah * An) * sg;
        dD2.x = fmaf(g, u, dD2.x);
        db2.x += dd;
        w_st2(rs.ddts, (uint32_t)ci * 2u, dso + sp * (G::ROW * 2), l3_cvt_pk(dd, 0.f));
        reinterpret_cast<uint16_t *>(dxq)[sp * G::ROW + ci] = (uint16_t)(l3_cvt_pk(du, 0.f) & 0xffffu);
        rT.x = fmaf(dh, vu, rT.x);                    // the row's dB / dC sums: one pair of registers (not the LDS strip: two
        rT.y = fmaf(g, h, rT.y);                      // read-modify-write round trips in the row's dependent chain)
    };
    // i: route-order chunk row -- an int in the rolled loops, a std::integral_constant for the peeled rows and for maps of two chunk
    // rows (28 x 28, 24 x 24): there every test on the row index folds away, and so does the compare-and-skip of the accumulate chain
    auto row = [&](auto tail_tag, auto next1_tag, auto iv) {
        constexpr bool is_tail = decltype(tail_tag)::value;
        constexpr bool is_static = !std::is_same<decltype(iv), int>::value;
        const int i = iv;
        const int sp = REV ? NSEG - 1 - i : i;        // physical chunk row
        // ---- consume the raw vectors: everything the row needs from them is in fp32 registers below
        l3f2 v[4], u[4], g[4], Bq[4], Cq[4];
        l3_unpack<REV>(op.d, v);
        w_pairs<REV>(op.b0, op.b1, Bq);
        w_pairs<REV>(op.c0, op.c1, Cq);
        l3_unpack<REV>(lx, u);
        l3_unpack<REV>(lg, g);
        const float hin = op.h;
        __builtin_amdgcn_sched_barrier(0);
        dma_next_tile(i);
        request_next(next1_tag, i);
        __builtin_amdgcn_sched_barrier(0);
        // ---- element-wise part
        l3f2 a[4], vu[4], bb[4], cg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = l3_exp2(v[q] * A2);
            vu[q] = v[q] * u[q];
            bb[q] = vu[q] * Bq[q];
            cg[q] = Cq[q] * g[q];
        }
        // ---- the lane's adjoint map E_out = Q E_in + R over its chunk (walked against the route; E_t = a_t dh_t,
        // dh_t = C_t g_t + E_{t+1}), scanned over the lanes
        l3f2 acg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acg[q] = a[q] * cg[q];
        float R = 0.f;
#pragma unroll
        for (int q = 3; q >= 0; --q) {
            R = fmaf(a[q].y, R, acg[q].y);
            R = fmaf(a[q].x, R, acg[q].x);
        }
        float Q = (a[0].x * a[0].y) * (a[1].x * a[1].y) * ((a[2].x * a[2].y) * (a[3].x * a[3].y));
        l3f2 h[4], dh[4];
        w_scan_replay(Q, R, a, bb, hin, h);
        const float tE = fmaf(Q, Ec, R);
        float E = dpp_mov<kWaveShr1>(Ec, tE);         // adjoint entering this lane's chunk = the map of the lanes below on the carry
        Ec = bcast_lane<63>(tE);
        __builtin_amdgcn_sched_barrier(0);
        lds_next(next1_tag, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 3; q >= 0; --q) {
            dh[q].y = cg[q].y + E;
            E = fmaf(a[q].y, E, acg[q].y);
            dh[q].x = cg[q].x + E;
            E = fmaf(a[q].x, E, acg[q].x);
        }
        // ---- per position pair: the outputs, packed as they come
        uint32_t wd[4], wu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const l3f2 sg = 1.f - l3_exp2(v[q] * (-kLog2e));   // sigmoid(raw) = 1 - exp(-softplus(raw)): the step sizes arrive activated
            const l3f2 ah = h[q] - bb[q];
            const l3f2 s1 = dh[q] * Bq[q];
            const l3f2 dhah = dh[q] * ah;
            dA2 = __builtin_elementwise_fma(v[q], dhah, dA2);
            const l3f2 du = __builtin_elementwise_fma(v[q], s1, g[q] * Dr);
            const l3f2 dd = __builtin_elementwise_fma(u[q], s1, dhah * An) * sg;
            dD2 = __builtin_elementwise_fma(g[q], u[q], dD2);
            db2 += dd;
            // (traversal pair q is physical dword 3 - q, halves swapped, of a descending route)
            wd[REV ? 3 - q : q] = REV ? l3_cvt_pk(dd.y, dd.x) : l3_cvt_pk(dd.x, dd.y);
            wu[REV ? 3 - q : q] = REV ? l3_cvt_pk(du.y, du.x) : l3_cvt_pk(du.x, du.y);
        }
        w_st16(rs.ddts, is_tail ? lo16t : lo16, dso + sp * (G::ROW * 2), make_uint4(wd[0], wd[1], wd[2], wd[3]));
        if (!is_tail || tail_live)
            *reinterpret_cast<uint4 *>(dxql + sp * G::ROW) = make_uint4(wu[0], wu[1], wu[2], wu[3]);   // this route's private dx plane
        // (the store's data registers stay allocated until here: nothing may be written into them right behind it -- w_st16)
        asm volatile("" ::"v"(wd[0]), "v"(wd[1]), "v"(wd[2]), "v"(wd[3]));
        // ---- dB += dh (delta u), dC += g h, summed over the planes of this workgroup (traversal order; un-permuted at the flush)
        if (G::LSZ > 0 && sp >= G::NREG) {
            if (!is_tail || tail_live) {
                float *tb = ldsacc + (sp - G::NREG) * G::ROW + ci * 8, *tc = tb + G::LSZ;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    float4 b4 = *reinterpret_cast<float4 *>(tb + 2 * q), c4 = *reinterpret_cast<float4 *>(tc + 2 * q);
                    const l3f2 b0 = dh[q] * vu[q], b1 = dh[q + 1] * vu[q + 1], c0 = g[q] * h[q], c1 = g[q + 1] * h[q + 1];
                    b4.x += b0.x; b4.y += b0.y; b4.z += b1.x; b4.w += b1.y;
                    c4.x += c0.x; c4.y += c0.y; c4.z += c1.x; c4.w += c1.y;
                    *reinterpret_cast<float4 *>(tb + 2 * q) = b4;
                    *reinterpret_cast<float4 *>(tc + 2 * q) = c4;
                }
            }
        }
        {
            // One statement per register row: a scalar compare-and-skip around eight packed FMAs INSIDE the statement, so that the
            // compiler sees a plain read-modify-write of the row's sums (as a switch over the rows it double-buffers all of them)
#define W_ACC_OPS(SS)                                                                                           \
                     : [b0] "+v"(rB[SS < G::NREG ? SS : 0][0]), [b1] "+v"(rB[SS < G::NREG ? SS : 0][1]),        \
                       [b2] "+v"(rB[SS < G::NREG ? SS : 0][2]), [b3] "+v"(rB[SS < G::NREG ? SS : 0][3]),        \
                       [c0] "+v"(rC[SS < G::NREG ? SS : 0][0]), [c1] "+v"(rC[SS < G::NREG ? SS : 0][1]),        \
                       [c2] "+v"(rC[SS < G::NREG ? SS : 0][2]), [c3] "+v"(rC[SS < G::NREG ? SS : 0][3])         \
                     : [d0] "v"(dh[0]), [d1] "v"(dh[1]), [d2] "v"(dh[2]), [d3] "v"(dh[3]), [u0] "v"(vu[0]),     \
                       [u1] "v"(vu[1]), [u2] "v"(vu[2]), [u3] "v"(vu[3]), [g0] "v"(g[0]), [g1] "v"(g[1]),       \
                       [g2] "v"(g[2]), [g3] "v"(g[3]), [h0] "v"(h[0]), [h1] "v"(h[1]), [h2] "v"(h[2]),          \
                       [h3] "v"(h[3])
#define W_ACC_FMAS                                                                                              \
                     "v_pk_fma_f32 %[b0], %[d0], %[u0], %[b0]\n\t"                                              \
                     "v_pk_fma_f32 %[c0], %[g0], %[h0], %[c0]\n\t"                                              \
                     "v_pk_fma_f32 %[b1], %[d1], %[u1], %[b1]\n\t"                                              \
                     "v_pk_fma_f32 %[c1], %[g1], %[h1], %[c1]\n\t"                                              \
                     "v_pk_fma_f32 %[b2], %[d2], %[u2], %[b2]\n\t"                                              \
                     "v_pk_fma_f32 %[c2], %[g2], %[h2], %[c2]\n\t"                                              \
                     "v_pk_fma_f32 %[b3], %[d3], %[u3], %[b3]\n\t"                                              \
                     "v_pk_fma_f32 %[c3], %[g3], %[h3], %[c3]\n\t"
#define W_ACC_ROW(SS)                                                                                           \
    if constexpr (SS < G::NREG) {                                                                               \
        if constexpr (is_static) {                                                                              \
            if (sp == SS) asm volatile(W_ACC_FMAS W_ACC_OPS(SS));                                               \
        } else {                                                                                                \
            asm volatile("s_cmp_lg_u32 %[sp], " #SS "\n\t"                                                      \
                         "s_cbranch_scc1 1f\n\t" W_ACC_FMAS "1:\n\t" W_ACC_OPS(SS), [sp] "s"(sp)                \
                         : "scc");                                                                              \
        }                                                                                                       \
    }
            W_ACC_ROW(0) W_ACC_ROW(1) W_ACC_ROW(2) W_ACC_ROW(3) W_ACC_ROW(4) W_ACC_ROW(5) W_ACC_ROW(6) W_ACC_ROW(7)
#undef W_ACC_ROW
#undef W_ACC_FMAS
#undef W_ACC_OPS
        }
    };
    constexpr std::false_type no{};
    constexpr std::true_type yes{};
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IL = std::integral_constant<int, NSEG - 1>;
    if constexpr (!G::HAS_TAIL) {
#pragma unroll 1
        for (int i = NSEG - 1; i >= 0; --i) row(no, no, i);
    } else if constexpr (T1 && !REV) {
        // the tail first; the plane's last row (i = 0) requests the next plane's tail
        row1(NSEG - 1);
#pragma unroll 1
        for (int i = NSEG - 2; i >= 1; --i) row(no, no, i);
        row(no, yes, I0{});
    } else if constexpr (T1) {
        // the tail last, requested by row 1
#pragma unroll 1
        for (int i = NSEG - 1; i >= 2; --i) row(no, no, i);
        row(no, yes, I1{});
        row1(0);
    } else if constexpr (NSEG == 2) {
        // two chunk rows per plane (28 x 28): both spelled out
        if constexpr (!REV) {
            row(yes, no, I1{});
            row(no, no, I0{});
        } else {
            row(no, no, I1{});
            row(yes, no, I0{});
        }
    } else if constexpr (!REV) {
        row(yes, no, IL{});
#pragma unroll 1
        for (int i = NSEG - 2; i >= 0; --i) row(no, no, i);
    } else {
#pragma unroll 1
        for (int i = NSEG - 1; i >= 1; --i) row(no, no, i);
        row(yes, no, I0{});
    }
    dA_acc = dA2.x + dA2.y;
    dD_acc = dD2.x + dD2.y;
    dbias_acc = db2.x + db2.y;
}

// first operands of a plane (route-order row NSEG - 1), requested ahead of the tile's staging
template <int HW, bool REV>
__device__ __forceinline__ void w_bwd_first(WOps &op, W1Ops &op1, const WBuf &rs, const int dso, const int bso, const int cso, const int lane) {
    using G = L3Geom<HW>;
    constexpr int NSEG = G::NSEG, sp = REV ? 0 : NSEG - 1;
    constexpr bool tail = G::HAS_TAIL && sp == NSEG - 1;
    const int ci = REV ? lane : 63 - lane;
    if constexpr (tail && WTail1<HW>::on) {
        // the one-position tail is the plane's first row: its operands, and those of the full row behind it
        op1.d = w_ld2(rs.dts, (uint32_t)ci * 2u, dso + sp * (G::ROW * 2));
        op1.b = w_ld4(rs.Bs, (uint32_t)ci * 4u, bso + sp * (G::ROW * 4));
        op1.c = w_ld4(rs.Cs, (uint32_t)ci * 4u, bso + sp * (G::ROW * 4));
        op1.h = w_ld4(rs.chk, (uint32_t)(63 - lane) * 4u, cso + (NSEG - 1) * 256);
        constexpr int spf = NSEG - 2;
        op.d = w_ld16(rs.dts, (uint32_t)ci * 16u, dso + spf * (G::ROW * 2));
        op.b0 = w_ld16(rs.Bs, (uint32_t)ci * 32u, bso + spf * (G::ROW * 4));
        op.b1 = w_ld16(rs.Bs, (uint32_t)ci * 32u, bso + spf * (G::ROW * 4) + 16);
        op.c0 = w_ld16(rs.Cs, (uint32_t)ci * 32u, bso + spf * (G::ROW * 4));
        op.c1 = w_ld16(rs.Cs, (uint32_t)ci * 32u, bso + spf * (G::ROW * 4) + 16);
        op.h = w_ld4(rs.chk, (uint32_t)(63 - lane) * 4u, cso + (NSEG - 2) * 256);
        return;
    }
    const bool live = !tail || ci < G::TAILV;
    const uint32_t lo16 = live ? (uint32_t)ci * 16u : kWDead, lo4 = live ? (uint32_t)(63 - lane) * 4u : kWDead;
    const uint32_t lo32 = live ? (uint32_t)ci * 32u : kWDead;
    op.d = w_ld16(rs.dts, lo16, dso + sp * (G::ROW * 2));
    op.b0 = w_ld16(rs.Bs, lo32, bso + sp * (G::ROW * 4));
    op.b1 = w_ld16(rs.Bs, lo32, bso + sp * (G::ROW * 4) + 16);
    op.c0 = w_ld16(rs.Cs, lo32, bso + sp * (G::ROW * 4));
    op.c1 = w_ld16(rs.Cs, lo32, bso + sp * (G::ROW * 4) + 16);
    op.h = w_ld4(rs.chk, lo4, cso + (NSEG - 1) * 256);
}

template <int HW, int PPT, bool REV>
__device__ __forceinline__ void w_bwd_body(const LeanArgs &a, float *smem, const int wave, const int lane) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, PL = PPT * L, NSEG = G::NSEG;
    const int D = a.D_;
    const int tiles_pb = D / PPT;
    const int groups_pb = tiles_pb / a.pli;
    int b, tg;
    l3_block_map(a, groups_pb, b, tg);
    // (uniform by construction, but formed on the vector ALU -- integer division -- and a vector register in a buffer
    //  instruction's scalar-offset slot is legalised with a readfirstlane LOOP per access)
    b = __builtin_amdgcn_readfirstlane(b);
    tg = __builtin_amdgcn_readfirstlane(tg);
    bf16_t *xN = reinterpret_cast<bf16_t *>(smem), *xT = xN + PL, *gN = xT + PL, *gT = gN + PL, *DX = gT + PL;
    float *ldsacc = smem + (8 * (size_t)PL * 2) / 4 + wave * 2 * G::LSZ;
    // behind the strips: the second natural x image and the raw dy of the tile being fetched (l3_dma_tile)
    bf16_t *xN1 = reinterpret_cast<bf16_t *>(smem + (8 * (size_t)PL * 2) / 4 + 4 * 2 * G::LSZ);
    float *graw = reinterpret_cast<float *>(xN1 + PL);
    for (int e = lane; e < 2 * G::LSZ; e += 64) ldsacc[e] = 0.f;
    l3f2 rB[G::NACC][4], rC[G::NACC][4];
    l3f2 rT = {0.f, 0.f};                              // WTail1: dB / dC sums of this lane's position of the tail row
#pragma unroll
    for (int s = 0; s < G::NACC; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) rB[s][q] = rC[s][q] = l3f2{0.f, 0.f};
    const bool col = wave >> 1;
    const int k = (REV ? 2 : 0) + (wave >> 1);
    const bf16_t *gq = col ? gT : gN;
    bf16_t *dxq = DX + (size_t)wave * PL;
    const int route = b * 4 + k;
    const WBuf rs = w_bufs(a, NSEG, true);
    const int bso = __builtin_amdgcn_readfirstlane(route * (L * 4));          // (fp32 rows)
    const int ci = REV ? lane : 63 - lane;
    {
        const int64_t po0 = ((int64_t)b * D + (int64_t)tg * a.pli * PPT) * L;
        l3_dma_tile<PL>((const bf16_t *)a.x + po0, (const float *)a.dy + po0, l3_lds_addr(xN), l3_lds_addr(graw), wave, lane);
    }
    WOps op;
    W1Ops op1;
#pragma unroll 1
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * PPT;
        const int64_t po = ((int64_t)b * D + d0) * L;
        int tz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));    // opaque zero: keeps the per-thread tile positions out of registers
        const int tid = threadIdx.x + tz;
        bf16_t *xNc = (it & 1) ? xN1 : xN;            // this tile's natural x image
        const bf16_t *xq = col ? xT : xNc;
        // tile `it` was sent for during tile it - 1 (or above): wait for this wave's pieces, then for everybody's
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the operands of the tile's first row: in flight under the staging
        w_bwd_first<HW, REV>(op, op1, rs, __builtin_amdgcn_readfirstlane((route * D + d0) * (L * 2)), bso,
                             __builtin_amdgcn_readfirstlane((route * D + d0) * (NSEG * 256)), lane);
        __syncthreads();                               // (also: the previous tile's merge has read the private planes)
        l3_stage_lds<HW, PPT>(xNc, xT, graw, gN, gT, tid);
        L3Next nx{};
        nx.x = (const bf16_t *)a.x + po + PL;
        nx.g = (const float *)a.dy + po + PL;
        nx.xdst = l3_lds_addr((it & 1) ? xN : xN1);
        nx.gdst = l3_lds_addr(graw);
        nx.wave = wave;
        nx.go = it + 1 < a.pli;
        __syncthreads();
#pragma unroll 1
        for (int pl = 0; pl < PPT; ++pl) {
            const int d = d0 + pl, row = k * D + d;
            const float An = a.A[row], Dr = a.D[row];
            float dA_acc, dD_acc, dbias_acc;
            w_bwd_plane<HW, REV, PL>(rs, __builtin_amdgcn_readfirstlane((route * D + d) * (L * 2)), bso,
                                     __builtin_amdgcn_readfirstlane((route * D + d) * (NSEG * 256)), pl + 1 < PPT, An, Dr, xq + pl * L, gq + pl * L, dxq + pl * L, ldsacc, rB, rC, rT,
                                     dA_acc, dD_acc, dbias_acc, lane, op, op1, nx);
            nx.go = false;
            // the plane's three parameter-gradient sums: DPP adds (no LDS round trips: a plane is 2 chunk rows at 28 x 28), and ONE
            // atomic instruction of three lanes (the atomic unit retires instructions, not lanes)
            w_wave_sum3(dA_acc, dD_acc, dbias_acc);
            {
                const float tA = bcast_lane<63>(dA_acc), tD = bcast_lane<63>(dD_acc), tb = bcast_lane<63>(dbias_acc);
                if (lane < 3) {
                    float *dst = (lane == 0 ? a.dA : (lane == 1 ? a.dD : a.dbias)) + row;
                    atomicAdd(dst, lane == 0 ? tA : (lane == 1 ? tD : tb));
                }
            }
        }
        __syncthreads();
        {
            int tz2;
            asm volatile("v_mov_b32 %0, 0" : "=v"(tz2));
            if constexpr (HW % 8 == 0) {
                // the sums into the (now free) natural x image in the conflict-free order, then out in memory order
                l3_merge_store<HW, PPT>(xNc, DX, DX + PL, DX + 2 * PL, DX + 3 * PL, threadIdx.x + tz2);
                __syncthreads();
                for (int v = threadIdx.x + tz2; v < PL / 8; v += 256)
                    *reinterpret_cast<uint4 *>((bf16_t *)a.dx + po + v * 8) = *reinterpret_cast<const uint4 *>(xNc + v * 8);
            } else {
                l3_merge_store<HW, PPT>((bf16_t *)a.dx + po, DX, DX + PL, DX + 2 * PL, DX + 3 * PL, threadIdx.x + tz2);
            }
        }
    }
    // ---- flush: registers hold [chunk row][traversal element] of this lane.  The sums are first laid out by position in LDS (the
    // plane region is free), then leave as plain coalesced stores of this workgroup's partial rows (ss2d_l3_parts_kernel adds the
    // groups of a sample up) or, without a workspace, as contiguous float atomics (only fast when a wave instruction covers
    // contiguous bytes).
    // (Tried and withdrawn: the partial sums leaving the registers as they sit -- chunk row, lane, traversal element -- with a
    //  summing kernel that undoes the order: 3 us faster at 56 x 56 and, with a second process on the GPU, wrong dB / dC values
    //  at chunks 48 .. 63 of the register rows in 0.1 - 25 % of the launches, in every store form tried (16- / 8-byte global,
    //  buffer stores from untouched registers, host synchronisation between the two kernels); this form: 0 of 12 000.)
    __syncthreads();
    float *dBg = a.dBs + (int64_t)route * L, *dCg = a.dCs + (int64_t)route * L;
    float *stage = smem + (size_t)wave * L;            // 4 waves x L floats <= 8 PL bf16 (PPT >= 1: 16 L bytes)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int s = 0; s < G::NREG; ++s) {
            const int tp0 = s * G::ROW + ci * 8;
            if (WTail1<HW>::on && s == NSEG - 1) continue;      // (that row's sums: rT, below)
            if (tp0 < L) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const l3f2 v = pass ? rC[s][q] : rB[s][q];
                    stage[tp0 + (REV ? 7 - 2 * q : 2 * q)] = v.x;
                    stage[tp0 + (REV ? 6 - 2 * q : 2 * q + 1)] = v.y;
                }
            }
        }
        // the LDS strip holds [chunk][traversal element]; one position per lane (WTail1): the tail row's sums are a register pair
        if constexpr (WTail1<HW>::on) {
            stage[(NSEG - 1) * G::ROW + ci] = pass ? rT.y : rT.x;
        } else {
            for (int e = lane; e < G::LSZ; e += 64)
                stage[G::NREG * G::ROW + (e & ~7) + (REV ? 7 - (e & 7) : (e & 7))] = ldsacc[pass * G::LSZ + e];
        }
        wave_sync();
        if (a.parts) {
            float *dp = a.parts + ((((int64_t)b * groups_pb + tg) * 4 + k) * 2 + pass) * L;
            for (int e = lane; e < L; e += 64) dp[e] = stage[e];
        } else {
            float *dstp = pass ? dCg : dBg;
            for (int e = lane; e < L; e += 64) atomicAdd(dstp + e, stage[e]);
        }
        wave_sync();
    }
}

template <int HW, int PPT>
__global__ void __launch_bounds__(256, 2) ss2d_w_bwd_kernel(const LeanArgs a) {
    static_assert(L3Geom<HW>::L % 8 == 0 && L3Geom<HW>::NSEG <= 8, "map size not covered");
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave & 1) w_bwd_body<HW, PPT, true>(a, smem, wave, lane);
    else w_bwd_body<HW, PPT, false>(a, smem, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, one route over one plane: as l3_fwd_plane (mode 2) with the chunk rows unrolled statically, and the state entering
// every lane's chunk stored for the backward (chk[(route * D + d) * NSEG * 64 + row * 64 + lane], route order)
// ---------------------------------------------------------------------------------------------------------------------
struct WFOps { uint4 d, b, c; };

template <int HW, bool REV>
__device__ __forceinline__ void w_fwd_first(WFOps &op, const WBuf &rs, const int dso, const int bso, const int lane) {
    using G = L3Geom<HW>;
    constexpr int NSEG = G::NSEG, sp = REV ? NSEG - 1 : 0;
    constexpr bool tail = G::HAS_TAIL && sp == NSEG - 1;
    const int ci = REV ? 63 - lane : lane;
    if constexpr (tail && WTail1<HW>::on) {
        op.d.x = w_ld2(rs.dts, (uint32_t)ci * 2u, dso + sp * (G::ROW * 2));
        op.b.x = w_ld2(rs.Bs, (uint32_t)ci * 2u, bso + sp * (G::ROW * 2));
        op.c.x = w_ld2(rs.Cs, (uint32_t)ci * 2u, bso + sp * (G::ROW * 2));
        return;
    }
    const uint32_t lo16 = (!tail || ci < G::TAILV) ? (uint32_t)ci * 16u : kWDead;
    op.d = w_ld16(rs.dts, lo16, dso + sp * (G::ROW * 2));
    op.b = w_ld16(rs.Bs, lo16, bso + sp * (G::ROW * 2));
    op.c = w_ld16(rs.Cs, lo16, bso + sp * (G::ROW * 2));
}

template <int HW, bool REV>
__device__ __forceinline__ void w_fwd_plane(const WBuf &rs, const int dso, const int bso, const int cso, const bool chain,
                                            const float A2, const float Dr, const bf16_t *xq, bf16_t *yq, const int lane,
                                            WFOps &op) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, NSEG = G::NSEG;
    constexpr bool T1 = WTail1<HW>::on;
    const uint16_t *xq16 = reinterpret_cast<const uint16_t *>(xq);
    uint16_t *yq16 = reinterpret_cast<uint16_t *>(yq);
    const int ci = REV ? 63 - lane : lane;            // the lanes run WITH the route
    const bool tail_live = !G::HAS_TAIL || ci < G::TAILV;
    const uint32_t lo16 = (uint32_t)ci * 16u, lo4 = (uint32_t)lane * 4u;
    const uint32_t lo16t = tail_live ? lo16 : kWDead, lo4t = tail_live ? lo4 : kWDead;
    const bf16_t *xql = xq + ci * 8;
    bf16_t *yql = yq + ci * 8;
    float hc = 0.f;                                   // state entering the chunk row
    uint4 lx;
    {
        constexpr int sp0 = REV ? NSEG - 1 : 0;
        if constexpr (T1 && sp0 == NSEG - 1) {
            lx = make_uint4(0, 0, 0, 0);
            lx.x = xq16[sp0 * G::ROW + ci];
        } else {
            lx = (G::HAS_TAIL && sp0 == NSEG - 1 && !tail_live) ? make_uint4(0, 0, 0, 0)
                                                                : *reinterpret_cast<const uint4 *>(xql + sp0 * G::ROW);
        }
    }
    w_static_for<NSEG>([&](auto ic) {
        constexpr int i = decltype(ic)::value;        // route-order chunk row
        constexpr int sp = REV ? NSEG - 1 - i : i;
        constexpr bool is_tail = G::HAS_TAIL && sp == NSEG - 1;
        constexpr bool one = T1 && is_tail;           // this row runs one position per lane
        l3f2 v[4], u[4], Bq[4], Cq[4];
        float v1 = 0.f, u1 = 0.f, B1 = 0.f, C1 = 0.f;
        if constexpr (one) {
            v1 = __uint_as_float(op.d.x << 16);
            B1 = __uint_as_float(op.b.x << 16);
            C1 = __uint_as_float(op.c.x << 16);
            u1 = __uint_as_float(lx.x << 16);
        } else {
            l3_unpack<REV>(op.d, v);
            l3_unpack<REV>(op.b, Bq);
            l3_unpack<REV>(op.c, Cq);
            l3_unpack<REV>(lx, u);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + 1 < NSEG) {
            constexpr int spn = REV ? NSEG - 2 - i : i + 1;
            constexpr bool tail_n = G::HAS_TAIL && spn == NSEG - 1;
            if constexpr (T1 && tail_n) {
                op.d.x = w_ld2(rs.dts, (uint32_t)ci * 2u, dso + spn * (G::ROW * 2));
                op.b.x = w_ld2(rs.Bs, (uint32_t)ci * 2u, bso + spn * (G::ROW * 2));
                op.c.x = w_ld2(rs.Cs, (uint32_t)ci * 2u, bso + spn * (G::ROW * 2));
                lx.x = xq16[spn * G::ROW + ci];
            } else {
                op.d = w_ld16(rs.dts, tail_n ? lo16t : lo16, dso + spn * (G::ROW * 2));
                op.b = w_ld16(rs.Bs, tail_n ? lo16t : lo16, bso + spn * (G::ROW * 2));
                op.c = w_ld16(rs.Cs, tail_n ? lo16t : lo16, bso + spn * (G::ROW * 2));
                if (!tail_n || tail_live) lx = *reinterpret_cast<const uint4 *>(xql + spn * G::ROW);
                else lx = make_uint4(0, 0, 0, 0);
            }
        } else {
            if (chain) w_fwd_first<HW, REV>(op, rs, dso + L * 2, bso, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (one) {
            const float a1 = exp2_fast(v1 * A2), bb1 = v1 * u1 * B1;
            float P = a1, S = bb1;
            l3_scan_up(P, S);
            const float th = fmaf(P, hc, S);          // state after this position
            const float hh = dpp_mov<kWaveShr1>(hc, th);
            hc = bcast_lane<63>(th);
            w_st4(rs.chk, lo4, cso + i * 256, hh);    // the state ENTERING every position of the row
            const float y1 = fmaf(C1, th, u1 * Dr);
            yq16[sp * G::ROW + ci] = (uint16_t)(l3_cvt_pk(y1, 0.f) & 0xffffu);
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        l3f2 a[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = l3_exp2(v[q] * A2);
            bb[q] = v[q] * u[q] * Bq[q];
        }
        float P = 1.f, S = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            S = fmaf(a[q].x, S, bb[q].x);
            S = fmaf(a[q].y, S, bb[q].y);
        }
        P = (a[0].x * a[0].y) * (a[1].x * a[1].y) * ((a[2].x * a[2].y) * (a[3].x * a[3].y));
        l3_scan_up(P, S);
        const float th = fmaf(P, hc, S);              // state after this lane's chunk
        float hh = dpp_mov<kWaveShr1>(hc, th);        // state entering it
        hc = bcast_lane<63>(th);
        w_st4(rs.chk, is_tail ? lo4t : lo4, cso + i * 256, hh);
        l3f2 y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hh = fmaf(a[q].x, hh, bb[q].x);
            const float h0 = hh;
            hh = fmaf(a[q].y, hh, bb[q].y);
            y[q] = __builtin_elementwise_fma(Cq[q], l3f2{h0, hh}, u[q] * Dr);
        }
        if (!is_tail || tail_live) *reinterpret_cast<uint4 *>(yql + sp * G::ROW) = l3_pack<REV>(y);
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int HW, int PPT, bool REV>
__device__ __forceinline__ void w_fwd_body(const LeanArgs &a, float *smem, const int wave, const int lane) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, PL = PPT * L, NSEG = G::NSEG;
    const int D = a.D_;
    const int tiles_pb = D / PPT;
    const int groups_pb = tiles_pb / a.pli;
    int b, tg;
    l3_block_map(a, groups_pb, b, tg);
    // (uniform by construction, but formed on the vector ALU -- integer division -- and a vector register in a buffer
    //  instruction's scalar-offset slot is legalised with a readfirstlane LOOP per access)
    b = __builtin_amdgcn_readfirstlane(b);
    tg = __builtin_amdgcn_readfirstlane(tg);
    // LDS: xN | xT | 4 private y planes (bf16: the per-route partial sums are rounded to the I/O precision once, before
    // the fixed-order fp32 merge, as in the lean kernels)
    bf16_t *xN = reinterpret_cast<bf16_t *>(smem), *xT = xN + PL, *Y = xT + PL;
    const bool col = wave >> 1;
    const int k = (REV ? 2 : 0) + (wave >> 1);
    const bf16_t *xq = col ? xT : xN;
    bf16_t *yq = Y + (size_t)wave * PL;
    const int route = b * 4 + k;
    const WBuf rs = w_bufs(a, NSEG, false);
    const int bso = __builtin_amdgcn_readfirstlane(route * (L * 2));
    WFOps op;
    constexpr int NVX = 2;
    static_assert(PL <= 4096, "tile beyond the staging registers");
#pragma unroll 1
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * PPT;
        const int64_t po = ((int64_t)b * D + d0) * L;
        int tz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));
        const int tid = threadIdx.x + tz;
        // the operands of the tile's first row: in flight under the staging
        w_fwd_first<HW, REV>(op, rs, __builtin_amdgcn_readfirstlane((route * D + d0) * (L * 2)), bso, lane);
        if constexpr (HW % 8 == 0) {
            constexpr int nvx = PL / 8;
            static_assert(NVX == 2, "staging registers are spelled out");
            const bf16_t *xs_ = (const bf16_t *)a.x + po;
            const int v0 = tid, v1 = tid + 256;
            const uint4 px0 = *reinterpret_cast<const uint4 *>(xs_ + (v0 < nvx ? v0 : 0) * 8);
            const uint4 px1 = *reinterpret_cast<const uint4 *>(xs_ + (v1 < nvx ? v1 : 0) * 8);
            __syncthreads();                           // (the previous tile's output pass has read xN / xT)
            if (v0 < nvx) *reinterpret_cast<uint4 *>(xN + v0 * 8) = px0;
            if (v1 < nvx) *reinterpret_cast<uint4 *>(xN + v1 * 8) = px1;
            __syncthreads();
            l3_transpose_pass<HW, PPT>(xN, xT, tid);
        } else {
            PlaneRegs<bf16_t, 8, NVX> px;
            l3_planes_issue<HW, PPT, bf16_t, 8, NVX>(px, (const bf16_t *)a.x + po, tid);
            __syncthreads();
            l3_planes_commit<HW, PPT, bf16_t, 8, NVX>(px, xN, xT, tid);
        }
        __syncthreads();
#pragma unroll 1
        for (int pl = 0; pl < PPT; ++pl) {
            const int d = d0 + pl, row = k * D + d;
            const float A2 = a.A[row] * kLog2e, Dr = a.D[row];
            w_fwd_plane<HW, REV>(rs, __builtin_amdgcn_readfirstlane((route * D + d) * (L * 2)), bso,
                                 __builtin_amdgcn_readfirstlane((route * D + d) * (NSEG * 256)), pl + 1 < PPT, A2, Dr, xq + pl * L, yq + pl * L, lane, op);
        }
        __syncthreads();
        int tz2;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz2));
        if constexpr (HW % 8 == 0 && PPT == 1) {
            // the fp32 sums into the (now free) x images in the conflict-free order, then out in memory order
            float *stage = reinterpret_cast<float *>(xN);                       // xN | xT = L floats
            l3_merge_y<HW, PPT>(stage, Y, Y + PL, Y + 2 * PL, Y + 3 * PL, threadIdx.x + tz2);
            __syncthreads();
            for (int v = threadIdx.x + tz2; v < PL / 4; v += 256)
                *reinterpret_cast<float4 *>((float *)a.y + po + v * 4) = *reinterpret_cast<const float4 *>(stage + v * 4);
        } else {
            l3_merge_y<HW, PPT>((float *)a.y + po, Y, Y + PL, Y + 2 * PL, Y + 3 * PL, threadIdx.x + tz2);
        }
    }
}

template <int HW, int PPT>
__global__ void __launch_bounds__(256, L3_WPE_FWD) ss2d_w_fwd_kernel(const LeanArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave & 1) w_fwd_body<HW, PPT, true>(a, smem, wave, lane);
    else w_fwd_body<HW, PPT, false>(a, smem, wave, lane);
}

}  // namespace xfm

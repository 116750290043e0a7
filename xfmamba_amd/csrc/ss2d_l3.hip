// ss2d_l3.hip -- fused SS2D scan kernels for the WIDE maps of the trunk (56 x 56, 28 x 28; d_state 1, bf16 I/O), built
// for TWO waves per SIMD.  Same algorithm, layout contract and checkpoint format as ss2d_lean.hpp (reference
// models/fusion_vmamba.py:1145-1174; adjoint per selective_scan_bwd_kernel.cuh:141-273), so either forward pairs with
// either backward.  What changed against the lean kernels (one wave per SIMD: 100 KB of dB/dC sums in LDS, 388 VGPRs):
//   * the map size is a template parameter: chunk-row count, tail lanes and every address stride are constants;
//   * the dB / dC sums of a route over the planes a workgroup walks live in REGISTERS (one jump per chunk row into a
//     block of static-register adds; the 8-lane tail row of a 56 x 56 map in a 512-byte LDS strip), which leaves
//     50 KB of LDS per workgroup -> two workgroups per CU, and the kernel is held to 256 VGPRs;
//   * both wave scans of a chunk row (states ascending, adjoints descending) are ONE block of DPP-fused VALU
//     instructions (v_fmac_f32_dpp / v_mul_f32_dpp: a scan step is two instructions, not a move pair plus two), the
//     two chains interleaved so that no DPP source is read within two instructions of its write; the descending
//     cross-row steps run under EXEC row masks with the row totals in SGPRs (v_readlane);
//   * the carries enter through one FMA and a one-lane wave shift instead of exclusive-prefix shifts of both halves
//     of the map;
//   * softplus handling is a template parameter (1: in-kernel with bias, 2: the step sizes arrive activated,
//     3: dt_proj INSIDE the kernel -- SURVEY 8(f) rank 1, reference models/fusion_vmamba.py:1147-1150: the step size
//     softplus(W_dt . xr + bias) is formed per position from the dt_proj input rows `xrt` (batch, 4, L, Rp) -- position-major,
//     route order, the same for all channels of a route, so they come out of L2 -- with v_dot2_f32_bf16 against the channel's
//     Rp weights held as bf16 pairs in scalar registers: Rp / 2 instructions per position, no unpack.  The (B, 4, D, L) step
//     sizes are never stored: 8 of the 14 (forward) / 24 (backward) bytes per element at the "dts" boundary are gone).
// Roofline: HBM (24 B per (b,d,p) element backward, 14 B forward at this boundary: ss2d_kernels.hpp).
#include <cstdlib>

#include "ss2d_kernels.hpp"

namespace xfm {

#ifndef L3_NREG
#define L3_NREG 6
#endif
#ifndef L3_WPE_FWD
#define L3_WPE_FWD 4
#endif
#ifndef L3_WPE
#define L3_WPE 2
#endif
#ifndef L3_PREFETCH
#define L3_PREFETCH 1                                  // backward: the next tile's planes by LDS-direct loads (l3_dma_tile)
#endif
typedef float l3f2 __attribute__((ext_vector_type(2)));

// RP2: dt_rank pairs of the mode-3 kernels (0 otherwise); they hold 2 RP2 operand vectors per chunk row in flight, so the
// 56 x 56 backward keeps one chunk row less of the dB / dC sums in registers (row 5 joins the tail row in the LDS strip:
// 68.6 KB per workgroup, still two per CU)
template <int HW, int RP2 = 0> struct L3Geom {
    static constexpr int L = HW * HW;
    static constexpr int ROW = 512;                             // positions per chunk row: 64 lanes x 8
    static constexpr int NSEG = (L + ROW - 1) / ROW;
    static constexpr int TAILV = (L - (NSEG - 1) * ROW) / 8;    // live lanes of the last chunk row
    static constexpr bool HAS_TAIL = TAILV < 64;
    // dB / dC sums of a route over the planes of a workgroup: the first NREG chunk rows in registers (16 per row), the
    // rest in a wave-private LDS strip (56 x 56: row 5 and the 8-lane tail row -- 256 registers hold five rows next to
    // the working set, and 4.6 KB per wave still leave two workgroups per CU)
    static constexpr int NREG_MAX = L3_NREG - ((RP2 > 0 && NSEG > 5) ? 1 : 0);
    static constexpr int NREG = NSEG < NREG_MAX ? NSEG : NREG_MAX;
    static constexpr int NACC = NREG;
    static constexpr int LSZ = NSEG > NREG ? L - NREG * ROW : 0;   // positions whose sums live in LDS
};

// Inclusive scan of the affine maps (P, S) over ascending lanes and of (Q, R) over descending lanes.
// A step of the ascending scan is  S += S[lane - d] * P ; P *= P[lane - d]  (lanes without a source are disabled by the
// DPP bound control: the identity), the descending one mirrors it with row_shl.  The two chains alternate, so every DPP
// source was written at least three instructions earlier (the hardware needs two wait states).  EXEC must be all ones.
__device__ __forceinline__ void l3_scan_pair(float &P, float &S, float &Q, float &R) {
    uint32_t q1, r1, q2, r2, q3, r3;
#define L3_STEP(N)                                                                   \
    "v_fmac_f32_dpp %[S], %[S], %[P] row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_fmac_f32_dpp %[R], %[R], %[Q] row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t"  \
    "v_mul_f32_dpp %[P], %[P], %[P] row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
    "v_mul_f32_dpp %[Q], %[Q], %[Q] row_shl:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile(
        "s_nop 1\n\t"
        L3_STEP(1) L3_STEP(2) L3_STEP(4) L3_STEP(8)
        // ascending cross-row steps (lane 15 of a row -> the next row; lane 31 -> rows 2, 3), the row totals of the
        // descending scan (first lane of rows 1..3) read in between
        "v_fmac_f32_dpp %[S], %[S], %[P] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_readlane_b32 %[q1], %[Q], 16\n\t"
        "v_readlane_b32 %[r1], %[R], 16\n\t"
        "v_mul_f32_dpp %[P], %[P], %[P] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_readlane_b32 %[q3], %[Q], 48\n\t"
        "v_readlane_b32 %[r3], %[R], 48\n\t"
        "v_fmac_f32_dpp %[S], %[S], %[P] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_dpp %[P], %[P], %[P] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        // descending cross-row steps under EXEC row masks: rows 0 / 2 take the totals of rows 1 / 3, then rows 0, 1
        // take the total of rows 2, 3
        "s_mov_b32 exec_lo, 0xffff\n\t"
        "s_mov_b32 exec_hi, 0\n\t"
        "v_fmac_f32 %[R], %[r1], %[Q]\n\t"
        "v_mul_f32 %[Q], %[q1], %[Q]\n\t"
        "s_mov_b32 exec_lo, 0\n\t"
        "s_mov_b32 exec_hi, 0xffff\n\t"
        "v_fmac_f32 %[R], %[r3], %[Q]\n\t"
        "v_mul_f32 %[Q], %[q3], %[Q]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_readlane_b32 %[q2], %[Q], 32\n\t"
        "v_readlane_b32 %[r2], %[R], 32\n\t"
        "s_mov_b32 exec_hi, 0\n\t"
        "v_fmac_f32 %[R], %[r2], %[Q]\n\t"
        "v_mul_f32 %[Q], %[q2], %[Q]\n\t"
        "s_mov_b64 exec, -1\n\t"
        : [P] "+v"(P), [S] "+v"(S), [Q] "+v"(Q), [R] "+v"(R), [q1] "=&s"(q1), [r1] "=&s"(r1), [q2] "=&s"(q2),
          [r2] "=&s"(r2), [q3] "=&s"(q3), [r3] "=&s"(r3));
#undef L3_STEP
}

// ascending scan only (forward kernel)
__device__ __forceinline__ void l3_scan_up(float &P, float &S) {
#define L3_STEP(N)                                                                   \
    "v_fmac_f32_dpp %[S], %[S], %[P] row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"  \
    "s_nop 0\n\t"                                                                    \
    "v_mul_f32_dpp %[P], %[P], %[P] row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"   \
    "s_nop 0\n\t"
    asm volatile(
        "s_nop 1\n\t"
        L3_STEP(1) L3_STEP(2) L3_STEP(4) L3_STEP(8)
        "v_fmac_f32_dpp %[S], %[S], %[P] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %[P], %[P], %[P] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_fmac_f32_dpp %[S], %[S], %[P] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %[P], %[P], %[P] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        : [P] "+v"(P), [S] "+v"(S));
#undef L3_STEP
}

// 8 bf16 of a 16-byte vector -> four pairs in TRAVERSAL order (a descending route walks the vector backwards)
template <bool REV> __device__ __forceinline__ void l3_unpack(const uint4 &r, l3f2 (&o)[4]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = w[REV ? 3 - q : q];
        const float lo = __uint_as_float(x << 16), hi = __uint_as_float(x & 0xffff0000u);
        o[q] = REV ? l3f2{hi, lo} : l3f2{lo, hi};
    }
}
// (the conversion names its two sources itself: through a two-element vector conversion the descending routes, whose pairs sit
//  in registers high element first, paid a v_pk_mov_b32 half swap in front of every v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t l3_cvt_pk(const float lo, const float hi) {
    uint32_t w;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
    return w;
}
template <bool REV> __device__ __forceinline__ uint4 l3_pack(const l3f2 (&v)[4]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const l3f2 p = v[REV ? 3 - i : i];
        w[i] = REV ? l3_cvt_pk(p.y, p.x) : l3_cvt_pk(p.x, p.y);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ l3f2 l3_exp2(const l3f2 t) { return l3f2{exp2_fast(t.x), exp2_fast(t.y)}; }

// operands of one chunk row, requested one row ahead.  d: the step-size vector (NX = 1), or in mode 3 the lane's piece of
// the dt_proj input rows: 8 positions x Rp bf16 = NX = Rp / 4 vectors, contiguous
template <int NX> struct L3Ops { uint4 d[NX]; uint4 b, c; float h; };
// Mode 3 (NX > 1): the rows are stored BLOCKED by this file's chunk geometry -- (chunk row, piece j, chunk c, 8 values) with
// piece j of chunk c = values 8 j .. 8 j + 7 of the chunk's 8 positions x Rp ranks (position-major) -- so that load
// instruction j of a wave is one contiguous KB (lane = chunk).  Position-major rows read directly put a lane's 16-byte pieces
// 16 Rp bytes apart: 48 different 128-byte lines per instruction, each fetched by 6 instructions -- measured 126 vs 88 us for the
// 56 x 56 forward, the texture-address path being the bound.
template <int NX> __device__ __forceinline__ void l3_ops_load(L3Ops<NX> &op, const bf16_t *drow, const bf16_t *Brow,
                                                              const bf16_t *Crow, const int tp) {
    if constexpr (NX == 1) {
        op.d[0] = *reinterpret_cast<const uint4 *>(drow + tp);
    } else {
        const bf16_t *blk = drow + (int64_t)(tp >> 9) * (512 * NX) + ((tp >> 3) & 63) * 8;   // chunk row, chunk
#pragma unroll
        for (int j = 0; j < NX; ++j) op.d[j] = *reinterpret_cast<const uint4 *>(blk + j * 512);
    }
    op.b = *reinterpret_cast<const uint4 *>(Brow + tp);
    op.c = *reinterpret_cast<const uint4 *>(Crow + tp);
}
template <int NX> __device__ __forceinline__ void l3_ops_zero(L3Ops<NX> &op) {
#pragma unroll
    for (int j = 0; j < NX; ++j) op.d[j] = make_uint4(0, 0, 0, 0);
    op.b = op.c = make_uint4(0, 0, 0, 0);
}

// mode 3: raw step size of the lane's 8 positions in TRAVERSAL order = bias + sum_j <xr pair j of the position, weight pair j>
typedef __bf16 l3bf2 __attribute__((ext_vector_type(2)));
template <int RP2, bool REV>
__device__ __forceinline__ void l3_delta_raw(const uint4 (&d)[2 * RP2], const uint32_t (&wp)[RP2], const float bias, l3f2 (&v)[4]) {
    uint32_t w[8 * RP2];
#pragma unroll
    for (int j = 0; j < 2 * RP2; ++j) {
        w[4 * j] = d[j].x; w[4 * j + 1] = d[j].y; w[4 * j + 2] = d[j].z; w[4 * j + 3] = d[j].w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int p = REV ? 7 - e : e;
        float s = bias;
#pragma unroll
        for (int j = 0; j < RP2; ++j)
            s = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const l3bf2 *>(&w[p * RP2 + j]),
                                                *reinterpret_cast<const l3bf2 *>(&wp[j]), s, false);
        if (e & 1) v[e >> 1].y = s; else v[e >> 1].x = s;
    }
}
// softplus (threshold 20, reference models/csms6s.py:49-50) of a step size that was formed from bf16 operands: log2(1 + z)
// straight from v_log_f32 (relative error ~6e-8 / z: far inside the bf16 bound for every step size above 1e-4), no series
// Without selects: with r = x log2(e), log2(1 + 2^r) >= r and equals r in fp32 from r = 25 on, so max(log2(1 + 2^min(r, 64)), r)
// IS the thresholded softplus to the last bit that matters (log1p(e^x) - x < 2.1e-9 for x > 20) and never overflows.
__device__ __forceinline__ float l3_softplus(const float x) {
    const float r = x * kLog2e;
    const float t = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(fminf(r, 64.f)));
    return fmaxf(t, r) * 0.6931471805599453f;
}
// ... and d softplus / d x = 1 - 1 / (1 + 2^r)  (1 for large r: rcp of 2^64 is 5e-20)
__device__ __forceinline__ float l3_softplus_sig(const float x, float &sig) {
    const float r = x * kLog2e;
    const float zp1 = 1.0f + __builtin_amdgcn_exp2f(fminf(r, 64.f));
    sig = 1.0f - __builtin_amdgcn_rcpf(zp1);
    return fmaxf(__builtin_amdgcn_logf(zp1), r) * 0.6931471805599453f;
}
// the RP2 weight pairs of channel row `row` (wave-uniform: scalar loads, scalar registers)
template <int RP2> __device__ __forceinline__ void l3_weight_pairs(const void *dtw, const int row, uint32_t (&wp)[RP2 ? RP2 : 1]) {
    if constexpr (RP2 > 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(dtw) + (int64_t)__builtin_amdgcn_readfirstlane(row) * RP2;
#pragma unroll
        for (int j = 0; j < RP2; ++j) wp[j] = __builtin_amdgcn_readfirstlane(src[j]);
    } else {
        wp[0] = 0;
    }
}
struct L3Lds { uint4 x, g; };

// ---------------------------------------------------------------------------------------------------------------------
// plane staging (natural + transposed bf16 copies in LDS) and the merge of the four private planes, for a compile-time
// map size.  `tid` is the thread index plus an opaque zero taken inside the tile loop: the per-thread element positions
// are a handful of constant divisions, and hoisted out of the tile loop they would sit in ~40 registers through the sweeps.
// ---------------------------------------------------------------------------------------------------------------------
// element position of vector v (VS elements) of a tile: when the rows are whole vectors consecutive lanes take the SAME
// column block of consecutive rows, so that the 2-byte writes of the transposed copy (and the 2-byte reads of the merge)
// of a wave instruction fall on consecutive addresses (ss2d_lean.hpp: lean_vec_pos)
template <int HW, int VS> __device__ __forceinline__ void l3_vec_pos(const int v, int &e0, int &pl, int &h, int &w) {
    constexpr int L = HW * HW;
    const int idx = v * VS;
    pl = idx / L;
    const int ep = idx - pl * L;
    if constexpr (HW % VS == 0) {
        const int r = ep / VS;
        const int wb = r / HW;
        h = r - wb * HW;
        w = wb * VS;
    } else {
        h = ep / HW;
        w = ep - h * HW;
    }
    e0 = pl * L + h * HW + w;
}

template <int HW, int PPT, typename S, int VS, int NV>
__device__ __forceinline__ void l3_planes_issue(PlaneRegs<S, VS, NV> &r, const S *src, const int tid) {
    constexpr int nvec = PPT * HW * HW / VS;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int v = tid + m * 256;
        if (v < nvec) {
            int e0, pl, h, w;
            l3_vec_pos<HW, VS>(v, e0, pl, h, w);
            r.v[m] = *reinterpret_cast<const typename VecIO<S, VS>::V *>(src + e0);
        }
    }
}

template <int HW, int PPT, typename S, int VS, int NV>
__device__ __forceinline__ void l3_planes_commit(const PlaneRegs<S, VS, NV> &r, bf16_t *nat, bf16_t *tr, const int tid) {
    constexpr int L = HW * HW, nvec = PPT * L / VS;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int v = tid + m * 256;
        if (v >= nvec) continue;
        int e0, pl, h, w;
        l3_vec_pos<HW, VS>(v, e0, pl, h, w);
        float f[VS];
        VecIO<S, VS>::unpack(r.v[m], f);
        *reinterpret_cast<typename VecIO<bf16_t, VS>::V *>(nat + e0) = VecIO<bf16_t, VS>::pack(f);
#pragma unroll
        for (int q = 0; q < VS; ++q) {
            tr[pl * L + w * HW + h] = from_float<bf16_t>(f[q]);
            if constexpr (HW % VS != 0) {
                if (++w == HW) {
                    w = 0;
                    ++h;
                }
            } else {
                ++w;
            }
        }
    }
}

// Maps whose rows are whole 8-element vectors (56 x 56): two passes.  Pass A loads the tile in MEMORY order (1 KB per wave
// instruction; the lean kernels' one-pass form asks for 16-byte pieces a row apart and spends 77 us of a 300 us launch
// there) and writes the natural images; pass B re-reads them with consecutive lanes on the same column block of
// consecutive rows (16-byte reads 112 bytes apart: conflict-free per 16-lane group) and scatters the transposed images
// with 2-byte writes that fall on consecutive addresses.
template <int HW, int PPT>
__device__ __forceinline__ void l3_transpose_pass(const bf16_t *nat, bf16_t *tr, const int tid) {
    constexpr int L = HW * HW, nvec = PPT * L / 8;
    for (int v = tid; v < nvec; v += 256) {
        int e0, pl, h, w;
        l3_vec_pos<HW, 8>(v, e0, pl, h, w);
        const uint4 r = *reinterpret_cast<const uint4 *>(nat + e0);
        const uint32_t wd[4] = {r.x, r.y, r.z, r.w};
        uint16_t *t = reinterpret_cast<uint16_t *>(tr) + pl * L + w * HW + h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t[(2 * q) * HW] = (uint16_t)(wd[q] & 0xffffu);
            t[(2 * q + 1) * HW] = (uint16_t)(wd[q] >> 16);
        }
    }
}

// dx = sum of the four private planes (routes 0 / 2 natural, 1 / 3 transposed), fixed order, 16-byte stores
template <int HW, int PPT>
__device__ __forceinline__ void l3_merge_store(bf16_t *out, const bf16_t *P0, const bf16_t *P1, const bf16_t *P2,
                                               const bf16_t *P3, const int tid) {
    constexpr int L = HW * HW, PL = PPT * L;
    for (int v = tid; v < PL / 8; v += 256) {
        int e0, pl, h, w;
        l3_vec_pos<HW, 8>(v, e0, pl, h, w);
        float n0[8], n1[8], o[8];
        VecIO<bf16_t, 8>::unpack(*reinterpret_cast<const uint4 *>(P0 + e0), n0);
        VecIO<bf16_t, 8>::unpack(*reinterpret_cast<const uint4 *>(P1 + e0), n1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t0 = pl * L + w * HW + h;
            o[q] = (n0[q] + n1[q]) + (ldf<bf16_t>(P2 + t0) + ldf<bf16_t>(P3 + t0));
            if constexpr (HW % 8 != 0) {
                if (++w == HW) {
                    w = 0;
                    ++h;
                }
            } else {
                ++w;
            }
        }
        *reinterpret_cast<uint4 *>(out + e0) = VecIO<bf16_t, 8>::pack(o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The NEXT tile's planes by LDS-direct loads (backward; modes 0-2).  A tile's x (bf16) and dy (fp32) are PL contiguous
// elements each: x lands in the other natural x image as it is, dy in a raw fp32 strip; no register is held and the issuing
// wave waits for nothing -- the round trip runs under the current tile's sweeps (register staging of the next tile before the
// sweeps was tried twice: 24 live registers put the kernel past 256).  The loads are inline asm on purpose: for the builtin
// the compiler orders every later LDS read behind the load (vmcnt(0) in front of the sweeps' first ds_read).  As asm they are
// invisible to its vmcnt bookkeeping, which can only make a later counted wait stricter (in-order return), never unsafe; the
// issue point is chosen so that it does not: right after the wave's operands of the tile's first chunk row have arrived.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l3_dma16(const void *src, const uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory", "m0");
}

template <int PL>
__device__ __forceinline__ void l3_dma_tile(const bf16_t *xs, const float *gs, const uint32_t xdst, const uint32_t gdst,
                                            const int wave, const int lane) {
    constexpr int NVX = PL * 2 / 16, NVG = PL * 4 / 16;              // 16-byte vectors
    constexpr int NXP = (NVX + 63) / 64, NGP = (NVG + 63) / 64;      // pieces of one wave instruction (1 KB)
#pragma unroll
    for (int q = 0; q < (NXP + NGP + 3) / 4; ++q) {
        const int p = 4 * q + wave;                                  // (uniform)
        if (p < NXP) {
            const int v = 64 * p + lane;
            if (v < NVX) l3_dma16(reinterpret_cast<const uint4 *>(xs) + v, xdst + 1024u * p);
        } else if (p < NXP + NGP) {
            const int pg = p - NXP, v = 64 * pg + lane;
            if (v < NVG) l3_dma16(reinterpret_cast<const uint4 *>(gs) + v, gdst + 1024u * pg);
        }
    }
}

__device__ __forceinline__ uint32_t l3_lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// tile start with the planes already in LDS (x natural, dy raw fp32): dy -> bf16 natural image, both transposed images.
// Consecutive lanes take the same column block of consecutive rows (l3_vec_pos): 2-byte scatters on consecutive addresses.
template <int HW, int PPT>
__device__ __forceinline__ void l3_stage_lds(const bf16_t *xN, bf16_t *xT, const float *graw, bf16_t *gN, bf16_t *gT,
                                             const int tid) {
    constexpr int L = HW * HW, VS = HW % 8 == 0 ? 8 : 4, nvec = PPT * L / VS;
    static_assert(HW % VS == 0, "rows must be whole vectors");
    for (int v = tid; v < nvec; v += 256) {
        int e0, pl, h, w;
        l3_vec_pos<HW, VS>(v, e0, pl, h, w);
        uint16_t *tx = reinterpret_cast<uint16_t *>(xT) + pl * L + w * HW + h;
        uint16_t *tg = reinterpret_cast<uint16_t *>(gT) + pl * L + w * HW + h;
        uint32_t xw[VS / 2], gw[VS / 2];
        if constexpr (VS == 8) {
            const uint4 r = *reinterpret_cast<const uint4 *>(xN + e0);
            const float4 g0 = *reinterpret_cast<const float4 *>(graw + e0), g1 = *reinterpret_cast<const float4 *>(graw + e0 + 4);
            xw[0] = r.x; xw[1] = r.y; xw[2] = r.z; xw[3] = r.w;
            gw[0] = pack_bf16x2(g0.x, g0.y); gw[1] = pack_bf16x2(g0.z, g0.w);
            gw[2] = pack_bf16x2(g1.x, g1.y); gw[3] = pack_bf16x2(g1.z, g1.w);
            *reinterpret_cast<uint4 *>(gN + e0) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        } else {
            const uint2 r = *reinterpret_cast<const uint2 *>(xN + e0);
            const float4 g0 = *reinterpret_cast<const float4 *>(graw + e0);
            xw[0] = r.x; xw[1] = r.y;
            gw[0] = pack_bf16x2(g0.x, g0.y); gw[1] = pack_bf16x2(g0.z, g0.w);
            *reinterpret_cast<uint2 *>(gN + e0) = make_uint2(gw[0], gw[1]);
        }
#pragma unroll
        for (int q = 0; q < VS / 2; ++q) {
            tx[(2 * q) * HW] = (uint16_t)(xw[q] & 0xffffu);
            tx[(2 * q + 1) * HW] = (uint16_t)(xw[q] >> 16);
            tg[(2 * q) * HW] = (uint16_t)(gw[q] & 0xffffu);
            tg[(2 * q + 1) * HW] = (uint16_t)(gw[q] >> 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward, one route over one plane
// ---------------------------------------------------------------------------------------------------------------------
template <int HW, bool REV, int MODE, int RP2>
__device__ __forceinline__ void l3_fwd_plane(const bf16_t *__restrict__ dts_row, const bf16_t *__restrict__ Brow,
                                             const bf16_t *__restrict__ Crow, float *__restrict__ chk_row,
                                             const bool more_planes, const float A2, const float Dr, const float bias,
                                             const uint32_t (&wp)[RP2 ? RP2 : 1], const bf16_t *xq, bf16_t *yq, const int lane,
                                             L3Ops<(MODE == 3 ? 2 * RP2 : 1)> &op) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, NSEG = G::NSEG, NX = MODE == 3 ? 2 * RP2 : 1;
    const int ci = REV ? 63 - lane : lane;
    const bool tail_live = !G::HAS_TAIL || ci < G::TAILV;
    float hc = 0.f;                                   // state entering the chunk row
    uint4 xn;
    {
        const int sp0 = REV ? NSEG - 1 : 0;
        xn = (G::HAS_TAIL && sp0 == NSEG - 1 && !tail_live) ? make_uint4(0, 0, 0, 0)
                                                            : *reinterpret_cast<const uint4 *>(xq + sp0 * G::ROW + ci * 8);
    }
#pragma unroll 1
    for (int i = 0; i < NSEG; ++i) {                  // chunk rows along the route
        const int sp = REV ? NSEG - 1 - i : i;
        const int tp0 = sp * G::ROW + ci * 8;
        const bool is_tail = G::HAS_TAIL && sp == NSEG - 1;
        l3f2 v[4];
        if constexpr (MODE == 3) l3_delta_raw<RP2, REV>(op.d, wp, bias, v);       // (dt_proj: consumed before the registers are re-requested)
        else l3_unpack<REV>(op.d[0], v);
        const uint4 bv = op.b, cv = op.c, xv = xn;
        {   // request the next row (this plane's row i + 1, or the first row of the next plane)
            const int in_ = i + 1 < NSEG ? i + 1 : 0;
            const int spn = REV ? NSEG - 1 - in_ : in_;
            const int tpn = spn * G::ROW + ci * 8;
            const bf16_t *drow = (i + 1 < NSEG || MODE == 3) ? dts_row : dts_row + L;   // (mode 3: the route's rows serve every plane)
            const bool dead = G::HAS_TAIL && spn == NSEG - 1 && !tail_live;
            if (i + 1 < NSEG || more_planes) {
                if (!dead) l3_ops_load<NX>(op, drow, Brow, Crow, tpn);
                else l3_ops_zero<NX>(op);
            }
            if (i + 1 < NSEG) xn = dead ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4 *>(xq + tpn);
        }
        l3f2 u[4], Bq[4], Cq[4];
        l3_unpack<REV>(xv, u);
        l3_unpack<REV>(bv, Bq);
        l3_unpack<REV>(cv, Cq);
        if constexpr (MODE != 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = v[q].x, v1 = v[q].y;
                if constexpr (MODE != 3) {
                    v0 += bias;
                    v1 += bias;
                }
                if constexpr (MODE == 1) {
                    v0 = softplus20(v0);
                    v1 = softplus20(v1);
                }
                if constexpr (MODE == 3) {
                    v0 = l3_softplus(v0);
                    v1 = l3_softplus(v1);
                }
                v[q] = l3f2{v0, v1};
            }
            if (is_tail) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = tail_live ? v[q] : l3f2{0.f, 0.f};
            }
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
            P *= a[q].x;
            S = fmaf(a[q].y, S, bb[q].y);
            P *= a[q].y;
        }
        l3_scan_up(P, S);
        const float th = fmaf(P, hc, S);              // state after this lane's chunk
        float hh = dpp_mov<kWaveShr1>(hc, th);
        hc = bcast_lane<63>(th);
        if (NSEG > 1 && lane == 63) chk_row[i] = hc;
        l3f2 y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hh = fmaf(a[q].x, hh, bb[q].x);
            const float h0 = hh;
            hh = fmaf(a[q].y, hh, bb[q].y);
            y[q] = __builtin_elementwise_fma(Cq[q], l3f2{h0, hh}, u[q] * Dr);
        }
        if (!is_tail || tail_live) *reinterpret_cast<uint4 *>(yq + tp0) = l3_pack<REV>(y);
    }
}

// y = sum of the four private planes (routes 0 / 2 natural, 1 / 3 transposed), fixed order, fp32 out
template <int HW, int PPT>
__device__ __forceinline__ void l3_merge_y(float *out, const bf16_t *P0, const bf16_t *P1, const bf16_t *P2,
                                           const bf16_t *P3, const int tid) {
    constexpr int L = HW * HW, PL = PPT * L;
    for (int v = tid; v < PL / 8; v += 256) {
        int e0, pl, h, w;
        l3_vec_pos<HW, 8>(v, e0, pl, h, w);
        float n0[8], n1[8], o[8];
        VecIO<bf16_t, 8>::unpack(*reinterpret_cast<const uint4 *>(P0 + e0), n0);
        VecIO<bf16_t, 8>::unpack(*reinterpret_cast<const uint4 *>(P1 + e0), n1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t0 = pl * L + w * HW + h;
            o[q] = (n0[q] + n1[q]) + (ldf<bf16_t>(P2 + t0) + ldf<bf16_t>(P3 + t0));
            if constexpr (HW % 8 != 0) {
                if (++w == HW) {
                    w = 0;
                    ++h;
                }
            } else {
                ++w;
            }
        }
        *reinterpret_cast<float4 *>(out + e0) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(out + e0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// Workgroup -> (sample, tile group).  Workgroups are handed to the 8 XCDs round-robin, and every workgroup re-reads its
// sample's B / C rows (4 routes x 2 x L bf16 = 50 KB at 56 x 56) once per plane.  With consecutive ids walking the groups of
// one sample, each XCD's L2 sees ALL samples at once (64 x 50 KB next to the streaming planes: the rows are evicted between
// uses -- PMC traffic of the 56 x 56 kernels was 1.8x / 2.1x the algorithmic bytes).  With xmap the samples are dealt to the
// XCDs (sample b lives on XCD b % 8): an XCD holds the rows of the batch / 8 samples its workgroups walk, fetched once.
__device__ __forceinline__ void l3_block_map(const LeanArgs &a, const int groups_pb, int &b, int &tg) {
    const int bid = blockIdx.x;
    if (a.xmap) {
        const int j = bid >> 3, q = j / groups_pb;
        b = (bid & 7) + 8 * q;
        tg = j - q * groups_pb;
    } else {
        b = bid / groups_pb;
        tg = bid - b * groups_pb;
    }
}

template <int HW, int PPT, int MODE, int RP2, bool REV>
__device__ __forceinline__ void l3_fwd_body(const LeanArgs &a, float *smem, const int wave, const int lane) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, PL = PPT * L, NSEG = G::NSEG, NX = MODE == 3 ? 2 * RP2 : 1;
    const int D = a.D_;
    const int tiles_pb = D / PPT;
    const int groups_pb = tiles_pb / a.pli;
    int b, tg;
    l3_block_map(a, groups_pb, b, tg);
    // LDS: xN | xT | 4 private y planes (bf16: the per-route partial sums are rounded to the I/O precision once, before
    // the fixed-order fp32 merge, as in the lean kernels)
    bf16_t *xN = reinterpret_cast<bf16_t *>(smem), *xT = xN + PL, *Y = xT + PL;
    const bool col = wave >> 1;
    const int k = (REV ? 2 : 0) + (wave >> 1);
    const bf16_t *xq = col ? xT : xN;
    bf16_t *yq = Y + (size_t)wave * PL;
    const int64_t route = (int64_t)b * 4 + k;
    const bf16_t *Brow = (const bf16_t *)a.Bs + route * L, *Crow = (const bf16_t *)a.Cs + route * L;
    const int n_planes = a.pli * PPT;
    const int ci = REV ? 63 - lane : lane;
    // mode 3: the route's dt_proj input rows (L, Rp) serve every plane; otherwise the step sizes of the first plane
    const bf16_t *drow0 = MODE == 3 ? (const bf16_t *)a.xrt + route * (NSEG * 512 * (2 * RP2))
                                    : (const bf16_t *)a.dts + (route * D + (int64_t)tg * a.pli * PPT) * L;
    L3Ops<NX> op;
    {   // operands of the first chunk row of the first plane
        const int sp = REV ? NSEG - 1 : 0;
        const int tp = sp * G::ROW + ci * 8;
        if (!(G::HAS_TAIL && sp == NSEG - 1) || ci < G::TAILV) l3_ops_load<NX>(op, drow0, Brow, Crow, tp);
        else l3_ops_zero<NX>(op);
        op.h = 0.f;
    }
    constexpr int NVX = 2;
    static_assert(PL <= 4096, "tile beyond the staging registers");
#pragma unroll 1
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * PPT;
        const int64_t po = ((int64_t)b * D + d0) * L;
        int tz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));
        const int tid = threadIdx.x + tz;
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
            const float A2 = a.A[row] * kLog2e, Dr = a.D[row], bias = MODE == 2 ? 0.f : a.bias[row];
            const bool more = it * PPT + pl + 1 < n_planes;
            uint32_t wp[RP2 ? RP2 : 1];
            l3_weight_pairs<RP2>(a.dtw, row, wp);
            l3_fwd_plane<HW, REV, MODE, RP2>(MODE == 3 ? drow0 : (const bf16_t *)a.dts + (route * D + d) * L, Brow, Crow,
                                             a.chk + (route * D + d) * NSEG, more, A2, Dr, bias, wp, xq + pl * L, yq + pl * L,
                                             lane, op);
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

// (forward: four waves per SIMD at 128 registers; with 12 operand vectors per chunk row in flight -- dt_rank 12 -- three at 168)
template <int HW, int PPT, int MODE, int RP2 = 0>
__global__ void __launch_bounds__(256, (RP2 > 4 ? 3 : L3_WPE_FWD)) ss2d_l3_fwd_kernel(const LeanArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave & 1) l3_fwd_body<HW, PPT, MODE, RP2, true>(a, smem, wave, lane);
    else l3_fwd_body<HW, PPT, MODE, RP2, false>(a, smem, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, one route over one plane
// ---------------------------------------------------------------------------------------------------------------------
// what the first chunk row of a tile's first plane sends for: the next tile's planes (l3_dma_tile)
struct L3Next {
    const bf16_t *x;
    const float *g;
    uint32_t xdst, gdst;
    int wave;
    bool go;
};

template <int HW, bool REV, int MODE, int RP2, int PFPL = 0>
__device__ __forceinline__ void l3_bwd_plane(const bf16_t *__restrict__ dts_row, bf16_t *__restrict__ ddts_row,
                                             const bf16_t *__restrict__ Brow, const bf16_t *__restrict__ Crow,
                                             const float *__restrict__ chk_row, const bool more_planes, const float An,
                                             const float Dr, const float bias, const uint32_t (&wp)[RP2 ? RP2 : 1],
                                             const bf16_t *xq, const bf16_t *gq,
                                             bf16_t *dxq, float *ldsacc, l3f2 (&rB)[L3Geom<HW, RP2>::NACC][4],
                                             l3f2 (&rC)[L3Geom<HW, RP2>::NACC][4], float &dA_acc, float &dD_acc,
                                             float &dbias_acc, const int lane, L3Ops<(MODE == 3 ? 2 * RP2 : 1)> &op,
                                             const int dbg, const L3Next &nx) {
    using G = L3Geom<HW, RP2>;
    constexpr int L = G::L, NSEG = G::NSEG, NX = MODE == 3 ? 2 * RP2 : 1;
    const float A2 = An * kLog2e;
    const int ci = REV ? 63 - lane : lane;
    const bool tail_live = !G::HAS_TAIL || ci < G::TAILV;
    float Ec = 0.f;                                   // adjoint flowing in from the chunk row processed before
    l3f2 dA2 = {0.f, 0.f}, dD2 = dA2, db2 = dA2;
    // x / dy of the chunk row to process, read from LDS one row ahead (a wave has one partner on its SIMD: an exposed LDS
    // round trip at the head of every row is not covered by anything)
    // (mode 3 reads them at the head of their own row instead: the dt_proj dot products run under the round trip, and the 8
    //  registers of the look-ahead are what the 56 x 56 kernel needs for the dt_proj operand vectors)
    L3Lds ln;
    if constexpr (MODE != 3) {
        const int sp0 = REV ? 0 : NSEG - 1;
        if (G::HAS_TAIL && sp0 == NSEG - 1 && !tail_live) {
            ln.x = ln.g = make_uint4(0, 0, 0, 0);
        } else {
            ln.x = *reinterpret_cast<const uint4 *>(xq + sp0 * G::ROW + ci * 8);
            ln.g = *reinterpret_cast<const uint4 *>(gq + sp0 * G::ROW + ci * 8);
        }
    }
#pragma unroll 1
    for (int i = NSEG - 1; i >= 0; --i) {             // chunk rows against the route
        const int sp = REV ? NSEG - 1 - i : i;        // physical chunk row
        const int tp0 = sp * G::ROW + ci * 8;
        const bool is_tail = G::HAS_TAIL && sp == NSEG - 1;
        if constexpr (MODE == 3) {
            if (is_tail && !tail_live) {
                ln.x = ln.g = make_uint4(0, 0, 0, 0);
            } else {
                ln.x = *reinterpret_cast<const uint4 *>(xq + tp0);
                ln.g = *reinterpret_cast<const uint4 *>(gq + tp0);
            }
        }
        l3f2 v[4];
        if constexpr (MODE == 3) l3_delta_raw<RP2, REV>(op.d, wp, bias, v);       // (dt_proj: consumed before the registers are re-requested)
        else l3_unpack<REV>(op.d[0], v);
        const uint4 bv = op.b, cv = op.c;
        const float hin = op.h;
        if constexpr (PFPL > 0) {
            // the next tile's planes: sent for once this row's operands are here (nothing of this wave is in flight behind the
            // explicit wait, so the asm loads cannot make a compiler-counted wait stricter than it is)
            if (nx.go && i == NSEG - 1) {
                int lz;                               // (opaque: the piece addresses are formed here, not held through the loop)
                asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, 0" : "=v"(lz)::"memory");
                l3_dma_tile<PFPL>(nx.x, nx.g, nx.xdst, nx.gdst, nx.wave, lane + lz);
            }
        }
        // ---- request the next row to process (this plane's row i - 1, or the last row of the next plane)
        {
            const int in_ = i > 0 ? i - 1 : NSEG - 1;
            const int spn = REV ? NSEG - 1 - in_ : in_;
            const int tpn = spn * G::ROW + ci * 8;
            const bf16_t *drow = (i > 0 || MODE == 3) ? dts_row : dts_row + L;     // (mode 3: the route's rows serve every plane)
            const float *crow = i > 0 ? chk_row : chk_row + NSEG;
            if ((i > 0 || more_planes) && !(dbg & 16)) {
                const bool ok = !(G::HAS_TAIL && spn == NSEG - 1) || tail_live;
                if (ok) l3_ops_load<NX>(op, drow, Brow, Crow, tpn);
                else l3_ops_zero<NX>(op);
                op.h = in_ > 0 ? crow[in_ - 1] : 0.f;
            }
        }
        const uint4 xv = ln.x, gv = ln.g;
        if (MODE != 3 && i > 0) {
            const int spn = REV ? NSEG - i : i - 1;
            if (G::HAS_TAIL && spn == NSEG - 1 && !tail_live) {
                ln.x = ln.g = make_uint4(0, 0, 0, 0);
            } else {
                ln.x = *reinterpret_cast<const uint4 *>(xq + spn * G::ROW + ci * 8);
                ln.g = *reinterpret_cast<const uint4 *>(gq + spn * G::ROW + ci * 8);
            }
        }
        l3f2 u[4], g[4], Bq[4], Cq[4], sg[4];
        l3_unpack<REV>(xv, u);
        l3_unpack<REV>(gv, g);
        l3_unpack<REV>(bv, Bq);
        l3_unpack<REV>(cv, Cq);
        if constexpr (MODE == 2) {                     // the step sizes arrive activated: sigmoid(raw) = 1 - exp(-softplus)
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[q] = 1.f - l3_exp2(v[q] * (-kLog2e));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s0 = 1.f, s1 = 1.f;
                float v0 = v[q].x, v1 = v[q].y;
                if constexpr (MODE != 3) {
                    v0 += bias;
                    v1 += bias;
                }
                if constexpr (MODE == 1) {
                    v0 = softplus20_sig(v0, s0);
                    v1 = softplus20_sig(v1, s1);
                }
                if constexpr (MODE == 3) {
                    v0 = l3_softplus_sig(v0, s0);
                    v1 = l3_softplus_sig(v1, s1);
                }
                v[q] = l3f2{v0, v1};
                sg[q] = l3f2{s0, s1};
            }
            if (is_tail) {                             // dead lanes of the tail row: identity elements
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = tail_live ? v[q] : l3f2{0.f, 0.f};
                    sg[q] = tail_live ? sg[q] : l3f2{0.f, 0.f};
                }
            }
        }
        l3f2 a[4], vu[4], bb[4], cg[4], acg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = l3_exp2(v[q] * A2);
            vu[q] = v[q] * u[q];
            bb[q] = vu[q] * Bq[q];
            cg[q] = Cq[q] * g[q];
            acg[q] = a[q] * cg[q];
        }
        float P = 1.f, S = 0.f, R = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            S = fmaf(a[q].x, S, bb[q].x);
            P *= a[q].x;
            S = fmaf(a[q].y, S, bb[q].y);
            P *= a[q].y;
        }
#pragma unroll
        for (int q = 3; q >= 0; --q) {
            R = fmaf(a[q].y, R, acg[q].y);
            R = fmaf(a[q].x, R, acg[q].x);
        }
        float Q = P;
        l3_scan_pair(P, S, Q, R);
        // state entering this lane's chunk = the inclusive map of the lane below applied to the row's incoming state;
        // adjoint entering it = the inclusive map of the lane above applied to the adjoint carried in
        const float th = fmaf(P, hin, S);
        float hh = dpp_mov<kWaveShr1>(hin, th);
        const float tE = fmaf(Q, Ec, R);
        float E = dpp_mov<kWaveShl1>(Ec, tE);
        Ec = bcast_lane<0>(tE);
        l3f2 h[4], dh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hh = fmaf(a[q].x, hh, bb[q].x);
            h[q].x = hh;
            hh = fmaf(a[q].y, hh, bb[q].y);
            h[q].y = hh;
        }
#pragma unroll
        for (int q = 3; q >= 0; --q) {
            dh[q].y = cg[q].y + E;
            E = fmaf(a[q].y, E, acg[q].y);
            dh[q].x = cg[q].x + E;
            E = fmaf(a[q].x, E, acg[q].x);
        }
        l3f2 du[4], dd[4], dBq[4], dCq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const l3f2 ah = h[q] - bb[q];
            const l3f2 s1 = dh[q] * Bq[q];
            const l3f2 dhah = dh[q] * ah;
            dA2 = __builtin_elementwise_fma(v[q], dhah, dA2);
            dBq[q] = dh[q] * vu[q];
            dCq[q] = g[q] * h[q];
            du[q] = __builtin_elementwise_fma(v[q], s1, g[q] * Dr);
            dd[q] = __builtin_elementwise_fma(u[q], s1, dhah * An) * sg[q];
            dD2 = __builtin_elementwise_fma(g[q], u[q], dD2);
            db2 += dd[q];
        }
        if ((!is_tail || tail_live) && !(dbg & 32)) {
            *reinterpret_cast<uint4 *>(ddts_row + tp0) = l3_pack<REV>(dd);
            *reinterpret_cast<uint4 *>(dxq + tp0) = l3_pack<REV>(du);      // this route's private dx plane
        }
        // ---- dB / dC sums over the planes of this workgroup: one jump per chunk row into static-register adds
        // (values stay in traversal order and are un-permuted once, at the flush)
        if (G::LSZ > 0 && sp >= G::NREG) {
            if (!is_tail || tail_live) {
                float *tb = ldsacc + (sp - G::NREG) * G::ROW + ci * 8, *tc = tb + G::LSZ;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    float4 b4 = *reinterpret_cast<float4 *>(tb + 2 * q), c4 = *reinterpret_cast<float4 *>(tc + 2 * q);
                    b4.x += dBq[q].x; b4.y += dBq[q].y; b4.z += dBq[q + 1].x; b4.w += dBq[q + 1].y;
                    c4.x += dCq[q].x; c4.y += dCq[q].y; c4.z += dCq[q + 1].x; c4.w += dCq[q + 1].y;
                    *reinterpret_cast<float4 *>(tb + 2 * q) = b4;
                    *reinterpret_cast<float4 *>(tc + 2 * q) = c4;
                }
            }
        }
        {
            // (always executed: inside an else the compiler merges two copies of every sum at the join)
            // One statement per register row: a scalar compare-and-skip around eight packed adds INSIDE the statement,
            // so that the compiler sees a plain read-modify-write of the row's sums (as a switch over the rows it
            // double-buffers all of them and copies 96 registers on every path of every chunk row).
#define L3_ACC_ROW(SS)                                                                                          \
    if constexpr (SS < G::NREG)                                                                                 \
        asm volatile("s_cmp_lg_u32 %[sp], " #SS "\n\t"                                                          \
                     "s_cbranch_scc1 1f\n\t"                                                                    \
                     "v_pk_add_f32 %[b0], %[b0], %[d0]\n\t"                                                     \
                     "v_pk_add_f32 %[b1], %[b1], %[d1]\n\t"                                                     \
                     "v_pk_add_f32 %[b2], %[b2], %[d2]\n\t"                                                     \
                     "v_pk_add_f32 %[b3], %[b3], %[d3]\n\t"                                                     \
                     "v_pk_add_f32 %[c0], %[c0], %[e0]\n\t"                                                     \
                     "v_pk_add_f32 %[c1], %[c1], %[e1]\n\t"                                                     \
                     "v_pk_add_f32 %[c2], %[c2], %[e2]\n\t"                                                     \
                     "v_pk_add_f32 %[c3], %[c3], %[e3]\n\t"                                                     \
                     "1:\n\t"                                                                                   \
                     : [b0] "+v"(rB[SS < G::NREG ? SS : 0][0]), [b1] "+v"(rB[SS < G::NREG ? SS : 0][1]),        \
                       [b2] "+v"(rB[SS < G::NREG ? SS : 0][2]), [b3] "+v"(rB[SS < G::NREG ? SS : 0][3]),        \
                       [c0] "+v"(rC[SS < G::NREG ? SS : 0][0]), [c1] "+v"(rC[SS < G::NREG ? SS : 0][1]),        \
                       [c2] "+v"(rC[SS < G::NREG ? SS : 0][2]), [c3] "+v"(rC[SS < G::NREG ? SS : 0][3])         \
                     : [d0] "v"(dBq[0]), [d1] "v"(dBq[1]), [d2] "v"(dBq[2]), [d3] "v"(dBq[3]), [e0] "v"(dCq[0]), \
                       [e1] "v"(dCq[1]), [e2] "v"(dCq[2]), [e3] "v"(dCq[3]), [sp] "s"(sp)                        \
                     : "scc");
            L3_ACC_ROW(0) L3_ACC_ROW(1) L3_ACC_ROW(2) L3_ACC_ROW(3) L3_ACC_ROW(4) L3_ACC_ROW(5) L3_ACC_ROW(6) L3_ACC_ROW(7)
#undef L3_ACC_ROW
        }

    }
    dA_acc = dA2.x + dA2.y;
    dD_acc = dD2.x + dD2.y;
    dbias_acc = db2.x + db2.y;
}

// the whole walk of a workgroup for one direction (a template parameter from the top: the ascending and the descending
// waves share no code path below this point, so the accumulators never meet at a join)
template <int HW, int PPT, int MODE, int RP2, bool REV>
__device__ __forceinline__ void l3_bwd_body(const LeanArgs &a, float *smem, const int wave, const int lane) {
    using G = L3Geom<HW, RP2>;
    constexpr int L = G::L, PL = PPT * L, NSEG = G::NSEG, NX = MODE == 3 ? 2 * RP2 : 1;
    const int D = a.D_;
    const int tiles_pb = D / PPT;
    const int groups_pb = tiles_pb / a.pli;
    int b, tg;
    l3_block_map(a, groups_pb, b, tg);
    bf16_t *xN = reinterpret_cast<bf16_t *>(smem), *xT = xN + PL, *gN = xT + PL, *gT = gN + PL, *DX = gT + PL;
    float *ldsacc = smem + (8 * (size_t)PL * 2) / 4 + wave * 2 * G::LSZ;
    // (modes 0-2) behind the strips: the second natural x image and the raw dy of the tile being fetched (l3_dma_tile)
    constexpr bool PF = MODE != 3 && L3_PREFETCH;
    bf16_t *xN1 = reinterpret_cast<bf16_t *>(smem + (8 * (size_t)PL * 2) / 4 + 4 * 2 * G::LSZ);
    float *graw = reinterpret_cast<float *>(xN1 + PL);
    for (int e = lane; e < 2 * G::LSZ; e += 64) ldsacc[e] = 0.f;
    l3f2 rB[G::NACC][4], rC[G::NACC][4];
#pragma unroll
    for (int s = 0; s < G::NACC; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) rB[s][q] = rC[s][q] = l3f2{0.f, 0.f};
    const bool col = wave >> 1;
    const int k = (REV ? 2 : 0) + (wave >> 1);
    const bf16_t *gq = col ? gT : gN;
    bf16_t *dxq = DX + (size_t)wave * PL;
    const int64_t route = (int64_t)b * 4 + k;
    const bf16_t *Brow = (const bf16_t *)a.Bs + route * L, *Crow = (const bf16_t *)a.Cs + route * L;
    const int n_planes = a.pli * PPT;
    const int ci = REV ? 63 - lane : lane;
    // operands of the first chunk row to process (the last one, in route order, of the first plane)
    const bf16_t *drow0 = MODE == 3 ? (const bf16_t *)a.xrt + route * (NSEG * 512 * (2 * RP2))
                                    : (const bf16_t *)a.dts + (route * D + (int64_t)tg * a.pli * PPT) * L;
    L3Ops<NX> op;
    {
        const int64_t r0 = route * D + (int64_t)tg * a.pli * PPT;
        const int sp = REV ? 0 : NSEG - 1;
        const int tp = sp * G::ROW + ci * 8;
        if (!(G::HAS_TAIL && sp == NSEG - 1) || ci < G::TAILV) l3_ops_load<NX>(op, drow0, Brow, Crow, tp);
        else l3_ops_zero<NX>(op);
        op.h = NSEG > 1 ? a.chk[r0 * NSEG + NSEG - 2] : 0.f;
    }
    constexpr int NVX = 2, NVG = 4;                    // PL <= 4096 elements: 512 / 1024 vectors over 256 threads
    static_assert(PL <= 4096, "tile beyond the staging registers");
    if constexpr (PF) {
        const int64_t po0 = ((int64_t)b * D + (int64_t)tg * a.pli * PPT) * L;
        if (!(a.dbg & 2))
            l3_dma_tile<PL>((const bf16_t *)a.x + po0, (const float *)a.dy + po0, l3_lds_addr(xN), l3_lds_addr(graw), wave, lane);
    }
#pragma unroll 1
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * PPT;
        const int64_t po = ((int64_t)b * D + d0) * L;
        int tz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(tz));    // opaque zero: keeps the per-thread tile positions out of registers
        const int tid = threadIdx.x + tz;
        bf16_t *xNc = (PF && (it & 1)) ? xN1 : xN;    // this tile's natural x image
        const bf16_t *xq = col ? xT : xNc;
        L3Next nx{};
        if constexpr (PF) {
            // tile `it` was sent for during tile it - 1 (or above): wait for this wave's pieces, then for everybody's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // (also: the previous tile's merge has read the private planes)
            if (!(a.dbg & 8)) l3_stage_lds<HW, PPT>(xNc, xT, graw, gN, gT, tid);
            nx.x = (const bf16_t *)a.x + po + PL;
            nx.g = (const float *)a.dy + po + PL;
            nx.xdst = l3_lds_addr((it & 1) ? xN : xN1);
            nx.gdst = l3_lds_addr(graw);
            nx.wave = wave;
            nx.go = it + 1 < a.pli && !(a.dbg & 2);
        } else if constexpr (HW % 8 == 0) {
            // Register staging in memory order.  (Named scalars, not arrays: hipcc left a uint4 / float4 array as an alloca --
            // 48 bytes of scratch per lane and plane, the vectors stored and re-read around the barrier: 1.9x the algorithmic
            // HBM bytes in the PMC counters.  Requesting the NEXT tile's vectors before the sweeps was tried: the 24 live
            // registers push the kernel over 256 and the spills cost what the hidden round trip gains.)
            constexpr int nvx = PL / 8, nvg = PL / 4;
            static_assert(NVX == 2 && NVG == 4, "staging registers are spelled out");
            const bf16_t *xs_ = (const bf16_t *)a.x + po;
            const float *gs_ = (const float *)a.dy + po;
            const int v0 = tid, v1 = tid + 256, v2 = tid + 512, v3 = tid + 768;
            const uint4 px0 = *reinterpret_cast<const uint4 *>(xs_ + (v0 < nvx ? v0 : 0) * 8);
            const uint4 px1 = *reinterpret_cast<const uint4 *>(xs_ + (v1 < nvx ? v1 : 0) * 8);
            const float4 pg0 = *reinterpret_cast<const float4 *>(gs_ + (v0 < nvg ? v0 : 0) * 4);
            const float4 pg1 = *reinterpret_cast<const float4 *>(gs_ + (v1 < nvg ? v1 : 0) * 4);
            const float4 pg2 = *reinterpret_cast<const float4 *>(gs_ + (v2 < nvg ? v2 : 0) * 4);
            const float4 pg3 = *reinterpret_cast<const float4 *>(gs_ + (v3 < nvg ? v3 : 0) * 4);
            __syncthreads();                           // (the previous tile's merge has read the planes)
            if (v0 < nvx) *reinterpret_cast<uint4 *>(xN + v0 * 8) = px0;
            if (v1 < nvx) *reinterpret_cast<uint4 *>(xN + v1 * 8) = px1;
            auto put_g = [&](const int v, const float4 g4) {
                if (v < nvg) *reinterpret_cast<uint2 *>(gN + v * 4) = make_uint2(pack_bf16x2(g4.x, g4.y), pack_bf16x2(g4.z, g4.w));
            };
            put_g(v0, pg0); put_g(v1, pg1); put_g(v2, pg2); put_g(v3, pg3);
            __syncthreads();
            l3_transpose_pass<HW, PPT>(xN, xT, tid);
            l3_transpose_pass<HW, PPT>(gN, gT, tid);
        } else {
            PlaneRegs<bf16_t, 8, NVX> px;
            PlaneRegs<float, 4, NVG> pg;
            if (!(a.dbg & 2)) {
                l3_planes_issue<HW, PPT, bf16_t, 8, NVX>(px, (const bf16_t *)a.x + po, tid);
                l3_planes_issue<HW, PPT, float, 4, NVG>(pg, (const float *)a.dy + po, tid);
            }
            __syncthreads();                           // (the previous tile's merge has read the planes)
            if (!(a.dbg & 2)) {
                l3_planes_commit<HW, PPT, bf16_t, 8, NVX>(px, xN, xT, tid);
                l3_planes_commit<HW, PPT, float, 4, NVG>(pg, gN, gT, tid);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int pl = 0; pl < ((a.dbg & 1) ? 0 : PPT); ++pl) {
            const int d = d0 + pl, row = k * D + d;
            const int64_t ro = (route * D + d) * L;
            const float *chk_row = a.chk + (route * D + d) * NSEG;
            const float An = a.A[row], Dr = a.D[row], bias = MODE == 2 ? 0.f : a.bias[row];
            const bool more = it * PPT + pl + 1 < n_planes;
            float dA_acc, dD_acc, dbias_acc;
            uint32_t wp[RP2 ? RP2 : 1];
            l3_weight_pairs<RP2>(a.dtw, row, wp);
            l3_bwd_plane<HW, REV, MODE, RP2, (PF ? PL : 0)>(MODE == 3 ? drow0 : (const bf16_t *)a.dts + ro, (bf16_t *)a.ddts + ro,
                                                            Brow, Crow, chk_row, more, An, Dr, bias, wp, xq + pl * L, gq + pl * L,
                                                            dxq + pl * L, ldsacc, rB, rC, dA_acc, dD_acc, dbias_acc, lane, op,
                                                            a.dbg, nx);
            nx.go = false;
            for (int o = 32; o > 0; o >>= 1) {
                dA_acc += __shfl_xor(dA_acc, o, 64);
                dD_acc += __shfl_xor(dD_acc, o, 64);
                dbias_acc += __shfl_xor(dbias_acc, o, 64);
            }
            if (lane == 0) {
                atomicAdd(a.dA + row, dA_acc);
                atomicAdd(a.dD + row, dD_acc);
                atomicAdd(a.dbias + row, dbias_acc);
            }
        }
        __syncthreads();
        if (!(a.dbg & 4)) {
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
    // ---- flush: registers hold [chunk row][traversal element] of this lane.  Atomics are only fast when a wave
    // instruction covers contiguous bytes, so the sums are first laid out by position in LDS (the plane region is free)
    __syncthreads();
    float *dBg = a.dBs + route * L, *dCg = a.dCs + route * L;
    float *stage = smem + (size_t)wave * L;            // 4 waves x L floats <= 8 PL bf16 (PPT >= 1: 16 L bytes)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int s = 0; s < G::NREG; ++s) {
            const int tp0 = s * G::ROW + ci * 8;
            if (tp0 < L) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const l3f2 v = pass ? rC[s][q] : rB[s][q];
                    stage[tp0 + (REV ? 7 - 2 * q : 2 * q)] = v.x;
                    stage[tp0 + (REV ? 6 - 2 * q : 2 * q + 1)] = v.y;
                }
            }
        }
        // the LDS strip holds [chunk][traversal element]
        for (int e = lane; e < G::LSZ; e += 64)
            stage[G::NREG * G::ROW + (e & ~7) + (REV ? 7 - (e & 7) : (e & 7))] = ldsacc[pass * G::LSZ + e];
        wave_sync();
        if (a.parts) {
            // plain coalesced stores of this workgroup's sums; ss2d_l3_parts_kernel adds the groups of a sample up
            float *dst = a.parts + ((((int64_t)b * groups_pb + tg) * 4 + k) * 2 + pass) * L;
            for (int e = lane; e < L; e += 64) dst[e] = stage[e];
        } else {
            float *dst = pass ? dCg : dBg;
            for (int e = lane; e < L; e += 64) atomicAdd(dst + e, stage[e]);
        }
        wave_sync();
    }
}

// dBs / dCs (batch, 4, L) = sum over the workgroups of a sample of their partial sums (batch, groups, 4, 2, L)
__global__ void __launch_bounds__(256) ss2d_l3_parts_kernel(const float *__restrict__ parts, float *__restrict__ dBs,
                                                            float *__restrict__ dCs, const int groups, const int L) {
    const int bk = blockIdx.y, b = bk >> 2, k = bk & 3;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= L) return;
    float4 sB = {0.f, 0.f, 0.f, 0.f}, sC = sB;
    for (int g = 0; g < groups; ++g) {
        const float *p = parts + ((((int64_t)b * groups + g) * 4 + k) * 2) * L + e;
        const float4 vb = *reinterpret_cast<const float4 *>(p), vc = *reinterpret_cast<const float4 *>(p + L);
        sB.x += vb.x; sB.y += vb.y; sB.z += vb.z; sB.w += vb.w;
        sC.x += vc.x; sC.y += vc.y; sC.z += vc.z; sC.w += vc.w;
    }
    *reinterpret_cast<float4 *>(dBs + (int64_t)bk * L + e) = sB;
    *reinterpret_cast<float4 *>(dCs + (int64_t)bk * L + e) = sC;
}

// kernel: wave w owns route {0,2,1,3}[w];
// LDS: xN | xT | gN | gT (bf16, PPT planes) | 4 private dx planes (bf16) | 4 x [dB | dC] strips (LSZ fp32 each)
template <int HW, int PPT, int MODE, int RP2 = 0>
__global__ void __launch_bounds__(256, L3_WPE) ss2d_l3_bwd_kernel(const LeanArgs a) {
    static_assert(L3Geom<HW>::L % 8 == 0 && L3Geom<HW>::NSEG <= 8, "map size not covered");
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave & 1) l3_bwd_body<HW, PPT, MODE, RP2, true>(a, smem, wave, lane);
    else l3_bwd_body<HW, PPT, MODE, RP2, false>(a, smem, wave, lane);
}

}  // namespace xfm

#include "ss2d_w.hpp"

namespace xfm {

// fourth-generation kernels (ss2d_w.hpp) for the activated-step-size mode; XFM_SS2D_W=0: the kernels of this file (A/B switch)
static bool w_enabled() {
    static const bool on = [] { const char *e = getenv("XFM_SS2D_W"); return !(e && e[0] == '0'); }();
    return on;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int l3_pli(int batch, int tiles_pb, bool bwd, int fwd_wpe = L3_WPE_FWD) {
    // tiles per workgroup: ONE round of the resident workgroups (2 per CU backward, 4 forward).  The dB / dC flush of a
    // workgroup is 8 L floats; as float atomics (~1.3 TB/s chip-wide) 1024 workgroups spend 79 us of a 56 x 56 launch there.
    const int resident = bwd ? 512 : 256 * fwd_wpe;
    int pli = (int)(((int64_t)batch * tiles_pb + resident - 1) / resident);
    if (pli < 1) pli = 1;
    if (pli > tiles_pb) pli = tiles_pb;
    static const int env_pli = [] { const char *e = getenv("XFM_L3_PLI"); return e ? atoi(e) : 0; }();   // tuning hook, read once
    if (env_pli > 0) pli = std::max(1, std::min(tiles_pb, env_pli));
    while (tiles_pb % pli) --pli;
    return pli;
}

// the dt_rank pairs (Rp / 2) the mode-3 kernels are built for, per map size (0: none)
static int l3_rp2(const int H, const int rank_p) {
    if (rank_p <= 0 || (rank_p & 1)) return 0;
    const int rp2 = rank_p / 2;
    if ((H == 56 && rp2 == 3) || (H == 28 && rp2 == 6)) return rp2;       // XFMamba-T / -S stages 0 / 1: dt_rank 6 / 12
    return 0;
}

template <int HW, int PPT, int RP2> static const void *l3_fn3(const bool bwd) {
    return bwd ? (const void *)ss2d_l3_bwd_kernel<HW, PPT, 3, RP2> : (const void *)ss2d_l3_fwd_kernel<HW, PPT, 3, RP2>;
}

int ss2d_l3_nseg(int batch, int D, int H, int W, int N, int in_dtype);
// ss2d_w.hpp serves this shape (activated step sizes, fp32 B / C rows): byte offsets inside every tensor fit its 32-bit
// buffer addressing
int ss2d_w_covers(int batch, int D, int H, int W, int N, int in_dtype) {
    if (!w_enabled() || !ss2d_l3_nseg(batch, D, H, W, N, in_dtype)) return 0;
    const long long planes = (long long)batch * 4 * D, L = (long long)H * W;
    const long long chk = planes * ((L + 511) / 512) * 256;
    return (planes * L * 2 < (1ll << 31) && chk < (1ll << 31)) ? 1 : 0;
}

template <int HW, int PPT> static int l3_launch(const xfm_ss2d_params_t &p, bool bwd, hipStream_t s, float *ws, size_t ws_bytes) {
    using G = L3Geom<HW>;
    constexpr int L = G::L, PL = PPT * L;
    const int D = p.d_inner;
    if (D % PPT) return XFM_ELIMIT;
    LeanArgs la{};
    la.x = p.x; la.dts = p.dts; la.Bs = p.Bs; la.Cs = p.Cs; la.Bs32 = p.Bs32; la.Cs32 = p.Cs32;
    la.A = p.A; la.D = p.D; la.bias = p.delta_bias;
    la.y = p.y; la.chk = p.chk; la.dy = p.dy; la.dx = p.dx; la.ddts = p.ddts;
    la.dBs = p.dBs; la.dCs = p.dCs; la.dA = p.dA; la.dD = p.dD; la.dbias = p.ddelta_bias;
    la.batch = p.batch; la.D_ = D; la.H = HW; la.W = HW; la.L = L;
    la.nseg = G::NSEG; la.ppt = PPT; la.softplus = p.delta_softplus;
    la.magicW = (uint32_t)((0x100000000ull + HW - 1) / HW);
    la.magicL = (uint32_t)((0x100000000ull + L - 1) / L);
    la.magicH = la.magicW;
    la.dbg = 0;                                        // timing-only switches: 1 skip sweeps, 2 skip plane staging, 4 skip merge,
    static const int env_dbg = [] { const char *e = getenv("XFM_L3_DBG"); return e ? atoi(e) : 0; }();
    la.dbg = env_dbg;                                  // 16 no operand prefetch, 32 no ddts / dx stores
    const int tiles_pb = D / PPT;
    const int rp2 = p.delta_softplus == 3 ? l3_rp2(HW, p.dt_rank_p) : 0;
    const int pli = l3_pli(p.batch, tiles_pb, bwd, rp2 > 4 ? 3 : L3_WPE_FWD);
    la.pli = pli;
    static const bool env_no_xmap = getenv("XFM_L3_NO_XMAP") != nullptr;
    la.xmap = (p.batch % 8 == 0 && !env_no_xmap) ? 1 : 0;
    const int groups = tiles_pb / pli;
    const size_t need = (size_t)p.batch * groups * 4 * 2 * L * sizeof(float);
    la.parts = (bwd && ws && ws_bytes >= need && groups > 1 && L % 4 == 0) ? ws : nullptr;
    size_t lds = bwd ? (size_t)8 * PL * 2 + (size_t)4 * 2 * G::LSZ * sizeof(float) : (size_t)6 * PL * 2;
    if (bwd && p.delta_softplus != 3 && L3_PREFETCH) lds += (size_t)PL * 2 + (size_t)PL * 4;   // second x image | raw dy
    const void *fn;
    if (p.delta_softplus == 3) {
        la.xrt = p.xrt; la.dtw = p.dt_w;
        constexpr int RP2 = HW == 56 ? 3 : (HW == 28 ? 6 : 0);           // the (map, dt_rank) pairs built: l3_rp2
        if constexpr (RP2 > 0) {
            if (rp2 != RP2) return XFM_ELIMIT;
            fn = l3_fn3<HW, PPT, RP2>(bwd);
            if (bwd) lds = (size_t)8 * PL * 2 + (size_t)4 * 2 * L3Geom<HW, RP2>::LSZ * sizeof(float);
        } else {
            return XFM_ELIMIT;
        }
    } else if (p.delta_softplus == 2 && p.bc_f32) {
        if (!ss2d_w_covers(p.batch, D, HW, HW, 1, p.in_dtype) || !p.Bs32 || !p.Cs32) return XFM_EINVAL;
        fn = bwd ? (const void *)ss2d_w_bwd_kernel<HW, PPT> : (const void *)ss2d_w_fwd_kernel<HW, PPT>;
    } else if (bwd)
        fn = p.delta_softplus == 2 ? (const void *)ss2d_l3_bwd_kernel<HW, PPT, 2>
                                   : (p.delta_softplus == 1 ? (const void *)ss2d_l3_bwd_kernel<HW, PPT, 1>
                                                            : (const void *)ss2d_l3_bwd_kernel<HW, PPT, 0>);
    else
        fn = p.delta_softplus == 2 ? (const void *)ss2d_l3_fwd_kernel<HW, PPT, 2>
                                   : (p.delta_softplus == 1 ? (const void *)ss2d_l3_fwd_kernel<HW, PPT, 1>
                                                            : (const void *)ss2d_l3_fwd_kernel<HW, PPT, 0>);
    static const int env_pad = [] { const char *e = getenv("XFM_L3_LDS_PAD"); return e ? atoi(e) : 0; }();   // occupancy experiments
    lds += (size_t)env_pad;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned grid = (unsigned)((int64_t)p.batch * (tiles_pb / pli));
    void *kargs[] = {&la};
    prof_before_main(s);
    const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, lds, s);
    prof_after_main(s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    if (la.parts)
        hipLaunchKernelGGL(ss2d_l3_parts_kernel, dim3((L / 4 + 255) / 256, p.batch * 4), dim3(256), 0, s, la.parts, p.dBs, p.dCs,
                           groups, L);
    return check_launch();
}

// XFM_ELIMIT: shape / dtype not covered here, the caller falls back to the lean / generic kernels
// planes per tile of the map sizes built here (0: not covered)
static int l3_ppt(const int H) { return (H == 56 || H == 48) ? 1 : ((H == 28 || H == 24) ? 4 : 0); }

int ss2d_l3_run(const xfm_ss2d_params_t *p, bool bwd, hipStream_t s, float *ws, size_t ws_bytes) {
    static const bool enabled = [] {
        const char *e = getenv("XFM_SS2D_L3");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return XFM_ELIMIT;
    static const bool fwd_enabled = [] {
        const char *e = getenv("XFM_SS2D_L3_FWD");
        return !(e && e[0] == '0');
    }();
    if (!bwd && !fwd_enabled) return XFM_ELIMIT;
    if (p->in_dtype != XFM_BF16 || p->out_dtype != XFM_F32 || p->dstate != 1 || p->H != p->W) return XFM_ELIMIT;
    if (p->bc_f32 && p->delta_softplus != 2) return XFM_EINVAL;
    if (p->delta_softplus < 0 || p->delta_softplus > 3) return XFM_ELIMIT;
    if (p->delta_softplus == 3 && !l3_rp2(p->H, p->dt_rank_p)) return XFM_ELIMIT;
    const int ppt = l3_ppt(p->H);
    if (!ppt || p->d_inner % ppt) return XFM_ELIMIT;
    if (!p->chk) return XFM_ELIMIT;            // no checkpoint buffer: the generic path decides (EINVAL if it needs one too)
    switch (p->H) {
        case 56: return l3_launch<56, 1>(*p, bwd, s, ws, ws_bytes);
        case 28: return l3_launch<28, 4>(*p, bwd, s, ws, ws_bytes);
        case 48: return l3_launch<48, 1>(*p, bwd, s, ws, ws_bytes);   // XFMamba-B at 384 x 384: stages 1 and 2
        case 24: return l3_launch<24, 4>(*p, bwd, s, ws, ws_bytes);
    }
    return XFM_ELIMIT;
}

// checkpoints per (route, channel) row the kernels of this file write / read: chk[(route * D + d) * NSEG + i], NSEG from THEIR
// geometry.  0: shape not dispatched here.  xfm_ss2d_plan reports at least this many chunks, so a caller that sizes chk
// from the plan can never be short whichever kernel family the plan search itself would pick.
int ss2d_l3_nseg(int batch, int D, int H, int W, int N, int in_dtype) {
    (void)batch;
    const char *e = getenv("XFM_SS2D_L3");
    if ((e && e[0] == '0') || in_dtype != XFM_BF16 || N != 1 || H != W) return 0;
    const int ppt = l3_ppt(H);
    if (!ppt || D % ppt) return 0;
    // ss2d_w.hpp keeps the state entering every LANE's chunk (64 per chunk row); the kernels of this file index the same
    // buffer with one entry per chunk row
    return (H * W + 511) / 512 * (w_enabled() ? 64 : 1);
}

// padded dt_rank for dt_proj inside the kernels of this file (0: no such kernel)
int ss2d_l3_dtfused_rank(int batch, int D, int H, int W, int N, int R, int in_dtype) {
    static const bool off = [] { const char *e = getenv("XFM_L3_DTFUSED"); return e && e[0] == '0'; }();   // A/B switch, read once
    if (off || !ss2d_l3_nseg(batch, D, H, W, N, in_dtype) || R <= 0) return 0;
    const int rp = (R + 1) & ~1;
    return l3_rp2(H, rp) ? rp : 0;
}

// xr (B4, R, L) -> xrt (B4, nrow, Rp, 64, 8), nrow = ceil(L / 512): the blocked rows the mode-3 kernels read (l3_ops_load).
// A thread owns one chunk (8 positions): R 16-byte reads (contiguous across the wave), Rp 16-byte writes (contiguous
// across the wave per piece); positions past L and ranks past R are zero.
template <int RP> __global__ void __launch_bounds__(64) l3_xr_rows_kernel(const uint16_t *__restrict__ xr, uint16_t *__restrict__ xrt,
                                                                          const int R, const int L) {
    const int64_t bk = blockIdx.y;
    const int sp = blockIdx.x, c = threadIdx.x, nrow = gridDim.x;
    const int t0 = sp * 512 + c * 8;
    uint16_t v[8 * RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (r < R && t0 < L) q = *reinterpret_cast<const uint4 *>(xr + (bk * R + r) * L + t0);      // (L % 8 == 0: whole chunks)
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[(2 * i) * RP + r] = (uint16_t)(w[i] & 0xffffu);
            v[(2 * i + 1) * RP + r] = (uint16_t)(w[i] >> 16);
        }
    }
    uint16_t *dst = xrt + ((bk * nrow + sp) * RP * 64 + c) * 8;
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)v[8 * j + 2 * i] | ((uint32_t)v[8 * j + 2 * i + 1] << 16);
        *reinterpret_cast<uint4 *>(dst + j * 512) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// bytes of the partial-sum workspace the backward can use (0: shape not covered here)
size_t ss2d_l3_ws_bytes(const xfm_ss2d_params_t *p) {
    if (p->in_dtype != XFM_BF16 || p->dstate != 1 || p->H != p->W) return 0;
    const int ppt = l3_ppt(p->H);
    if (!ppt || p->d_inner % ppt) return 0;
    const int tiles_pb = p->d_inner / ppt;
    const int groups = tiles_pb / l3_pli(p->batch, tiles_pb, true);
    return groups > 1 ? (size_t)p->batch * groups * 4 * 2 * p->H * p->W * sizeof(float) : 0;
}

}  // namespace xfm

extern "C" {
int xfm_ss2d_xr_rows(const void *xr, void *xrt, long long B4, int R, int Rp, int L, int dtype, void *stream) {
    using namespace xfm;
    if (!xr || !xrt || B4 <= 0 || B4 > 65535 || R <= 0 || Rp < R || L <= 0) return XFM_EINVAL;
    if (dtype != XFM_BF16 && dtype != XFM_F16) return XFM_EDTYPE;                 // (16-bit values are moved, not converted)
    if (L % 8 != 0 || (Rp != 6 && Rp != 12)) return XFM_ELIMIT;                   // the ranks l3_rp2 builds kernels for
    const dim3 grid((unsigned)((L + 511) / 512), (unsigned)B4);
    hipStream_t s = (hipStream_t)stream;
    if (Rp == 6) hipLaunchKernelGGL(l3_xr_rows_kernel<6>, grid, dim3(64), 0, s, (const uint16_t *)xr, (uint16_t *)xrt, R, L);
    else hipLaunchKernelGGL(l3_xr_rows_kernel<12>, grid, dim3(64), 0, s, (const uint16_t *)xr, (uint16_t *)xrt, R, L);
    return check_launch();
}
}

// wgrad_gemm.hip -- token-contracting product on the matrix cores: the weight gradient of every channel projection,
//
//     dW[m][n] += sum over samples b, tokens l of  A[b, l, m] * B[b, l, n]          (fp32 accumulation, fp32 result)
//
// with each operand either TOKEN-major (batch, L, C) -- the residual stream, Mlp activations, x_dbl -- or PLANE-major
// (batch, C, L) -- the scan path -- at a caller-given sample stride.  The framework formulation (per-sample or
// per-slice GEMMs into fp32 partial products, then a sum over them: proj.py) writes and re-reads tens of MB of partials
// per weight; here a workgroup owns a 128 x 128 tile of dW and a contiguous slice of the token axis, accumulates in
// registers and adds its tile to dW once (contiguous 128-byte atomic runs; the slice count is chosen so that the adds
// stay a small fraction of the operand bytes).
//
// The contraction index of v_mfma_f32_32x32x16_bf16 must be contiguous inside a lane's operand (8 values of k for one
// row).  A plane-major tile [channel][token] has that natively (16-byte LDS reads).  A token-major tile [token][channel]
// is stored as it comes from HBM (16-byte writes, no transposing scatter) and read with ds_read_b64_tr_b16: per group of
// 16 lanes a 4-token x 16-channel block arrives transposed, two reads = the 8 tokens of a lane's operand
// (cdna_hip_programming.md T10; rows of 256 bytes with the chunk XOR of its image (b), conflict-free for both the
// 16-byte row writes and the transposed reads).
#include "xfm_common.hpp"

#include <algorithm>

namespace xfm {

typedef __bf16 wg_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x4_t __attribute__((ext_vector_type(4)));
typedef float wg_f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t wg_u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t wg_u32x2_t __attribute__((ext_vector_type(2)));

constexpr int kWgTile = 128;            // channels of each operand per workgroup
// tokens per step: BK = 64, or 128 for token-major x token-major products over long token runs (the Mlp weights): half the
// barriers and LDS-fill phases per token, twice the loads in flight
constexpr int kWgPlanePitch = 144;      // bytes per channel row of a plane-major tile (64 tokens: 128 + 16, conflict-free reads)
constexpr int kWgPlaneBytes = kWgTile * kWgPlanePitch;

struct WgradArgs {
    const uint16_t *a, *b;              // operands (bf16)
    float *dw;                          // (M, N) fp32, accumulated into
    int M, N;                           // channels of A / B
    int batch, L;                       // samples, tokens per sample
    int64_t a_bs, b_bs;                 // sample strides (elements)
    int steps_per_sample;               // ceil(L / 64)
    int total_steps, steps_per_slice;   // batch * steps_per_sample; steps a workgroup walks
    int dbg;                            // timing switches (XFM_WGRAD_DBG): 1 no atomics, 2 no MFMA / fragment reads, 4 no global loads
    // register-staged kernel only: row pitches of token-major operands (elements; 0 = M / N: dense rows) and a GROUP axis of
    // independent products in one launch (operand / result pointers advance by *_gs per group): the dt_proj weight gradient of
    // the four routes of an SS2D block contracts ddts (B, 4, L, D) against column blocks of the x_proj rows (B, L, 4 C2p)
    int lda, ldb, groups;
    int64_t a_gs, b_gs, dw_gs;
    // slices of UNEQUAL length (a linear ramp, +-stagger around the mean): equal slices all reach their 64 KB of atomic adds at
    // the same moment and the chip retires those at ~1.3 TB/s -- the tail of every workgroup waits behind everyone else's;
    // staggered ends put most of that traffic under the MFMA phase of the slices still running
    int nslices;
    float stagger;
    // XCD-aware placement: the grid is rounded up to a multiple of 8 and workgroup b (dispatched to XCD b % 8) takes the
    // virtual index (b % 8) * grid / 8 + b / 8, so an XCD runs CONSECUTIVE virtual indices -- tiles of the same token slice,
    // whose operand rows then meet in that XCD's L2 instead of being fetched once per tile (wgs = workgroups with work)
    int wgs, xcd_map;
    long long *prof;                    // debugging: 4 timestamps per workgroup (xfm_dbg_wgrad_prof), or null
    // LDS-direct kernel, CONV instance: operand B is the token-major INPUT MAP x (batch, H, W, C) of a 3 x 3 stride-2 padding-1
    // convolution and its N = 9 C columns are the taps (kh, kw, c) of the window of output token t -- the weight gradient
    // dW (O, 3, 3, C) = dy^T . windows(x) without the windows ever being written (csrc/conv_tok.hip)
    int cv_H, cv_W, cv_C, cv_OH, cv_OW;
};

__device__ uint4 wg_zero_page[4];     // what a tap in the padding reads

static long long *g_wgrad_prof = nullptr;

__device__ __forceinline__ int wg_virtual_id(const WgradArgs &a) {
    const int b = blockIdx.x;
    return a.xcd_map ? (b & 7) * (int)(gridDim.x >> 3) + (b >> 3) : b;
}

// first step of slice i (i = nslices: one past the last step)
__device__ __forceinline__ int wg_slice_start(const WgradArgs &a, const int i) {
    if (a.stagger <= 0.f) return min(i * a.steps_per_slice, a.total_steps);
    const float x = (float)i / (float)a.nslices;
    const int v = (int)((float)a.total_steps * (x * (1.f - a.stagger) + a.stagger * x * x) + 0.5f);
    return i >= a.nslices ? a.total_steps : min(v, a.total_steps);
}

// byte offset of 16-byte chunk ch (0..15) of token row `row` in a token-major tile (image (b) of the guide)
__device__ __forceinline__ int wg_tok_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// Global loads the compiler does not see (saddr form: uniform 64-bit base + 32-bit byte offset per lane).  Its own counter
// bookkeeping put a full s_waitcnt vmcnt(0) between the request for step st + 2 and the use of step st + 1 -- the operand
// pipeline ran one step deep and every step exposed a whole memory round trip (1500 of 3300 cycles).  With these the waits
// are explicit (wg_vm_wait) and counted: vector-memory operations return in order, each step requests the same number.
__device__ __forceinline__ void wg_gload(wg_u32x4_t &d, const uint16_t *sbase, const uint32_t byte_off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(byte_off), "s"(sbase) : "memory");
}
__device__ __forceinline__ void wg_gload(wg_u32x2_t &d, const uint16_t *sbase, const uint32_t byte_off) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(byte_off), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wg_vm_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool PL, int BK>
struct WgOperand {
    static_assert(!PL || BK == 64, "plane-major tiles are 64 tokens");
    static constexpr int NTV = BK / 16;   // 16-byte vectors per thread of a token-major tile (BK rows x 16 chunks / 256)
    // global -> registers: this thread's share of one 32-token x 128-channel tile
    //   token-major: BK / 16 vectors of 16 bytes (row = idx >> 4, chunk = idx & 15, idx = tid + 256 v)
    //   plane-major: 8 vectors of 8 bytes (channel = idx >> 4, 4 tokens at 4 * (idx & 15), idx = tid + 256 v)
    wg_u32x4_t tv[PL ? 1 : NTV];
    wg_u32x2_t pv[PL ? 8 : 1];

    // byte offset of vector v from (sample base + first token of the step) -- lane constants, set once (channels clamped)
    uint32_t boff[PL ? 8 : NTV];
    __device__ __forceinline__ void init(const int C, const int L, const int c0, const int tid, const int ld) {
        if constexpr (!PL) {
#pragma unroll
            for (int v = 0; v < NTV; ++v) {
                const int idx = tid + 256 * v, row = idx >> 4, ch = idx & 15;
                boff[v] = 2u * (uint32_t)(row * ld + min(c0 + 8 * ch, C - 8));
            }
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int idx = tid + 256 * v, chn = idx >> 4;
                boff[v] = 2u * (uint32_t)(min(c0 + chn, C - 1) * L + 4 * (idx & 15));
            }
        }
    }
    // a step that lies inside its sample whole: `sl` = sample base + first token of the step (uniform)
    __device__ __forceinline__ void load_full(const uint16_t *sl) {
        if constexpr (!PL) {
#pragma unroll
            for (int v = 0; v < NTV; ++v) wg_gload(tv[v], sl, boff[v]);
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) wg_gload(pv[v], sl, boff[v]);
        }
    }
    // RAG: plane-major rows whose length is not a multiple of 4 (7 x 7 maps) -- element loads under per-element bounds.
    // Otherwise the loads are UNCONDITIONAL at clamped addresses and rows past the end of the sample are zeroed by selects
    // on the way to LDS (store(): a select next to the load would wait for it):
    // with zero-fill-then-load-under-a-branch the compiler's counter bookkeeping gave up at the joins and put
    // s_waitcnt vmcnt(0) in front of every step's loads -- the step in flight was drained before the next was requested
    // (1500 of 3300 cycles per step).  Clamped channels (tiles past M / N) fill tile rows that are never written back.
    template <bool RAG>
    __device__ __forceinline__ void load(const uint16_t *base, const int64_t bs, const int C, const int L, const int c0,
                                         const int sample, const int l0, const int tid, const int ld) {
        const uint16_t *sb = base + sample * bs;
        if constexpr (!PL) {
#pragma unroll
            for (int v = 0; v < NTV; ++v) {
                const int idx = tid + 256 * v, row = idx >> 4, ch = idx & 15;
                const int l = l0 + row, c = min(c0 + 8 * ch, C - 8);
                const uint32_t off = (uint32_t)(min(l, L - 1) * ld + c);
                if constexpr (RAG) tv[v] = *reinterpret_cast<const wg_u32x4_t *>(sb + off);      // (compiler-counted: see RAG)
                else wg_gload(tv[v], sb, 2 * off);                         // (rows past L: zeroed in store())
            }
        } else if constexpr (!RAG) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int idx = tid + 256 * v, chn = idx >> 4, l = l0 + 4 * (idx & 15);
                const int c = min(c0 + chn, C - 1);
                const uint32_t off = (uint32_t)(c * L + min(l, L - 4));
                wg_gload(pv[v], sb, 2 * off);                              // (L % 4 == 0: 4 tokens in or out whole)
            }
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int idx = tid + 256 * v, chn = idx >> 4, l = l0 + 4 * (idx & 15);
                const int c = c0 + chn;
                pv[v] = wg_u32x2_t{0, 0};
                if (c < C && l < L) {
                    const uint16_t *p = sb + (int64_t)c * L + l;
                    const uint32_t e0 = p[0], e1 = l + 1 < L ? p[1] : 0u, e2 = l + 2 < L ? p[2] : 0u, e3 = l + 3 < L ? p[3] : 0u;
                    pv[v] = wg_u32x2_t{e0 | (e1 << 16), e2 | (e3 << 16)};
                }
            }
        }
    }
    // ---- the same in PARTS, one per k16-step of a step (token-major: one vector, plane-major: two): requests, LDS writes and
    // MFMAs of a step interleave, so that the address path (64 B / clk), the LDS and the matrix pipe work at the same time
    static constexpr int NS = BK / 16;                    // parts
    static constexpr int VP = (PL ? 8 : NTV) / NS;        // vectors per part
    uint32_t soff;                                        // LDS byte offset of vector 0 inside a tile (vector v: + SSTEP v)
    static constexpr int SSTEP = PL ? 16 * kWgPlanePitch : 4096;
    __device__ __forceinline__ void init_store(const int tid) {
        soff = PL ? (uint32_t)((tid >> 4) * kWgPlanePitch + 8 * (tid & 15)) : (uint32_t)wg_tok_off(tid >> 4, tid & 15);
    }
    template <int PART> __device__ __forceinline__ void load_full_part(const uint16_t *sl) {
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            constexpr int v0 = PART * VP;
            if constexpr (!PL) wg_gload(tv[v0 + j], sl, boff[v0 + j]);
            else wg_gload(pv[v0 + j], sl, boff[v0 + j]);
        }
    }
    // a step that runs past the end of its sample: token rows clamped to the last one (zeroed on the way to LDS)
    template <int PART>
    __device__ __forceinline__ void load_clamped_part(const uint16_t *sb, const int C, const int L, const int c0, const int l0,
                                                      const int tid, const int ld) {
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            const int v = PART * VP + j, idx = tid + 256 * v;
            if constexpr (!PL) {
                const int row = idx >> 4, ch = idx & 15;
                wg_gload(tv[v], sb, 2u * (uint32_t)(min(l0 + row, L - 1) * ld + min(c0 + 8 * ch, C - 8)));
            } else {
                const int chn = idx >> 4, l = l0 + 4 * (idx & 15);
                wg_gload(pv[v], sb, 2u * (uint32_t)(min(c0 + chn, C - 1) * L + min(l, L - 4)));
            }
        }
    }
    template <int PART> __device__ __forceinline__ void landed_part() {
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            if constexpr (!PL) asm volatile("" : "+v"(tv[PART * VP + j]));
            else asm volatile("" : "+v"(pv[PART * VP + j]));
        }
    }
    // registers -> LDS (asm: the position of the writes among the fragment reads is what the counted lgkmcnt waits assume)
    template <int PART, bool ZERO>
    __device__ __forceinline__ void store_part(const uint32_t tile, const int tid, const int l0, const int L) const {
        const uint32_t ad = tile + soff;
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            constexpr int v0 = PART * VP;
            const int idx = tid + 256 * (v0 + j);
            if constexpr (!PL) {
                wg_u32x4_t t = tv[v0 + j];
                if (ZERO && l0 + (idx >> 4) >= L) t = wg_u32x4_t{0, 0, 0, 0};
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ad), "v"(t), "n"(SSTEP * (v0 + j)) : "memory");
            } else {
                wg_u32x2_t t = pv[v0 + j];
                if (ZERO && l0 + 4 * (idx & 15) >= L) t = wg_u32x2_t{0, 0};
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ad), "v"(t), "n"(SSTEP * (v0 + j)) : "memory");
            }
        }
    }
    // after wg_vm_wait: every later use of the registers is ordered behind it
    __device__ __forceinline__ void landed() {
        if constexpr (!PL) {
#pragma unroll
            for (int v = 0; v < NTV; ++v) asm volatile("" : "+v"(tv[v]));
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) asm volatile("" : "+v"(pv[v]));
        }
    }
    static constexpr int NLOADS = PL ? 8 : NTV;           // vector loads per thread and step
    // l0: first token of the step these registers hold (tokens >= L of a sample's last step become zeros here)
    template <bool RAG>
    __device__ __forceinline__ void store(uint8_t *tile, const int tid, const int l0, const int L) const {
        if constexpr (!PL) {
#pragma unroll
            for (int v = 0; v < NTV; ++v) {
                const int idx = tid + 256 * v;
                wg_u32x4_t t = tv[v];
                if (l0 + (idx >> 4) >= L) t = wg_u32x4_t{0, 0, 0, 0};
                *reinterpret_cast<wg_u32x4_t *>(tile + wg_tok_off(idx >> 4, idx & 15)) = t;
            }
        } else {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int idx = tid + 256 * v;
                wg_u32x2_t t = pv[v];
                if (!RAG && l0 + 4 * (idx & 15) >= L) t = wg_u32x2_t{0, 0};
                *reinterpret_cast<wg_u32x2_t *>(tile + (idx >> 4) * kWgPlanePitch + 8 * (idx & 15)) = t;
            }
        }
    }
    // MFMA operand of k16-step s (tokens 16 s .. 16 s + 15 of the tile) for the 32 channels starting at ct:
    // lane (r = lane & 31, kb = lane >> 5) gets channel ct + r, tokens 16 s + 8 kb .. + 7.  Token-major tiles: the two
    // transposed reads are only ISSUED here (lo = tokens + 0..3, hi = + 4..7); wg_wait() below makes them usable.
    static __device__ __forceinline__ void frag(const uint8_t *tile, const int ct, const int s, const int lane,
                                                wg_bf16x4_t &lo, wg_bf16x4_t &hi) {
        const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)tile;
        if constexpr (PL) {
            const uint32_t ad = base + (ct + (lane & 31)) * kWgPlanePitch + 32 * s + 16 * (lane >> 5);
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=&v"(lo), "=&v"(hi) : "v"(ad) : "memory");
        } else {
            // 16-lane group g = lane >> 4: channels ct + 16 (g & 1) .. + 15, tokens 16 s + 8 (g >> 1) + 4 h .. + 3 (read h)
            const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
            const int c0 = (ct + 16 * (g & 1)) >> 3;                        // first 16-byte chunk of the block
            const int r0 = 16 * s + 8 * (g >> 1);
            const uint32_t a0 = base + wg_tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
            const uint32_t a1 = base + wg_tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1) : "memory");
        }
    }
};

// The fragment reads are inline asm (the compiler does not count them in lgkmcnt): explicit waits, tied to the eight
// registers they make valid.  LDS reads complete in order; every k16-step issues exactly 8 of them (two per operand
// tile), so "all but the 8 youngest" = the previous step's fragments while the next step's are in flight.
template <int OUTSTANDING>
__device__ __forceinline__ void wg_wait(wg_bf16x4_t (&lo)[4], wg_bf16x4_t (&hi)[4]) {
    if constexpr (OUTSTANDING == 0)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
    else
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}

template <bool APL, bool BPL, int BK, bool RAG = false>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradArgs a) {
    extern __shared__ __align__(16) uint8_t wg_lds[];
    constexpr int kWgTokBytes = BK * 256;
    constexpr int ABYTES = APL ? kWgPlaneBytes : kWgTokBytes, BBYTES = BPL ? kWgPlaneBytes : kWgTokBytes;
    auto At = [&](const int buf) { return wg_lds + buf * ABYTES; };
    auto Bt = [&](const int buf) { return wg_lds + 2 * ABYTES + buf * BBYTES; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbm = (a.M + kWgTile - 1) / kWgTile, nbn = (a.N + kWgTile - 1) / kWgTile;
    const int vid = wg_virtual_id(a);
    if (vid >= a.wgs) return;
    const int per_group = a.wgs / a.groups, grp = vid / per_group, bid = vid - grp * per_group;
    const int tile_id = bid % (nbm * nbn), slice = bid / (nbm * nbn);
    const int m0 = (tile_id / nbn) * kWgTile, n0 = (tile_id % nbn) * kWgTile;
    const int st0 = wg_slice_start(a, slice), st1 = wg_slice_start(a, slice + 1);
    if (st0 >= st1) return;                                // (uniform per workgroup)
    if (a.prof && tid == 0) a.prof[4 * blockIdx.x] = a.prof[4 * blockIdx.x + 1] = wall_clock64();
    const uint16_t *const pa = a.a + grp * a.a_gs, *const pb = a.b + grp * a.b_gs;
    float *const pdw = a.dw + grp * a.dw_gs;
    const int wm = wave >> 1, wn = wave & 1;               // wave -> 64 x 64 of the tile
    wg_f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // Operand tiles travel HBM -> registers -> LDS TWO steps ahead of their MFMAs: with one workgroup per CU (the slice
    // rule of the launcher) nothing else hides HBM latency, and one step of MFMAs (~0.3 us) does not cover it.  Register
    // set P = parity of the step it holds; set (st & 1) is refilled with step st + 2 right after its old content (step st)
    // went to LDS in the previous iteration.
    WgOperand<APL, BK> ra[2];
    WgOperand<BPL, BK> rb[2];
    auto issue = [&](auto par, const int st) {
        constexpr int P = decltype(par)::value;
        const int sample = st / a.steps_per_sample, l0 = (st - sample * a.steps_per_sample) * BK;
        if (a.dbg & 4) return;
        if (!RAG && l0 + BK <= a.L) {                       // (uniform) the step lies inside its sample: lane-constant offsets
            ra[P].load_full(pa + sample * a.a_bs + (int64_t)l0 * (APL ? 1 : a.lda));
            rb[P].load_full(pb + sample * a.b_bs + (int64_t)l0 * (BPL ? 1 : a.ldb));
        } else {
            ra[P].template load<RAG>(pa, a.a_bs, a.M, a.L, m0, sample, l0, tid, a.lda);
            rb[P].template load<RAG>(pb, a.b_bs, a.N, a.L, n0, sample, l0, tid, a.ldb);
        }
    };
    if constexpr (!RAG) {
        ra[0].init(a.M, a.L, m0, tid, a.lda);
        rb[0].init(a.N, a.L, n0, tid, a.ldb);
#pragma unroll
        for (int v = 0; v < WgOperand<APL, BK>::NLOADS; ++v) ra[1].boff[v] = ra[0].boff[v];
#pragma unroll
        for (int v = 0; v < WgOperand<BPL, BK>::NLOADS; ++v) rb[1].boff[v] = rb[0].boff[v];
    }
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto l0_of = [&](const int st) { return (st - (st / a.steps_per_sample) * a.steps_per_sample) * BK; };
    constexpr int NLD = WgOperand<APL, BK>::NLOADS + WgOperand<BPL, BK>::NLOADS;     // loads per thread and step
    // (requests in the part order of the step loop -- A part s, B part s -- which its counted waits assume)
    auto issue_in_parts = [&](auto par, const int st) {
        constexpr int P = decltype(par)::value;
        const int sample = st / a.steps_per_sample, l0 = (st - sample * a.steps_per_sample) * BK;
        if (a.dbg & 4) return;
        const uint16_t *sa = pa + sample * a.a_bs, *sb = pb + sample * a.b_bs;
        const bool full = l0 + BK <= a.L;
        const uint16_t *sla = sa + (int64_t)l0 * (APL ? 1 : a.lda), *slb = sb + (int64_t)l0 * (BPL ? 1 : a.ldb);
        auto one = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            if (full) {
                ra[P].template load_full_part<S>(sla);
                rb[P].template load_full_part<S>(slb);
            } else {
                ra[P].template load_clamped_part<S>(sa, a.M, a.L, m0, l0, tid, a.lda);
                rb[P].template load_clamped_part<S>(sb, a.N, a.L, n0, l0, tid, a.ldb);
            }
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
        if constexpr (BK == 128) {
            one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
            one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
        }
    };
    if constexpr (RAG) {
        issue(P0{}, st0);
        if (st0 + 1 < st1) issue(P1{}, st0 + 1);
    } else {
        issue_in_parts(P0{}, st0);
        if (st0 + 1 < st1) issue_in_parts(P1{}, st0 + 1);
    }
    if constexpr (!RAG) {
        if (st0 + 1 < st1) wg_vm_wait<NLD>(); else wg_vm_wait<0>();
        ra[0].landed();
        rb[0].landed();
    }
    ra[0].template store<RAG>(At(0), tid, l0_of(st0), a.L);
    rb[0].template store<RAG>(Bt(0), tid, l0_of(st0), a.L);
    __syncthreads();
    auto stamp = [&](const int st, const int k) {
        if (a.prof && blockIdx.x == 8 && lane == 0 && st - st0 < 40) a.prof[2048 + ((st - st0) * 4 + wave) * 5 + k] = clock64();
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)wg_lds;
    if constexpr (!RAG) {
        ra[0].init_store(tid); rb[0].init_store(tid);
        ra[1].soff = ra[0].soff; rb[1].soff = rb[0].soff;
    }
    // ---- a step in parts (see WgOperand): per k16-step s -- fragment reads of s + 1, requests for part s of step st + 2,
    // wait for the fragments of s, its four MFMAs, then part s of step st + 1 on its way to the other LDS buffer
    // (sample, first token) of steps st + 1 and st + 2 and the sample bases of st + 2 are CARRIED by the loop: with one wave per
    // SIMD every instruction costs an issue slot of >= 4 cycles and a step's 16 MFMAs cover 128 of them; derived from the step
    // index every step, the scalar address arithmetic alone was 140 of a step's 360 instructions.  (Four specialised step bodies
    // -- selects / clamps only where a step ends past its sample -- were tried: the accumulators changed register class at the
    // joins, 64-128 copies per step, slower.)
    int l0n = 0, l02 = 0;                                // first token of steps st + 1 / st + 2 inside their samples
    const uint16_t *sa2 = pa, *sb2 = pb;                 // sample bases of step st + 2
    {
        const int s1 = (st0 + 1) / a.steps_per_sample, s2 = (st0 + 2) / a.steps_per_sample;
        l0n = (st0 + 1 - s1 * a.steps_per_sample) * BK;
        l02 = (st0 + 2 - s2 * a.steps_per_sample) * BK;
        sa2 = pa + s2 * a.a_bs;
        sb2 = pb + s2 * a.b_bs;
    }
    const int span = a.steps_per_sample * BK;
    auto step_parts = [&](auto par, const bool c2, const int st) {
        constexpr int P = decltype(par)::value, NS = BK / 16;
        constexpr bool ZN = true;
        constexpr int LP = NLD / NS;                        // loads per part
        const int buf = P;
        const bool req = st + 2 < st1 && !(a.dbg & 4), put = st + 1 < st1;
        const uint16_t *sla = sa2 + (int64_t)l02 * (APL ? 1 : a.lda), *slb = sb2 + (int64_t)l02 * (BPL ? 1 : a.ldb);
        const uint32_t At1 = lds0 + (buf ^ 1) * ABYTES, Bt1 = lds0 + 2 * ABYTES + (buf ^ 1) * BBYTES;
        wg_bf16x4_t lo[2][4], hi[2][4];                     // [ring][0,1: A channel tiles, 2,3: B channel tiles]
        auto frags = [&](const int ring, const int s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                WgOperand<APL, BK>::frag(At(buf), wm * 64 + i * 32, s, lane, lo[ring][i], hi[ring][i]);
                WgOperand<BPL, BK>::frag(Bt(buf), wn * 64 + i * 32, s, lane, lo[ring][2 + i], hi[ring][2 + i]);
            }
        };
        auto part = [&](auto sc) {
            constexpr int S = decltype(sc)::value, r = S & 1;
            if constexpr (S + 1 < NS) frags(r ^ 1, S + 1);
            if (req) {
                if (!c2) {
                    ra[P].template load_full_part<S>(sla);
                    rb[P].template load_full_part<S>(slb);
                } else {
                    ra[P].template load_clamped_part<S>(sa2, a.M, a.L, m0, l02, tid, a.lda);
                    rb[P].template load_clamped_part<S>(sb2, a.N, a.L, n0, l02, tid, a.ldb);
                }
            }
            if constexpr (S + 1 < NS) wg_wait<8>(lo[r], hi[r]);
            else wg_wait<0>(lo[r], hi[r]);
            wg_bf16x8_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = __builtin_shufflevector(lo[r][i], hi[r][i], 0, 1, 2, 3, 4, 5, 6, 7);
                bf[i] = __builtin_shufflevector(lo[r][2 + i], hi[r][2 + i], 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (put) {
                // part S of step st + 1 has landed when only what was requested after it remains: its own later parts and,
                // if this step requests, parts 0..S of step st + 2 -- together one step's worth
                if (req) wg_vm_wait<NLD>(); else wg_vm_wait<LP * (NS - 1 - S)>();
                ra[P ^ 1].template landed_part<S>();
                rb[P ^ 1].template landed_part<S>();
                ra[P ^ 1].template store_part<S, ZN>(At1, tid, l0n, a.L);
                rb[P ^ 1].template store_part<S, ZN>(Bt1, tid, l0n, a.L);
            }
        };
        frags(0, 0);
        part(std::integral_constant<int, 0>{});
        part(std::integral_constant<int, 1>{});
        part(std::integral_constant<int, 2>{});
        part(std::integral_constant<int, 3>{});
        if constexpr (NS == 8) {
            part(std::integral_constant<int, 4>{});
            part(std::integral_constant<int, 5>{});
            part(std::integral_constant<int, 6>{});
            part(std::integral_constant<int, 7>{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the asm LDS writes are not in the compiler's books)
        __syncthreads();
    };
    auto step_any = [&](auto par, const int st) {
        step_parts(par, l02 + BK > a.L, st);                // (uniform) does step st + 2 run past the end of its sample?
        l0n = l02;                                          // step st + 2 becomes st + 1; advance st + 2
        l02 += BK;
        if (l02 >= span) {
            l02 = 0;
            sa2 += a.a_bs;
            sb2 += a.b_bs;
        }
    };
    auto step = [&](auto par, const int st) {
        constexpr int P = decltype(par)::value;               // parity of (st - st0): LDS buffer and register set of step st
        const int buf = P;
        stamp(st, 0);
        if (st + 2 < st1) issue(par, st + 2);
        stamp(st, 1);
        wg_bf16x4_t lo[2][4], hi[2][4];                     // [ring][0,1: A channel tiles, 2,3: B channel tiles]
        auto frags = [&](const int ring, const int s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                WgOperand<APL, BK>::frag(At(buf), wm * 64 + i * 32, s, lane, lo[ring][i], hi[ring][i]);
                WgOperand<BPL, BK>::frag(Bt(buf), wn * 64 + i * 32, s, lane, lo[ring][2 + i], hi[ring][2 + i]);
            }
        };
        if (!(a.dbg & 2)) {
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int r = s & 1;
            if (s + 1 < BK / 16) {
                frags(r ^ 1, s + 1);
                wg_wait<8>(lo[r], hi[r]);
            } else {
                wg_wait<0>(lo[r], hi[r]);
            }
            wg_bf16x8_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = __builtin_shufflevector(lo[r][i], hi[r][i], 0, 1, 2, 3, 4, 5, 6, 7);
                bf[i] = __builtin_shufflevector(lo[r][2 + i], hi[r][2 + i], 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        }
        stamp(st, 2);
        if (st + 1 < st1) {
            const int l0n = l0_of(st + 1);
            if constexpr (!RAG) {                           // step st + 1 has landed when only step st + 2's loads remain
                if (st + 2 < st1 && !(a.dbg & 4)) wg_vm_wait<NLD>(); else wg_vm_wait<0>();
                ra[P ^ 1].landed();
                rb[P ^ 1].landed();
            }
            ra[P ^ 1].template store<RAG>(At(buf ^ 1), tid, l0n, a.L);
            rb[P ^ 1].template store<RAG>(Bt(buf ^ 1), tid, l0n, a.L);
        }
        stamp(st, 3);
        __syncthreads();
        stamp(st, 4);
    };
    for (int st = st0; st < st1; st += 2) {
        if constexpr (RAG) {
            step(P0{}, st);
            if (st + 1 < st1) step(P1{}, st + 1);
        } else {
            step_any(P0{}, st);
            if (st + 1 < st1) step_any(P1{}, st + 1);
        }
    }
    if (a.prof && tid == 0) a.prof[4 * blockIdx.x + 2] = wall_clock64();
    // D[m][n]: column n = lane & 31, row m = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): 32 lanes = 128 contiguous bytes of a row
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + c;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m < a.M && n < a.N && !(a.dbg & 1)) atomicAdd(pdw + (int64_t)m * a.N + n, acc[i][j][v]);
            }
        }
    if (a.prof && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.prof[4 * blockIdx.x + 3] = wall_clock64();
    }
}

// ---- token-major x token-major, operand tiles HBM -> LDS directly ------------------------------------------------------
// The register-staged kernel above spends a third of a launch in its LDS-fill phases (timing switches: ~15 of 50 us on
// the 12544 x 1536 x 384 product) and keeps one or two tiles in flight.  Here a tile never touches a VGPR:
// global_load_lds_dwordx4 writes lane l's 16 bytes at LDS base + 16 l -- a lane-linear image -- so the chunk XOR of the
// transposed-read layout is applied on the GLOBAL side (lane l of the instruction that fills rows 4 i .. 4 i + 3 reads
// chunk (l & 15) ^ swz(row) of row 4 i + (l >> 4)); four 64-token stages of both operands (128 KB) ring through LDS, three
// in flight while one feeds the MFMAs; one workgroup barrier per stage.  Needs whole stages (tokens % 64 == 0: the Mlp
// products) and whole 128-channel tiles only in the sense that rows past M / N read clamped addresses: such rows and
// columns of the tile are never written back.
constexpr int kGlStages = 4, kGlBK = 64, kGlTile = kGlBK * 256;     // bytes of one operand stage

// transposed fragment read at a compile-time byte offset from a per-lane address
template <int OFF> __device__ __forceinline__ void wg_tr_read(wg_bf16x4_t &d, const uint32_t ad) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(OFF) : "memory");
}

// NW waves per workgroup: 4 (a wave owns 64 x 64 of the tile) or 8 (32 A-channels x 64 B-channels: two waves per SIMD behind the same
// four-stage ring).  Eight is an experiment switch (XFM_WGRAD_NW=8), not the default: 12544 x 1536 x 384 28.0 -> 29.9 us, the
// fusion blocks' 9408 x 1536 x 1536 81.6 -> 88.7, the small products unchanged -- a wave's 64 x 64 block already reads 1 KB of
// fragments per MFMA, i.e. LDS reads take as long as the MFMAs, and a 32 x 64 block reads 1.5 KB.  (Timing switches at
// 1536 x 384: no operand loads and no atomics 25.4 us, no fragment reads / MFMAs 24.6, everything 28 -- the two halves are equally
// long and overlap; the atomics cost 0.7 us there and 6 ... 7 us of the 18 ... 25 us small products.)
template <int NW> __device__ __forceinline__ void wg_wait_next(wg_bf16x4_t (&lo)[4], wg_bf16x4_t (&hi)[4]) {
    if constexpr (NW == 4)
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
    else
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo[0]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[2]), "+v"(hi[3]));
}
template <int NW> __device__ __forceinline__ void wg_wait_all(wg_bf16x4_t (&lo)[4], wg_bf16x4_t (&hi)[4]) {
    if constexpr (NW == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[2]), "+v"(hi[3]));
}

template <bool DBG, bool CONV = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW, 1) wgrad_tt_glds_kernel(const WgradArgs a) {
    extern __shared__ __align__(16) uint8_t wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dbg = DBG ? a.dbg : 0;                       // timing switches only in the debugging instance
    const int nbm = (a.M + kWgTile - 1) / kWgTile, nbn = (a.N + kWgTile - 1) / kWgTile;
    const int vid = wg_virtual_id(a);
    if (vid >= a.wgs) return;
    if (DBG && a.prof && tid == 0) a.prof[4 * blockIdx.x] = wall_clock64();
    const int tile_id = vid % (nbm * nbn), slice = vid / (nbm * nbn);
    const int m0 = (tile_id / nbn) * kWgTile, n0 = (tile_id % nbn) * kWgTile;
    const int st0 = wg_slice_start(a, slice), st1 = wg_slice_start(a, slice + 1);
    if (st0 >= st1) return;
    constexpr int NI = NW == 4 ? 2 : 1;                    // 32-channel A blocks of a wave
    constexpr int NP = 16 / NW;                            // 1 KB pieces of a stage (per operand) a wave requests
    const int wm = wave >> 1, wn = wave & 1;
    wg_f32x16_t acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // stage buffer b: A tile at 2 b kGlTile, B tile kGlTile behind it
    // this lane's share of a stage: instruction i (0..3) of this wave fills rows 4 (4 wave + i) .. + 3 of the tile
    int64_t offa[NP], offb[NP];                           // element offsets inside a 64-token stage of A / B
    // CONV: the window of the token this lane's row of part i holds in the NEXT stage to request -- sample, output row / column
    // -- and the lane's tap: element offset from the window's centre-row, left-column pixel (2 oh, 2 ow) of the sample, and whether
    // the tap lies in the top / left padding for oh / ow = 0
    int cvn[NP], cvh[NP], cvw[NP], cvtap[NP], cvpad[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = 4 * (NP * wave + i) + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const int ca = min(m0 + 8 * ch, a.M - 8), cb = min(n0 + 8 * ch, a.N - 8);      // clamped: see above
        offa[i] = (int64_t)row * a.M + ca;
        offb[i] = (int64_t)row * a.N + cb;
        if constexpr (CONV) {
            const int tap = cb / a.cv_C, c = cb - tap * a.cv_C, kh = tap / 3, kw = tap - 3 * kh;
            cvtap[i] = ((kh - 1) * a.cv_W + (kw - 1)) * a.cv_C + c;
            cvpad[i] = (kh == 0 ? 1 : 0) | (kw == 0 ? 2 : 0);
            const int t = st0 * kGlBK + row, per = a.cv_OH * a.cv_OW;
            cvn[i] = t / per;
            const int r = t - cvn[i] * per;
            cvh[i] = r / a.cv_OW;
            cvw[i] = r - cvh[i] * a.cv_OW;
        }
    }
    const int cv_qo = CONV ? kGlBK / a.cv_OW : 0, cv_ro = CONV ? kGlBK - cv_qo * a.cv_OW : 0;   // a stage further: 64 tokens
    // The loads of a stage are issued in FOUR parts, one per k16-step of the stage being multiplied: a 1 KB LDS-direct load
    // keeps the CU's address path busy for 16 cycles (64 B / clk) -- the 32 of a stage (32 KB) for as long as the stage's 16
    // MFMAs per wave take -- and a wave that issues its eight loads back to back sits in the issue queue for that long
    // before its first MFMA (measured: 600 of 2400 cycles per stage).
    // (scalar stage bases are carried by the loop: derived from the stage index at every use they cost ~25 scalar
    //  instructions per part, and with one wave per SIMD every instruction of any kind takes an issue slot of ~4 cycles:
    //  an MFMA covers 8 of them)
    auto issue_part = [&](const uint16_t *pa, const uint16_t *pb, const int buf, const int i) {
        if (dbg & 4) return;
        uint8_t *dst = wg_lds + buf * 2 * kGlTile + (NP * wave + i) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pa + offa[i]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        const uint16_t *src = pb + offb[i];
        if constexpr (CONV) {
            const bool pad = ((cvpad[i] & 1) && cvh[i] == 0) || ((cvpad[i] & 2) && cvw[i] == 0);
            const int pix = ((cvn[i] * a.cv_H + 2 * cvh[i]) * a.cv_W + 2 * cvw[i]) * a.cv_C + cvtap[i];
            src = pad ? reinterpret_cast<const uint16_t *>(wg_zero_page) : a.b + pix;
            // this row's token one stage on (a stage is fewer than OH output rows: one wrap each at most)
            cvw[i] += cv_ro;
            const int c1 = cvw[i] >= a.cv_OW ? 1 : 0;
            cvw[i] -= c1 * a.cv_OW;
            cvh[i] += cv_qo + c1;
            const int c2 = cvh[i] >= a.cv_OH ? 1 : 0;
            cvh[i] -= c2 * a.cv_OH;
            cvn[i] += c2;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(dst + kGlTile), 16, 0, 0);
    };
    const int64_t sa = (int64_t)kGlBK * a.M, sb = (int64_t)kGlBK * a.N;       // elements per stage (batch == 1: one token run)
    const uint16_t *pan = a.a + (int64_t)st0 * sa, *pbn = a.b + (int64_t)st0 * sb;
    int bufn = 0;                                         // pan / pbn / bufn: the next stage to request
#pragma unroll
    for (int q = 0; q < kGlStages - 1; ++q)
        if (st0 + q < st1) {
#pragma unroll
            for (int i = 0; i < NP; ++i) issue_part(pan, pbn, bufn, i);
            pan += sa; pbn += sb; bufn = (bufn + 1) & (kGlStages - 1);
        }
    if (DBG && a.prof && tid == 0) a.prof[4 * blockIdx.x + 1] = wall_clock64();
    // fragment addresses of k16-step 0 in stage buffer 0 (wg_tok_off: the XOR term does not depend on the k16-step, so
    // step s is a constant 4096 s bytes further and the B tile kGlTile: immediate offsets of the reads)
    uint32_t fr[4][2];                                    // [0, 1: A channel tiles, 2, 3: B channel tiles][lo, hi]
    {
        const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)wg_lds;
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int r0 = 8 * (g >> 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ct = t < 2 ? wm * 32 * NI + (t & 1) * 32 : wn * 64 + (t & 1) * 32;
            const int c0 = (ct + 16 * (g & 1)) >> 3;
            fr[t][0] = base + wg_tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
            fr[t][1] = base + wg_tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
        }
    }
    static_assert((kGlStages & (kGlStages - 1)) == 0, "stage ring is a power of two");
    int buf = 0;
    for (int st = st0; st < st1; ++st) {
        // stage st has landed when at most the 2 NP loads of each later stage in flight remain outstanding
        const int later = min(st1 - 1 - st, kGlStages - 2);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // everyone's share of stage st is in LDS; everyone is done with st - 1.  A BARE barrier: __syncthreads() carries a
        // workgroup fence, and for LDS-direct loads the compiler turns that into s_waitcnt vmcnt(0) -- every stage would wait
        // for ALL loads in flight, i.e. run with no prefetch at all (the fragment reads of st - 1 were drained by the last
        // wg_wait<0>, the loads of st by the counted wait above)
        __builtin_amdgcn_s_barrier();
        const bool more = st + kGlStages - 1 < st1;       // stage st + 3 refills the buffer stage st - 1 used
        if (dbg & 2) {
            if (more)
                for (int i = 0; i < NP; ++i) issue_part(pan, pbn, bufn, i);
        } else {
            uint32_t ad[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ad[t][0] = fr[t][0] + buf * 2 * kGlTile;
                ad[t][1] = fr[t][1] + buf * 2 * kGlTile;
            }
            wg_bf16x4_t lo[2][4], hi[2][4];
            auto frags = [&](const int ring, auto sc) {
                constexpr int S = decltype(sc)::value;
                wg_tr_read<4096 * S>(lo[ring][0], ad[0][0]);
                wg_tr_read<4096 * S>(hi[ring][0], ad[0][1]);
                wg_tr_read<4096 * S + kGlTile>(lo[ring][2], ad[2][0]);
                wg_tr_read<4096 * S + kGlTile>(hi[ring][2], ad[2][1]);
                if constexpr (NI == 2) {
                    wg_tr_read<4096 * S>(lo[ring][1], ad[1][0]);
                    wg_tr_read<4096 * S>(hi[ring][1], ad[1][1]);
                }
                wg_tr_read<4096 * S + kGlTile>(lo[ring][3], ad[3][0]);
                wg_tr_read<4096 * S + kGlTile>(hi[ring][3], ad[3][1]);
            };
            auto k16 = [&](auto sc) {
                constexpr int S = decltype(sc)::value, r = S & 1;
                if constexpr (S + 1 < kGlBK / 16) {
                    frags(r ^ 1, std::integral_constant<int, S + 1>{});
                    if constexpr (S < NP) {
                        if (more) issue_part(pan, pbn, bufn, S);
                    }
                    wg_wait_next<NW>(lo[r], hi[r]);
                } else {
                    if constexpr (S < NP) {
                        if (more) issue_part(pan, pbn, bufn, S);
                    }
                    wg_wait_all<NW>(lo[r], hi[r]);
                }
                wg_bf16x8_t af[NI], bf[2];
#pragma unroll
                for (int i = 0; i < NI; ++i) af[i] = __builtin_shufflevector(lo[r][i], hi[r][i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int i = 0; i < 2; ++i) bf[i] = __builtin_shufflevector(lo[r][2 + i], hi[r][2 + i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            };
            frags(0, std::integral_constant<int, 0>{});
            k16(std::integral_constant<int, 0>{});
            k16(std::integral_constant<int, 1>{});
            k16(std::integral_constant<int, 2>{});
            k16(std::integral_constant<int, 3>{});
        }
        pan += sa; pbn += sb;
        bufn = (bufn + 1) & (kGlStages - 1);
        buf = (buf + 1) & (kGlStages - 1);
    }
    if (DBG && a.prof && tid == 0) a.prof[4 * blockIdx.x + 2] = wall_clock64();
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + c;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 32 * NI + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m < a.M && n < a.N && !(dbg & 1)) atomicAdd(a.dw + (int64_t)m * a.N + n, acc[i][j][v]);
            }
        }
    if (DBG && a.prof && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.prof[4 * blockIdx.x + 3] = wall_clock64();
    }
}

static bool wg_xcd_map() {
    static const bool on = [] { const char *e = getenv("XFM_WGRAD_XCD"); return !e || atoi(e) != 0; }();
    return on;
}

template <bool APL, bool BPL, int BK, bool RAG = false>
static int wgrad_launch(WgradArgs a, int nslices, hipStream_t s) {
    constexpr int kWgTokBytes = BK * 256;
    constexpr int ABYTES = APL ? kWgPlaneBytes : kWgTokBytes, BBYTES = BPL ? kWgPlaneBytes : kWgTokBytes;
    const size_t lds = 2 * (ABYTES + BBYTES);
    const int nbm = (a.M + kWgTile - 1) / kWgTile, nbn = (a.N + kWgTile - 1) / kWgTile;
    if (lds > 64 * 1024) {
        static LdsOptIn attr;
        if (!lds_opt_in(attr, (const void *)wgrad_kernel<APL, BPL, BK, RAG>, lds)) return XFM_ELAUNCH;
    }
    a.wgs = nbm * nbn * nslices;
    a.xcd_map = wg_xcd_map() ? 1 : 0;
    const int grid = a.xcd_map ? (a.wgs + 7) / 8 * 8 : a.wgs;
    hipLaunchKernelGGL((wgrad_kernel<APL, BPL, BK, RAG>), dim3((unsigned)grid), dim3(256), lds, s, a);
    return check_launch();
}

}  // namespace xfm

namespace xfm {
// Token-major x token-major with row pitches and a group axis (see WgradArgs): dW[g] (M, N) += sum_{b, l} A[g][b, l, :M] (x)
// B[g][b, l, :N].  Used by xfm_ss2dc_post for the dt_proj weight gradient of the four routes in ONE launch.  M, N % 8 == 0,
// 16-byte aligned rows.  (xfm namespace, not part of the C ABI.)
int wgrad_grouped(const void *a, const void *b, float *dw, int M, int N, int batch, int L, int64_t a_bs, int64_t b_bs, int lda,
                  int ldb, int groups, int64_t a_gs, int64_t b_gs, int64_t dw_gs, hipStream_t s) {
    if (M % 8 || N % 8 || lda % 8 || ldb % 8 || a_bs % 8 || b_bs % 8 || a_gs % 8 || b_gs % 8) return XFM_ELIMIT;
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15)) return XFM_EINVAL;
    WgradArgs w{};
    w.a = (const uint16_t *)a; w.b = (const uint16_t *)b; w.dw = dw;
    w.M = M; w.N = N; w.batch = batch; w.L = L; w.a_bs = a_bs; w.b_bs = b_bs;
    w.lda = lda; w.ldb = ldb; w.groups = groups; w.a_gs = a_gs; w.b_gs = b_gs; w.dw_gs = dw_gs;
    constexpr int BK = 64;
    w.steps_per_sample = (L + BK - 1) / BK;
    w.total_steps = batch * w.steps_per_sample;
    const int tiles = ((M + kWgTile - 1) / kWgTile) * ((N + kWgTile - 1) / kWgTile);
    // two workgroups per CU; a workgroup's atomic tail is its (128 x N) corner only, so short slices are fine
    int nsl = std::max(1, std::min(512 / (tiles * groups), w.total_steps / 4));
    w.steps_per_slice = (w.total_steps + nsl - 1) / nsl;
    nsl = (w.total_steps + w.steps_per_slice - 1) / w.steps_per_slice;
    w.nslices = nsl;
    return wgrad_launch<false, false, 64>(w, nsl * groups, s);
}
}  // namespace xfm

extern "C" {

/* debugging (tools/wgradprof.py): device buffer of 4 x int64 per workgroup that the LDS-direct kernel stamps with the 100 MHz
 * wall clock at entry, after its prologue loads, after its last stage and after its adds have drained; null switches it off */
void xfm_dbg_wgrad_prof(void *buf) { xfm::g_wgrad_prof = (long long *)buf; }

int xfm_wgrad_supported(int M, int N, int L, int a_planes, int b_planes) {
    if (M <= 0 || N <= 0 || L <= 0) return 0;
    if (!a_planes && M % 8 != 0) return 0;                  // token-major rows are read in 16-byte vectors
    if (!b_planes && N % 8 != 0) return 0;
    return 1;
}

int xfm_wgrad(const void *a, const void *b, float *dw, int M, int N, int batch, int L, int64_t a_bs, int64_t b_bs,
              int a_planes, int b_planes, void *stream) {
    using namespace xfm;
    if (!a || !b || !dw || batch <= 0) return XFM_EINVAL;
    if (!xfm_wgrad_supported(M, N, L, a_planes, b_planes)) return XFM_ELIMIT;
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15)) return XFM_EINVAL;
    const bool rag = (L & 3) != 0;                           // plane-major rows of ragged length: no alignment needed
    if ((!a_planes && (a_bs % 8)) || (!b_planes && (b_bs % 8)) || (a_planes && !rag && (a_bs % 4)) || (b_planes && !rag && (b_bs % 4)))
        return XFM_EINVAL;
    if (rag && (((uintptr_t)a | (uintptr_t)b) & 1)) return XFM_EINVAL;
    WgradArgs w{};
    w.a = (const uint16_t *)a; w.b = (const uint16_t *)b; w.dw = dw;
    w.M = M; w.N = N; w.batch = batch; w.L = L; w.a_bs = a_bs; w.b_bs = b_bs;
    w.lda = M; w.ldb = N; w.groups = 1;
    const bool glds = !a_planes && !b_planes && batch == 1 && L % kGlBK == 0 && L >= 2048 && M >= 8 && N >= 8 &&
                      !getenv("XFM_WGRAD_NO_GLDS");
    // (128-token steps for long token-major runs were dropped with the step-in-parts pipeline: eight parts in four variants
    //  spilled, and a spilled destination of an asynchronous asm load is saved before the data arrives)
    const int BK = 64;
    w.steps_per_sample = (L + BK - 1) / BK;
    if (const char *env = getenv("XFM_WGRAD_DBG")) w.dbg = atoi(env);
    w.prof = g_wgrad_prof;
    w.total_steps = batch * w.steps_per_sample;
    // Slices of the token axis.  Every workgroup ends with 64 KB of fp32 atomic adds, and the chip retires those at
    // ~1.3 TB/s against ~5+ TB/s of operand streaming: the adds of ALL workgroups (tiles * slices * 64 KB) are the floor of
    // the launch.  So: no more workgroups than CUs (one round), and at least 1024 tokens (256 KB of operands) per workgroup.
    const int tiles = ((M + kWgTile - 1) / kWgTile) * ((N + kWgTile - 1) / kWgTile);
    int cap = 256;
    if (const char *env = getenv("XFM_WGRAD_WGS")) cap = atoi(env);
    static const int min_tokens = [] { const char *e = getenv("XFM_WGRAD_MINTOK"); return e ? atoi(e) : 512; }();
    int nsl = std::max(1, std::min(cap / tiles, w.total_steps / std::max(1, min_tokens / BK)));
    w.steps_per_slice = (w.total_steps + nsl - 1) / nsl;
    nsl = (w.total_steps + w.steps_per_slice - 1) / w.steps_per_slice;
    w.nslices = nsl;
    // (measured after the stage loop went from 40 to 28 us: many tiles x few slices -- 384 x 1536 over 12544 tokens -- 30.3 /
    //  28.5 / 26.3 us at stagger 0 / 0.25 / 0.5; few tiles x many slices -- 96 x 384 over 200704 -- 37.0 / 36.0 / 40.4 at 0 /
    //  0.15 / 0.5; nothing on the register-staged kernels)
    static const float stag = [] { const char *e = getenv("XFM_WGRAD_STAGGER"); return e ? (float)atof(e) : -1.f; }();
    w.stagger = (glds && nsl >= 4) ? (stag >= 0.f ? stag : (tiles >= 16 ? 0.5f : 0.2f)) : 0.f;
    hipStream_t s = (hipStream_t)stream;
    if (glds) {
        const size_t lds = (size_t)kGlStages * 2 * kGlTile;
        static xfm::LdsOptIn attr[3];
        if (!xfm::lds_opt_in(attr[0], (const void *)wgrad_tt_glds_kernel<false>, lds) ||
            !xfm::lds_opt_in(attr[1], (const void *)wgrad_tt_glds_kernel<true>, lds) ||
            !xfm::lds_opt_in(attr[2], (const void *)wgrad_tt_glds_kernel<false, false, 8>, lds))
            return XFM_ELAUNCH;
        w.wgs = tiles * nsl;
        w.xcd_map = wg_xcd_map() ? 1 : 0;
        const int grid = w.xcd_map ? (w.wgs + 7) / 8 * 8 : w.wgs;
        static const int env_nw = [] { const char *e = getenv("XFM_WGRAD_NW"); return e ? atoi(e) : 4; }();
        if (w.dbg || w.prof) hipLaunchKernelGGL(wgrad_tt_glds_kernel<true>, dim3((unsigned)grid), dim3(256), lds, s, w);
        else if (env_nw == 8) hipLaunchKernelGGL((wgrad_tt_glds_kernel<false, false, 8>), dim3((unsigned)grid), dim3(512), lds, s, w);
        else hipLaunchKernelGGL(wgrad_tt_glds_kernel<false>, dim3((unsigned)grid), dim3(256), lds, s, w);
        return check_launch();
    }
    if (rag && (a_planes || b_planes)) {
        if (a_planes) return b_planes ? wgrad_launch<true, true, 64, true>(w, nsl, s) : wgrad_launch<true, false, 64, true>(w, nsl, s);
        return wgrad_launch<false, true, 64, true>(w, nsl, s);
    }
    if (a_planes) return b_planes ? wgrad_launch<true, true, 64>(w, nsl, s) : wgrad_launch<true, false, 64>(w, nsl, s);
    if (b_planes) return wgrad_launch<false, true, 64>(w, nsl, s);
    return wgrad_launch<false, false, 64>(w, nsl, s);
}

/* Weight gradient of a 3 x 3, stride-2, padding-1 convolution on token-major maps straight from the input map (see
 * xfm_conv3x3s2_tokens_* in include/xfm_hip.h): dweight (O, 3, 3, C) fp32, ZEROED by the caller, += dy^T . windows(x) with
 * dy (B, H/2, W/2, O) and x (B, H, W, C) bf16 -- the LDS-direct token x token kernel with the rows of its second operand
 * gathered from x (a tap in the padding reads a zero page), no (tokens, 9 C) workspace.  _supported: whole 64-token stages
 * (B H/2 W/2 % 64 == 0, at least 2048 tokens), maps whose 64-token stage spans fewer than H/2 - 1 output rows, C, O % 8 == 0. */
int xfm_conv3x3s2_tokens_bwd_weight_x_supported(int B, int H, int W, int C, int O) {
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || C % 8 || O < 8 || O % 8) return 0;
    const long long T = (long long)B * (H / 2) * (W / 2);
    if (T % xfm::kGlBK || T < 2048 || T > 0x7fffffffll || (long long)B * H * W * C >= 0x7fffffffll) return 0;
    return (H / 2) >= xfm::kGlBK / (W / 2) + 2 ? 1 : 0;
}

int xfm_conv3x3s2_tokens_bwd_weight_x(const void *dy, const void *x, float *dweight, int B, int H, int W, int C, int O,
                                      void *stream) {
    using namespace xfm;
    if (!dy || !x || !dweight) return XFM_EINVAL;
    if (!xfm_conv3x3s2_tokens_bwd_weight_x_supported(B, H, W, C, O)) return XFM_ELIMIT;
    if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return XFM_EINVAL;
    WgradArgs w{};
    w.a = (const uint16_t *)dy; w.b = (const uint16_t *)x; w.dw = dweight;
    w.M = O; w.N = 9 * C; w.batch = 1; w.L = (int)((long long)B * (H / 2) * (W / 2));
    w.lda = w.M; w.ldb = w.N; w.groups = 1;
    w.cv_H = H; w.cv_W = W; w.cv_C = C; w.cv_OH = H / 2; w.cv_OW = W / 2;
    w.steps_per_sample = w.L / kGlBK;
    w.total_steps = w.steps_per_sample;
    const int tiles = ((w.M + kWgTile - 1) / kWgTile) * ((w.N + kWgTile - 1) / kWgTile);
    int nsl = std::max(1, std::min(256 / tiles, w.total_steps / 8));     // (as xfm_wgrad: one round of workgroups, >= 512 tokens each)
    w.steps_per_slice = (w.total_steps + nsl - 1) / nsl;
    nsl = (w.total_steps + w.steps_per_slice - 1) / w.steps_per_slice;
    w.nslices = nsl;
    w.stagger = nsl >= 4 ? (tiles >= 16 ? 0.5f : 0.2f) : 0.f;
    const size_t lds = (size_t)kGlStages * 2 * kGlTile;
    static xfm::LdsOptIn attr[2];
    if (!xfm::lds_opt_in(attr[0], (const void *)wgrad_tt_glds_kernel<false, true>, lds) ||
        !xfm::lds_opt_in(attr[1], (const void *)wgrad_tt_glds_kernel<false, true, 8>, lds))
        return XFM_ELAUNCH;
    w.wgs = tiles * nsl;
    w.xcd_map = wg_xcd_map() ? 1 : 0;
    const int grid = w.xcd_map ? (w.wgs + 7) / 8 * 8 : w.wgs;
    static const int env_nw = [] { const char *e = getenv("XFM_WGRAD_NW"); return e ? atoi(e) : 4; }();
    if (env_nw == 8) hipLaunchKernelGGL((wgrad_tt_glds_kernel<false, true, 8>), dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, w);
    else hipLaunchKernelGGL((wgrad_tt_glds_kernel<false, true>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, w);
    return check_launch();
}

}  // extern "C"

// scan_core.hpp -- wave-level building blocks shared by selective_scan.hip and ss2d_fused.hip.
#pragma once
#include "xfm_common.hpp"

namespace xfm {

// ---------------------------------------------------------------------------------------------
// per-wave tile <-> register transposition
// buf layout: element j of row g lives at lane' = g*LPR + j/C, slot j%C -> buf[lane'*PC + slot]
// ---------------------------------------------------------------------------------------------
// `slot` = (row-in-tile << lg_lpr) + chunk index this lane consumes (== lane for ascending scans);
// REV reverses the element order inside the chunk (register j <- chunk element C-1-j).
template <typename T, int C, bool REV = false>
__device__ __forceinline__ void tile_load(float *buf, const T *base, int64_t row_stride, int G, int lg_lpr, int s0,
                                          int L, int lane, int slot, float (&v)[C]) {
    constexpr int PC = C | 1;
    const int SL = C << lg_lpr;
    for (int g = 0; g < G; ++g) {
        const T *row = base + (int64_t)g * row_stride;
        for (int j = lane; j < SL; j += 64) {
            const int t = s0 + j;
            const float val = t < L ? ldf<T>(row + t) : 0.f;
            const int q = j / C;
            buf[((g << lg_lpr) + q) * PC + (j - q * C)] = val;
        }
    }
    wave_sync();
#pragma unroll
    for (int jj = 0; jj < C; ++jj) v[jj] = buf[slot * PC + (REV ? C - 1 - jj : jj)];
    wave_sync();
}

template <typename T, int C, bool REV = false>
__device__ __forceinline__ void tile_store(float *buf, T *base, int64_t row_stride, int G, int lg_lpr, int s0, int L,
                                           int lane, int slot, const float (&v)[C]) {
    constexpr int PC = C | 1;
    const int SL = C << lg_lpr;
#pragma unroll
    for (int jj = 0; jj < C; ++jj) buf[slot * PC + (REV ? C - 1 - jj : jj)] = v[jj];
    wave_sync();
    for (int g = 0; g < G; ++g) {
        T *row = base + (int64_t)g * row_stride;
        for (int j = lane; j < SL; j += 64) {
            const int t = s0 + j;
            const int q = j / C;
            if (t < L) stf<T>(row + t, buf[((g << lg_lpr) + q) * PC + (j - q * C)]);
        }
    }
    wave_sync();
}

// Segmented inclusive scan of affine maps over the LPR lanes of a row, ascending lane order.
// (P,S) represents h -> P*h + S; on return lane i holds the composition of lanes 0..i.
__device__ __forceinline__ void seg_scan_up(float &P, float &S, int i, int LPR) {
    for (int d = 1; d < LPR; d <<= 1) {
        const float Pp = __shfl_up(P, d, LPR);
        const float Sp = __shfl_up(S, d, LPR);
        if (i >= d) {
            S = fmaf(P, Sp, S);
            P *= Pp;
        }
    }
}
// Same in descending lane order: lane i holds the composition of lanes LPR-1..i (later lanes first).
__device__ __forceinline__ void seg_scan_down(float &P, float &S, int i, int LPR) {
    for (int d = 1; d < LPR; d <<= 1) {
        const float Pn = __shfl_down(P, d, LPR);
        const float Sn = __shfl_down(S, d, LPR);
        if (i + d < LPR) {
            S = fmaf(P, Sn, S);
            P *= Pn;
        }
    }
}

}  // namespace xfm

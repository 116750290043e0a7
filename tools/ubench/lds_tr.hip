// Micro-benchmark (gfx950): cost of the fragment reads of the token-contracting GEMM (csrc/wgrad_gemm.hip) -- ds_read_b64_tr_b16
// on the chunk-XOR token-major image -- against plain ds_read_b64 on the same addresses and ds_read_b128 on a linear image,
// with and without MFMAs between the reads, one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_tr.hip -o tools/ubench/lds_tr.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tok_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// MODE 0: tr reads, xor image;  1: plain b64 reads at the same addresses;  2: tr reads, lgkmcnt(0) after every 8;
// 3: tr reads + 4 MFMAs per 8 reads (the kernel's loop);  4: MFMAs only
template <int MODE> __global__ void __launch_bounds__(256) k(float *out, int iters) {
    extern __shared__ __align__(16) uint8_t sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<uint32_t *>(sm)[i] = 0x3f803f80u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)sm;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    uint32_t ad[4][4][2];                         // [s][operand tile][lo / hi]
    for (int s = 0; s < 4; ++s)
        for (int t = 0; t < 4; ++t) {
            const int ct = (t < 2 ? wm : wn) * 64 + (t & 1) * 32;
            const int c0 = (ct + 16 * (g & 1)) >> 3, r0 = 16 * s + 8 * (g >> 1);
            const uint32_t tb = base + (t < 2 ? 0 : 16384);
            ad[s][t][0] = tb + tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
            ad[s][t][1] = tb + tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
        }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    bf16x4 lo[4], hi[4];
    for (int t = 0; t < 4; ++t) lo[t] = hi[t] = bf16x4{0, 0, 0, 0};
    uint32_t sink = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE != 4) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (MODE == 1) {
                        asm volatile("ds_read_b64 %0, %1" : "=v"(lo[t]) : "v"(ad[s][t][0]) : "memory");
                        asm volatile("ds_read_b64 %0, %1" : "=v"(hi[t]) : "v"(ad[s][t][1]) : "memory");
                    } else {
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[t]) : "v"(ad[s][t][0]) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[t]) : "v"(ad[s][t][1]) : "memory");
                    }
                }
                if (MODE == 2 || MODE == 3)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
            }
            if (MODE == 3 || MODE == 4) {
                bf16x8 af[2], bf[2];
                for (int t = 0; t < 2; ++t) {
                    af[t] = __builtin_shufflevector(lo[t], hi[t], 0, 1, 2, 3, 4, 5, 6, 7);
                    bf[t] = __builtin_shufflevector(lo[2 + t], hi[2 + t], 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a * 2 + b], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
    const long long t1 = clock64();
    for (int t = 0; t < 4; ++t) sink ^= ((uint32_t *)&lo[t])[0] ^ ((uint32_t *)&hi[t])[1];
    float a = 0.f;
    for (int t = 0; t < 4; ++t) a += acc[t][0] + acc[t][7];
    if (lane == 0) out[blockIdx.x * 4 + wave] = (float)(t1 - t0) / (iters * 4) + a * 1e-30f + (float)(sink & 1) * 1e-30f;
}

template <int MODE> void run(const char *name) {
    float *d;
    const int blocks = 256;
    hipMalloc(&d, blocks * 4 * sizeof(float));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, d, 2000);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, d, 2000);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    float h[1024];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-44s %8.1f clock64 ticks per k16-step (8 reads / 4 MFMAs) per wave;  %7.1f ns per k16-step wall\n", name, s / 1024,
           ms * 1e6 / (2000 * 4));
    hipFree(d);
}


// the stage loop of wgrad_tt_glds_kernel without its loads: 4 k16-steps per stage, fragment reads one k16-step ahead
// (lgkmcnt(8)), a barrier per stage, NBUF stage buffers of 32 KB
template <int NBUF, bool BARRIER> __global__ void __launch_bounds__(256, 1) kp(float *out, int iters) {
    extern __shared__ __align__(16) uint8_t sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NBUF * 32768 / 4; i += 256) reinterpret_cast<uint32_t *>(sm)[i] = 0x3f803f80u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)sm;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    auto frag = [&](const uint32_t tb, const int ct, const int s, bf16x4 &lo, bf16x4 &hi) {
        const int c0 = (ct + 16 * (g & 1)) >> 3, r0 = 16 * s + 8 * (g >> 1);
        const uint32_t a0 = tb + tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
        const uint32_t a1 = tb + tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1) : "memory");
    };
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const uint32_t At = base + (it % NBUF) * 32768, Bt = At + 16384;
        if (BARRIER) __builtin_amdgcn_s_barrier();
        bf16x4 lo[2][4], hi[2][4];
        auto frags = [&](const int ring, const int s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                frag(At, wm * 64 + t * 32, s, lo[ring][t], hi[ring][t]);
                frag(Bt, wn * 64 + t * 32, s, lo[ring][2 + t], hi[ring][2 + t]);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = s & 1;
            if (s + 1 < 4) {
                frags(r ^ 1, s + 1);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(lo[r][0]), "+v"(lo[r][1]), "+v"(lo[r][2]), "+v"(lo[r][3]), "+v"(hi[r][0]), "+v"(hi[r][1]), "+v"(hi[r][2]), "+v"(hi[r][3]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[r][0]), "+v"(lo[r][1]), "+v"(lo[r][2]), "+v"(lo[r][3]), "+v"(hi[r][0]), "+v"(hi[r][1]), "+v"(hi[r][2]), "+v"(hi[r][3]));
            }
            bf16x8 af[2], bf[2];
            for (int t = 0; t < 2; ++t) {
                af[t] = __builtin_shufflevector(lo[r][t], hi[r][t], 0, 1, 2, 3, 4, 5, 6, 7);
                bf[t] = __builtin_shufflevector(lo[r][2 + t], hi[r][2 + t], 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a * 2 + b], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float a = 0.f;
    for (int t = 0; t < 4; ++t) a += acc[t][0] + acc[t][7];
    if (lane == 0) out[blockIdx.x * 4 + wave] = (float)(t1 - t0) / iters + a * 1e-30f;
}

template <int NBUF, bool BARRIER> void runp(const char *name) {
    float *d;
    const int blocks = 256;
    hipMalloc(&d, blocks * 4 * sizeof(float));
    hipFuncSetAttribute((const void *)kp<NBUF, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * 32768);
    hipLaunchKernelGGL((kp<NBUF, BARRIER>), dim3(blocks), dim3(256), NBUF * 32768, 0, d, 500);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kp<NBUF, BARRIER>), dim3(blocks), dim3(256), NBUF * 32768, 0, d, 500);
    hipDeviceSynchronize();
    float h[1024];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-60s %8.1f cycles per stage (16 MFMAs = 512)\n", name, s / 1024);
    hipFree(d);
}

int main() {
    runp<1, false>("pipelined stage loop, 1 buffer, no barrier");
    runp<1, true>("pipelined stage loop, 1 buffer, barrier per stage");
    runp<4, true>("pipelined stage loop, 4 buffers (128 KB), barrier per stage");
    run<0>("ds_read_b64_tr_b16, xor image, no waits");
    run<1>("ds_read_b64, same addresses, no waits");
    run<2>("ds_read_b64_tr_b16, lgkmcnt(0) per 8 reads");
    run<3>("tr reads + wait + 4 MFMA");
    run<4>("4 MFMA only");
    return 0;
}

// Micro-benchmark (gfx950): vector-ALU issue rate per SIMD as a function of the waves resident on it.
// Each wave runs a stream of INDEPENDENT instructions of one kind (16 accumulators); reported: cycles per instruction as one
// wave sees it, and SIMD cycles per instruction (= that / waves per SIMD), from the wall time of the launch and the shader clock
// estimated with s_memtime.  Decides whether a kernel at two waves per SIMD that issues one VALU instruction per ~5 cycles and
// wave is at the SIMD's limit (4 cycles per wave64 instruction) or at half of it (2 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, long long *ticks, int iters, float seed) {
    float a[16];
    f2 p[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i + threadIdx.x;
        p[i] = f2{seed + i, seed - i};
        u[i] = threadIdx.x * 77 + i;
    }
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const f2 pm = {m, m}, pc = {c, c};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {                    // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 1) {             // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 2) {             // v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 3) {             // DPP fmac
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 4) {             // integer shift (the bf16 unpack)
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 5) {             // v_cvt_pk_bf16_f32
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(m));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 6) {             // v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 7) {             // dependent fma chain (latency)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (MODE == 8) {             // mix: 3 fma : 1 exp
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(a[(i + 1) & 15]) : "v"(m), "v"(c)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(i + 2) & 15]) : "v"(m), "v"(c)); \
             asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 3) & 15]));
            REP16(X)
#undef X
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char *name, int wps) {
    // blocks of 256 threads = one wave per SIMD each; wps blocks per CU through the LDS request (160 KB / wps each)
    const int blocks = 256 * wps, iters = 2000;
    float *d;
    long long *tk;
    hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
    hipMalloc(&tk, (size_t)blocks * 4 * sizeof(long long));
    size_t lds = (size_t)(160 * 1024) / wps - 1024;
    if (wps > 4) lds = 160 * 1024 / wps - 512;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, tk, 10, 1.0f);     // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, tk, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), tk, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double n = (double)iters * 64;
    const double tick_per = s / h.size() / n;                       // s_memtime ticks per instruction, one wave
    const double ns_per = ms * 1e6 / n;                             // wall ns per instruction per wave
    printf("%-14s waves/SIMD %d : %6.2f ticks/instr/wave  %6.3f ns/instr/wave  -> SIMD: %6.3f ns/instr (%.2f cyc @2.4GHz)\n", name, wps,
           tick_per, ns_per, ns_per / wps, ns_per / wps * 2.4);
    hipFree(d); hipFree(tk);
}

int main() {
    for (int w : {1, 2, 3, 4, 8}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<6>("v_pk_mul_f32", w);
        run<2>("v_exp_f32", w);
        run<3>("v_fmac_dpp", w);
        run<4>("v_lshlrev", w);
        run<5>("v_cvt_pk_bf16", w);
        run<7>("fma chain", w);
        run<8>("3fma+1exp", w);
    }
    return 0;
}

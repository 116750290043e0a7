// Micro-benchmark (gfx950): cost of LDS read-modify-write flavours per wave instruction, at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench/lds_atomics.hip -o gpurun_out/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE> __global__ void k(float *out, int iters, int stride) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *base = sm + wave * 64 * 200;
    for (int i = threadIdx.x; i < (int)(blockDim.x / 64) * 64 * 200; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    float *p = base + lane * stride;
    float v = (float)lane, acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) atomicAdd(p + j, v);                                           // ds_add_f32 (no return)
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(p + j), 3u);        // ds_add_u32
            else if (MODE == 2) p[j] += v;                                                 // read, add, write
            else if (MODE == 3) p[j] = v + j;                                              // plain write
            else if (MODE == 4) acc += atomicAdd(p + j, v);                                // ds_add_rtn_f32
            else if (MODE == 5) acc += p[j];                                               // plain read
        }
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = (float)(t1 - t0) / (iters * 16) + acc * 1e-30f;
}

template <int MODE> void run(const char *name, int waves_per_cu, int stride) {
    float *d;
    const int blocks = 256, threads = 64 * waves_per_cu;
    hipMalloc(&d, blocks * waves_per_cu * sizeof(float));
    const size_t lds = (size_t)waves_per_cu * 64 * 200 * 4;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), lds > 160 * 1024 ? 160 * 1024 : lds, 0, d, 200, stride);
    hipDeviceSynchronize();
    float h[256 * 16];
    hipMemcpy(h, d, blocks * waves_per_cu * sizeof(float), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * waves_per_cu; ++i) s += h[i];
    printf("%-22s waves/CU %2d stride %3d : %7.1f clock64 ticks per wave-instruction (per wave)\n", name, waves_per_cu, stride,
           s / (blocks * waves_per_cu));
    hipFree(d);
}

int main() {
    for (int w : {1, 4, 8}) {
        for (int stride : {197, 1}) {
            if (w * 64 * 200 * 4 > 160 * 1024) continue;
            run<0>("ds_add_f32", w, stride);
            run<1>("ds_add_u32", w, stride);
            run<2>("read+add+write", w, stride);
            run<3>("ds_write_b32", w, stride);
            run<4>("ds_add_rtn_f32", w, stride);
            run<5>("ds_read_b32", w, stride);
        }
    }
    return 0;
}

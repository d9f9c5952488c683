// Developer probe: issue rate of v_mfma_f32_16x16x4_f32 from ONE wave per SIMD (256-thread workgroups, one per CU),
// 16 independent accumulators, nothing else in the loop.  Prints ns per MFMA per wave -> effective matrix-core clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks, int threads, int iters) {
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = 1e6 * ms / ((double)iters * NACC);
    const int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
    printf("blocks %4d threads %4d nacc %2d: %.1f us total, %.2f ns per MFMA per wave (x%d waves/SIMD) -> %.1f TFLOP/s chip-equivalent at 256 CUs\n",
           blocks, threads, NACC, 1e3 * ms, per, waves_per_simd, 2048.0 * 4 * waves_per_simd * 256 / per * 1e-3);
    hipFree(out);
}

int main() {
    run<16>(64, 256, 20000);
    run<16>(256, 256, 20000);
    run<16>(256, 512, 20000);
    run<4>(256, 256, 80000);
    run<4>(256, 512, 80000);
    run<16>(1024, 256, 20000);
    return 0;
}

// Developer probe: ticks of s_memtime per v_mfma_f32_16x16x4_f32 at 1 and 2 waves per SIMD, and wall time
// (=> real shader clock under fp32-MFMA load vs the s_memtime tick rate).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, unsigned long long* ticks, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&ticks, 1024 * 8);
    const int iters = 4096;
    for (int threads : {256, 512}) for (int blocks : {1, 170, 256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, ticks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[1024]; hipMemcpy(h, ticks, blocks * 8, hipMemcpyDeviceToHost);
        double mf = 4.0 * iters * (threads / 256);   // MFMAs per SIMD
        printf("threads %d blocks %3d: %7.1f us wall, %8llu ticks, %.2f ticks/MFMA(per SIMD), %.2f ns/MFMA => clk if 32cyc: %.2f GHz, tick rate %.2f GHz\n",
               threads, blocks, ms * 1e3, h[0], h[0] / mf, ms * 1e6 / mf, 32.0 / (ms * 1e6 / mf), h[0] / (ms * 1e6) );
    }
    return 0;
}

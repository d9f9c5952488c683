// Developer probe: does v_mfma_f32_16x16x4_f32 keep its 32-cycle issue interval when A/B operands
// rotate over distinct VGPRs (as in the real k-loop) instead of two fixed registers?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, const float* in, unsigned long long* ticks, int iters) {
    f32x4 acc[4] = {};
    f32x4 a = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 4);
    f32x4 b[4];
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f32x4*>(in + 1024 + i * 256 + threadIdx.x * 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0][0], acc[t], 0, 0, 0);
                if (MODE == 1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[t][q], acc[t], 0, 0, 0);
            }
        if (MODE == 1) { asm volatile("" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float *out, *in; unsigned long long* ticks;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20); hipMalloc(&ticks, 1024 * 8);
    const int iters = 1024;
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, in, ticks, iters);
        else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, in, ticks, iters);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.2f ticks per MFMA\n", mode, h / (16.0 * iters));
    }
    return 0;
}

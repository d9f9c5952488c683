// Developer probe (round 6): does one LDS read slotted behind EVERY v_mfma_f32_16x16x4_f32 (the weight-gradient kernel's
// k-slice: 8 MFMAs + 6 ds_read_b32, operands used two slices later) cost matrix-core issue slots?  512-thread workgroups
// (two waves per SIMD), one per CU; variants: 0 = MFMAs only, 1 = + a ds_read_b32 behind every MFMA, 2 = + one ds_read_b128
// per 4 MFMAs (the same bytes), 3 = 1 with the MFMA as inline asm on AGPR accumulators (gnf_train.hip's mfma_inplace).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 1e-3f * (i & 63);
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p = lds + (threadIdx.x & 63);
    float a[3][2], b[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        a[s][0] = p[64 * s], a[s][1] = p[64 * s + 16];
#pragma unroll
        for (int n = 0; n < 4; ++n) b[s][n] = p[64 * s + 32 + 8 * n];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const float* q = p + 256 * ((it * 6 + ks) & 15);
            int li = 0;
            f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (V == 3) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[2 * n + m]) : "v"(a[ks % 3][m]), "v"(b[ks % 3][n]));
                    } else {
                        acc[2 * n + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks % 3][m], b[ks % 3][n], acc[2 * n + m], 0, 0, 0);
                    }
                    if ((V == 1 || V == 3) && li < 6) {
                        const float t = q[16 * li + 64 * ((ks + 2) % 3)];
                        if (li < 2) a[(ks + 2) % 3][li] = t; else b[(ks + 2) % 3][li - 2] = t;
                        ++li;
                    }
                    if (V == 2 && (li++ & 3) == 0) {
                        v4 = *reinterpret_cast<const f32x4*>(q + 4 * ((threadIdx.x & 15) + 16 * (li >> 2)));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (V == 2) a[(ks + 2) % 3][0] += v4[0] * 0.f, b[(ks + 2) % 3][0] += v4[1] * 0.f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][i & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
static void run(int iters) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix<V>, dim3(256), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<V>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 48.0 * 2048.0;
    printf("variant %d: %.1f us, %.1f TFLOP/s = %.3f of the 157.3 peak\n", V, 1e3 * ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
    hipFree(out);
}

int main() {
    run<0>(4000), run<1>(4000), run<2>(4000), run<3>(4000);
    return 0;
}

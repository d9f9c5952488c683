// Developer probe (round 6): does a second stream's kernel get CU slots beside a persistent one-workgroup-per-CU kernel?
// A: 256 workgroups x 512 threads, RA registers per wave (forced), LA bytes of dynamic LDS, spins for ~300 us.
// B: 1408 workgroups x 512 threads, ~100 registers, 55 KB of static LDS, ~5 us of work each.  Prints B's time alone, and
// launched right behind A on another stream, for a few (RA, LA).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RA>
__global__ __launch_bounds__(512) void k_a(float* out, long long cycles) {
    extern __shared__ float dyn[];
    float r[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) r[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < RA; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % RA];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RA; ++i) s += r[i];
    dyn[threadIdx.x] = s;
    out[blockIdx.x * 512 + threadIdx.x] = s + dyn[(threadIdx.x + 1) & 511];
}

__global__ __launch_bounds__(512) void k_b(float* out, long long cycles) {
    __shared__ float st[55 * 256];
    float r[80];
#pragma unroll
    for (int i = 0; i < 80; ++i) r[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 80; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % 80];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 80; ++i) s += r[i];
    st[threadIdx.x] = s;
    out[blockIdx.x * 512 + threadIdx.x] = s + st[(threadIdx.x + 1) & 511];
}

template <int RA>
static void run(size_t la) {
    float *oa, *ob;
    hipMalloc(&oa, 4 * 256 * 512), hipMalloc(&ob, 4 * 1408 * 512);
    hipStream_t s1, s2;
    hipStreamCreate(&s1), hipStreamCreate(&s2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_a<RA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const long long ca = 720000, cb = 12000;   // ~300 us, ~5 us at 2.4 GHz
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            if (mode) hipLaunchKernelGGL(k_a<RA>, dim3(256), dim3(512), la, s1, oa, ca);
            hipEventRecord(e0, s2);
            hipLaunchKernelGGL(k_b, dim3(1408), dim3(512), 0, s2, ob, cb);
            hipEventRecord(e1, s2);
            hipDeviceSynchronize();
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("A: %3d float regs forced, %6zu B dyn LDS | B %s: %.1f us\n", RA, la, mode ? "behind A on another stream" : "alone", 1e3 * ms);
    }
}

int main() {
    run<40>(67584), run<56>(67584), run<64>(67584), run<72>(67584), run<80>(67584), run<88>(67584), run<96>(67584), run<100>(67584), run<140>(67584);
    return 0;
}

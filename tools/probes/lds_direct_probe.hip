// Developer probe: semantics of LDS-direct buffer loads on gfx950 (buffer_load_dword / dwordx4 ... lds):
// lane l's element lands at lds_base + l * size; out-of-range lanes (descriptor range check) write zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* g, float* out, int n_valid) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n_valid * 4, 0x00020000);
    const int l = threadIdx.x;
    // dwordx4: lane l fetches floats [4 * perm(l), +4) with perm(l) = (l * 7) % 64; LDS chunk at float offset 0
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, ((l * 7) % 64) * 16, 0, 0, 0);
    // dword: lane l fetches float 300 + (63 - l); LDS chunk at float offset 512
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + 512), 4, (300 + 63 - l) * 4, 0, 0, 0);
    // dword with soffset + immediate offset: float 100 + l + 8 (imm 32 bytes), LDS at float offset 1024
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(lds + 1024), 4, l * 4, 400, 32, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    const int N = 1024, valid = 340;  // floats [340, ...) are out of range
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *g, *o;
    hipMalloc(&g, N * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(g, h.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, valid);
    std::vector<float> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q) { float want = (float)(4 * ((l * 7) % 64) + q); if (r[4 * l + q] != want) ++bad; }
    for (int l = 0; l < 64; ++l) { int idx = 300 + 63 - l; float want = idx < valid ? (float)idx : 0.f; if (r[512 + l] != want) ++bad; }
    for (int l = 0; l < 64; ++l) { int idx = 100 + l + 8; float want = (float)idx; if (r[1024 + l] != want) { ++bad; } }
    printf("lds-direct probe: %d mismatches; x4 lane1 -> %.0f %.0f, dword lane0 -> %.0f (want 0: out of range), lane 30 -> %.0f, soffset lane 3 -> %.0f, untouched %.0f\n",
           bad, r[4], r[5], r[512], r[512 + 30], r[1024 + 3], r[300]);
    return bad != 0;
}

// Process-wide developer options of libgnf_hip.so and the per-device one-time setup helper.
//
// The library never reads the environment.  Everything that used to be an A/B environment switch is a named
// integer here, 0 (= "let the library decide") unless gnf_set_option() changed it; the launch paths read the
// table with relaxed atomic loads, so changing an option between two calls is well defined.  The table and the
// per-device "dynamic LDS attribute already raised" bitmaps below are the only state the library keeps.
#pragma once
#include <atomic>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace gnf {

enum OptionId {
    OPT_FORCE_SHAPE = 0,     // fused forward workgroup shape <MT><NETS>, e.g. 21; 0 = by batch size
    OPT_FUSED_VARIANT,       // fused forward kernel: A/B bits (0 = shipped behaviour): 1 no thin-chunk form, 4 attention front-end as
                             // its own launch, 8 batch-norm bijectors as their own pass per half-step, 64 no closing round of small workgroups in the large-batch kernel
    OPT_FLOW_NO_OOP,         // out-of-place flows: always copy first, then walk in place (A/B of the fused first step)
    OPT_ATTN_EDGE_TILED,     // attention forward: always the edge-tiled kernel
    OPT_ATTN_ROWS,           // attention forward: always the rows kernel
    OPT_GEMM_NO_BUF,         // generic GEMM: bounds-checked fetch instead of buffer descriptors
    OPT_GEMM_NO_SPLITK,      // generic GEMM: never split thin launches over the reduction
    OPT_DW_GROUPED,          // weight gradients: always the grouped kernel
    OPT_DW_WIDE_UNITS,       // weight gradients: wide kernel with this many workgroups
    OPT_DW_WIDE_LDS,         // ... and this LDS request per workgroup (bytes)
    OPT_DW_NO_STREAMK,       // ... whole chunks instead of stream-K runs
    OPT_DW_NO_BUF,           // ... bounds-checked fetch
    OPT_DW_DEBUG,            // bit 1: print the dW launch plan to stderr (first two launches); bits 2 / 4 / 8 / 16: timing ablations
                             // of the merged backward + dW launch (gnf_train.hip, launch_half_bwd_dw); A/B of round 3: 32 dagg
                             // through a GEMM launch, 64 scalar dL/dx_cond kernel, 128 run-time head geometry in the attention
                             // backward edge kernels, 256 the batch-norm bijector's backward pass as its own launch
    OPT_BWD_GENERIC,         // backward pass through the generic GEMM path even where the fused kernel fits
    OPT_DW_UNMERGED,         // small batches: dW GEMMs on the auxiliary stream (round-1 scheme) instead of inside the backward launch
    OPT_NO_MLP_STASH,        // ignore GnfFlow.mlp_stash (the backward walk recomputes the MLP rows)
    OPT_ATTN_BWD_ROWS,       // attention rows kernels: 64 / 32 (backward also 16) rows per workgroup (0 = by batch size / mean degree)
    OPT_ATTN_BWD_SPLIT,      // attention backward on sparse batches: receiver and sender pass as two launches (A/B)
    OPT_COUNT
};

extern std::atomic<int64_t> g_options[OPT_COUNT];
inline int64_t opt(OptionId id) { return g_options[id].load(std::memory_order_relaxed); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: BODY runs once per device
// (a lost race runs it twice, which is harmless), keyed by the call site.
#define GNF_ONCE_PER_DEVICE(...)                                                     \
    do {                                                                               \
        static std::atomic<uint64_t> once_mask_{0};                                    \
        int once_dev_ = 0;                                                             \
        (void)hipGetDevice(&once_dev_);                                                \
        const uint64_t once_bit_ = 1ull << (once_dev_ & 63);                           \
        if (!(once_mask_.load(std::memory_order_acquire) & once_bit_)) {               \
            __VA_ARGS__;                                                               \
            once_mask_.fetch_or(once_bit_, std::memory_order_release);                 \
        }                                                                              \
    } while (0)

}  // namespace gnf

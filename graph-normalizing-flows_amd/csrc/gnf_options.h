// Process-wide developer options of libgnf_hip.so and the per-device one-time setup helper.
//
// The library never reads the environment.  The launch shapes it picks by batch size can be forced - that is how the
// parity tests reach every kernel instance on small batches - through a named integer here, 0 (= "let the library
// decide") unless gnf_set_option() changed it; the launch paths read the
// table with relaxed atomic loads, so changing an option between two calls is well defined.  The table and the
// per-device "dynamic LDS attribute already raised" bitmaps below are the only state the library keeps.
#pragma once
#include <atomic>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace gnf {

enum OptionId {
    OPT_FORCE_SHAPE = 0,     // fused forward workgroup shape <MT><NETS>, e.g. 21; 40 / 30 / 20 / 10: the large-batch kernel with that many
                             // row tiles per workgroup at most; 49: that kernel (cap 4) with the split tiles' hand-over flag withheld -
                             // fault injection, every split tile takes the lost-partner branch (NaN rows, NaN partial sums); 0 = by batch size
    OPT_ATTN_KERNEL,         // attention forward: 1 always the rows kernel, 2 always the edge-tiled kernel, 3 always the matrix-core
                             // attention core (each keeps the front-end out of the fused kernel's prologue); 0 = by batch / geometry
    OPT_ATTN_BWD_ROWS,       // attention rows kernels: 64 / 32 (backward also 16 and 3264) rows per workgroup; 0 = by batch size / mean degree
    OPT_BWD_GENERIC,         // backward pass through the generic GEMM path even where the fused kernel fits
    OPT_DW_GROUPED,          // weight gradients: always the grouped kernel (on the auxiliary stream, not inside the backward launch)
    OPT_DW_WIDE_UNITS,       // weight gradients: wide kernel with this many workgroups at most
    OPT_DW_THIN_ON_DW,       // merged backward + dW launch: 1 = the thin layers' units stay with the dW workgroups (not the tile workgroups)
    OPT_COUNT
};

static constexpr int64_t kForceBigLostPartner = 49;
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

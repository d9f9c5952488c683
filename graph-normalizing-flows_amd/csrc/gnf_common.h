// Internal declarations shared by the HIP translation units of libgnf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gnf.h"
#include "gnf_options.h"

namespace gnf {

void set_error(const char* fmt, ...);

#define GNF_HIP_TRY(expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            gnf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                           __LINE__);                                                         \
            return GNF_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define GNF_LAUNCH_CHECK(name)                                                                \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            gnf::set_error("launch of %s failed: %s", name, hipGetErrorString(_e));           \
            return GNF_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

// Everything one coupling half-step needs, resolved to raw pointers (host struct, passed by value
// to the launchers).
struct HalfStep {
    const int32_t* rowptr;
    const int32_t* col;
    int64_t n_nodes;
    const float* x_cond;  // first column of the conditioning half
    float* x_upd;         // first column of the half being updated
    int64_t ld;
    int32_t H;
    int32_t direction;
    GnfGnnSpec gnn;
    const GnfMlp* s_net;  // host
    const GnfMlp* t_net;  // host
    double* partials;     // device: one fp64 partial sum(s) per workgroup of the epilogue kernel
    int32_t* n_partials;  // host out: how many partials this launch writes
    float* attn_region;   // NULL: the attention front-end works in the scratch; else q | k | v and h0 of both nets go
                          // here (a slot of GnfFlow.attn_stash) and stay for the backward pass
    int64_t n_edges;      // 0: unknown (only used to pick between kernel generations by mean degree)
    // out-of-place first half-step of a flow (the caller's functional x -> z without a separate copy pass): when
    // x_src != NULL the conditioning half is read from x_src + cond_off, the old value of the updated half from
    // x_src + upd_off (same leading dimension ld), and the launch also copies its rows of the conditioning half into
    // cond_copy (= the destination buffer's conditioning half).  Only the fused both-nets kernel implements it
    // (fused_supports_oop); everything else copies first.
    const float* x_upd_src = nullptr;
    float* cond_copy = nullptr;
    // attention nets: fragment-order copies of this half-step's two attention blocks (launch_attn_pack), or NULL
    const float* attn_packed[2] = {nullptr, nullptr};
    // ... and the sender window (lo, hi) of every 16-row tile of the batch (launch_attn_tiles, once per flow call), or NULL
    const int32_t* attn_tiles = nullptr;
    // large-batch kernel, split row tiles (gnf_fused_big.hip): > 0 = the caller zeroed big_split_flags(scratch, ...) at the
    // start of its call and hands every half-step launch a value of its own (1, 2, ...); 0 = no split tiles
    int32_t split_epoch = 0;
    // training forward: this half-step's slot of GnfFlow.mlp_stash (mlp_stash_layout), or NULL
    float* mlp_stash = nullptr;
    // forward, the flow's last two half-steps (their outputs are z): room for one fp64 partial of sum(x_upd_new^2) per
    // workgroup; *n_sq = how many the launch wrote (0: this path does not - the caller runs k_gauss over z instead)
    double* sq_partials = nullptr;
    int32_t* n_sq = nullptr;
    // forward with a batch-norm bijector in front of the NEXT half-step: room for one [H][2] fp64 row of column sums /
    // sums of squares of the updated half per workgroup (that bijector's batch moments); *n_bn = rows written (0: none)
    double* bn_part = nullptr;
    int32_t* n_bn = nullptr;
    // forward with the batch-norm bijectors applied where the rows are read (the fused kernel's attention instance only:
    // fused_bn_on_load_ok): bnc = the bijector in front of THIS half-step - bnc_nparts partial rows (sum x, sum x^2) of
    // the conditioning half in bnc_part, its batch moments / log-det term / (scale, shift) pairs go to bnc->batch_mean /
    // batch_variance, bnc_logdet, bnc_const ([2][H]); bnu_const = the previous half-step's pairs (its conditioning half
    // is the half this one rewrites), or NULL
    const GnfBatchNorm* bnc = nullptr;
    const double* bnc_part = nullptr;
    int32_t bnc_nparts = 0;
    double* bnc_logdet = nullptr;
    float* bnc_const = nullptr;
    const float* bnu_const = nullptr;
    // inverse pass, the same idea: the bijector behind the PREVIOUS half-step of the walk, applied where this one reads the
    // half it rewrites (moving statistics: no moments to gather), or NULL
    const GnfBatchNorm* bnu_inv = nullptr;
};
// may a forward flow with batch-norm bijectors hand them to the half-step kernels this way? (the first half-step decides)
bool fused_bn_on_load_ok(const HalfStep& hs);
// the pass bn_on_load leaves for the end of the flow: x[:, :H] = x * scale + shift with the last bijector's pairs
int launch_bn_affine(float* x, int64_t ld, int64_t n, int32_t H, const float* scale_shift, hipStream_t st);
// partial rows (sum x, sum x^2 per feature, fp64) of x[:, :H] - the first bijector's moments; returns the row count
int launch_bn_stats(const float* x, int64_t ld, int64_t n, int32_t H, double* part, hipStream_t st, int* rows_out);

// MLP-row stash (GnfFlow.mlp_stash, ABI v8): the rows of a half-step the backward walk would otherwise recompute
struct MlpStashLayout {
    size_t h0, act, act_each, st, st_each, mask, slot;  // float offsets inside a slot / floats per slot
    int ld_act;                                   // row pitch of the hidden activations (widest hidden layer)
    int mld, mask_words;                          // act' ballot words: 16-column tiles per mask row, 64-bit words per 16-node tile
};
MlpStashLayout mlp_stash_layout(const GnfMlp* net, int64_t n, int32_t H);
bool fused_stash_shape(const GnfMlp* s, const GnfMlp* t, int64_t n);   // forward side (gnf_fused.hip)
// both sides: message-passing nets on the fused forward kernel's (1,2) shape AND the merged backward launch
bool mlp_stash_supported(const GnfFlow* flow, int64_t n, int32_t H);   // gnf_train.hip
// ... its second mode: nets on the layered forward / generic backward path (too wide for the fused kernels): the forward's
// layer outputs and s, t go straight into the slot, the backward skips its recompute of both MLPs
bool layered_stash_mode(const GnfFlow* flow, int64_t n, int32_t H);
bool fused_supports_oop(const HalfStep& hs);
// floats of one half-step's slot in GnfFlow.attn_stash ( = attn_scratch_floats: [2][n][P] q|k|v, [2][n][in0] h0,
// [2][n][heads*v] attended values, [2][n][3*heads] softmax statistics)
size_t attn_stash_slot_floats(const GnfFlow* flow, int64_t n_nodes);

// ---- layout of the caller-provided workspace --------------------------------------------------
// [ fp64 partial sums | float scratch of the layered path ]
static constexpr int kFinalizeBlock = 256;
static constexpr int kMaxGaussBlocks = 1024;
static constexpr int kBnBlocksMax = 256;  // workgroups of the batch-norm moment pass
static constexpr int kBnPartRowsMax = 1024;  // [H][2] fp64 partial rows the moment buffers hold (a producer kernel may leave one per workgroup)

inline int64_t coupling_blocks_max(int64_t n_nodes) { return (n_nodes + 15) / 16 + 1; }
// per half-step: the coupling partials + one slot for the batch-norm log-det term
inline int64_t partials_per_halfstep(int64_t n_nodes) { return coupling_blocks_max(n_nodes) + 1; }

// layered path: ping-pong activation buffers per net x two nets side by side (grouped GEMM launches)
static constexpr int kLayeredActBufs = 4;

struct WorkspacePlan {
    int64_t partial_stride;  // doubles per half-step
    int64_t n_halfsteps;
    size_t partial_bytes;    // (n_halfsteps * stride + kMaxGaussBlocks + batch-norm moment partials) * 8, 256-aligned
    size_t bn_offset;        // doubles: start of the batch-norm moment partials inside the fp64 region
    size_t bn_offset2;       // ... a second set of partial rows (batch norm on load: a kernel reads one set while its epilogue fills the other)
    size_t bn_const_offset;  // ... and the (scale, shift) pairs of every bijector of the call: n_halfsteps x [2][H] floats
    size_t attn_pack_offset; // floats: fragment-order attention weights of every net of the call, behind the attention region
    size_t attn_pack_per_net;
    size_t attn_tile_offset; // floats: [tiles][2] ints, the attention front-end's per-tile sender windows (behind the packed weights)
    size_t scratch_floats;   // layered path activations (+ attention front-end region at its end)
    size_t base_floats;      // offset of the attention region inside the scratch
    size_t total_bytes;
};
WorkspacePlan plan_workspace(int64_t n_nodes, int32_t H, const GnfMlp* net, int32_t combine,
                             int64_t n_halfsteps);

// ---- launchers (each returns GNF_OK / GNF_E*) --------------------------------------------------
bool fused_supported(const HalfStep& hs);
bool fused_fits_lds(const GnfMlp* m);      // forward kernel, smallest shape
bool fused_bwd_fits_lds(const GnfMlp* m);  // backward kernel
int launch_half_fused(const HalfStep& hs, float* scratch, hipStream_t st);
// large-batch form of the fused kernel (gnf_fused_big.hip): workgroups of up to `cap` (<= 4) row tiles of 16 nodes, dealt
// out evenly over rounds x 2 x CUs workgroups (big_plan); *n_wg_out = workgroups launched (= fp64 partials written)
struct FusedArgs;
bool big_supported(const GnfMlp* s, int32_t H);
int big_cu_count();  // multiProcessorCount of the current device (cached per device)
// split row tiles of the large-batch kernel: [kBigSplitMax flags | kBigSplitMax x 16 x 128 floats of s rows] behind the layer-0
// rows at the head of the half-step scratch (message-passing nets; room = the layered path's activation buffers)
static constexpr int kBigSplitMax = 128;
inline size_t big_split_offset(int64_t n_nodes, int in0) { return ((size_t)n_nodes * (size_t)in0 + 63) / 64 * 64; }
inline size_t big_split_floats() { return (size_t)kBigSplitMax + (size_t)kBigSplitMax * 16 * 128; }
int big_plan(int64_t n_nodes, int cus, int cap, int32_t* seg_n, int32_t* seg_sz, int32_t* seg_kind = nullptr, int32_t* xg0 = nullptr);  // -> workgroups; runs of (count, row tiles)
int launch_half_big(FusedArgs& a, int64_t n_nodes, int cap, hipStream_t st, int* n_wg_out);
// s, t as the last layer's partial products (launch_linear_big_fused): n_slab dense [N, H] slabs `stride` floats apart
// behind the s / t pointers, + the layer's bias; s_out / t_out (nullable): where the summed rows are also written (the
// training forward's stash).  n_slab == 0: s, t are the finished rows.
struct SlabSrc {
    int32_t n_slab;
    int64_t stride;
    const float *bias_s, *bias_t;
    float *s_out, *t_out;
};
// coupling epilogue from global s / t [N, H] buffers (writes hs.partials, *hs.n_partials)
int launch_coupling(const float* s, const float* t, const HalfStep& hs, const float* xres, hipStream_t st, const SlabSrc* slabs = nullptr);
int launch_half_layered(const HalfStep& hs, float* scratch, hipStream_t st);
int launch_attn_pair(const HalfStep& hs, float* scratch, float** h0_pair, hipStream_t st);
int launch_gnn_layered(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, const float* x,
                       int64_t ldx, int32_t H, const GnfGnnSpec& g, const GnfMlp* mlp, float* out,
                       int64_t ldo, float* scratch, hipStream_t st);
int launch_aggregate(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, const float* x,
                     int64_t ldx, int32_t H, int32_t mean, int32_t mode, float eps, float* out,
                     int64_t ldo, hipStream_t st);
int launch_gauss_partials(const float* z, int64_t n_nodes, int32_t D, int64_t ld, double* partials,
                          int32_t* n_partials, hipStream_t st);
// out[0] (+)= sum of a[0..na) ; out[1] = sum of b[0..nb)   (fixed order, fp64, single workgroup)
int launch_finalize(const double* a, int64_t na, const double* b, int64_t nb, double* out,
                    int accumulate_a, int write_b, hipStream_t st);
int launch_copy_rows(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t n, int32_t W, hipStream_t st);
int launch_pack_mlp(const GnfMlp* mlp, float* packed, hipStream_t st);
int64_t packed_floats(const GnfMlp* mlp);

// fused backward half-step (gnf_fused_bwd.hip)
bool fused_bwd_supported(const GnfMlp* s, const GnfMlp* t);
void fused_bwd_launch_shape(const GnfMlp* s, int64_t n, int64_t* tiles, size_t* lds);
int launch_half_bwd_fused(const int32_t* rowptr, const int32_t* col, int64_t n, const GnfGnnSpec& gnn,
                          const GnfMlp* s, const GnfMlp* t, const float* x_cond, float* y_upd, int64_t ld,
                          float* g_upd, int64_t ldg, int32_t H, float* h0_out, const float* const* h0_in, float* const* hin,
                          int64_t ldh, float* const* dP, int64_t lddp, float* const* gst, float* const* dh0,
                          hipStream_t st);
// a batch-norm bijector sits in front of this half-step: the last kernel of the attention backward writes the final
// dL/dy rows of the conditioning half, so it also leaves the bijector's backward moments (sum G, sum G x^ per column)
// as one [H][2] fp64 row per workgroup in `part`; *n_parts = rows written
struct AttnBnFold {
    const float* y;      // the normalised conditioning half [n, H], leading dimension ld
    int64_t ld;
    const float* gamma;
    const float* beta;
    double* part;
    int32_t* n_parts;
};
// attention front-end, backwards (gnf_attn_bwd.hip)
int launch_attn_backward(const GnfAttn* const* at, int64_t n, int32_t H, int32_t in0, const int32_t* rowptr,
                         const int32_t* col, const int32_t* rowptr_t, const int32_t* col_t, const float* const* qkv,
                         const float* const* dh0, const float* const* gst, float* const* dqkv, float* const* agg,
                         float* const* dagg, float* const* stats, float* g_cond, int64_t ldg, hipStream_t st,
                         int64_t n_edges = 0,   // 0 = unknown (picks between tile sizes by mean degree)
                         const AttnBnFold* bn = nullptr,
                         // xc_dst != NULL: the last kernel also copies the conditioning half [n, H] (leading dimension xc_ld)
                         // into xc_dst [n][H] for the dW GEMMs that run after it has changed
                         const float* xc_src = nullptr, int64_t xc_ld = 0, float* xc_dst = nullptr,
                         // wct != NULL: [Wq | Wk | Wv]^T of the two blocks as packed MFMA fragments (attn_wct_floats each,
                         // k_pack_wot): dL/dx_cond += dqkv Wcat^T runs on the matrix cores (k_attn_bwd_dx_mfma)
                         const float* const* wct = nullptr);
// both passes of the attention backward on the matrix cores for heads wider than the per-(row, head) thread kernels hold
// (gnf_attn_core_bwd.hip); needs the forward's agg / stats; 1 = not its geometry
int launch_attn_core_backward(const GnfAttn* a0, int64_t n, const int32_t* rowptr, const int32_t* col, const int32_t* rowptr_t,
                              const int32_t* col_t, const float* const* qkv, const float* const* dagg, const float* const* agg,
                              float* const* stats, float* const* dqkv, hipStream_t st);
// [Wq | Wk | Wv]^T ([P, H], P = 2 heads kq + v) as fragments: Bp[kg][nt][lane][q] = Wcat[16 nt + (lane & 15)][16 kg + 4 (lane >> 4) + q]
size_t attn_wct_floats(const GnfAttn* at, int32_t H);


// y = act(x W + b) through the generic GEMM (gnf_train.hip; thin launches split over the reduction); GNF_OK or a GNF_E* code
// (b[q] == NULL: no bias)
int launch_linear_splitk(const float* const* x, int64_t ldx, const float* const* W, const float* const* b, float* const* y,
                         int64_t ldy, int nj, int64_t n, int32_t I, int32_t O, int act, float alpha, int apply_act,
                         float* const* sk, size_t sk_floats, hipStream_t st);

// which layers the wide-layer kernel takes: y = x W reads Wp of a layer with a long reduction and a wide output; the
// backward dX = dY W^T reads WpT where the roles are swapped.  gnf_pack_flow keeps exactly these copies of a net that is
// too wide for the fused kernels in step with its weights.
// (short reductions - the 100 / 164 -> 2048 first layer - measured level with the generic tile through k_linear_big as well,
// 51 vs 49.5 us in round 5: they stay on the generic tile, which keeps this kernel's profile one shape)
inline bool linear_big_fwd_layer(int I, int O) { return I >= 512 && O >= 256; }
inline bool linear_big_bwd_layer(int I, int O) { return O >= 512 && I >= 256; }
// ... and a short reduction into a wide layer (the first layer of such a net) runs k_linear_short off Wp (whole reduction in
// registers: at most 13 k-groups; activation rows loaded as 16-byte operand fragments: widths in whole float4s)
inline bool linear_short_fwd_layer(int I, int O) { return I <= 208 && (I & 3) == 0 && O >= 256; }
// wide y = act(x W_j + b_j) of a pair of nets from their packed weights (gnf_linear_big.hip); 1 = not its case
int launch_linear_big(const GnfMlp* const* nets, int nj, int j, const float* const* x, int64_t ldx, float* const* y, int64_t ldy,
                      int64_t n, int act, float alpha, int apply_act, hipStream_t st);
// layers j and j + 1 = the last in one launch (the thin last layer out of the wide one's accumulators): slab[q] receives
// *n_slabs = linear_big_fused_slabs(O_j) dense partial [n, O_{j+1}] products whose sum + b_{j+1} is the net's output;
// y[q] == NULL: layer j's own output is not kept.  1 = not its case
static constexpr int kLinearBigFusedMaxOut = 128;
inline bool linear_big_fused_last(const GnfMlp* m, int j) {
    return j >= 1 && j == m->num_layers - 1 && m->dims[j + 1] <= kLinearBigFusedMaxOut && linear_big_fwd_layer(m->dims[j - 1], m->dims[j]);
}
int launch_linear_short(const GnfMlp* const* nets, int nj, int j, const float* const* x, int64_t ldx, float* const* y, int64_t ldy,
                        int64_t n, int act, float alpha, int apply_act, hipStream_t st);
int launch_linear_big_fused(const GnfMlp* const* nets, int nj, int j, const float* const* x, int64_t ldx, float* const* y, int64_t ldy,
                            float* const* slab, int32_t* n_slabs, int64_t n, int act, float alpha, hipStream_t st);
int linear_big_fused_slabs(int O);
// dX[q] = (dY[q] W_j^T) * act'(h[q]) of a pair of nets from their transposed packed fragments (h == NULL: no mask); 1 = not its case
int launch_linear_big_dx(const GnfMlp* const* nets, int nj, int j, const float* const* dy, int64_t lddy, float* const* dx, int64_t lddx,
                         const float* const* h, int64_t ldh, int64_t n, int act, float alpha, hipStream_t st);

// batch-norm bijector (gnf_bn.hip)
int validate_bn(const GnfBatchNorm* bn, int direction, const char* what, int q);
// cross-rank moments (GnfFlow.bn_allreduce): fold the per-workgroup partials into flow->bn_sync_buf (and, when
// local_copy != NULL, into a second copy that stays local), then call the hook.  part: [nparts][H][2] doubles.
int bn_sync_exchange(const GnfFlow* flow, const double* part, int nparts, int64_t n, int32_t H, double* local_copy,
                     hipStream_t st);
int launch_bn_normalize(const GnfFlow* flow, const GnfBatchNorm* bn, float* x, int64_t ld, int64_t n, int32_t H,
                        double* part, double* logdet_slot, hipStream_t st, int pre_parts = 0);
int launch_bn_denormalize(const GnfBatchNorm* bn, float* z, int64_t ld, int64_t n, int32_t H, hipStream_t st);

// batch-norm bijector, backwards (gnf_bn_bwd.hip).  pre_parts > 0: `part` already holds that many [H][2] partial rows
int launch_bn_backward(const GnfFlow* flow, const GnfBatchNorm* bn, const GnfBatchNorm* gbn, float* y, int64_t ld,
                       float* gy, int64_t ldg, int64_t n, int32_t H, double* part, hipStream_t st, int pre_parts = 0);

int validate_mlp(const GnfMlp* m, const char* what);
int validate_flow_call(const GnfCsr* csr, const GnfFlow* flow, int64_t ld, int32_t D, const char* what);
// attention front-end (gnf_attn.hip).  THE limit of the block's geometry (include/gnf.h, GnfAttn): heads <= 64,
// heads * kq <= 256, heads * v <= 256, and pad16(2 heads kq + v) + pad16(H) + H <= 1272 (the backward pass's dL/dx_cond
// product keeps sixteen rows of both nets' projections, products and operands in one CU's LDS, k_attn_bwd_dx_mfma) -
// validate_attn enforces all of it for the forward / inverse AND the backward entry points
static constexpr int kAttnMaxHeads = 64;
static constexpr int kAttnMaxWidth = 256;
static constexpr int kAttnMaxRowFloats = 1272;
inline bool attn_geometry_ok(int heads, int kq, int v, int H) {
    const int64_t nq = (int64_t)heads * kq, nv = (int64_t)heads * v, P = 2 * nq + v;
    return heads >= 1 && heads <= kAttnMaxHeads && kq >= 1 && v >= 1 && nq <= kAttnMaxWidth && nv <= kAttnMaxWidth &&
           ((P + 15) & ~(int64_t)15) + ((H + 15) & ~15) + H <= kAttnMaxRowFloats;
}
int validate_attn(const GnfAttn* at, const GnfMlp* mlp, int32_t H, const char* what);
size_t attn_scratch_floats(const GnfAttn* at, int64_t n_nodes, int32_t in0);
// need_qkv: the caller reads the per-node q | k | v block of `scratch` afterwards (backward pass, attention stash)
int launch_attn_front(const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx,
                      int32_t H, const GnfAttn* const* at, int nets, int32_t in0, float* scratch,
                      float* const* h0_out, hipStream_t st, int64_t n_edges = 0, bool need_qkv = true,
                      const float* const* packed = nullptr, float* const* agg_out = nullptr, float* const* mz_out = nullptr);
// Scratch contract: q | k | v of net q at scratch + q n P.  With agg_out == NULL the kernels that need the attended values
// in memory (gnf_attn_core.hip) put them at scratch + 2 n (P + in0) - attn_scratch_floats' layout, which every caller that
// passes NULL provides (plan_workspace); a caller that passes only the q | k | v block (the backward recompute) passes agg_out.
// agg_out / mz_out (training): per net [N, heads*v] attended values and [N, 3*heads] softmax statistics (running max at [h],
// denominator at [heads + h]; the third block is scratch of the backward pass), kept for launch_attn_backward
// matrix-core attention core for wide heads (gnf_attn_core.hip): qkv[q] -> agg[q] ([N, heads v], normalised), mz[q]
// (nullable), and h0[q][:, 0:H) = x for concat blocks; the output projection is the caller's next launch
int launch_attn_core(const GnfAttn* a0, int nets, const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx,
                     int32_t H, int32_t in0, const float* const* qkv, float* const* agg, float* const* mz, float* const* h0,
                     hipStream_t st);
// q | k | v = x [Wq | Wk | Wv] of 1 or 2 nets on the matrix cores, any widths (gnf_attn_core.hip); qkv[q]: [N, P]
// h0_concat != NULL: h0_concat[q][r, 0:H) = x[r, :] rides along (the concat half of a block's layer-0 rows, row stride in0)
int launch_attn_proj_mfma(const GnfAttn* const* at, int nets, int64_t n, const float* x, int64_t ldx, int32_t H, float* const* qkv,
                          hipStream_t st, float* const* h0_concat = nullptr, int32_t in0 = 0);
// one-launch front-end for sparse batches (gnf_attn_front.hip), weights pre-packed into fragment order once per flow call
bool attn_front_fused_ok(const GnfAttn* at, int32_t H);
size_t attn_pack_floats(const GnfAttn* at, int32_t H);
int launch_attn_pack(const GnfAttn* const* at, int count, int32_t H, float* out, hipStream_t st);
// out[tiles][2]: sender window (lo, hi) of every 16-row tile (an empty tile: 0x7fffffff, -1)
int launch_attn_tiles(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t* out, hipStream_t st);
int launch_attn_front_fused(const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx, int32_t H,
                            const GnfAttn* const* at, int nets, int32_t in0, const float* const* packed,
                            float* const* qkv_out, float* const* h0_out, hipStream_t st, float* const* agg_out = nullptr,
                            float* const* mz_out = nullptr);
// snt.LayerNorm over the feature axis of a block's output (gnn.py:550-552), one or two [N, W] blocks per launch:
//   u = in (+ xres);  y = (u - mean_f u) / sqrt(var_f u + GNF_LN_EPS) * gamma + beta   (biased variance)
// y may alias in.  u (may be NULL, may alias in; leading dimension ldin) keeps the un-normalised rows for the backward pass.
struct LnJob {
    const float* in;
    float* y;
    float* u;
    const float* gamma;
    const float* beta;
};
struct LnArgs {
    LnJob job[2];
    int64_t ldin, ldy;
    const float* xres;  // residual rows added first (gnn.py:547-548), or NULL
    int64_t ldx;
    int64_t n;
    int32_t W;
};
int launch_layer_norm(const LnArgs& a, int nets, hipStream_t st);
int launch_half_layer_norm(const HalfStep& hs, float* sbuf, float* tbuf, const float* xres, hipStream_t st);
// dst[r, 0:W) += src[r, 0:W)
int launch_add_rows(float* dst, int64_t ldd, const float* src, int64_t lds_, int64_t n, int32_t W, hipStream_t st);

}  // namespace gnf

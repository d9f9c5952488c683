/*
 * gnf.h - C ABI of libgnf_hip.so: the MI355X (gfx950) implementation of the GRevNet affine-coupling
 * forward / inverse + log-det hot path of jliu/graph-normalizing-flows.
 *
 * The reference has no FFI: its "operator API" for this path is the Sonnet-module call surface of
 * /root/reference/gnn.py, executed by TensorFlow kernels.  Each entry point below names the
 * reference interface (file:line) whose arithmetic it replaces.  A maintainer binds them with
 * ctypes; graph-normalizing-flows_amd/_abi.py is that binding.
 *
 * Conventions
 *   - plain C types only; every pointer that is not marked "host" is a DEVICE pointer into memory
 *     owned by the caller (torch tensors' data_ptr()).  The library allocates nothing persistent.  Its
 *     only state: a thread-local error string, the table of developer options behind gnf_set_option
 *     (all 0 = automatic by default; the library never reads the environment) and, per device, the
 *     one-time "dynamic LDS limit raised" marks of its kernels.  Several host threads, several devices
 *     in one process (hipSetDevice before the call, as for any HIP library) and several streams may
 *     call concurrently.
 *   - every entry point is asynchronous on the caller's stream (hipStream_t passed as void*), does
 *     no host synchronisation, and is legal inside hipGraph stream capture once each kernel it
 *     launches has run once outside a capture on that device (the first launch raises the kernel's
 *     dynamic-LDS limit with hipFuncSetAttribute): tests/test_graph_capture_gpu.py captures
 *     gnf_grevnet_f32 (both directions) and gnf_grevnet_backward_f32 and replays them bitwise.
 *   - return 0 on success, a negative GNF_E* code on failure; gnf_last_error() gives the message.
 *     Nothing throws across the boundary.
 *   - node features live in ONE row-major [N, D] fp32 buffer with leading dimension ld >= D; the two
 *     coupling halves are the column ranges [0, D/2) and [D/2, D), so tf.split / tf.concat
 *     (gnn.py:306,340,344,373) are zero-copy views.
 *   - the graph is given as CSR sorted by RECEIVER: for node r, col[rowptr[r] .. rowptr[r+1]) are
 *     the SENDERS of its incoming edges in original edge order (stable) - the order in which
 *     tf.unsorted_segment_sum (CPU) adds a receiver's edges up on the reference's edge list.
 *     Summation order actually taken: the fused kernels' in-kernel gathers add every row up sequentially in that
 *     order; the stand-alone aggregation kernel (gnf_aggregate_f32, and the launch in front of the large-batch
 *     kernel) does so for rows of up to 32 edges, and adds a longer row up as contiguous edge segments whose
 *     partial sums are combined in a FIXED segment order (gnf_layered.hip: deterministic and bitwise reproducible,
 *     but not the sequential sum - it differs from it in rounding only).
 */
#ifndef GNF_H
#define GNF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNF_ABI_VERSION 9
#define GNF_MAX_LAYERS 8 /* Linear layers per MLP (gnn.py:165-166 builds num_layers of them) */

typedef void* gnf_stream_t; /* hipStream_t */

enum GnfStatus {
    GNF_OK = 0,
    GNF_EINVAL = -1,      /* null pointer / bad enum / bad flag */
    GNF_ESHAPE = -2,      /* dimension mismatch (odd D, ld < D, MLP in/out dims inconsistent ...) */
    GNF_EWORKSPACE = -3,  /* workspace too small: call gnf_workspace_bytes */
    GNF_EHIP = -4,        /* a HIP runtime call / kernel launch failed */
    GNF_EUNSUPPORTED = -5 /* valid request outside what this build implements */
};

enum GnfAgg { GNF_AGG_SUM = 0, GNF_AGG_MEAN = 1 };             /* tf.unsorted_segment_sum / _mean (gnn.py:239,245,251,256) */
enum GnfCombine { GNF_COMBINE_EPS = 0, GNF_COMBINE_CONCAT = 1 }; /* AggThenMLPBlock gnn.py:122-126 / ConcatThenMLPBlock gnn.py:107-111 */
enum GnfAct { GNF_ACT_RELU = 0, GNF_ACT_LEAKY_RELU = 1 };       /* tf.nn.relu / tf.nn.leaky_relu(alpha) (run_grevnet.py:158,179,205) */
enum GnfDirection { GNF_FORWARD = 0, GNF_INVERSE = 1 };          /* x*exp(s)+t (gnn.py:323,338) / (z-t)*exp(-s) (gnn.py:359,372) */

/* Graph topology of the whole batch (block-diagonal union of graphs), receiver-sorted CSR. */
typedef struct GnfCsr {
    const int32_t* rowptr; /* [n_nodes + 1] */
    const int32_t* col;    /* [n_edges] sender node ids */
    int64_t n_nodes;       /* N = sum(n_node)  (run_grevnet.py:298) */
    int64_t n_edges;       /* E = sum(n_edge), self loops included */
} GnfCsr;

/* Optional attention front-end of a net: DMSelfAttentionMLP (gnn.py:480-553) around DMSelfAttention
 * (gnn.py:385-477), the reference drivers' default make_gnn_fn (run_grevnet.py:56,199-211).  With it,
 * the MLP input is [x || new] (concat) or new, new = reshape(attended, heads*v) @ Wo:
 *   q = x Wq, k = x Wk [N, heads, kq]; v = x Wv [N, v] repeated over heads;
 *   logit[e,h] = <q[sender e, h], k[receiver e, h]> (/ sqrt(kq) if kq_dim_division);
 *   softmax over each receiver's incoming edges; attended[r,h] = sum_e w[e,h] v[sender e].
 * All weights device, row-major [in, out] like snt.Linear(use_bias=False) (gnn.py:509-540).
 * GnfGnnSpec.agg / combine / epsilon are ignored for such a net; activation / alpha still apply to
 * the MLP.  residual: MLP output += x (gnn.py:547-548).  layer_norm: snt.LayerNorm() over the feature axis of
 * the block's output after the residual add (gnn.py:550-552): (h - mean) / sqrt(var + GNF_LN_EPS) * ln_gamma
 * + ln_beta with the biased per-row variance; ln_gamma / ln_beta have the MLP's output width. */
#define GNF_LN_EPS 1e-5f
typedef struct GnfAttn {
    /* ONE limit, checked by every entry point that takes the block (forward, inverse, backward): num_heads in 1..64 and
     * num_heads * kq_dim <= 256 and num_heads * v_dim <= 256 and, with H = D / 2 the width of the conditioning half,
     * pad16(2 * num_heads * kq_dim + v_dim) + pad16(H) + H <= 1272 (sixteen rows of the backward pass's dL/dx_cond product in
     * one CU's LDS; the drivers' geometries: H <= 544 / H <= 536) - GNF_ESHAPE otherwise.  Inside it every geometry runs; the
     * drivers' defaults have kernels of their own (run_grevnet.py:74-77: 8 heads, kq = v = 10, C = 80;
     * train_grevnet_with_data.py:40-46: 1 head, kq = v = 64, C = 64).  The forward and inverse kernels alone could take
     * some geometries beyond it; the limit is deliberately the same for all three entry points, so that a flow that evaluates
     * or samples is also a flow that trains (the reference's DMSelfAttentionMLP has no bound: gnn.py:480-553). */
    int32_t num_heads;
    int32_t kq_dim;
    int32_t v_dim;
    int32_t out_dim;         /* concat_heads_output_dim */
    int32_t concat;          /* attn_concat */
    int32_t kq_dim_division; /* divide logits by sqrt(kq_dim) */
    int32_t residual;
    int32_t layer_norm;
    const float* Wq; /* [H, num_heads*kq_dim] */
    const float* Wk; /* [H, num_heads*kq_dim] */
    const float* Wv; /* [H, v_dim] */
    const float* Wo; /* [num_heads*v_dim, out_dim] */
    const float* ln_gamma; /* [MLP output width], read when layer_norm != 0 (else may be NULL) */
    const float* ln_beta;  /* [MLP output width] */
} GnfAttn;

/* One snt.nets.MLP (gnn.py:159-180): num_layers Linear layers, y = x @ W + b, W row-major [in,out];
 * activation between layers, none after the last (activate_final=False, gnn.py:179). */
typedef struct GnfMlp {
    int32_t num_layers;               /* K >= 1 */
    int32_t dims[GNF_MAX_LAYERS + 1]; /* dims[0] = input width, dims[j+1] = output width of layer j */
    const float* W[GNF_MAX_LAYERS];   /* device, [dims[j], dims[j+1]] row-major */
    const float* b[GNF_MAX_LAYERS];   /* device, [dims[j+1]] */
    const float* packed;              /* device buffer written by gnf_pack_mlp (MFMA fragment order), or NULL */
    const GnfAttn* attn;              /* HOST pointer: attention front-end of this net, or NULL for the
                                         message-passing blocks described by GnfGnnSpec */
} GnfMlp;

/* What a make_gnn_fn() product does around its MLP (gnn.py:238-257). */
typedef struct GnfGnnSpec {
    int32_t agg;        /* GnfAgg */
    int32_t combine;    /* GnfCombine */
    float epsilon;      /* AggThenMLPBlock.epsilon (gnn.py:120,123); ignored for concat */
    int32_t activation; /* GnfAct */
    float alpha;        /* leaky_relu slope (TF default 0.2) */
} GnfGnnSpec;

/* Parameter structure of GRevNet.__init__ (gnn.py:274-302).  s_nets / t_nets are HOST arrays of
 * GnfMlp descriptors: 2*T entries indexed [half*T + i] (weight_sharing = 0, gnn.py:288-296) or 2
 * entries indexed [half] (weight_sharing = 1, gnn.py:284-286).  half 0 nets read columns [0,D/2)
 * and update [D/2,D); half 1 nets the reverse (gnn.py:320-338). */
/* One batch-norm bijector of the flow: make_batch_norm() (gnn.py:260-263) = tfb.BatchNormalization(
 * tf.layers.BatchNormalization(axis=-1, gamma_constraint=relu + 1e-6), training=True).  All pointers are
 * device fp32 [D/2].  GNF_FORWARD (f) normalises the conditioning half with THIS batch's moments over the
 * node axis and adds N * sum_f(log gamma_f - 0.5 log(var_f + epsilon)) to the log-det (gnn.py:310-313,
 * 325-328); the moments are handed back in batch_mean / batch_variance (may be NULL) for the moving-average
 * update the training step runs (UPDATE_OPS, run_grevnet.py:360).  GNF_INVERSE (g) de-normalises with the
 * MOVING statistics (bn.forward, gnn.py:356-358, 369-371).  TFP-0.7 / tf.layers semantics restated from
 * the upstream sources (absent here: unpinned). */
typedef struct GnfBatchNorm {
    const float* gamma;
    const float* beta;
    const float* moving_mean;
    const float* moving_variance;
    float* batch_mean;
    float* batch_variance;
    float epsilon; /* tf.layers.BatchNormalization default 1e-3 */
    int32_t reserved;
} GnfBatchNorm;

typedef struct GnfFlow {
    int32_t num_timesteps; /* T */
    int32_t weight_sharing;
    const GnfMlp* s_nets; /* host */
    const GnfMlp* t_nets; /* host */
    GnfGnnSpec gnn;
    /* ABI v4: NULL (use_batch_norm=False), or host array of 2*T bijectors indexed [half*T + i] - one per
     * half-step even with weight sharing (gnn.py:298-299) */
    const GnfBatchNorm* bns;
    /* ABI v5: cross-rank batch statistics.  The reference runs on ONE device, so its bijector sees the moments of the
     * whole batch (gnn.py:310-313).  Under graph sharding (one rank per GPU) each rank holds a part of the batch: with
     * bn_allreduce == NULL the moments are per shard (ordinary data-parallel batch norm); with a hook they are the
     * whole batch's again.  Per bijector call the library puts this rank's [sum_f, sum-of-squares_f] pairs (f < D/2,
     * interleaved) followed by its node count into bn_sync_buf (2*(D/2)+1 doubles of device memory, caller-owned),
     * calls bn_allreduce(ctx, bn_sync_buf, 2*(D/2)+1, stream) - which must enqueue an in-place SUM all-reduce of that
     * buffer over the ranks, ordered on `stream` (ncclAllReduce on it, or torch.distributed on the current stream) and
     * return 0 - and normalises with the reduced moments; the backward pass does the same with its two sums.  Every
     * rank must make the same sequence of calls (a rank with an empty shard cannot take part). */
    int (*bn_allreduce)(void* ctx, double* device_buf, int64_t count, gnf_stream_t stream);
    void* bn_allreduce_ctx;
    double* bn_sync_buf;
    /* ABI v5: optional stash of the attention front-end (attention GNNs only; NULL = none).  The reversible backward
     * pass recomputes every half-step's activations from the reconstructed inputs; for the attention front-end
     * (q | k | v projections, edge softmax, attended values: the costliest part of such a half-step) that recompute
     * can be traded for memory: gnf_grevnet_f32(GNF_FORWARD) then leaves each half-step's q | k | v, layer-0 inputs,
     * attended values and softmax statistics of both nets in attn_stash, and gnf_grevnet_backward_f32 - called next with the SAME flow, graph and the z that
     * forward produced - reads them instead of recomputing.  gnf_attn_stash_bytes() sizes it (2T slots). */
    float* attn_stash;
    size_t attn_stash_bytes;
    /* ABI v8: optional stash of the MLP rows (batches of up to 256 16-node tiles; NULL = none).  Same trade for the
     * MLPs: gnf_grevnet_f32 / gnf_grevnet_from_f32(GNF_FORWARD) then leaves every half-step's layer-0 input, hidden
     * activations, their sign bits and s, t of both nets in mlp_stash (what TensorFlow keeps for tf.gradients anyway), and
     * gnf_grevnet_backward_f32 - called next with the SAME flow, graph and the z that forward produced - skips the
     * recompute half of its fused kernel and feeds the weight-gradient GEMMs from the stash.  gnf_mlp_stash_bytes()
     * sizes it and returns 0 where the library would not use one (batches of more than one 16-node tile per CU on the fused
     * kernels, attention blocks that end in LayerNorm): pass NULL then.  Nets too wide for the fused kernels (the data
     * driver's 2048 x 3 MLPs) use it too since round 5: their layered forward writes every hidden activation and s, t
     * into the slot and the backward pass skips its recompute of both MLPs.  The stash is OPTIONAL at every size: it is memory
     * for time (2T slots of every hidden activation - about 20 GB for a 30 k-node batch of those nets), gnf_mlp_stash_bytes()
     * reports what it would take (0 beyond 48 GB) and the caller decides whether the device has it; with mlp_stash = NULL the
     * same flow trains through the recomputing walk.  (The Python trainer takes it within half of the free device memory,
     * or the caller's cap: train.py, GRevNetTrainer.mlp_stash_max_bytes.) */
    float* mlp_stash;
    size_t mlp_stash_bytes;
} GnfFlow;

int gnf_abi_version(void);
/* Launch-shape forcing (ABI v6; the list was cut from 18 A/B switches to 6 in round 4, a 7th came in round 6).  The library picks its kernel
 * instances by batch size; the parity tests force them on small batches through these named integers.  value 0 =
 * automatic.  force_shape: fused forward workgroup shape <MT><NETS>, e.g. 21; 40 / 30 / 20 / 10: the large-batch kernel
 * with that many row tiles per workgroup at most (49: cap 4 with the split row tiles' hand-over flag withheld - fault
 * injection for the lost-partner branch, tests only).  attn_kernel: 1 the attention rows kernels, 2 the edge-tiled kernel
 * (either keeps the front-end out of the fused kernel's prologue).  attn_bwd_rows: 64 / 32 / 16 / 3264 rows per workgroup
 * of the attention rows / edge kernels.  bwd_generic: the backward pass through the generic GEMM path.  dw_grouped: weight
 * gradients through the grouped kernel.  dw_wide_units: the wide weight-gradient kernel with that many workgroups at most.
 * dw_thin_on_dw: 1 = in the merged backward + weight-gradient launch of a training step with the MLP-row stash, the thin first /
 * last layers' units stay with the weight-gradient workgroups instead of the backward-tile workgroups (the round-5 dealing).
 * Process-wide, relaxed atomics: takes effect for calls made after it returns.  Unknown name: GNF_EINVAL.  Nothing in the
 * reference corresponds to these. */
int gnf_set_option(const char* name, int64_t value);
int64_t gnf_get_option(const char* name);
/* Bytes of GnfFlow.attn_stash for n_nodes nodes of width D (0 when the flow's nets have no attention front-end). */
size_t gnf_attn_stash_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow);
/* ABI v8: bytes of GnfFlow.mlp_stash for n_nodes nodes of width D (2T slots), or 0 where it would not be used. */
size_t gnf_mlp_stash_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow);
const char* gnf_last_error(void); /* thread-local, valid until the next failing call on this thread */

/* Number of floats gnf_pack_mlp writes for this MLP (host computation, no device access). */
int64_t gnf_packed_floats(const GnfMlp* mlp);

/* Re-lay one MLP's weights into the zero-padded MFMA-fragment order the fused kernel streams
 * (one-off, like any inference engine's weight pre-pack; replaces nothing in the reference -
 * Sonnet keeps W as a [in,out] tf.Variable, gnn.py:167-174).  `packed` must hold
 * gnf_packed_floats(mlp) floats. */
int gnf_pack_mlp(const GnfMlp* mlp, float* packed, gnf_stream_t stream);

/* Build the receiver-sorted CSR from a GraphsTuple edge list (senders/receivers with global node
 * ids, graphs contiguous, n_node/n_edge per graph: the fields of gn.graphs.GraphsTuple built at
 * train_grevnet_with_data.py:265-271 / graph_data.py:122).  Stable within each receiver.
 * ws: gnf_csr_workspace_bytes(n_graphs, N) bytes. */
size_t gnf_csr_workspace_bytes(int64_t n_graphs, int64_t n_nodes);
int gnf_build_csr(const int32_t* senders, const int32_t* receivers, const int32_t* n_node,
                  const int32_t* n_edge, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                  int32_t* rowptr, int32_t* col, void* ws, size_t ws_bytes, gnf_stream_t stream);

/* Kernel A alone: out[r, 0:H) = reduce_{e: recv[e]=r} x[send[e], 0:H)   (sum, or sum/max(deg,1)).
 * Replaces EdgeBlock gather + ReceivedEdgesToNodesAggregator (gnn.py:103-104,117-118,151-156). */
int gnf_aggregate_f32(const GnfCsr* csr, const float* x, int64_t ldx, int32_t H, int32_t agg,
                      float* out, int64_t ldo, gnf_stream_t stream);

/* One GNN module call on its own, outside a coupling: out[N, dims[K]] = MLP(combine(x, agg(x))),
 * i.e. NodeBlockGNN._build (gnn.py:155-156) -> AggThenMLPBlock / ConcatThenMLPBlock._build
 * (gnn.py:122-126 / 107-111).  x is [N, H] (stride ldx), out is [N, mlp->dims[K]] (stride ldo).
 * ws: gnf_gnn_workspace_bytes(N, H, mlp, combine). */
size_t gnf_gnn_workspace_bytes(int64_t n_nodes, int32_t H, const GnfMlp* mlp, int32_t combine);
int gnf_gnn_apply_f32(const GnfCsr* csr, const GnfMlp* mlp, const GnfGnnSpec* gnn, const float* x,
                      int64_t ldx, int32_t H, float* out, int64_t ldo, void* ws, size_t ws_bytes,
                      gnf_stream_t stream);

/* Workspace (bytes) needed by gnf_coupling_half_f32 / gnf_grevnet_f32 for this problem. */
size_t gnf_workspace_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow);

/* One coupling half-step (one s-net + one t-net on the same conditioning half):
 *   s = S(x_cond), t = T(x_cond);  x_upd <- x_upd*exp(s)+t  (FORWARD)  or  (x_upd - t)*exp(-s)  (INVERSE)
 * x_cond / x_upd point at the first column of the two halves inside the [N, D] buffer (stride ld).
 * If logdet_accum != NULL, sum(s) over all nodes and features is ADDED to *logdet_accum (device
 * fp64; caller zeroes it).  Replaces gnn.py:320-323 (or 335-338 / 347-359 / 361-372). */
int gnf_coupling_half_f32(const GnfCsr* csr, const GnfMlp* s_net, const GnfMlp* t_net,
                          const GnfGnnSpec* gnn, const float* x_cond, float* x_upd, int64_t ld,
                          int32_t H, int32_t direction, double* logdet_accum, void* ws,
                          size_t ws_bytes, gnf_stream_t stream);

/* The whole flow, in place on x[N, D]:
 *   FORWARD  = GRevNet.f (gnn.py:304-341): x -> z; sums[0] = log_det_jacobian, sums[1] = sum(z^2)
 *              (the data term of run_grevnet.py:292-294), both device fp64, written (not accumulated).
 *   INVERSE  = GRevNet.g (gnn.py:343-373): z -> x; sums may be NULL (nothing written).
 * With flow->bns != NULL (use_batch_norm=True) every half-step of f is preceded by its batch-norm bijector
 * (gnn.py:310-313, 325-328) and every half-step of g followed by it (gnn.py:356-358, 369-371); see GnfBatchNorm. */
int gnf_grevnet_f32(const GnfCsr* csr, const GnfFlow* flow, float* x, int64_t ld, int32_t D,
                    int32_t direction, double* sums, void* ws, size_t ws_bytes, gnf_stream_t stream);
/* ABI v6: the same, out of place - what the reference's functional TF graph does (GRevNet.f / .g return NEW tensors,
 * gnn.py:340-341, 373; the input graph stays intact): reads x_src[N, D] (leading dimension ld_src), leaves the result in
 * x[N, D].  x_src == NULL or x_src == x: in place.  The first coupling half-step reads the source and writes the
 * destination directly where the fused kernel runs it (no separate copy pass); otherwise one copy, then in place. */
int gnf_grevnet_from_f32(const GnfCsr* csr, const GnfFlow* flow, const float* x_src, int64_t ld_src, float* x, int64_t ld,
                         int32_t D, int32_t direction, double* sums, void* ws, size_t ws_bytes, gnf_stream_t stream);

/* Kernel D alone: *out = sum_{n,j} z[n,j]^2 in fp64 (device).  log_prob_zs =
 * -0.5 * (*out) - 0.5 * D * ln(2*pi) * N   (tfd.MultivariateNormalDiag(0,1).log_prob summed,
 * run_grevnet.py:292-294; train_grevnet_with_data.py:348-351). ws: gnf_workspace_bytes(...) or
 * at least 8 * 1024 bytes. */
int gnf_gauss_sumsq_f32(const float* z, int64_t n_nodes, int32_t D, int64_t ld, double* out,
                        void* ws, size_t ws_bytes, gnf_stream_t stream);

/* Decoder that follows the sampling pass: pred_adj(graph, scaled_hacky_sigmoid_l2)
 * (loss.py:45-53 distance, 131-151 block-diagonal mask, 154-159 pred_adj with remove_diag; called at
 * train_grevnet_with_data.py:415-416 and thresholded at 0.5 at :532-533):
 *   P[i,j] = sigmoid(10 * (1 - ||z_i - z_j||^2 / sqrt(D)))  for i != j in the same graph, 0 on the diagonal.
 * The reference builds a dense masked [N,N] matrix; only the per-graph [n_g, n_g] blocks are produced
 * here: out_blocks holds them concatenated (sum n_g^2 floats), block g starts at block_off[g]
 * (block_off: device int64[n_graphs + 1], written by this call).  max_nodes_per_graph: any upper
 * bound on n_node (sizes the launch; nothing is read from the host).  Any D (up to 1024 a tile of sixteen rows sits in
 * LDS, beyond that the rows are read from global memory; same order of additions).
 * ws: gnf_pred_adj_workspace_bytes(n_graphs). */
size_t gnf_pred_adj_workspace_bytes(int64_t n_graphs);
int gnf_pred_adj_f32(const float* z, int64_t ld, int32_t D, const int32_t* n_node, int64_t n_graphs,
                     int32_t max_nodes_per_graph, float* out_blocks, int64_t* block_off, void* ws,
                     size_t ws_bytes, gnf_stream_t stream);

/* ---- training step ---------------------------------------------------------------
 * Replaces optimizer.compute_gradients(total_loss) (run_grevnet.py:361-362) for
 *   total_loss = -(sum_n log N(z_n; 0, I) + log_det_jacobian)           (run_grevnet.py:291-295)
 * by reversible back-propagation through GRevNet.f (gnn.py:304-341): the half-steps are walked in
 * reverse, each one's input is rebuilt from its output with the inverse update (gnn.py:359,372) and
 * its two GNNs are recomputed - no forward activation is kept (the drivers' use_efficient_backprop,
 * run_grevnet.py:46,282-288).
 *   csr    the batch's CSR by receiver (gnf_build_csr)
 *   csr_t  the same edges grouped by SENDER: gnf_build_csr(receivers, senders, ...) (arguments swapped)
 *   flow   the nets, exactly as passed to gnf_grevnet_f32 (raw W/b are read; `packed` is ignored)
 *   grad   a GnfFlow of the same shape whose W[j]/b[j] point at the GRADIENT buffers ([in,out] / [out],
 *          overwritten; with weight_sharing the T uses of a net are summed); packed/attn ignored
 *          With batch-norm bijectors (flow->bns): grad->bns[q].gamma / .beta point at their gradient buffers,
 *          and flow->bns[q].batch_mean / batch_variance must still hold the moments of the forward pass
 *          (tf.gradients differentiates through the batch moments and through the bijector's log-det term).
 *   z      in: f(x) as left by gnf_grevnet_f32(GNF_FORWARD); out: x again (the reconstruction)
 *   aux_stream  NULL, or a second stream: the weight-gradient GEMMs of a half-step then overlap the next
 *          half-step's fused kernel (fork / join by events; everything is complete on `stream` order)
 * Message-passing and attention GNNs, with or without batch-norm bijectors.
 * ws: gnf_backward_workspace_bytes(n_nodes, D, flow). */
size_t gnf_backward_workspace_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow);
int gnf_grevnet_backward_f32(const GnfCsr* csr, const GnfCsr* csr_t, const GnfFlow* flow, const GnfFlow* grad,
                             float* z, int64_t ld, int32_t D, void* ws, size_t ws_bytes, gnf_stream_t stream,
                             gnf_stream_t aux_stream);

/* Re-pack EVERY net of a flow (W/b -> `packed`, nets with packed == NULL skipped) in a handful of
 * launches; call after an optimiser step.  Nothing in the reference (weight pre-pack).
 * Nets too wide for the LDS-resident kernels (they run layer by layer): only what those kernels read of `packed` follows the
 * weights - the wide layers' fragments in both orientations, the fragments of a short first layer and of a thin last layer,
 * and every bias row; the other regions of such a net's `packed` are read by nothing. */
int gnf_pack_flow(const GnfFlow* flow, gnf_stream_t stream);

/* ABI v6: what follows the optimiser step for the batch-norm bijectors of a flow, all 2T of them in one launch:
 * gamma <- relu(gamma) + 1e-6 (the gamma_constraint of make_batch_norm, gnn.py:261-262, which TF projects after
 * apply_gradients) and the moving-average update tf.layers.BatchNormalization registers in UPDATE_OPS
 * (run_grevnet.py:360): moving <- moving * momentum + batch * (1 - momentum), with the batch moments the last
 * gnf_grevnet_f32(GNF_FORWARD) left in batch_mean / batch_variance.  H = D/2.  No-op without bijectors. */
int gnf_bn_post_step_f32(const GnfFlow* flow, int32_t H, float momentum, gnf_stream_t stream);

/* ABI v9: the path's collective without a host language in the launch path.  The reference is single-device (no
 * counterpart); under graph sharding the only exchange INSIDE the flow's launch sequence is the batch-norm
 * bijector's cross-rank moments (GnfFlow.bn_allreduce above, 2 D/2 + 1 doubles per bijector call).
 * gnf_rccl_allreduce_sum_f64 has exactly that hook's signature with an RCCL communicator as its context:
 *     flow.bn_allreduce = gnf_rccl_allreduce_sum_f64;  flow.bn_allreduce_ctx = comm;
 * enqueues ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream) on the caller's stream.  `comm` is an
 * ncclComm_t: the caller's own, or one made here - rank 0 calls gnf_rccl_unique_id, hands the 128 bytes to the other
 * ranks by whatever means it has (MPI, a file, torch.distributed), every rank calls gnf_rccl_comm_create on its device.
 * librccl.so is resolved with dlopen at the first of these calls (no link-time dependency: GNF_EINVAL + gnf_last_error
 * when it cannot be found); a process that already holds RCCL (PyTorch bundles its own copy) gets that copy. */
#define GNF_RCCL_UNIQUE_ID_BYTES 128
int gnf_rccl_unique_id(char* id_out /* GNF_RCCL_UNIQUE_ID_BYTES */);
int gnf_rccl_comm_create(const char* id, int32_t n_ranks, int32_t rank, void** comm_out);
int gnf_rccl_comm_destroy(void* comm);
int gnf_rccl_allreduce_sum_f64(void* comm, double* device_buf, int64_t count, gnf_stream_t stream);

/* tf.train.AdamOptimizer.apply_gradients on one flat fp32 parameter vector (run_grevnet.py:352-356, 375):
 *   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  w <- w - lr_t m / (sqrt(v) + epsilon)
 * lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) is computed by the caller (TF does it on the host side too). */
int gnf_adam_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                 float epsilon, gnf_stream_t stream);
/* tf.clip_by_value on a flat gradient (run_grevnet.py:363-367). */
int gnf_clip_by_value_f32(float* g, int64_t n, float lo, float hi, gnf_stream_t stream);
/* tf.clip_by_norm per gradient tensor (run_grevnet.py:369-372): tensor i = g[offsets[i] .. offsets[i+1])
 * (offsets: device int64[n_tensors + 1]) is scaled by clip_norm / max(||t||_2, clip_norm).
 * ws: NULL (one workgroup per tensor: fine for small tensors) or gnf_clip_workspace_bytes(n_tensors) bytes of device
 * scratch (ABI v5: 64 workgroups per tensor in two passes; a 2048 x 2048 gradient in one workgroup takes 10 ms). */
size_t gnf_clip_workspace_bytes(int32_t n_tensors);
int gnf_clip_by_norm_f32(float* g, const int64_t* offsets, int32_t n_tensors, float clip_norm, void* ws, size_t ws_bytes,
                         gnf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GNF_H */

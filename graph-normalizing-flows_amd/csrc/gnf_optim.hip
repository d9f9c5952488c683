// What follows the gradients in a training iteration (run_grevnet.py:352-377, 440-447): Adam on flat parameter
// vectors, the two clipping modes, the batch-norm bijectors' post-step (gamma constraint + moving averages), and the
// re-pack of every net's MFMA fragment copy after the weights moved.  Split out of gnf_train.hip in round 3.
#include <string.h>

#include "gnf_common.h"

namespace gnf {

// ---- multi-tensor re-pack (after an optimiser step every net's MFMA fragment copy is stale) ----------
struct PackDesc {
    const float* W;
    const float* b;
    float* wout;
    float* bout;
    float* wtout;
    int32_t I, O, Ip, Op;
};
static constexpr int kPackBatch = 56;
struct PackBatch {
    PackDesc d[kPackBatch];
};

__global__ __launch_bounds__(256) void k_pack_multi(const PackBatch pb) {
    const PackDesc& d = pb.d[blockIdx.y];
    // (a layer with no reader of either fragment copy - wout == wtout == NULL - still gets its bias row: nw = 0)
    const int64_t nw = d.wout || d.wtout ? (int64_t)d.Ip * d.Op : 0;
    // a thread = the four k-consecutive elements of one lane's fragment (i = 4 t .. 4 t + 3): one 16-byte store per copy, and
    // for the transposed copy - whose four elements are consecutive in a row of W - one 16-byte load where rows are aligned
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = 4 * t;
    if (i < nw) {  // same fragment order as k_pack_layer (gnf_fused.hip)
        const int lane = (int)(t & 63);
        const int64_t blk = t >> 6;
        const int nts = d.Op >> 4;
        const int kg = (int)(blk / nts), nt = (int)(blk % nts);
        const int k = 16 * kg + 4 * (lane >> 4);
        const int c = 16 * nt + (lane & 15);
        float w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = (k + q < d.I && c < d.O) ? d.W[(int64_t)(k + q) * d.O + c] : 0.f;
        const int nts_t = d.Ip >> 4;
        const int kg_t = (int)(blk / nts_t), nt_t = (int)(blk % nts_t);
        const int ko = 16 * kg_t + 4 * (lane >> 4);
        const int ci = 16 * nt_t + (lane & 15);
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
        if (ci < d.I) {
            const float* row = d.W + (int64_t)ci * d.O + ko;
            if (ko + 3 < d.O && (d.O & 3) == 0 && (reinterpret_cast<uintptr_t>(d.W) & 15) == 0) {
                const float4 v = *reinterpret_cast<const float4*>(row);
                t4[0] = v.x, t4[1] = v.y, t4[2] = v.z, t4[3] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ko + q < d.O) t4[q] = row[q];
            }
        }
        // (wout / wtout == NULL: that copy of the layer has no reader - wide nets, see pack_flow)
        if (((reinterpret_cast<uintptr_t>(d.wout) | reinterpret_cast<uintptr_t>(d.wtout)) & 15) == 0) {
            if (d.wout) *reinterpret_cast<float4*>(d.wout + i) = make_float4(w4[0], w4[1], w4[2], w4[3]);
            if (d.wtout) *reinterpret_cast<float4*>(d.wtout + i) = make_float4(t4[0], t4[1], t4[2], t4[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (d.wout) d.wout[i + q] = w4[q];
                if (d.wtout) d.wtout[i + q] = t4[q];
            }
        }
    } else if (i < nw + 4 * ((d.Op + 3) / 4)) {
        for (int q = 0; q < 4; ++q) {
            const int c = (int)(i - nw) + q;
            if (c < d.Op) d.bout[c] = c < d.O ? d.b[c] : 0.f;
        }
    }
}

static inline int pad16i(int v) { return (v + 15) & ~15; }

static int pack_flow(const GnfFlow* flow, hipStream_t st) {
    const int n_nets = flow->weight_sharing ? 2 : 2 * flow->num_timesteps;
    PackBatch pb;
    int cnt = 0;
    int64_t maxtot = 0;
    auto flush = [&]() -> int {
        if (!cnt) return GNF_OK;
        hipLaunchKernelGGL(k_pack_multi, dim3((unsigned)((maxtot / 4 + 255) / 256), cnt), dim3(256), 0, st, pb);  // 4 elements per thread
        GNF_LAUNCH_CHECK("k_pack_multi");
        cnt = 0;
        maxtot = 0;
        return GNF_OK;
    };
    for (int kind = 0; kind < 2; ++kind)
        for (int q = 0; q < n_nets; ++q) {
            const GnfMlp* m = kind ? &flow->t_nets[q] : &flow->s_nets[q];
            if (!m->packed) continue;
            // nets too wide for LDS never run the fused kernels: of their fragment-order copies only what the wide-layer
            // kernel of the layered path reads (gnf_linear_big.hip: Wp of the layers linear_big_fwd_layer names and of a
            // thin last layer it multiplies out of its accumulators - linear_big_fused_last -, Wp of a short reduction into a wide layer - linear_short_fwd_layer -, WpT of the ones
            // linear_big_bwd_layer names, and the bias rows) follows the weights - the full re-pack of the
            // data-backed trainer's 2048-wide nets was 1.1 ms per step and 1.5 GB
            const bool wide_only = !fused_fits_lds(m) && !fused_bwd_fits_lds(m);
            int64_t woff = 0, boff = 0;
            for (int j = 0; j < m->num_layers; ++j) boff += (int64_t)pad16i(m->dims[j]) * pad16i(m->dims[j + 1]);
            int64_t toff = boff;
            for (int j = 0; j < m->num_layers; ++j) toff += pad16i(m->dims[j + 1]);
            for (int j = 0; j < m->num_layers; ++j) {
                const int I = m->dims[j], O = m->dims[j + 1], Ip = pad16i(I), Op = pad16i(O);
                float* pk = const_cast<float*>(m->packed);
                const bool need_w = !wide_only || linear_big_fwd_layer(I, O) || linear_big_fused_last(m, j) || linear_short_fwd_layer(I, O);
                const bool need_wt = !wide_only || linear_big_bwd_layer(I, O);
                {   // every layer's padded bias row follows its bias (include/gnf.h: "every bias row"), the fragment copies
                    // only where a kernel reads them
                    pb.d[cnt++] = PackDesc{m->W[j], m->b[j], need_w ? pk + woff : nullptr, pk + boff,
                                           need_wt ? pk + toff + woff : nullptr, I, O, Ip, Op};
                    const int64_t tot = (need_w || need_wt ? (int64_t)Ip * Op : 0) + Op;
                    maxtot = maxtot > tot ? maxtot : tot;
                }
                woff += (int64_t)Ip * Op;
                boff += Op;
                if (cnt == kPackBatch) {
                    const int rc = flush();
                    if (rc) return rc;
                }
            }
        }
    return flush();
}

// ---- optimiser ---------------------------------------------------------------------------------
// tf.train.AdamOptimizer (run_grevnet.py:352-356): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) is computed by the
// caller (fp64 on the host, like TF's python side);  m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;
// w <- w - lr_t m / (sqrt(v) + eps)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float lr_t, float b1, float b2,
                                              float eps) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gv = g[i];
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        w[i] = w[i] - lr_t * mv / (sqrtf(vv) + eps);
    }
}

__global__ __launch_bounds__(256) void k_clip_value(float* __restrict__ g, int64_t n, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        g[i] = fminf(fmaxf(g[i], lo), hi);
}

// tf.clip_by_norm per gradient tensor (run_grevnet.py:369-372): t * clip / max(||t||_2, clip).
// One workgroup per tensor; offsets[i] .. offsets[i+1] delimit tensor i inside the flat gradient.
__global__ __launch_bounds__(256) void k_clip_norm(float* __restrict__ g, const int64_t* __restrict__ offsets,
                                                   float clip) {
    __shared__ double red[256];
    const int64_t beg = offsets[blockIdx.x], end = offsets[blockIdx.x + 1];
    double s = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float nrm = (float)sqrt(red[0]);
    const float scale = clip / fmaxf(nrm, clip);
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) g[i] *= scale;
}

// The same in two passes with kClipSlices workgroups per tensor (a 2048 x 2048 gradient in ONE workgroup took 10 ms):
// pass 1: fp64 sum of squares of slice s of tensor t -> part[t][s];  pass 2: every slice's workgroup adds the
// tensor's partials in slice order (the same number in every workgroup) and scales its slice.
static constexpr int kClipSlices = 64;
__device__ __forceinline__ void clip_slice(const int64_t* __restrict__ offsets, int64_t& lo, int64_t& hi) {
    const int64_t beg = offsets[blockIdx.y], len = offsets[blockIdx.y + 1] - beg;
    const int64_t per = (len + kClipSlices - 1) / kClipSlices;
    lo = beg + (int64_t)blockIdx.x * per;
    hi = lo + per < beg + len ? lo + per : beg + len;
}
__global__ __launch_bounds__(256) void k_clip_norm_part(const float* __restrict__ g, const int64_t* __restrict__ offsets,
                                                        double* __restrict__ part) {
    __shared__ double red[256];
    int64_t lo, hi;
    clip_slice(offsets, lo, hi);
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * kClipSlices + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_clip_norm_scale(float* __restrict__ g, const int64_t* __restrict__ offsets,
                                                         const double* __restrict__ part, float clip) {
    int64_t lo, hi;
    clip_slice(offsets, lo, hi);
    if (lo >= hi) return;
    double tot = 0.0;
    for (int q = 0; q < kClipSlices; ++q) tot += part[(int64_t)blockIdx.y * kClipSlices + q];
    const float nrm = (float)sqrt(tot);
    const float scale = clip / fmaxf(nrm, clip);
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) g[i] *= scale;
}

}  // namespace gnf

using namespace gnf;

extern "C" {

// After an optimiser step, for every batch-norm bijector of the flow in ONE launch (was 4 torch elementwise launches
// per bijector): the gamma_constraint projection relu(gamma) + 1e-6 (gnn.py:261-262) and tf.layers' UPDATE_OPS
// moving <- moving * momentum + batch * (1 - momentum) (run_grevnet.py:360).
struct BnPostBatch {
    GnfBatchNorm bn[48];
    int32_t H;
    float momentum;
};
__global__ __launch_bounds__(256) void k_bn_post_step(const BnPostBatch b) {
    const GnfBatchNorm bn = b.bn[blockIdx.x];
    float* gamma = const_cast<float*>(bn.gamma);
    float* mm = const_cast<float*>(bn.moving_mean);
    float* mv = const_cast<float*>(bn.moving_variance);
    for (int f = threadIdx.x; f < b.H; f += 256) {
        gamma[f] = fmaxf(gamma[f], 0.f) + 1e-6f;
        mm[f] = mm[f] * b.momentum + bn.batch_mean[f] * (1.f - b.momentum);
        mv[f] = mv[f] * b.momentum + bn.batch_variance[f] * (1.f - b.momentum);
    }
}

int gnf_bn_post_step_f32(const GnfFlow* flow, int32_t H, float momentum, gnf_stream_t stream) {
    if (!flow || H < 1 || !(momentum >= 0.f && momentum <= 1.f)) {
        set_error("gnf_bn_post_step_f32: bad arguments");
        return GNF_EINVAL;
    }
    if (!flow->bns) return GNF_OK;
    const int total = 2 * flow->num_timesteps;
    for (int q = 0; q < total; ++q) {
        const GnfBatchNorm& bn = flow->bns[q];
        if (!bn.gamma || !bn.moving_mean || !bn.moving_variance || !bn.batch_mean || !bn.batch_variance) {
            set_error("gnf_bn_post_step_f32: bijector %d has a null pointer", q);
            return GNF_EINVAL;
        }
    }
    for (int base = 0; base < total; base += 48) {
        BnPostBatch b;
        const int nb = total - base < 48 ? total - base : 48;
        for (int q = 0; q < nb; ++q) b.bn[q] = flow->bns[base + q];
        b.H = H;
        b.momentum = momentum;
        hipLaunchKernelGGL(k_bn_post_step, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, b);
        GNF_LAUNCH_CHECK("k_bn_post_step");
    }
    return GNF_OK;
}

int gnf_pack_flow(const GnfFlow* flow, gnf_stream_t stream) {
    if (!flow || !flow->s_nets || !flow->t_nets || flow->num_timesteps < 0) {
        set_error("gnf_pack_flow: null flow / nets");
        return GNF_EINVAL;
    }
    const int n_nets = flow->weight_sharing ? 2 : 2 * flow->num_timesteps;
    for (int q = 0; q < n_nets; ++q) {
        int rc = validate_mlp(&flow->s_nets[q], "gnf_pack_flow s_net");
        if (rc) return rc;
        rc = validate_mlp(&flow->t_nets[q], "gnf_pack_flow t_net");
        if (rc) return rc;
    }
    return pack_flow(flow, (hipStream_t)stream);
}

int gnf_adam_f32(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                 float epsilon, gnf_stream_t stream) {
    if (n < 0 || (n > 0 && (!w || !g || !m || !v))) {
        set_error("gnf_adam_f32: null buffer or n=%lld", (long long)n);
        return GNF_EINVAL;
    }
    if (n == 0) return GNF_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, lr_t, beta1,
                       beta2, epsilon);
    GNF_LAUNCH_CHECK("k_adam");
    return GNF_OK;
}

int gnf_clip_by_value_f32(float* g, int64_t n, float lo, float hi, gnf_stream_t stream) {
    if (n < 0 || (n > 0 && !g) || !(lo <= hi)) {
        set_error("gnf_clip_by_value_f32: bad arguments");
        return GNF_EINVAL;
    }
    if (n == 0) return GNF_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_clip_value, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, lo, hi);
    GNF_LAUNCH_CHECK("k_clip_value");
    return GNF_OK;
}

size_t gnf_clip_workspace_bytes(int32_t n_tensors) {
    return n_tensors > 0 ? (size_t)n_tensors * kClipSlices * sizeof(double) : 0;
}

int gnf_clip_by_norm_f32(float* g, const int64_t* offsets, int32_t n_tensors, float clip_norm, void* ws, size_t ws_bytes,
                         gnf_stream_t stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!g || !offsets)) || !(clip_norm > 0.f)) {
        set_error("gnf_clip_by_norm_f32: bad arguments");
        return GNF_EINVAL;
    }
    if (n_tensors == 0) return GNF_OK;
    if (!ws) {  // no scratch: one workgroup per tensor (fine for small tensors)
        hipLaunchKernelGGL(k_clip_norm, dim3((unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, g, offsets, clip_norm);
        GNF_LAUNCH_CHECK("k_clip_norm");
        return GNF_OK;
    }
    if (ws_bytes < gnf_clip_workspace_bytes(n_tensors)) {
        set_error("gnf_clip_by_norm_f32: workspace %zu < %zu bytes", ws_bytes, gnf_clip_workspace_bytes(n_tensors));
        return GNF_EWORKSPACE;
    }
    const dim3 grid(kClipSlices, (unsigned)n_tensors);
    hipLaunchKernelGGL(k_clip_norm_part, grid, dim3(256), 0, (hipStream_t)stream, g, offsets, (double*)ws);
    GNF_LAUNCH_CHECK("k_clip_norm_part");
    hipLaunchKernelGGL(k_clip_norm_scale, grid, dim3(256), 0, (hipStream_t)stream, g, offsets, (const double*)ws, clip_norm);
    GNF_LAUNCH_CHECK("k_clip_norm_scale");
    return GNF_OK;
}

}  // extern "C"

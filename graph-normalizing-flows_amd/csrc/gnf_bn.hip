// Batch-norm bijector inside the flow: make_batch_norm() = tfb.BatchNormalization(
// batchnorm_layer=tf.layers.BatchNormalization(axis=-1, gamma_constraint=relu+1e-6), training=True)
// (gnn.py:260-263), applied to the conditioning half before each half-step of f (gnn.py:310-313,325-328)
// and undone after each half-step of g (gnn.py:356-358,369-371).
//
// Semantics restated from tensorflow-probability 0.7 / tf.layers (third party, absent here: UNPINNED):
//   f:  bn.inverse(x)  = (x - mean_B) / sqrt(var_B + eps) * gamma + beta      mean_B, var_B: moments of THIS batch
//                         over the node axis (tf.nn.moments: biased variance), training=True
//       bn.inverse_log_det_jacobian(x, event_ndims=2) = N * sum_f (log gamma_f - 0.5 log(var_B,f + eps))
//                         (the bijector's ildj is the per-node scalar; event_ndims=2 sums it over the N nodes)
//   g:  bn.forward(z)  = (z - beta) / gamma * sqrt(moving_var + eps) + moving_mean      (the MOVING statistics)
// The moving averages themselves are updated by the training step (UPDATE_OPS, run_grevnet.py:360), not here:
// the batch moments are handed back through GnfBatchNorm.batch_mean / batch_variance.
#include "gnf_common.h"

namespace gnf {

static constexpr int kBnRows = 16;  // rows per workgroup of the normalising pass

// pass 1: per-workgroup column sums and sums of squares (fp64), fixed order
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, int64_t ld, int64_t n, int H,
                                                  int64_t rows_per_block, double* __restrict__ part) {
    __shared__ double sh[2][256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < H; c0 += 256) {
        const int w = H - c0 < 256 ? H - c0 : 256;  // columns of this pass
        const int lanes = 256 / w;                   // row lanes per column
        const int c = tid % w, rs = tid / w;
        double s = 0.0, q = 0.0;
        if (rs < lanes)
            for (int64_t r = r0 + rs; r < r1; r += lanes) {
                const double v = (double)x[r * ld + c0 + c];
                s += v;
                q += v * v;
            }
        sh[0][tid] = s;
        sh[1][tid] = q;
        __syncthreads();
        if (tid < w) {
            double ts = 0.0, tq = 0.0;
            for (int k = 0; k < lanes; ++k) {
                ts += sh[0][k * w + tid];
                tq += sh[1][k * w + tid];
            }
            part[((int64_t)blockIdx.x * H + c0 + tid) * 2 + 0] = ts;
            part[((int64_t)blockIdx.x * H + c0 + tid) * 2 + 1] = tq;
        }
        __syncthreads();
    }
}

// cross-rank moments: the partials of this rank summed (same fixed order as pass 2) into out[H][2], the node count
// behind them; `copy` (nullable) receives the same pairs and stays local.
__global__ __launch_bounds__(256) void k_bn_fold(const double* __restrict__ part, int nparts, int H, int64_t n,
                                                 double* __restrict__ out, double* __restrict__ copy) {
    for (int c = threadIdx.x; c < H; c += 256) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nparts; ++b) {
            s += part[((int64_t)b * H + c) * 2 + 0];
            q += part[((int64_t)b * H + c) * 2 + 1];
        }
        out[2 * c + 0] = s;
        out[2 * c + 1] = q;
        if (copy) copy[2 * c + 0] = s, copy[2 * c + 1] = q;
    }
    if (threadIdx.x == 0) out[2 * H] = (double)n;
}

int bn_sync_exchange(const GnfFlow* flow, const double* part, int nparts, int64_t n, int32_t H, double* local_copy,
                     hipStream_t st) {
    if (!flow->bn_sync_buf) {
        set_error("GnfFlow.bn_allreduce is set but bn_sync_buf is NULL (needs 2*(D/2)+1 doubles of device memory)");
        return GNF_EINVAL;
    }
    hipLaunchKernelGGL(k_bn_fold, dim3(1), dim3(256), 0, st, part, nparts, H, n, flow->bn_sync_buf, local_copy);
    GNF_LAUNCH_CHECK("k_bn_fold");
    const int rc = flow->bn_allreduce(flow->bn_allreduce_ctx, flow->bn_sync_buf, 2 * (int64_t)H + 1, (gnf_stream_t)st);
    if (rc != 0) {
        set_error("GnfFlow.bn_allreduce hook failed with %d", rc);
        return GNF_EINVAL;
    }
    return GNF_OK;
}

// pass 2: every workgroup re-reduces the (few) partials, normalises its rows in place; workgroup 0 also
// writes the log-det term and the batch moments.  n_moments (nullable, device): the node count the moments are
// over when the partials are cross-rank sums (the log-det term still counts THIS rank's n nodes).
__global__ __launch_bounds__(256) void k_bn_apply(float* __restrict__ x, int64_t ld, int64_t n, int H,
                                                  int64_t rows_per_block, const double* __restrict__ part, int nparts,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float eps, float* __restrict__ batch_mean,
                                                  float* __restrict__ batch_var, double* __restrict__ logdet_slot,
                                                  const double* __restrict__ n_moments) {
    extern __shared__ float ss[];  // scale[H] | shift[H]
    __shared__ double red[256];
    __shared__ double gsum[512];   // [G][H][2] group sums of the partials (G * H <= 256)
    float* scale = ss;
    float* shift = ss + H;
    const int tid = threadIdx.x;
    double ld_local = 0.0;
    // column sums of the partials.  Few partials (k_bn_stats' ~16 row chunks): one thread per column walks them.  Many
    // (one per 16-node tile when the previous half-step's kernel left them): all 256 threads share the walk - thread
    // (g, c) sums partials g, g + G, ..; the G group sums are then added in order.  Fixed order either way.
    const int G = (nparts > 32 && H <= 128) ? 256 / H : 1;
    if (G > 1) {
        const int c = tid % H, g = tid / H;
        double s = 0.0, q = 0.0;
        if (g < G)
            for (int b0 = g; b0 < nparts; b0 += 16 * G) {
                double ps[16], pq_[16];  // sixteen partial pairs in flight per thread (eight: six dependent round trips for 340 partial rows)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int b = b0 + k * G < nparts ? b0 + k * G : g;
                    ps[k] = part[((int64_t)b * H + c) * 2 + 0];
                    pq_[k] = part[((int64_t)b * H + c) * 2 + 1];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (b0 + k * G < nparts) {
                        s += ps[k];
                        q += pq_[k];
                    }
            }
        __syncthreads();
        if (g < G) gsum[(g * H + c) * 2 + 0] = s, gsum[(g * H + c) * 2 + 1] = q;
        __syncthreads();
    }
    for (int c = tid; c < H; c += 256) {
        double s = 0.0, q = 0.0;
        if (G > 1) {
            for (int g = 0; g < G; ++g) s += gsum[(g * H + c) * 2 + 0], q += gsum[(g * H + c) * 2 + 1];
        } else
        for (int b0 = 0; b0 < nparts; b0 += 8) {  // eight partial pairs in flight, summed in order
            double ps[8], pq_[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int b = b0 + k < nparts ? b0 + k : nparts - 1;
                ps[k] = part[((int64_t)b * H + c) * 2 + 0];
                pq_[k] = part[((int64_t)b * H + c) * 2 + 1];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (b0 + k < nparts) {
                    s += ps[k];
                    q += pq_[k];
                }
        }
        const double nm = n_moments ? *n_moments : (double)n;
        const double mean = s / nm;
        double var = q / nm - mean * mean;
        if (var < 0.0) var = 0.0;
        const float g = gamma[c];
        const float sc = g / sqrtf((float)var + eps);
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        ld_local += log((double)g) - 0.5 * log(var + (double)eps);
        if (blockIdx.x == 0) {
            if (batch_mean) batch_mean[c] = (float)mean;
            if (batch_var) batch_var[c] = (float)var;
        }
    }
    if (blockIdx.x == 0) {
        red[tid] = ld_local;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) *logdet_slot = (double)n * red[0];
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const int64_t tot = (r1 - r0) * H;
    for (int64_t i = tid; i < tot; i += 256) {
        const int64_t r = r0 + i / H;
        const int c = (int)(i % H);
        float* p = x + r * ld + c;
        *p = *p * scale[c] + shift[c];
    }
}

// g direction: z <- (z - beta) / gamma * sqrt(moving_var + eps) + moving_mean
__global__ __launch_bounds__(256) void k_bn_denorm(float* __restrict__ z, int64_t ld, int64_t n, int H,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ mmean, const float* __restrict__ mvar,
                                                   float eps) {
    const int64_t tot = n * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / H;
        const int c = (int)(i % H);
        float* p = z + r * ld + c;
        *p = (*p - beta[c]) / gamma[c] * sqrtf(mvar[c] + eps) + mmean[c];
    }
}

// moment pass: few large row chunks (each workgroup's partial is re-read by every workgroup of the second pass)
int bn_blocks(int64_t n, int64_t* rows_per_block) {
    // about sixteen chunks, at least 32 rows each: few enough partials that the second pass (every workgroup adds them
    // up again, one after the other) stays short, enough workgroups that the moment pass is not two workgroups walking
    // 256 rows each (75 us on a 440-node batch); with 32-row chunks throughout, a 3200-node batch had 100 partials and
    // the second pass went from 10 to 25 us
    int64_t rpb = (n + 15) / 16;
    if (rpb < 32) rpb = 32;
    int64_t blocks = (n + rpb - 1) / rpb;
    if (blocks > kBnBlocksMax) {
        rpb = (n + kBnBlocksMax - 1) / kBnBlocksMax;
        blocks = (n + rpb - 1) / rpb;
    }
    if (blocks < 1) blocks = 1;
    *rows_per_block = rpb;
    return (int)blocks;
}

// pre_parts > 0: `part` already holds that many [H][2] partial rows of sum / sum of squares over x's rows (left by the
// fused kernel of the half-step that produced x): no moment pass
int launch_bn_normalize(const GnfFlow* flow, const GnfBatchNorm* bn, float* x, int64_t ld, int64_t n, int32_t H,
                        double* part, double* logdet_slot, hipStream_t st, int pre_parts) {
    if (n == 0) return GNF_OK;
    int64_t rpb;
    int blocks = pre_parts > 0 ? pre_parts : bn_blocks(n, &rpb);
    if (pre_parts <= 0) {
        hipLaunchKernelGGL(k_bn_stats, dim3(blocks), dim3(256), 0, st, x, ld, n, H, rpb, part);
        GNF_LAUNCH_CHECK("k_bn_stats");
    }
    const double* n_moments = nullptr;
    if (flow->bn_allreduce) {  // the moments of the whole batch, not of this rank's shard
        const int rc = bn_sync_exchange(flow, part, blocks, n, H, nullptr, st);
        if (rc) return rc;
        part = flow->bn_sync_buf;
        blocks = 1;
        n_moments = flow->bn_sync_buf + 2 * (int64_t)H;
    }
    const int64_t arows = kBnRows;
    const int64_t ablocks = (n + arows - 1) / arows;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)ablocks), dim3(256), 2 * H * sizeof(float), st, x, ld, n, H, arows, part,
                       blocks, bn->gamma, bn->beta, bn->epsilon, bn->batch_mean, bn->batch_variance, logdet_slot,
                       n_moments);
    GNF_LAUNCH_CHECK("k_bn_apply");
    return GNF_OK;
}

// x[:, :H] = x * scale + shift (scale_shift = [2][H] as the fused kernel's attention instance leaves them): the one
// bijector of a flow with batch norm on load whose conditioning half no later half-step rewrites
__global__ __launch_bounds__(256) void k_bn_affine(float* __restrict__ x, int64_t ld, int64_t n, int H,
                                                   const float* __restrict__ ss) {
    const int64_t total = n * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / H;
        const int c = (int)(i - r * H);
        float* p = x + r * ld + c;
        *p = *p * ss[c] + ss[H + c];
    }
}
int launch_bn_affine(float* x, int64_t ld, int64_t n, int32_t H, const float* scale_shift, hipStream_t st) {
    if (n == 0) return GNF_OK;
    int64_t blocks = (n * H + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_bn_affine, dim3((unsigned)blocks), dim3(256), 0, st, x, ld, n, H, scale_shift);
    GNF_LAUNCH_CHECK("k_bn_affine");
    return GNF_OK;
}
int launch_bn_stats(const float* x, int64_t ld, int64_t n, int32_t H, double* part, hipStream_t st, int* rows_out) {
    int64_t rpb;
    const int blocks = bn_blocks(n, &rpb);
    hipLaunchKernelGGL(k_bn_stats, dim3(blocks), dim3(256), 0, st, x, ld, n, H, rpb, part);
    GNF_LAUNCH_CHECK("k_bn_stats");
    *rows_out = blocks;
    return GNF_OK;
}

int launch_bn_denormalize(const GnfBatchNorm* bn, float* z, int64_t ld, int64_t n, int32_t H, hipStream_t st) {
    if (n == 0) return GNF_OK;
    int64_t blocks = (n * H + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_bn_denorm, dim3((unsigned)blocks), dim3(256), 0, st, z, ld, n, H, bn->gamma, bn->beta,
                       bn->moving_mean, bn->moving_variance, bn->epsilon);
    GNF_LAUNCH_CHECK("k_bn_denorm");
    return GNF_OK;
}

int validate_bn(const GnfBatchNorm* bn, int direction, const char* what, int q) {
    if (!bn->gamma || !bn->beta || (direction == GNF_INVERSE && (!bn->moving_mean || !bn->moving_variance))) {
        set_error("%s: batch-norm %d has null gamma / beta%s", what, q,
                  direction == GNF_INVERSE ? " / moving statistics" : "");
        return GNF_EINVAL;
    }
    if (!(bn->epsilon > 0.f)) {
        set_error("%s: batch-norm %d epsilon must be > 0", what, q);
        return GNF_EINVAL;
    }
    return GNF_OK;
}

}  // namespace gnf

// The batch-norm bijector backwards (training step, SURVEY.md 8f #2 + #4): what tf.gradients does to
// tfb.BatchNormalization.inverse + its inverse_log_det_jacobian inside total_loss (gnn.py:260-263, 310-313, 325-328;
// run_grevnet.py:361-362).  Forward: gnf_bn.hip.  Called from the reversible walk of gnf_grevnet_backward_f32
// (gnf_train.hip).  Split out of gnf_train.hip in round 3 (one concern per translation unit).
#include "gnf_common.h"

namespace gnf {

// ---- batch-norm bijector, backwards (forward: gnf_bn.hip) ------------------------------------------------
// y = (x - mu) / sigma * gamma + beta with the batch moments mu, var (sigma = sqrt(var + eps)) and the
// log-det term N * sum_f(log gamma_f - 0.5 log(var_f + eps)) inside L = -(log_prob_zs + logdet).  tf.gradients
// differentiates THROUGH the moments (they are functions of x).  With xh = (y - beta) / gamma, Gy = dL/dy:
//   dbeta = sum_n Gy          dgamma = sum_n Gy xh - N / gamma
//   dx    = [ gamma Gy - mean_n(gamma Gy) - xh mean_n(gamma Gy xh) + xh ] / sigma        (+xh: from 0.5 N log(var + eps))
//   x     = xh sigma + mu                                                              (the state, rebuilt)
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* __restrict__ y, int64_t ld,
                                                      const float* __restrict__ gy, int64_t ldg, int64_t n, int H,
                                                      int64_t rows_per_block, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, double* __restrict__ part) {
    __shared__ double sh[2][256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const int tid = threadIdx.x;
    for (int c0 = 0; c0 < H; c0 += 256) {
        const int w = H - c0 < 256 ? H - c0 : 256;
        const int lanes = 256 / w;
        const int c = tid % w, rs = tid / w;
        double s = 0.0, q = 0.0;
        if (rs < lanes) {
            const float ig = 1.f / gamma[c0 + c], b = beta[c0 + c];
            for (int64_t r = r0 + rs; r < r1; r += lanes) {
                const double gv = (double)gy[r * ldg + c0 + c];
                const double xh = (double)((y[r * ld + c0 + c] - b) * ig);
                s += gv;
                q += gv * xh;
            }
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

__global__ __launch_bounds__(256) void k_bn_bwd_apply(float* __restrict__ y, int64_t ld, float* __restrict__ gy,
                                                      int64_t ldg, int64_t n, int H, int64_t rows_per_block,
                                                      const double* __restrict__ part, int nparts,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bmean, const float* __restrict__ bvar,
                                                      float eps, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      int accumulate, const double* __restrict__ gsum,
                                                      const double* __restrict__ n_moments) {
    extern __shared__ float ss[];  // per column: m1 = mean(gamma Gy), m2 = mean(gamma Gy xh), gamma, beta, sigma, mu
    float* m1 = ss;
    float* m2 = ss + H;
    float* sg = ss + 2 * H;
    float* sb = ss + 3 * H;
    float* ssig = ss + 4 * H;
    float* smu = ss + 5 * H;
    const int tid = threadIdx.x;
    // many partial rows (one per workgroup of the kernel that left them): the 256 threads share the walk, thread (g, c)
    // sums rows g, g + G, ..; the G group sums are then added in order (as k_bn_apply does, gnf_bn.hip)
    __shared__ double gsum2[512];
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
        if (g < G) gsum2[(g * H + c) * 2 + 0] = s, gsum2[(g * H + c) * 2 + 1] = q;
        __syncthreads();
    }
    for (int c = tid; c < H; c += 256) {
        double s = 0.0, q = 0.0;
        if (G > 1) {
            for (int g = 0; g < G; ++g) s += gsum2[(g * H + c) * 2 + 0], q += gsum2[(g * H + c) * 2 + 1];
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
        const float g = gamma[c];
        // the means in dx are over the whole batch (cross-rank sums when gsum != NULL); d gamma / d beta below are
        // THIS rank's share - the gradient all-reduce adds the shares up
        const double nm = n_moments ? *n_moments : (double)n;
        const double sg_ = gsum ? gsum[2 * c + 0] : s, qg_ = gsum ? gsum[2 * c + 1] : q;
        m1[c] = (float)(sg_ / nm) * g;
        m2[c] = (float)(qg_ / nm) * g;
        sg[c] = g;
        sb[c] = beta[c];
        ssig[c] = sqrtf(bvar[c] + eps);
        smu[c] = bmean[c];
        if (blockIdx.x == 0) {
            const float db = (float)s;
            const float dg = (float)(q - (double)n / (double)g);
            dbeta[c] = accumulate ? dbeta[c] + db : db;
            dgamma[c] = accumulate ? dgamma[c] + dg : dg;
        }
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const int64_t tot = (r1 - r0) * H;
    for (int64_t i = tid; i < tot; i += 256) {
        const int64_t r = r0 + i / H;
        const int c = (int)(i % H);
        float* py = y + r * ld + c;
        float* pg = gy + r * ldg + c;
        const float xh = (*py - sb[c]) / sg[c];
        const float sig = ssig[c];
        *pg = (sg[c] * *pg - m1[c] - xh * m2[c] + xh) / sig;
        *py = xh * sig + smu[c];
    }
}

// pre_parts > 0: `part` already holds that many [H][2] partial rows (sum G, sum G x^) left by the kernel that wrote the
// final gy rows (k_attn_bwd_dx): no moment pass
int launch_bn_backward(const GnfFlow* flow, const GnfBatchNorm* bn, const GnfBatchNorm* gbn, float* y, int64_t ld,
                              float* gy, int64_t ldg, int64_t n, int32_t H, double* part, hipStream_t st, int pre_parts) {
    int64_t rpb = (n + 15) / 16;  // about sixteen chunks of at least 32 rows (see bn_blocks in gnf_bn.hip)
    if (rpb < 32) rpb = 32;
    int64_t blocks = (n + rpb - 1) / rpb;
    if (blocks > kBnBlocksMax) {
        rpb = (n + kBnBlocksMax - 1) / kBnBlocksMax;
        blocks = (n + rpb - 1) / rpb;
    }
    if (pre_parts > 0) {
        blocks = pre_parts;
    } else {
        hipLaunchKernelGGL(k_bn_bwd_stats, dim3((unsigned)blocks), dim3(256), 0, st, y, ld, gy, ldg, n, H, rpb, bn->gamma,
                           bn->beta, part);
        GNF_LAUNCH_CHECK("k_bn_bwd_stats");
    }
    const double *gsum = nullptr, *n_moments = nullptr;
    if (flow->bn_allreduce) {  // sum G, sum G x^ over the whole batch; this rank's own sums stay behind the partials
        double* local = part + (size_t)kBnPartRowsMax * H * 2;
        const int rc = bn_sync_exchange(flow, part, (int)blocks, n, H, local, st);
        if (rc) return rc;
        part = local;
        blocks = 1;
        gsum = flow->bn_sync_buf;
        n_moments = flow->bn_sync_buf + 2 * (int64_t)H;
    }
    const int64_t arows = 16;
    const int64_t ablocks = (n + arows - 1) / arows;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)ablocks), dim3(256), 6 * H * sizeof(float), st, y, ld, gy, ldg, n,
                       H, arows, part, (int)blocks, bn->gamma, bn->beta, bn->batch_mean, bn->batch_variance, bn->epsilon,
                       const_cast<float*>(gbn->gamma), const_cast<float*>(gbn->beta), 0, gsum, n_moments);
    GNF_LAUNCH_CHECK("k_bn_bwd_apply");
    return GNF_OK;
}

}  // namespace gnf

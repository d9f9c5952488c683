// extern "C" entry points of libgnf_hip.so (declared in include/gnf.h): host-side validation,
// workspace planning and dispatch onto the gfx950 kernels.  No torch types, no allocation, no
// host synchronisation; every launch goes to the caller's stream.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "gnf_common.h"

namespace gnf {

static thread_local char g_err[512] = "";

std::atomic<int64_t> g_options[OPT_COUNT];  // zero-initialised: every option "auto"

static const char* const kOptionNames[OPT_COUNT] = {
    "force_shape", "attn_kernel", "attn_bwd_rows", "bwd_generic", "dw_grouped", "dw_wide_units", "dw_thin_on_dw"};

static int option_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, kOptionNames[i])) return i;
    return -1;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int validate_mlp(const GnfMlp* m, const char* what) {
    if (!m) {
        set_error("%s: null GnfMlp", what);
        return GNF_EINVAL;
    }
    if (m->num_layers < 1 || m->num_layers > GNF_MAX_LAYERS) {
        set_error("%s: num_layers=%d outside [1,%d]", what, m->num_layers, GNF_MAX_LAYERS);
        return GNF_ESHAPE;
    }
    for (int j = 0; j <= m->num_layers; ++j)
        if (m->dims[j] < 1) {
            set_error("%s: dims[%d]=%d must be >= 1", what, j, m->dims[j]);
            return GNF_ESHAPE;
        }
    for (int j = 0; j < m->num_layers; ++j)
        if (!m->W[j] || !m->b[j]) {
            set_error("%s: layer %d has a null W or b pointer", what, j);
            return GNF_EINVAL;
        }
    return GNF_OK;
}

static int validate_spec(const GnfGnnSpec* g) {
    if (!g) {
        set_error("null GnfGnnSpec");
        return GNF_EINVAL;
    }
    if (g->agg != GNF_AGG_SUM && g->agg != GNF_AGG_MEAN) {
        set_error("GnfGnnSpec.agg=%d is not GNF_AGG_SUM/GNF_AGG_MEAN", g->agg);
        return GNF_EINVAL;
    }
    if (g->combine != GNF_COMBINE_EPS && g->combine != GNF_COMBINE_CONCAT) {
        set_error("GnfGnnSpec.combine=%d is not GNF_COMBINE_EPS/GNF_COMBINE_CONCAT", g->combine);
        return GNF_EINVAL;
    }
    if (g->activation != GNF_ACT_RELU && g->activation != GNF_ACT_LEAKY_RELU) {
        set_error("GnfGnnSpec.activation=%d is not GNF_ACT_RELU/GNF_ACT_LEAKY_RELU", g->activation);
        return GNF_EINVAL;
    }
    return GNF_OK;
}

static int validate_csr(const GnfCsr* c) {
    if (!c) {
        set_error("null GnfCsr");
        return GNF_EINVAL;
    }
    if (c->n_nodes < 0 || c->n_edges < 0 || c->n_nodes > INT32_MAX || c->n_edges > INT32_MAX) {
        set_error("GnfCsr: n_nodes=%lld n_edges=%lld out of int32 range", (long long)c->n_nodes,
                  (long long)c->n_edges);
        return GNF_ESHAPE;
    }
    if (c->n_nodes > 0 && (!c->rowptr || (c->n_edges > 0 && !c->col))) {
        set_error("GnfCsr: null rowptr/col");
        return GNF_EINVAL;
    }
    return GNF_OK;
}

// s/t nets of one half-step must agree with H and the combine mode (gnn.py:107-126: the MLP input
// is [x|agg] (2H) or eps*x+agg (H); its output feeds exp(s) / +t on an H-wide half, gnn.py:323).
static int validate_pair(const GnfMlp* s, const GnfMlp* t, const GnfGnnSpec* g, int32_t H) {
    int rc = validate_mlp(s, "s_net");
    if (rc) return rc;
    rc = validate_mlp(t, "t_net");
    if (rc) return rc;
    if ((s->attn != nullptr) != (t->attn != nullptr)) {
        set_error("s_net and t_net must both have (or both lack) an attention front-end");
        return GNF_ESHAPE;
    }
    if (s->attn) {
        rc = validate_attn(s->attn, s, H, "s_net");
        if (rc) return rc;
        rc = validate_attn(t->attn, t, H, "t_net");
        if (rc) return rc;
        if (memcmp(s->attn, t->attn, 8 * sizeof(int32_t))) {  // one make_gnn_fn builds both nets (gnn.py:288-296)
            set_error("s_net and t_net attention front-ends must have identical hyper-parameters");
            return GNF_ESHAPE;
        }
    }
    if (s->num_layers != t->num_layers || memcmp(s->dims, t->dims, sizeof(int32_t) * (s->num_layers + 1))) {
        // one make_gnn_fn builds both nets of a coupling (gnn.py:288-296): identical layer widths
        set_error("s_net and t_net must have identical layer widths");
        return GNF_ESHAPE;
    }
    const int in0 = s->attn ? s->dims[0] : ((g->combine == GNF_COMBINE_CONCAT) ? 2 * H : H);
    const GnfMlp* nets[2] = {s, t};
    for (int q = 0; q < 2; ++q) {
        const GnfMlp* m = nets[q];
        if (m->dims[0] != in0 || m->dims[m->num_layers] != H) {
            set_error("%s: MLP maps %d -> %d but the coupling needs %d -> %d (H=%d, combine=%d)",
                      q ? "t_net" : "s_net", m->dims[0], m->dims[m->num_layers], in0, H, H, g->combine);
            return GNF_ESHAPE;
        }
    }
    return GNF_OK;
}

WorkspacePlan plan_workspace(int64_t n_nodes, int32_t H, const GnfMlp* net, int32_t combine,
                             int64_t n_halfsteps) {
    WorkspacePlan p;
    p.partial_stride = partials_per_halfstep(n_nodes);
    p.n_halfsteps = n_halfsteps;
    p.bn_offset = (size_t)(n_halfsteps * p.partial_stride + kMaxGaussBlocks);
    p.bn_offset2 = p.bn_offset + (size_t)kBnPartRowsMax * (size_t)H * 2;
    p.bn_const_offset = p.bn_offset2 + (size_t)kBnPartRowsMax * (size_t)H * 2;
    size_t pb = (p.bn_const_offset + ((size_t)(n_halfsteps > 0 ? n_halfsteps : 1) * 2 * (size_t)H + 1) / 2) * sizeof(double);
    p.partial_bytes = (pb + 255) / 256 * 256;
    int lmax = 1;
    if (net)
        for (int j = 1; j < net->num_layers; ++j) lmax = lmax > net->dims[j] ? lmax : net->dims[j];
    const int in0 = net ? net->dims[0] : ((combine == GNF_COMBINE_CONCAT) ? 2 * H : H);
    p.base_floats = (size_t)n_nodes * (size_t)(in0 + kLayeredActBufs * lmax + 2 * H);
    p.attn_pack_offset = p.base_floats + attn_scratch_floats(net ? net->attn : nullptr, n_nodes, in0);
    p.attn_pack_per_net = net ? (attn_pack_floats(net->attn, H) + 63) / 64 * 64 : 0;
    // (every net of the call: at most 2 per half-step and kind; weight sharing needs fewer)
    p.attn_tile_offset = p.attn_pack_offset + p.attn_pack_per_net * (size_t)(4 * (n_halfsteps > 0 ? n_halfsteps : 1));
    p.scratch_floats = p.attn_tile_offset + (net && net->attn ? ((size_t)2 * ((n_nodes + 15) / 16) + 63) / 64 * 64 : 0);
    p.total_bytes = p.partial_bytes + p.scratch_floats * sizeof(float);
    return p;
}

static int run_half(const HalfStep& hs, float* scratch, hipStream_t st) {
    if (fused_supported(hs)) return launch_half_fused(hs, scratch, st);
    return launch_half_layered(hs, scratch, st);
}

static const GnfMlp* pick(const GnfFlow* f, const GnfMlp* nets, int half, int i) {
    return f->weight_sharing ? &nets[half] : &nets[half * f->num_timesteps + i];
}

// Argument checks shared by the whole-flow entry points (forward / inverse / backward).
int validate_flow_call(const GnfCsr* csr, const GnfFlow* flow, int64_t ld, int32_t D, const char* what) {
    int rc = validate_csr(csr);
    if (rc) return rc;
    if (!flow || !flow->s_nets || !flow->t_nets) {
        set_error("%s: null flow / nets", what);
        return GNF_EINVAL;
    }
    rc = validate_spec(&flow->gnn);
    if (rc) return rc;
    if (flow->num_timesteps < 0) {
        set_error("%s: num_timesteps=%d", what, flow->num_timesteps);
        return GNF_ESHAPE;
    }
    if (D < 2 || (D & 1) || ld < D) {
        // tf.split(x, 2, axis=1) (gnn.py:306) requires an even feature width
        set_error("%s: D=%d must be even and >= 2, ld=%lld >= D", what, D, (long long)ld);
        return GNF_ESHAPE;
    }
    const int H = D / 2;
    const int T = flow->num_timesteps;
    const int n_nets = flow->weight_sharing ? 2 : 2 * T;
    for (int q = 0; q < n_nets; ++q) {
        rc = validate_pair(&flow->s_nets[q], &flow->t_nets[q], &flow->gnn, H);
        if (rc) return rc;
        // one make_gnn_fn builds every net (gnn.py:266-267): identical layer widths
        if (memcmp(flow->s_nets[q].dims, flow->s_nets[0].dims, sizeof(int32_t) * (GNF_MAX_LAYERS + 1)) ||
            flow->s_nets[q].num_layers != flow->s_nets[0].num_layers ||
            memcmp(flow->t_nets[q].dims, flow->s_nets[0].dims, sizeof(int32_t) * (GNF_MAX_LAYERS + 1)) ||
            flow->t_nets[q].num_layers != flow->s_nets[0].num_layers) {
            set_error("%s: net %d has different layer widths than net 0", what, q);
            return GNF_ESHAPE;
        }
    }
    return GNF_OK;
}

}  // namespace gnf

using namespace gnf;

extern "C" {

int gnf_abi_version(void) { return GNF_ABI_VERSION; }

int gnf_set_option(const char* name, int64_t value) {
    const int i = option_index(name);
    if (i < 0) {
        set_error("gnf_set_option: unknown option '%s'", name ? name : "(null)");
        return GNF_EINVAL;
    }
    g_options[i].store(value, std::memory_order_relaxed);
    return GNF_OK;
}

int64_t gnf_get_option(const char* name) {
    const int i = option_index(name);
    if (i < 0) {
        set_error("gnf_get_option: unknown option '%s'", name ? name : "(null)");
        return GNF_EINVAL;
    }
    return g_options[i].load(std::memory_order_relaxed);
}

size_t gnf_mlp_stash_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow) {
    if (n_nodes <= 0 || D < 2 || (D & 1) || !flow || !flow->s_nets || !flow->t_nets || flow->num_timesteps <= 0) return 0;
    if (!mlp_stash_supported(flow, n_nodes, D / 2)) return 0;
    return (size_t)2 * flow->num_timesteps * mlp_stash_layout(&flow->s_nets[0], n_nodes, D / 2).slot * sizeof(float);
}

size_t gnf_attn_stash_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow) {
    if (n_nodes <= 0 || D < 2 || !flow || !flow->s_nets || flow->num_timesteps <= 0) return 0;
    return (size_t)2 * flow->num_timesteps * attn_stash_slot_floats(flow, n_nodes) * sizeof(float);
}

const char* gnf_last_error(void) { return g_err; }

int64_t gnf_packed_floats(const GnfMlp* mlp) {
    if (!mlp || mlp->num_layers < 1 || mlp->num_layers > GNF_MAX_LAYERS) {
        set_error("gnf_packed_floats: bad GnfMlp");
        return GNF_EINVAL;
    }
    return packed_floats(mlp);
}

int gnf_pack_mlp(const GnfMlp* mlp, float* packed, gnf_stream_t stream) {
    int rc = validate_mlp(mlp, "gnf_pack_mlp");
    if (rc) return rc;
    if (!packed) {
        set_error("gnf_pack_mlp: null output buffer");
        return GNF_EINVAL;
    }
    return launch_pack_mlp(mlp, packed, (hipStream_t)stream);
}

int gnf_aggregate_f32(const GnfCsr* csr, const float* x, int64_t ldx, int32_t H, int32_t agg,
                      float* out, int64_t ldo, gnf_stream_t stream) {
    int rc = validate_csr(csr);
    if (rc) return rc;
    if (agg != GNF_AGG_SUM && agg != GNF_AGG_MEAN) {
        set_error("gnf_aggregate_f32: agg=%d", agg);
        return GNF_EINVAL;
    }
    if (H < 1 || ldx < H || ldo < H) {
        set_error("gnf_aggregate_f32: H=%d ldx=%lld ldo=%lld", H, (long long)ldx, (long long)ldo);
        return GNF_ESHAPE;
    }
    if (csr->n_nodes > 0 && (!x || !out)) {
        set_error("gnf_aggregate_f32: null x/out");
        return GNF_EINVAL;
    }
    return launch_aggregate(csr->rowptr, csr->col, csr->n_nodes, x, ldx, H, agg == GNF_AGG_MEAN, 2,
                            0.f, out, ldo, (hipStream_t)stream);
}

size_t gnf_gnn_workspace_bytes(int64_t n_nodes, int32_t H, const GnfMlp* mlp, int32_t combine) {
    if (n_nodes < 0 || H < 1 || !mlp) return 0;
    return plan_workspace(n_nodes, H, mlp, combine, 0).scratch_floats * sizeof(float);
}

int gnf_gnn_apply_f32(const GnfCsr* csr, const GnfMlp* mlp, const GnfGnnSpec* gnn, const float* x,
                      int64_t ldx, int32_t H, float* out, int64_t ldo, void* ws, size_t ws_bytes,
                      gnf_stream_t stream) {
    int rc = validate_csr(csr);
    if (rc) return rc;
    rc = validate_spec(gnn);
    if (rc) return rc;
    rc = validate_mlp(mlp, "gnf_gnn_apply_f32");
    if (rc) return rc;
    if (H < 1) {
        set_error("gnf_gnn_apply_f32: H=%d", H);
        return GNF_ESHAPE;
    }
    if (mlp->attn) {
        rc = validate_attn(mlp->attn, mlp, H, "gnf_gnn_apply_f32");
        if (rc) return rc;
    }
    const int in0 = mlp->attn ? mlp->dims[0] : ((gnn->combine == GNF_COMBINE_CONCAT) ? 2 * H : H);
    const int od = mlp->dims[mlp->num_layers];
    if (H < 1 || ldx < H || ldo < od || mlp->dims[0] != in0) {
        set_error("gnf_gnn_apply_f32: H=%d ldx=%lld ldo=%lld, MLP maps %d -> %d, needs input %d", H,
                  (long long)ldx, (long long)ldo, mlp->dims[0], od, in0);
        return GNF_ESHAPE;
    }
    if (csr->n_nodes == 0) return GNF_OK;
    if (!x || !out || !ws) {
        set_error("gnf_gnn_apply_f32: null x/out/ws");
        return GNF_EINVAL;
    }
    if (ws_bytes < gnf_gnn_workspace_bytes(csr->n_nodes, H, mlp, gnn->combine)) {
        set_error("gnf_gnn_apply_f32: workspace %zu < %zu bytes", ws_bytes,
                  gnf_gnn_workspace_bytes(csr->n_nodes, H, mlp, gnn->combine));
        return GNF_EWORKSPACE;
    }
    return launch_gnn_layered(csr->rowptr, csr->col, csr->n_nodes, x, ldx, H, *gnn, mlp, out, ldo,
                              (float*)ws, (hipStream_t)stream);
}

size_t gnf_workspace_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow) {
    if (n_nodes < 0 || D < 2 || !flow || !flow->s_nets) return 0;
    return plan_workspace(n_nodes, D / 2, &flow->s_nets[0], flow->gnn.combine,
                          2 * (int64_t)(flow->num_timesteps > 0 ? flow->num_timesteps : 1))
        .total_bytes;
}

int gnf_coupling_half_f32(const GnfCsr* csr, const GnfMlp* s_net, const GnfMlp* t_net,
                          const GnfGnnSpec* gnn, const float* x_cond, float* x_upd, int64_t ld,
                          int32_t H, int32_t direction, double* logdet_accum, void* ws,
                          size_t ws_bytes, gnf_stream_t stream) {
    int rc = validate_csr(csr);
    if (rc) return rc;
    rc = validate_spec(gnn);
    if (rc) return rc;
    if (H < 1 || ld < 2 * (int64_t)H) {
        set_error("gnf_coupling_half_f32: H=%d ld=%lld (need ld >= 2H)", H, (long long)ld);
        return GNF_ESHAPE;
    }
    rc = validate_pair(s_net, t_net, gnn, H);
    if (rc) return rc;
    if (direction != GNF_FORWARD && direction != GNF_INVERSE) {
        set_error("gnf_coupling_half_f32: direction=%d", direction);
        return GNF_EINVAL;
    }
    if (csr->n_nodes == 0) return GNF_OK;
    if (!x_cond || !x_upd || !ws) {
        set_error("gnf_coupling_half_f32: null x_cond/x_upd/ws");
        return GNF_EINVAL;
    }
    const WorkspacePlan p = plan_workspace(csr->n_nodes, H, s_net, gnn->combine, 1);
    if (ws_bytes < p.total_bytes) {
        set_error("gnf_coupling_half_f32: workspace %zu < %zu bytes", ws_bytes, p.total_bytes);
        return GNF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t nparts = 0;
    HalfStep hs{csr->rowptr, csr->col, csr->n_nodes, x_cond, x_upd, ld, H, direction, *gnn,
                s_net, t_net, (double*)ws, &nparts, nullptr, csr->n_edges};
    rc = run_half(hs, (float*)((char*)ws + p.partial_bytes), st);
    if (rc) return rc;
    if (logdet_accum)
        return launch_finalize((const double*)ws, nparts, nullptr, 0, logdet_accum, 1, 0, st);
    return GNF_OK;
}

int gnf_grevnet_f32(const GnfCsr* csr, const GnfFlow* flow, float* x, int64_t ld, int32_t D,
                    int32_t direction, double* sums, void* ws, size_t ws_bytes, gnf_stream_t stream) {
    return gnf_grevnet_from_f32(csr, flow, nullptr, 0, x, ld, D, direction, sums, ws, ws_bytes, stream);
}

int gnf_grevnet_from_f32(const GnfCsr* csr, const GnfFlow* flow, const float* x_src, int64_t ld_src, float* x, int64_t ld,
                         int32_t D, int32_t direction, double* sums, void* ws, size_t ws_bytes, gnf_stream_t stream) {
    int rc = validate_flow_call(csr, flow, ld, D, "gnf_grevnet_f32");
    if (rc) return rc;
    if (x_src == x) x_src = nullptr;
    if (x_src && ld_src < D) {
        set_error("gnf_grevnet_from_f32: ld_src=%lld < D=%d", (long long)ld_src, D);
        return GNF_ESHAPE;
    }
    if (direction != GNF_FORWARD && direction != GNF_INVERSE) {
        set_error("gnf_grevnet_f32: direction=%d", direction);
        return GNF_EINVAL;
    }
    const int H = D / 2;
    const int T = flow->num_timesteps;
    const int n_nets = flow->weight_sharing ? 2 : 2 * T;
    if (direction == GNF_FORWARD && !sums) {
        set_error("gnf_grevnet_f32: FORWARD needs a device sums[2] buffer");
        return GNF_EINVAL;
    }
    if (flow->bns)
        for (int q = 0; q < 2 * T; ++q) {
            rc = validate_bn(&flow->bns[q], direction, "gnf_grevnet_f32", q);
            if (rc) return rc;
        }
    const int64_t n = csr->n_nodes;
    if (n > 0 && (!x || !ws)) {
        set_error("gnf_grevnet_f32: null x/ws");
        return GNF_EINVAL;
    }
    const WorkspacePlan p = plan_workspace(n, H, n_nets ? &flow->s_nets[0] : nullptr,
                                           flow->gnn.combine, 2 * (int64_t)(T > 0 ? T : 1));
    if (n > 0 && ws_bytes < p.total_bytes) {
        set_error("gnf_grevnet_f32: workspace %zu < %zu bytes", ws_bytes, p.total_bytes);
        return GNF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double* partials = (double*)ws;
    float* scratch = (float*)((char*)ws + p.partial_bytes);
    float* half0 = x;       // columns [0, H)
    float* half1 = x + H;   // columns [H, D)
    int64_t used = 0;       // partial slots written so far (packed densely, fixed order)
    // Out of place (x_src -> x): the first half-step of the walk reads the source buffer and writes the destination
    // (its conditioning rows copied on the way) when the fused both-nets kernel runs it; otherwise one copy pass
    // up front and the in-place walk.  A batch-norm bijector in front of the first half-step rewrites the
    // conditioning half in place, so such flows copy first too.
    bool first_oop = false;
    if (x_src && n > 0) {
        if (T > 0 && ld_src == ld && !(direction == GNF_FORWARD && flow->bns)) {
            const int h0_ = direction == GNF_FORWARD ? 0 : 1, i0_ = direction == GNF_FORWARD ? 0 : T - 1;
            HalfStep probe{csr->rowptr, csr->col, n, x, x, ld, H, direction, flow->gnn,
                           pick(flow, flow->s_nets, h0_, i0_), pick(flow, flow->t_nets, h0_, i0_), nullptr, nullptr,
                           nullptr, csr->n_edges};
            first_oop = fused_supports_oop(probe);
        }
        if (!first_oop) {
            rc = launch_copy_rows(x_src, ld_src, x, ld, n, D, st);
            if (rc) return rc;
        }
    }
    // attention nets on a sparse batch: the weights of every net go into fragment order ONCE per call (one small launch)
    // and the one-launch front-end of gnf_attn_front.hip runs per half-step (the attn_kernel option keeps the two-launch
    // kernels)
    const float* attn_pack = nullptr;
    const int32_t* attn_tiles = nullptr;
    if (n > 0 && n_nets > 0 && flow->s_nets[0].attn && csr->n_edges > 0 && csr->n_edges < 24 * n &&
        !opt(OPT_ATTN_KERNEL) && attn_front_fused_ok(flow->s_nets[0].attn, H)) {
        const GnfAttn* all[2 * 64];
        if (2 * n_nets <= 128) {
            for (int q = 0; q < n_nets; ++q) {
                all[q] = flow->s_nets[q].attn;
                all[n_nets + q] = flow->t_nets[q].attn;
            }
            rc = launch_attn_pack(all, 2 * n_nets, H, scratch + p.attn_pack_offset, st);
            if (rc) return rc;
            attn_pack = scratch + p.attn_pack_offset;
            // ... and every tile's sender window, once for the 2 T half-steps of the call
            rc = launch_attn_tiles(csr->rowptr, csr->col, n, reinterpret_cast<int32_t*>(scratch + p.attn_tile_offset), st);
            if (rc) return rc;
            attn_tiles = reinterpret_cast<const int32_t*>(scratch + p.attn_tile_offset);
        }
    }
    auto mark_attn = [&](HalfStep& hs, int half, int i) {
        if (!attn_pack) return;
        const int q = flow->weight_sharing ? half : half * T + i;
        hs.attn_packed[0] = attn_pack + (size_t)q * p.attn_pack_per_net;
        hs.attn_packed[1] = attn_pack + (size_t)(n_nets + q) * p.attn_pack_per_net;
        hs.attn_tiles = attn_tiles;
    };
    // split row tiles of the large-batch kernel (message-passing nets, batches of more than two row tiles per CU): their
    // flags are zeroed once per call, every half-step launch gets a value of its own
    int32_t split_epoch = 0;
    if (n_nets > 0 && !flow->s_nets[0].attn && n > (int64_t)32 * big_cu_count()) {
        const int in0_ = flow->s_nets[0].dims[0];
        int lmax_ = 1;
        for (int j = 1; j < flow->s_nets[0].num_layers; ++j) lmax_ = lmax_ > flow->s_nets[0].dims[j] ? lmax_ : flow->s_nets[0].dims[j];
        if (big_split_offset(n, in0_) + big_split_floats() <= (size_t)n * (size_t)(in0_ + kLayeredActBufs * lmax_ + 2 * H)) {
            GNF_HIP_TRY(hipMemsetAsync(scratch + big_split_offset(n, in0_), 0, kBigSplitMax * sizeof(int), st));
            split_epoch = 1;
        }
    }
    bool first = true;
    auto mark_first = [&](HalfStep& hs, int half) {
        if (!first) return;
        first = false;
        if (first_oop) {
            const int co = half == 0 ? 0 : H, uo = half == 0 ? H : 0;
            hs.x_cond = x_src + co;
            hs.x_upd_src = x_src + uo;
            hs.cond_copy = x + co;
        }
    };
    // the forward pass leaves every half-step's attention front-end in the caller's stash for the backward pass
    const size_t stash_slot = attn_stash_slot_floats(flow, n);
    float* stash = nullptr;
    if (direction == GNF_FORWARD && flow->attn_stash && stash_slot > 0) {
        if (flow->attn_stash_bytes < (size_t)2 * T * stash_slot * sizeof(float)) {
            set_error("gnf_grevnet_f32: attn_stash %zu < %zu bytes", flow->attn_stash_bytes,
                      (size_t)2 * T * stash_slot * sizeof(float));
            return GNF_EWORKSPACE;
        }
        stash = flow->attn_stash;
    }
    // ... and, for message-passing nets on small batches, every row its MLP kernels would recompute (ABI v8)
    float* mstash = nullptr;
    size_t mstash_slot = 0;
    if (direction == GNF_FORWARD && flow->mlp_stash && n > 0 && mlp_stash_supported(flow, n, H)) {
        mstash_slot = mlp_stash_layout(&flow->s_nets[0], n, H).slot;
        if (flow->mlp_stash_bytes < (size_t)2 * T * mstash_slot * sizeof(float)) {
            set_error("gnf_grevnet_f32: mlp_stash %zu < %zu bytes", flow->mlp_stash_bytes,
                      (size_t)2 * T * mstash_slot * sizeof(float));
            return GNF_EWORKSPACE;
        }
        mstash = flow->mlp_stash;
    }

    // sum(z^2): per-workgroup partials out of the coupling epilogues of the last two half-steps instead of a k_gauss
    // launch over z (without batch norm - a bijector in front of the very last half-step still changes the other half -
    // and while two launches' partials fit the slots; paths that do not write them leave nsq at 0)
    double* const gpart = partials + 2 * (int64_t)(T > 0 ? T : 1) * p.partial_stride;
    int32_t nsq[2] = {0, 0};
    int32_t bn_pre = 0;  // partial rows the previous half-step's kernel left for the next bijector's moments (0: none)
    bool bn_on_load = false;  // the bijectors are applied by the half-step kernels themselves (decided at the walk's first half-step)
    const GnfBatchNorm* inv_pending = nullptr;  // inverse pass: the bijector the next half-step of the walk applies on load
    const bool sq_ok = direction == GNF_FORWARD && !flow->bns && T > 0 && 2 * ((n + 15) / 16) <= kMaxGaussBlocks;
    if (n > 0) {
        if (direction == GNF_FORWARD) {
            for (int i = 0; i < T; ++i) {  // gnn.py:309-338
                for (int half = 0; half < 2; ++half) {
                    // gnn.py:310-313, 325-328: the bijector normalises the conditioning half first - as a pass of its own, or
                    // (attention nets through the fused kernel's attention instance, fused_bn_on_load_ok) where that kernel
                    // reads the rows, the rows in memory staying raw until the NEXT half-step's kernel rewrites them
                    double* bn_slot = nullptr;
                    if (flow->bns) bn_slot = partials + used, used += 1;
                    int32_t np_ = 0;
                    HalfStep hs{csr->rowptr, csr->col, n, half == 0 ? half0 : half1,
                                half == 0 ? half1 : half0, ld, H, GNF_FORWARD, flow->gnn,
                                pick(flow, flow->s_nets, half, i), pick(flow, flow->t_nets, half, i),
                                partials + used, &np_,
                                stash ? stash + (size_t)(2 * i + half) * stash_slot : nullptr, csr->n_edges};
                    mark_first(hs, half);
                    mark_attn(hs, half, i);
                    if (split_epoch) hs.split_epoch = split_epoch++;
                    if (mstash) hs.mlp_stash = mstash + (size_t)(2 * i + half) * mstash_slot;
                    const int idx = 2 * i + half;
                    if (flow->bns && idx == 0) {
                        // decided once, for every half-step of the walk: ALL net pairs must fit the attention instance (a
                        // C-ABI caller's nets may differ from pair to pair when weight_sharing = 0), else the separate passes
                        bn_on_load = !flow->bn_allreduce;
                        for (int i2 = 0; i2 < T && bn_on_load; ++i2)
                            for (int h2 = 0; h2 < 2 && bn_on_load; ++h2) {
                                HalfStep probe = hs;
                                probe.s_net = pick(flow, flow->s_nets, h2, i2), probe.t_net = pick(flow, flow->t_nets, h2, i2);
                                mark_attn(probe, h2, i2);
                                bn_on_load = fused_bn_on_load_ok(probe);
                            }
                    }
                    double* const bn_rows_in = partials + ((idx & 1) && bn_on_load ? p.bn_offset2 : p.bn_offset);
                    double* const bn_rows_out = partials + (!(idx & 1) && bn_on_load ? p.bn_offset2 : p.bn_offset);
                    if (flow->bns && !bn_on_load) {
                        rc = launch_bn_normalize(flow, &flow->bns[half * T + i], half == 0 ? half0 : half1, ld, n, H,
                                                 partials + p.bn_offset, bn_slot, st, bn_pre);
                        if (rc) return rc;
                        bn_pre = 0;
                    } else if (flow->bns) {
                        int rows = bn_pre;
                        if (rows == 0) {  // (the first bijector, or a predecessor that left no column sums)
                            rc = launch_bn_stats(half == 0 ? half0 : half1, ld, n, H, bn_rows_in, st, &rows);
                            if (rc) return rc;
                        }
                        float* consts = reinterpret_cast<float*>(partials + p.bn_const_offset);
                        hs.bnc = &flow->bns[half * T + i];
                        hs.bnc_part = bn_rows_in, hs.bnc_nparts = rows;
                        hs.bnc_logdet = bn_slot;
                        hs.bnc_const = consts + (size_t)idx * 2 * H;
                        hs.bnu_const = idx > 0 ? consts + (size_t)(idx - 1) * 2 * H : nullptr;
                        bn_pre = 0;
                    }
                    // the half this half-step updates is what the NEXT bijector normalises: its column sums ride along
                    if (flow->bns && !(i == T - 1 && half == 1) && (n + 15) / 16 <= kBnPartRowsMax) {
                        hs.bn_part = bn_rows_out;
                        hs.n_bn = &bn_pre;
                    }
                    if (sq_ok && i == T - 1) {  // the outputs of the flow's last two half-steps are z: sum(z^2) rides along
                        hs.sq_partials = gpart + (half == 0 ? 0 : nsq[0]);
                        hs.n_sq = &nsq[half];
                    }
                    rc = run_half(hs, scratch, st);
                    if (rc) return rc;
                    used += np_;
                }
            }
            if (bn_on_load && T > 0) {  // the last bijector's conditioning half (columns [H, D)) is still raw in memory
                rc = launch_bn_affine(half1, ld, n, H, reinterpret_cast<const float*>(partials + p.bn_const_offset) + (size_t)(2 * T - 1) * 2 * H, st);
                if (rc) return rc;
            }
        } else {
            for (int i = T - 1; i >= 0; --i) {  // gnn.py:347-372
                for (int half = 1; half >= 0; --half) {
                    int32_t np_ = 0;
                    HalfStep hs{csr->rowptr, csr->col, n, half == 0 ? half0 : half1,
                                half == 0 ? half1 : half0, ld, H, GNF_INVERSE, flow->gnn,
                                pick(flow, flow->s_nets, half, i), pick(flow, flow->t_nets, half, i),
                                partials + used, &np_, nullptr, csr->n_edges};
                    mark_first(hs, half);
                    mark_attn(hs, half, i);
                    if (split_epoch) hs.split_epoch = split_epoch++;
                    // gnn.py:356-358, 369-371: bn.forward (moving statistics) on the conditioning half AFTER the half-step - a
                    // pass of its own, or (fused attention instance) left for the next half-step of the walk, which rewrites
                    // exactly that half, to apply where it reads the old value
                    if (flow->bns && i == T - 1 && half == 1) bn_on_load = fused_bn_on_load_ok(hs);
                    if (bn_on_load) hs.bnu_inv = inv_pending;
                    rc = run_half(hs, scratch, st);
                    if (rc) return rc;
                    // partial slots are reused: the inverse pass has no log-det (gnn.py:343-373)
                    if (flow->bns && bn_on_load) {
                        inv_pending = &flow->bns[half * T + i];
                    } else if (flow->bns) {
                        rc = launch_bn_denormalize(&flow->bns[half * T + i], half == 0 ? half0 : half1, ld, n, H, st);
                        if (rc) return rc;
                    }
                }
            }
            if (inv_pending) {  // the walk's last half-step (i = 0, half = 0): its conditioning half is columns [0, H)
                rc = launch_bn_denormalize(inv_pending, half0, ld, n, H, st);
                if (rc) return rc;
            }
        }
    }
    if (direction == GNF_FORWARD) {
        // (measured and dropped, tools/ab_options.py: k_gauss + k_finalize merged into one launch whose last-arriving
        // workgroup runs the final reduction - the release / ticket / acquire hand-off costs 1 us more than the launch
        // boundary it removes, CHANGELOG.md section 4.4)
        int32_t ng = 0;
        if (nsq[0] > 0 && nsq[1] > 0) {
            ng = nsq[0] + nsq[1];
        } else if (n > 0) {
            rc = launch_gauss_partials(x, n, D, ld, gpart, &ng, st);
            if (rc) return rc;
        }
        return launch_finalize(partials, used, gpart, ng, sums, 0, 1, st);
    }
    return GNF_OK;
}

int gnf_gauss_sumsq_f32(const float* z, int64_t n_nodes, int32_t D, int64_t ld, double* out,
                        void* ws, size_t ws_bytes, gnf_stream_t stream) {
    if (n_nodes < 0 || D < 1 || ld < D) {
        set_error("gnf_gauss_sumsq_f32: N=%lld D=%d ld=%lld", (long long)n_nodes, D, (long long)ld);
        return GNF_ESHAPE;
    }
    if (!out || !ws || (n_nodes > 0 && !z)) {
        set_error("gnf_gauss_sumsq_f32: null z/out/ws");
        return GNF_EINVAL;
    }
    if (ws_bytes < kMaxGaussBlocks * sizeof(double)) {
        set_error("gnf_gauss_sumsq_f32: workspace %zu < %zu bytes", ws_bytes,
                  kMaxGaussBlocks * sizeof(double));
        return GNF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t ng = 0;
    if (n_nodes > 0) {
        int rc = launch_gauss_partials(z, n_nodes, D, ld, (double*)ws, &ng, st);
        if (rc) return rc;
    }
    // a == b trick: write out[0] from the partials (no accumulate)
    return launch_finalize((const double*)ws, ng, nullptr, 0, out, 0, 0, st);
}

}  // extern "C"

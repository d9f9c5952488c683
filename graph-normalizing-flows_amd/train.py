"""Training step of the GRevNet drivers (/root/reference/run_grevnet.py:340-377, 440-447; same at
train_grevnet_with_data.py:356-395) on the MI355X kernels:

    grads_and_vars = optimizer.compute_gradients(total_loss)      -> gnf_grevnet_backward_f32 (reversible backprop)
    tf.clip_by_value / tf.clip_by_norm on every gradient           -> gnf_clip_by_value_f32 / gnf_clip_by_norm_f32
    tf.train.AdamOptimizer(lr, beta1, beta2, epsilon).apply_gradients -> gnf_adam_f32 + gnf_pack_flow

Every trainable variable of the flow lives in ONE flat fp32 device vector (`theta`; the MLPs' W / b are
views into it, in snt.Linear's own [in, out] layout), the gradient, and Adam's two moment vectors have
the same layout: the optimiser is one elementwise launch over the whole model, and the data-parallel
gradient exchange is one flat all-reduce (RCCL over xGMI), not one collective per variable.

The drivers' default learning-rate decay (tf.train.exponential_decay, run_grevnet.py:341-347) is one line of
`current_learning_rate`; any other schedule is the caller's business (`step(graph, learning_rate=...)`; the
--use_lr_schedule function lives in examples/driver_utils.py: out of this path's scope).

The trainer's state (what tf.train.Saver checkpoints for the drivers, run_grevnet.py:379,449-453: variables, Adam slots,
global_step, the bijectors' moving statistics) is `trainer_state` / `load_trainer_state` / `save_checkpoint` /
`load_checkpoint` below; bench.py restores it before every timed region of a training workload.
"""
import ctypes as C
import math

import torch

from . import _abi
from .gnn import _touch as _gnn_touch
from .flow import LN_2PI
from .graphs import csr_of


class _StepTerms(dict):
    """values_map of a training iteration (run_grevnet.py:290-302).  The two batch sums come out of the flow as device
    fp64 scalars; every derived scalar (log_prob_zs, total_loss, the *_per_node values ...) is a few 0-d torch
    operations that are only launched when somebody asks for the value (a training loop that logs every n-th step
    pays for them every n-th step, not ~10 small launches per iteration)."""
    _DERIVED = ("log_det_jacobian", "log_prob_zs", "log_prob_xs", "total_loss", "loss_per_node", "log_prob_xs_per_node",
                "log_prob_zs_per_node", "log_det_jacobian_per_node")

    def __init__(self, z_graph, reconstruction, sums, n, d):
        super().__init__(z_graph=z_graph, reconstruction=reconstruction, num_nodes=float(n), sums=sums)
        self._n, self._d = float(n), int(d)

    def __missing__(self, key):
        if key not in self._DERIVED:
            raise KeyError(key)
        sums, n = dict.__getitem__(self, "sums"), self._n
        logdet = sums[0]
        log_prob_zs = -0.5 * sums[1] - 0.5 * self._d * LN_2PI * n
        log_prob_xs = log_prob_zs + logdet
        vals = {"log_det_jacobian": logdet, "log_prob_zs": log_prob_zs, "log_prob_xs": log_prob_xs,
                "total_loss": -log_prob_xs, "loss_per_node": -log_prob_xs / n, "log_prob_xs_per_node": log_prob_xs / n,
                "log_prob_zs_per_node": log_prob_zs / n, "log_det_jacobian_per_node": logdet / n}
        return vals[key]   # not cached: `sums` may be rewritten in place (captured-graph replay)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._DERIVED

    def get(self, key, default=None):
        return self[key] if key in self else default


class GRevNetTrainer:
    """total_loss, its gradient and the Adam update for one GRevNet (message-passing GNNs with or without the
    batch-norm bijectors, and the edge-list attention GNN: every variable the reference's optimizer would see).

    Hyper-parameters default to the drivers' flags (run_grevnet.py:114-131): lr 1e-4, beta1 0.9,
    beta2 0.9, epsilon 1e-8, exponential lr decay (1000 steps, 0.96), no clipping."""

    def __init__(self, grevnet, lr=1e-4, adam_beta1=0.9, adam_beta2=0.9, adam_epsilon=1e-8, use_lr_decay=True,
                 lr_decay_steps=1000, lr_decay_rate=0.96, clip_gradient_by_value=False,
                 clip_gradient_value_lower=-1.0, clip_gradient_value_upper=5.0, clip_gradient_by_norm=False,
                 clip_gradient_norm=10.0):
        self.net = grevnet
        self.lr, self.beta1, self.beta2, self.epsilon = float(lr), float(adam_beta1), float(adam_beta2), float(adam_epsilon)
        self.use_lr_decay, self.lr_decay_steps, self.lr_decay_rate = bool(use_lr_decay), int(lr_decay_steps), float(lr_decay_rate)
        self.clip_by_value = bool(clip_gradient_by_value)
        self.clip_lo, self.clip_hi = float(clip_gradient_value_lower), float(clip_gradient_value_upper)
        self.clip_by_norm, self.clip_norm = bool(clip_gradient_by_norm), float(clip_gradient_norm)
        self.global_step = 0
        self.theta = self.grad = self.m = self.v = None
        self._grad_flow = None
        self._keep = None
        self._offsets = None
        self._ws = None
        self._bns = []
        self._attn_blocks = []
        self._clip_ws = None     # scratch of the two-pass clip_by_norm
        self._stash = None       # attention front-end stash (uint8 device buffer), see loss_and_grads
        self.stash_attention = True          # False: recompute the attention front-end in the backward walk
        self.stash_mlp_rows = True           # False: recompute the MLP rows too (the fully reversible walk)
        self._mlp_stash = None
        self._arena_versions = None   # version counters of every parameter container when the arena was (re)built
        self._arena_epoch = -1
        self._aux = None         # second HIP stream: the weight-gradient GEMMs overlap the backward walk
        self.overlap_weight_grads = True     # False: no auxiliary stream for the weight-gradient GEMMs
        # the MLP-row stash is memory for time (wide nets: 2T slots of every hidden activation - about 20 GB for 30 k nodes
        # at the data driver's defaults): it is taken only while it fits this budget; beyond it the walk recomputes.
        # None = at most `mlp_stash_free_fraction` of the memory the device has free when the stash is first needed.
        self.mlp_stash_max_bytes = None
        self.mlp_stash_free_fraction = 0.5
        self.mlp_stash_declined = None       # (bytes asked, reason) of the last stash that was NOT taken

    # ---- parameter arena ---------------------------------------------------------------------
    def _ensure_arena(self, hdim, device):
        net = self.net
        net._flow(hdim, device)                      # builds lazily-initialised MLPs (Sonnet-style first connect)
        from .gnn import _EPOCH
        if self.theta is not None and self.theta.device == torch.device(device) and self._arena_epoch == _EPOCH[0]:
            return                                   # no parameter container anywhere was touched since the arena was built
        mlps = net.mlps("s") + net.mlps("t")
        blocks = net.blocks("s") + net.blocks("t")
        attn_blocks = [b for b in blocks if getattr(b, "attn_params", None) is not None]
        bns = [b for half in net.bns for b in half] if net.use_batch_norm else []   # index half*T + i

        def versions():
            return (tuple(m.version for m in mlps), tuple(b.attn_version() for b in blocks), tuple(b.version for b in bns))
        if self.theta is not None and self.theta.device == torch.device(device):
            # the parameter containers were last (re)bound by this trainer: nothing to do.  Otherwise somebody called
            # set_params / set_attn_params (or moved the net) since: their tensors live OUTSIDE theta, so Adam would go on
            # updating a vector nothing reads - re-seat the arena on the new values (Adam moments kept when the
            # model size is unchanged)
            if self._arena_versions == versions():
                self._arena_epoch = _EPOCH[0]
                return
        old_m, old_v = self.m, self.v
        sizes = []
        for m in mlps:
            for (w, b) in m.params:
                sizes += [w.numel(), b.numel()]
        for blk in attn_blocks:                       # attention front-end: wq, wk, wv, wo (+ ln_gamma, ln_beta)
            sizes += [blk.attn_params[k].numel() for k in blk.attn_keys()]
        for b in bns:                                 # trainable: gamma, beta (the moving statistics are not)
            sizes += [b.gamma.numel(), b.beta.numel()]
        total = sum(sizes)
        theta = torch.empty(total, dtype=torch.float32, device=device)
        off, bounds = 0, [0]
        for m in mlps:
            views = []
            for (w, b) in m.params:
                wv = theta[off:off + w.numel()].view_as(w)
                wv.copy_(w)
                off += w.numel()
                bounds.append(off)
                bv = theta[off:off + b.numel()].view_as(b)
                bv.copy_(b)
                off += b.numel()
                bounds.append(off)
                views.append((wv, bv))
            m.params = views                          # the MLP now reads / is updated through the arena
            m.version += 1
            _gnn_touch()
        self._attn_off = off
        for blk in attn_blocks:
            views = {}
            for k in blk.attn_keys():
                old = blk.attn_params[k]
                view = theta[off:off + old.numel()].view_as(old)
                view.copy_(old)
                views[k] = view
                off += old.numel()
                bounds.append(off)
            blk.attn_params = views
            blk._attn_version += 1
            _gnn_touch()
        self._bn_off = off
        for b in bns:
            for name in ("gamma", "beta"):
                old = getattr(b, name)
                view = theta[off:off + old.numel()]
                view.copy_(old)
                setattr(b, name, view)
                off += old.numel()
                bounds.append(off)
            b.version += 1
            _gnn_touch()
        net._cache = None
        self.theta = theta
        self.grad = torch.zeros_like(theta)
        keep = old_m is not None and old_m.numel() == total and old_m.device == theta.device
        self.m = old_m if keep else torch.zeros_like(theta)
        self.v = old_v if keep else torch.zeros_like(theta)
        self._offsets = torch.tensor(bounds, dtype=torch.int64, device=device)
        # gradient flow descriptor: same shapes, W / b pointing into self.grad
        n = len(net.mlps("s"))
        gs, gt = (_abi.GnfMlp * n)(), (_abi.GnfMlp * n)()
        off = 0
        for arr, ms in ((gs, net.mlps("s")), (gt, net.mlps("t"))):
            for q, m in enumerate(ms):
                arr[q].num_layers = len(m.layer_sizes)
                for j, d in enumerate(m.dims()):
                    arr[q].dims[j] = d
                for j, (w, b) in enumerate(m.params):
                    arr[q].W[j] = self.grad.data_ptr() + 4 * off
                    off += w.numel()
                    arr[q].b[j] = self.grad.data_ptr() + 4 * off
                    off += b.numel()
        gattn = None
        if attn_blocks:                               # gradient GnfAttn per net, hung off the gradient GnfMlp
            gattn = (_abi.GnfAttn * len(attn_blocks))()
            for q, blk in enumerate(attn_blocks):     # order: s nets then t nets, like `blocks`
                ga = gattn[q]
                ga.num_heads, ga.kq_dim, ga.v_dim, ga.out_dim = blk.num_heads, blk.kq_dim, blk.v_dim, blk.concat_heads_output_dim
                ga.layer_norm = int(blk.layer_norm)
                ptrs = []
                for k in blk.attn_keys():
                    ptrs.append(self.grad.data_ptr() + 4 * off)
                    off += blk.attn_params[k].numel()
                ga.Wq, ga.Wk, ga.Wv, ga.Wo = ptrs[:4]
                if blk.layer_norm:
                    ga.ln_gamma, ga.ln_beta = ptrs[4:]
                arr = gs if q < n else gt
                arr[q % n].attn = C.cast(C.byref(gattn, q * C.sizeof(_abi.GnfAttn)), C.POINTER(_abi.GnfAttn))
        spec = net.blocks("s")[0].spec()
        gbn = None
        if bns:
            gbn = (_abi.GnfBatchNorm * len(bns))()
            for q, b in enumerate(bns):
                gbn[q].gamma = self.grad.data_ptr() + 4 * off
                off += b.gamma.numel()
                gbn[q].beta = self.grad.data_ptr() + 4 * off
                off += b.beta.numel()
        self._grad_flow = _abi.GnfFlow(net.num_timesteps, int(net.weight_sharing),
                                       C.cast(gs, C.POINTER(_abi.GnfMlp)), C.cast(gt, C.POINTER(_abi.GnfMlp)), spec,
                                       C.cast(gbn, C.POINTER(_abi.GnfBatchNorm)) if gbn is not None else None)
        self._keep = (gs, gt, gbn, gattn)
        self._bns = bns
        self._attn_blocks = attn_blocks
        self._arena_versions = versions()            # as left by the re-binding above
        self._arena_epoch = _EPOCH[0]

    def named_gradients(self):
        """Gradients in the oracle / fixture container layout ({"s": [[mlp]*T, [mlp]*T], "t": ...}; mlp =
        [(dW, db), ...]) as numpy arrays (host copy; for tests and summaries)."""
        net, out, off = self.net, {}, 0
        g = self.grad.detach().cpu().numpy()
        for kind in ("s", "t"):
            flat = []
            for m in net.mlps(kind):
                layers = []
                for (w, b) in m.params:
                    dw = g[off:off + w.numel()].reshape(tuple(w.shape)).copy()
                    off += w.numel()
                    db = g[off:off + b.numel()].copy()
                    off += b.numel()
                    layers.append((dw, db))
                flat.append(layers)
            t = net.num_timesteps
            out[kind] = flat if net.weight_sharing else [flat[:t], flat[t:]]
        if self._attn_blocks:      # attention nets: {"attn": {wq, wk, wv, wo}, "mlp": [...]} like the parameter container
            per_block = []
            for blk in self._attn_blocks:
                d = {}
                for k in blk.attn_keys():
                    w = blk.attn_params[k]
                    d[k] = g[off:off + w.numel()].reshape(tuple(w.shape)).copy()
                    off += w.numel()
                per_block.append(d)
            nb = len(net.mlps("s"))
            for ki, kind in enumerate(("s", "t")):
                flat = out[kind] if net.weight_sharing else out[kind][0] + out[kind][1]
                wrapped = [{"attn": per_block[ki * nb + q], "mlp": mlp} for q, mlp in enumerate(flat)]
                t = net.num_timesteps
                out[kind] = wrapped if net.weight_sharing else [wrapped[:t], wrapped[t:]]
        if self._bns:
            t, h = net.num_timesteps, self._bns[0].gamma.numel()
            flat = []
            for _ in self._bns:
                flat.append({"gamma": g[off:off + h].copy(), "beta": g[off + h:off + 2 * h].copy()})
                off += 2 * h
            out["bn"] = [flat[:t], flat[t:]]
        return out

    # ---- compute_gradients ---------------------------------------------------------------------
    def loss_and_grads(self, graph):
        """run_grevnet.py:291-302 + optimizer.compute_gradients(total_loss) (:361-362).  Leaves the gradient in
        self.grad (flat, arena layout) and returns the scalar terms (0-d fp64 device tensors)."""
        lib = _abi.lib()
        net = self.net
        x = graph.nodes
        if x.device.type != "cuda":
            raise _abi.GnfError("training runs on a HIP device only (no CPU path)")
        n, d = x.shape
        dev = x.device
        self._ensure_arena(d // 2, dev)
        # attention GNNs: the forward pass leaves every half-step's front-end (q | k | v, layer-0 inputs) in a stash
        # the backward pass reads instead of recomputing it (memory for time: 2T slots; stash_attention=False keeps
        # the fully reversible, recompute-everything walk)
        fwd_flow = net._flow(d // 2, dev)
        with torch.cuda.device(dev):   # (the planners behind these sizes read the CURRENT device's CU count)
            stash_bytes = lib.gnf_attn_stash_bytes(n, d, C.byref(fwd_flow)) if self.stash_attention else 0
            mlp_bytes = lib.gnf_mlp_stash_bytes(n, d, C.byref(fwd_flow)) if self.stash_mlp_rows else 0
        if stash_bytes:
            if self._stash is None or self._stash.numel() < stash_bytes or self._stash.device != dev:
                self._stash = torch.empty(stash_bytes, dtype=torch.uint8, device=dev)
            fwd_flow.attn_stash, fwd_flow.attn_stash_bytes = self._stash.data_ptr(), stash_bytes
        # message-passing nets on small batches: the same trade for the MLP rows (layer-0 inputs, hidden activations,
        # s and t of every half-step - what TensorFlow keeps for tf.gradients anyway): the backward kernels skip their
        # recompute half.  gnf_mlp_stash_bytes is 0 where the library would not use a stash.
        if mlp_bytes and not self._ensure_mlp_stash(mlp_bytes, dev):
            mlp_bytes = 0          # (GnfFlow.mlp_stash = NULL: the backward walk recomputes the MLP rows)
        if mlp_bytes:
            fwd_flow.mlp_stash, fwd_flow.mlp_stash_bytes = self._mlp_stash.data_ptr(), mlp_bytes
        try:
            return self._loss_and_grads(graph, n, d, dev)
        finally:   # plain forward calls of the same net must not write into (or rely on) the stashes
            fwd_flow.attn_stash, fwd_flow.attn_stash_bytes = None, 0
            fwd_flow.mlp_stash, fwd_flow.mlp_stash_bytes = None, 0

    def _ensure_mlp_stash(self, mlp_bytes, dev):
        """Make self._mlp_stash hold `mlp_bytes` on `dev` if the budget allows; False = train without the stash."""
        if self._mlp_stash is not None and self._mlp_stash.numel() >= mlp_bytes and self._mlp_stash.device == dev:
            return True
        self._mlp_stash = None     # (a smaller one is released before the larger one is asked for)
        free, _total = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)   # the caching allocator's idle blocks
        ok, why = mlp_stash_within_budget(mlp_bytes, free, self.mlp_stash_max_bytes, self.mlp_stash_free_fraction)
        if ok:
            try:
                self._mlp_stash = torch.empty(mlp_bytes, dtype=torch.uint8, device=dev)
                self.mlp_stash_declined = None
                return True
            except torch.OutOfMemoryError:
                why = "allocation failed (device out of memory)"
        self.mlp_stash_declined = (int(mlp_bytes), why)
        return False

    def _loss_and_grads(self, graph, n, d, dev):
        lib = _abi.lib()
        net = self.net
        z, sums = net._run(graph, _abi.GNF_FORWARD)               # f: fused forward kernels
        net.last_sums = sums
        z_graph = graph.replace(nodes=z)
        flow = net._flow(d // 2, dev)
        csr, csr_t = csr_of(graph), csr_of(graph, by_sender=True)
        ws_bytes = lib.gnf_backward_workspace_bytes(n, d, C.byref(flow))
        if self._ws is None or self._ws.numel() < ws_bytes or self._ws.device != dev:
            self._ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        if self.overlap_weight_grads and (self._aux is None or self._aux.device != dev):
            self._aux = torch.cuda.Stream(device=dev)
        if not self.overlap_weight_grads:
            self._aux = None
        state = z_graph.nodes.clone()                             # z in, x (reconstructed) out
        with torch.cuda.device(dev):
            _abi.check(lib.gnf_grevnet_backward_f32(C.byref(csr.desc), C.byref(csr_t.desc), C.byref(flow),
                                                    C.byref(self._grad_flow), _abi.ptr(state), state.stride(0), d,
                                                    _abi.ptr(self._ws), ws_bytes, _abi.stream_ptr(dev),
                                                    C.c_void_p(self._aux.cuda_stream if self._aux is not None else 0)),
                       "gnf_grevnet_backward_f32")
        return _StepTerms(z_graph, state, sums, n, d)

    # ---- apply_gradients -----------------------------------------------------------------------
    def current_learning_rate(self):
        if self.use_lr_decay:   # tf.train.exponential_decay(lr, global_step, decay_steps, decay_rate), run_grevnet.py:341-347
            return self.lr * self.lr_decay_rate ** (self.global_step / float(self.lr_decay_steps))
        return self.lr

    def apply_gradients(self, learning_rate=None):
        """Clipping (run_grevnet.py:363-373) then tf.train.AdamOptimizer.apply_gradients (:375) and the re-pack
        of the matrix-core weight copies."""
        lib = _abi.lib()
        dev = self.theta.device
        lr = self.current_learning_rate() if learning_rate is None else float(learning_rate)
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        n = self.theta.numel()
        with torch.cuda.device(dev):
            st = _abi.stream_ptr(dev)
            if self.clip_by_value:
                _abi.check(lib.gnf_clip_by_value_f32(_abi.ptr(self.grad), n, self.clip_lo, self.clip_hi, st),
                           "gnf_clip_by_value_f32")
            if self.clip_by_norm:
                nt = self._offsets.numel() - 1
                cb = lib.gnf_clip_workspace_bytes(nt)
                if self._clip_ws is None or self._clip_ws.numel() < cb or self._clip_ws.device != dev:
                    self._clip_ws = torch.empty(max(cb, 8), dtype=torch.uint8, device=dev)
                _abi.check(lib.gnf_clip_by_norm_f32(_abi.ptr(self.grad), _abi.ptr(self._offsets), nt, self.clip_norm,
                                                    _abi.ptr(self._clip_ws), cb, st), "gnf_clip_by_norm_f32")
            _abi.check(lib.gnf_adam_f32(_abi.ptr(self.theta), _abi.ptr(self.grad), _abi.ptr(self.m), _abi.ptr(self.v),
                                        n, lr_t, self.beta1, self.beta2, self.epsilon, st), "gnf_adam_f32")
            h = self.net.mlps("s")[0].layer_sizes[-1]
            flow = self.net._flow(h, dev)
            if self.net.fused:
                _abi.check(lib.gnf_pack_flow(C.byref(flow), st), "gnf_pack_flow")
            if self._bns:          # gamma_constraint projection (gnn.py:261-262) + UPDATE_OPS (run_grevnet.py:360), one launch
                _abi.check(lib.gnf_bn_post_step_f32(C.byref(flow), h, self._bns[0].momentum, st), "gnf_bn_post_step_f32")
        self.global_step = t

    def all_reduce_gradients(self, group=None):
        """Data parallelism: total_loss is a SUM over nodes (run_grevnet.py:295), so the gradient of the global
        batch is the sum of the shard gradients: one flat all-reduce of the whole gradient vector."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if self.grad.is_cuda and dist.get_backend(group) == "gloo":
                host = self.grad.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self.grad.copy_(host)
            else:
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def step(self, graph, learning_rate=None, all_reduce=False):
        """One iteration of the training loop (run_grevnet.py:440-447): returns the scalars of values_map."""
        out = self.loss_and_grads(graph)
        if all_reduce:
            self.all_reduce_gradients()
        self.apply_gradients(learning_rate)
        return out


def mlp_stash_within_budget(stash_bytes, free_bytes, max_bytes=None, free_fraction=0.5):
    """The trainer's rule for taking the MLP-row stash (pure arithmetic: tests/test_host_logic_cpu.py): within the caller's
    cap when there is one, else within `free_fraction` of the device memory currently free (the backward workspace, the
    attention stash and the next batch have to fit beside it).  Returns (ok, reason)."""
    if max_bytes is not None:
        if stash_bytes > max_bytes:
            return False, f"{stash_bytes} bytes > mlp_stash_max_bytes = {max_bytes}"
        return True, ""
    if stash_bytes > free_fraction * free_bytes:
        return False, f"{stash_bytes} bytes > {free_fraction:g} x {free_bytes} bytes free on the device"
    return True, ""


# ---- the trainer's state: what the drivers checkpoint through tf.train.Saver (run_grevnet.py:379, 449-453) --------------
def trainer_state(tr):
    """Everything a resumed run needs: parameters, Adam moments, step counter, batch-norm moving statistics (host copies)."""
    if tr.theta is None:
        raise RuntimeError("run a step (or loss_and_grads) first so that the variables exist")
    return {"theta": tr.theta.detach().cpu(), "m": tr.m.detach().cpu(), "v": tr.v.detach().cpu(),
            "global_step": tr.global_step,
            "bn_moving": [(b.moving_mean.detach().cpu(), b.moving_variance.detach().cpu()) for b in tr._bns]}


def load_trainer_state(tr, state):
    """Restore `trainer_state`'s dict into a connected trainer; the matrix-core weight copies follow (gnf_pack_flow)."""
    if tr.theta is None:
        raise RuntimeError("connect the trainer first (run loss_and_grads on a batch)")
    if state["theta"].numel() != tr.theta.numel():
        raise ValueError(f"checkpoint has {state['theta'].numel()} parameters, the flow has {tr.theta.numel()}")
    tr.theta.copy_(state["theta"])
    tr.m.copy_(state["m"])
    tr.v.copy_(state["v"])
    tr.global_step = int(state["global_step"])
    for b, (mm, mv) in zip(tr._bns, state["bn_moving"]):
        b.moving_mean.copy_(mm)
        b.moving_variance.copy_(mv)
    dev = tr.theta.device
    with torch.cuda.device(dev):
        flow = tr.net._flow(tr.net.mlps("s")[0].layer_sizes[-1], dev)
        if tr.net.fused:
            _abi.check(_abi.lib().gnf_pack_flow(C.byref(flow), _abi.stream_ptr(dev)), "gnf_pack_flow")


def save_checkpoint(tr, path):
    torch.save(trainer_state(tr), path)


def load_checkpoint(tr, path):
    load_trainer_state(tr, torch.load(path, map_location="cpu"))

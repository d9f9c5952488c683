"""Multi-GPU data parallelism for the hot path (new functionality named by BASELINE.json's
north_star; the reference is single-device).

A batch is a block-diagonal union of independent graphs, and with use_batch_norm=False (with it, the
batch-norm moments are per shard, as in ordinary data-parallel batch norm) the only
cross-graph quantities of the path are three batch-wide sums: log_prob_zs, log_det_jacobian
(gnn.py:322,337; run_grevnet.py:294) and sum(n_node) (run_grevnet.py:298).  So: whole graphs ->
ranks (balanced), weights replicated, each rank runs the single-GPU path on its shard, and ONE
all-reduce(sum) of a 3 x fp64 vector (RCCL over xGMI with backend "nccl"; gloo on CPU in the tests)
gives every rank the batch scalars.  No other collective exists on the path.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_graph_ids(n_node, n_edge, world_size, node_cost=1.0, edge_cost=0.05):
    """Greedy longest-processing-time partition of graphs by cost n_i*node_cost + e_i*edge_cost
    (the MLP cost scales with nodes, the aggregation with edges).  Deterministic; returns a list of
    `world_size` int arrays of graph positions (each sorted ascending so batches keep their order)."""
    n_node = np.asarray(n_node, np.float64)
    n_edge = np.asarray(n_edge, np.float64)
    cost = n_node * node_cost + n_edge * edge_cost
    order = np.argsort(-cost, kind="stable")
    loads = np.zeros(world_size)
    bins = [[] for _ in range(world_size)]
    for g in order:
        r = int(np.argmin(loads))          # ties -> lowest rank: deterministic
        bins[r].append(int(g))
        loads[r] += cost[g]
    return [np.array(sorted(b), dtype=np.int64) for b in bins]


def all_reduce_shard_sums(shard_sums, group=None):
    """The single collective of the path: sum [log_prob_zs, log_det_jacobian, num_nodes] over ranks.
    `shard_sums` is a 3-element fp64 tensor (flow.log_prob_terms(...)["shard_sums"]) on this rank's
    device (cuda for RCCL, cpu for gloo).  Returns the reduced tensor (in place)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if shard_sums.is_cuda and dist.get_backend(group) == "gloo":
            # debugging on a box without RCCL peers (several ranks on one GPU): stage through the host
            host = shard_sums.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            shard_sums.copy_(host)
        else:
            dist.all_reduce(shard_sums, op=dist.ReduceOp.SUM, group=group)
    return shard_sums


class _Reduced:
    """Handle of a reduction that is already complete (single rank, or the gloo debugging path)."""

    def wait(self):
        return True


def all_reduce_shard_sums_async(shard_sums, group=None, force=False):
    """The same collective, not waited for: returns a handle whose .wait() orders the CURRENT stream behind the
    reduction (torch.distributed Work semantics; no host blocking with RCCL).  A throughput loop over independent
    batches calls it right after batch i's forward, launches batch i + 1 on the compute stream, and only then waits
    and reads batch i's sums - the few tens of microseconds a 24-byte all-reduce over 8 GPUs takes are hidden behind
    the next batch instead of sitting between two of them.  `shard_sums` must stay untouched until .wait().
    force: issue the collective even in a one-rank group (bench.py --force-dist: the RCCL call on a 1-GPU box)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        if shard_sums.is_cuda and dist.get_backend(group) == "gloo":
            all_reduce_shard_sums(shard_sums, group)      # debugging path: staged through the host, synchronous
            return _Reduced()
        return dist.all_reduce(shard_sums, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return _Reduced()


def assemble_from_sums(sums):
    """run_grevnet.py:295-302 on the all-reduced [log_prob_zs, log_det_jacobian, num_nodes]."""
    log_prob_zs, logdet, n = sums[0], sums[1], sums[2]
    log_prob_xs = log_prob_zs + logdet
    return {
        "log_prob_zs": log_prob_zs, "log_det_jacobian": logdet, "log_prob_xs": log_prob_xs,
        "total_loss": -log_prob_xs, "num_nodes": n, "loss_per_node": -log_prob_xs / n,
        "log_prob_xs_per_node": log_prob_xs / n, "log_prob_zs_per_node": log_prob_zs / n,
        "log_det_jacobian_per_node": logdet / n,
    }


class RcclComm:
    """An RCCL communicator created by the library itself (gnf_rccl_comm_create, ABI v9) on the CURRENT device: what a
    non-Python host would hand to GnfFlow.bn_allreduce_ctx with bn_allreduce = gnf_rccl_allreduce_sum_f64.
    `unique_id`: the 128 bytes of rank 0's RcclComm.unique_id(), carried to the other ranks by any means
    (RcclComm.from_process_group uses torch.distributed for exactly that and nothing else)."""

    def __init__(self, unique_id, n_ranks, rank):
        import ctypes as C
        from . import _abi
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        h = C.c_void_p()
        _abi.check(_abi.lib().gnf_rccl_comm_create(bytes(unique_id), self.n_ranks, self.rank, C.byref(h)), "gnf_rccl_comm_create")
        self.handle = h.value

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _abi
        buf = C.create_string_buffer(128)
        _abi.check(_abi.lib().gnf_rccl_unique_id(buf), "gnf_rccl_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """One communicator over the ranks of a torch.distributed group (the id travels by broadcast_object_list)."""
        n, r = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if r == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], n, r)

    def all_reduce_sum_f64(self, t):
        """In-place SUM all-reduce of a float64 device tensor on the current stream (the hook's own entry point)."""
        from . import _abi
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        _abi.check(_abi.lib().gnf_rccl_allreduce_sum_f64(self.handle, t.data_ptr(), t.numel(), _abi.stream_ptr(t.device)),
                   "gnf_rccl_allreduce_sum_f64")
        return t

    def destroy(self):
        from . import _abi
        if self.handle:
            _abi.check(_abi.lib().gnf_rccl_comm_destroy(self.handle), "gnf_rccl_comm_destroy")
            self.handle = None

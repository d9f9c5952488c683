"""The N > 1 path of the PRODUCT (device kernels, bench.py's sharding / double-buffered async all-reduce / drain
logic, GRevNetTrainer.step(all_reduce=True)) under a real process group, on the one GPU a gpurun box has: N ranks
share cuda:0 (GNF_BENCH_ONE_DEVICE=1) and talk over gloo on 127.0.0.1.  What is NOT covered anywhere: RCCL with
N > 1 (needs N GPUs; the driver's 8-GPU run is the first) - test_parity_gpu.py only initialises a 1-rank RCCL
communicator.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(nranks, graphs_per_gpu, steps=6, warmup=2, extra=()):
    env = dict(os.environ, GNF_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    tail = ["--gpus", str(nranks), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-secondary",
            "--latency-steps", "0", "--kernel-timing-steps", "1", "--graphs-per-gpu", str(graphs_per_gpu), *extra]
    if nranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
               "--dist-backend", "gloo"] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_n_ranks_on_one_device_match_one_rank_over_the_same_batch(nranks):
    one = _bench(1, 64 * nranks)
    many = _bench(nranks, 64)
    assert many["n_gpus"] == nranks and one["n_gpus"] == 1
    assert many["config"]["nodes_total"] == one["config"]["nodes_total"]
    assert many["config"]["edges_total"] == one["config"]["edges_total"]
    # every rank's shard is about 1/N of the batch (greedy balance on nodes + edges)
    assert abs(many["config"]["nodes_rank0"] - one["config"]["nodes_total"] / nranks) <= 60
    # the all-reduced batch log-prob: same graphs, same features, same weights -> the 1-rank value (fp64 sums of
    # per-workgroup partials grouped differently: ~1e-13 relative)
    assert abs(many["log_prob_xs_per_node"] - one["log_prob_xs_per_node"]) <= 1e-9
    # every timed step's reduced sums reached the host buffer inside the timed region (drain logic)
    assert many["steps_landed_on_host"] == many["steps"] == 6
    assert one["steps_landed_on_host"] == 6


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("workload,graphs_per_rank,tol", [("config2", 16, 1e-9), ("config5", 8, 1e-9), ("default_flags", 16, 1e-6)])
def test_bench_eight_ranks_on_one_device_match_one_rank(workload, graphs_per_rank, tol):
    """EIGHT ranks - the rank count BASELINE.json's north_star names (configs 3 and 5) - through bench.py's sharded protocol
    on the one device of this box (gloo): whole graphs dealt to 8 ranks by the greedy balance, every rank's kernels on its
    shard, the asynchronous 3 x fp64 all-reduce, the drain.  config5: the large-batch kernel's hyper-parameters (D = 256,
    T = 16).  default_flags: the drivers' attention GNN + batch-norm bijectors with the batch moments taken over ALL ranks'
    nodes (GnfFlow.bn_allreduce: one exchange of 2 H + 1 doubles per bijector call, 8 ranks) - the all-reduced moments are
    fp64 sums grouped by rank, the (scale, shift) pairs derived from them are float32: a last-bit difference there moves the
    per-node log-prob by parts in 1e-8, hence the looser pin.  What this cannot show: RCCL itself with 8 peers."""
    extra = ("--workload", workload, "--repeats", "1", "--prewarm-ms", "0")   # (one timed region: this is a protocol test)
    one = _bench(1, 8 * graphs_per_rank, extra=extra)
    many = _bench(8, graphs_per_rank, extra=extra)
    assert many["n_gpus"] == 8 and many["nccl_ranks_seen"] == 8 and many["dist_backend"] == "gloo"
    assert many["config"]["nodes_total"] == one["config"]["nodes_total"]
    assert many["config"]["edges_total"] == one["config"]["edges_total"]
    assert len(many["config"]["nodes_per_rank"]) == 8 and sum(many["config"]["nodes_per_rank"]) == one["config"]["nodes_total"]
    imb = many["config"]["shard_imbalance_max_over_mean"]
    assert imb["nodes"] <= 1.05 and imb["edges"] <= 1.05, imb
    assert abs(many["log_prob_xs_per_node"] - one["log_prob_xs_per_node"]) <= tol, (many["log_prob_xs_per_node"], one["log_prob_xs_per_node"])
    assert many["steps_landed_on_host"] == many["steps"] == 6
    # (no `consistency` check here: eight processes share ONE device, a rank's HIP-event interval around its flow call
    # contains the other ranks' kernels - the timing cross-checks mean something only with a device per rank)


@pytest.mark.timeout(1200)
def test_bench_gpus_2_launches_its_own_ranks():
    """The driver's command shape for N > 1 is `python bench.py --gpus N --steps K --warmup W` with NO launcher in front:
    bench.py starts its N ranks itself (bench.self_launch).  On the one-GPU box the two ranks share cuda:0 and talk over
    gloo; everything else is the default line (no flags switching legs off)."""
    env = dict(os.environ, GNF_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dist-backend", "gloo"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1000, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["nccl_ranks_seen"] == 2 and d["dist_backend"] == "gloo"
    assert d["launcher"].startswith("bench.py itself")
    assert d["steps"] == 6 and d["warmup"] == 2 and d["steps_landed_on_host"] == 6
    assert len(d["config"]["nodes_per_rank"]) == 2 and sum(d["config"]["nodes_per_rank"]) == d["config"]["nodes_total"]
    assert d["consistency"] == []
    one = _bench(1, 128)
    assert one["config"]["nodes_total"] == d["config"]["nodes_total"]
    assert abs(d["log_prob_xs_per_node"] - one["log_prob_xs_per_node"]) <= 1e-9
    # a rank that dies takes the launch down with a non-zero exit code instead of leaving its peer in a collective
    r = subprocess.run(cmd, env=dict(env, GNF_BENCH_TEST_FAIL_RANK="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stdout[-500:])
    # RCCL cannot put two ranks on one device: refused up front with an `error` line, not a hang
    r = subprocess.run(cmd[:-2], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "error" in json.loads(r.stdout.splitlines()[-1])


@pytest.mark.timeout(900)
def test_trainer_step_all_reduce_two_ranks_match_one_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_train_check.py")], capture_output=True, text=True,
                       timeout=800, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "dp-train-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.timeout(900)
def test_bench_one_rank_through_the_rccl_branch_of_the_sharded_path():
    """`bench.py --force-dist --dist-backend nccl`: ONE rank through every branch of the N > 1 protocol over RCCL itself
    (init_process_group("nccl", device_id=...), the asynchronous 3 x fp64 all-reduce of the shard sums and its
    stream-ordered wait, the drain into the pinned host rows, barrier, the MAX all-reduce of the elapsed time, the
    shard-balance all-gather) - so that the first multi-GPU contact is not the first time those lines run.  The
    reduced log-prob must equal the plain single-GPU run's."""
    plain = _bench(1, 64)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "6",
           "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--latency-steps", "0", "--kernel-timing-steps", "1", "--repeats", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "graph-shard dp1" in d["config"]["parallelism"]
    assert d["config"]["nodes_per_rank"] == [d["config"]["nodes_total"]]
    assert d["config"]["shard_imbalance_max_over_mean"] == {"nodes": 1.0, "edges": 1.0}
    assert d["steps_landed_on_host"] == 6
    assert abs(d["log_prob_xs_per_node"] - plain["log_prob_xs_per_node"]) <= 1e-12
    assert d["consistency"] == []


def test_native_rccl_communicator_and_batch_norm_hook_one_rank(grid_small):
    """ABI v9: a communicator made by the library itself (gnf_rccl_unique_id / gnf_rccl_comm_create: ncclCommInitRank
    through dlopen'ed librccl.so) and gnf_rccl_allreduce_sum_f64 as GnfFlow.bn_allreduce - no Python callback and no
    torch.distributed anywhere in the launch path.  One rank: the all-reduce is the identity, so a batch-norm flow with
    sync_batch_norm over the native communicator must reproduce the unsynchronised flow bit for bit."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import graph_from_arrays, make_product_grevnet
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.sharding import RcclComm
    from oracle import gnf_oracle as O
    comm = RcclComm(RcclComm.unique_id(), 1, 0)
    try:
        t = torch.arange(7, dtype=torch.float64, device="cuda:0") * 1.5
        want = t.clone()
        comm.all_reduce_sum_f64(t)
        torch.cuda.synchronize()
        assert torch.equal(t, want)
        n_node, n_edge, sl, rl = grid_small
        nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, list(range(12)))
        n = int(nn.sum())
        hp = dict(D=8, latent=32, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False)
        p = O.make_grevnet_params(8, 4, 32, 3, 2, final_scale=0.3)
        p["bn"] = O.make_bn_params(9, 4, 2)
        x = (np.random.default_rng(31).standard_normal((n, 8)) * 1.5 + 0.5).astype(np.float32)
        graph = graph_from_arrays(nn, ne, s, r, x, "cuda:0")
        plain = log_prob_terms(make_product_grevnet(hp, p), graph)
        net = make_product_grevnet(hp, p)
        net.sync_batch_norm = True
        net.bn_rccl_comm = comm
        synced = log_prob_terms(net, graph)
        torch.cuda.synchronize()
        assert torch.equal(synced["z_graph"].nodes, plain["z_graph"].nodes)
        assert float(synced["log_prob_xs_per_node"]) == float(plain["log_prob_xs_per_node"])
        ref = O.Fp64Dense(s, r, n).log_prob(x, p, 2)
        assert abs(float(synced["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4
    finally:
        comm.destroy()

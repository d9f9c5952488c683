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


@pytest.mark.timeout(900)
def test_trainer_step_all_reduce_two_ranks_match_one_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_train_check.py")], capture_output=True, text=True,
                       timeout=800, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "dp-train-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])

"""bench.py pieces that run without a GPU: the algorithmic work model, the weight
generator, batch construction + sharding, and the workload table."""
import json
import os
import sys

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_work_matches_survey_worked_example():
    # config 2, D=64, L=256, K=5, agg-then: P_w = 212,992
    hp = dict(bench.HP)
    n, e = 2560, 29663
    flops, nbytes = bench.algorithmic_half_step(n, e, hp)
    p_w = 32 * 256 + 3 * 256 * 256 + 256 * 32
    assert p_w == 212992
    assert flops == n * 4 * p_w + e * 32 + 6 * n * 32
    assert nbytes == 12 * n * 32 + 4 * e + 4 * n + 8 * (p_w + 4 * 256 + 32)
    # per node-update (2 half-steps): F_nu ~ 1.705 MFLOP
    assert abs(2 * flops / n - 1.705e6) / 1.705e6 < 0.01


def test_make_params_layout_and_scale():
    hp = dict(bench.HP, T=2)
    p = bench.make_params(1, hp, 0.25)
    assert len(p["s"]) == 2 and len(p["s"][0]) == 2 and len(p["t"][1]) == 2
    mlp = p["s"][0][0]
    assert [w.shape for w, _ in mlp] == [(32, 256), (256, 256), (256, 256), (256, 256), (256, 32)]
    assert mlp[-1][0].std() < 0.5 * mlp[1][0].std()         # last layer scaled down
    assert mlp[0][0].dtype == np.float32


def test_make_batch_shards_cover_the_global_batch():
    bench.WORKLOAD = bench.WORKLOADS["config2"]
    bench.GRAPHS_PER_GPU = 8
    try:
        tot_n = 0
        for rank in range(2):
            dicts, n_global, e_global = bench.make_batch(2, rank)
            tot_n += sum(d["n_node"] for d in dicts)
            assert all(d["nodes"].shape == (d["n_node"], bench.HP["D"]) for d in dicts)
        assert tot_n == n_global
        d1, n1, e1 = bench.make_batch(1, 0)
        assert len(d1) == 8 and sum(d["n_node"] for d in d1) == n1
        assert sum(len(d["senders"]) for d in d1) == e1
    finally:
        bench.GRAPHS_PER_GPU = 64


def test_workload_table():
    assert set(bench.WORKLOADS) == {"config2", "config2_fc", "config2_attn", "config4", "config5", "wide_fc", "config2_train", "default_flags",
                                    "default_flags_train", "data_default_flags", "data_default_flags_train", "wide_fc_train"}
    assert set(bench.WORKLOAD_SOURCES) == set(bench.WORKLOADS)
    # the data driver's literal defaults (train_grevnet_with_data.py:40-46, 100-117)
    hp = bench.WORKLOADS["data_default_flags"]["hp"]
    assert (hp["D"], hp["latent"], hp["K"], hp["T"], hp["use_batch_norm"], hp["activation"]) == (200, 2048, 3, 10, True, "relu")
    assert hp["attn"] == dict(num_heads=1, kq_dim=64, v_dim=64, out_dim=64, concat=True, kq_dim_division=True, residual=False)
    assert bench.WORKLOADS["data_default_flags"]["fc"] and bench.WORKLOADS["data_default_flags_train"]["train"]
    assert bench.WORKLOADS["config5"]["hp"] == dict(D=256, T=16)
    assert bench.WORKLOADS["config4"]["inverse"] is True


def test_every_pmc_traffic_entry_names_its_kernel_sources():
    """profiles/pmc_traffic.json: each workload's entry carries the list of kernel sources it is evidence for (the
    workload's set in bench.WORKLOAD_SOURCES, every one an existing file of csrc/) and the stamp over them; the training /
    wide-layer workloads' sets contain the files their dominant kernels live in."""
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert "source_stamp" not in d                       # (round 4's single stamp over five forward files is gone)
    csrc = os.path.join(ROOT, "graph-normalizing-flows_amd", "csrc")
    for wl, e in d["workloads"].items():
        assert e["sources"] and len(e["source_stamp"]) == 16, wl
        assert all(os.path.exists(os.path.join(csrc, f)) for f in e["sources"]), wl
        if e["source_stamp"] == bench.kernel_source_stamp(wl):     # evidence for THIS build: taken on exactly the workload's set
            assert e["sources"] == sorted(set(bench.WORKLOAD_SOURCES[wl])), wl
        else:                                                      # a snapshot of an earlier build: the set it names was that build's
            assert set(e["sources"]) <= set(bench.WORKLOAD_SOURCES[wl]) | {"gnf_attn_front_dev.h"}, wl
    assert {"gnf_linear_big.hip", "gnf_train.hip"} <= set(bench.WORKLOAD_SOURCES["wide_fc"])
    for wl in bench.WORKLOADS:
        if wl.endswith("_train"):
            assert {"gnf_train.hip", "gnf_fused_bwd.hip", "gnf_fused_bwd_dev.h"} <= set(bench.WORKLOAD_SOURCES[wl]), wl
        if "default_flags_train" in wl:
            assert {"gnf_attn_bwd.hip", "gnf_bn_bwd.hip"} <= set(bench.WORKLOAD_SOURCES[wl]), wl
        assert len(bench.kernel_source_stamp(wl)) == 16


def test_line_consistency_checks():
    """kernel_us x launches_per_step <= ms_per_step (round 2's attention line had 78.8 us x 16 = 1.26 ms against a 0.866 ms
    step: its kernel-timing leg ran a path the step does not), on synthetic lines and on every round-3 / round-4 line
    committed under profiles/ (the default line's secondary workloads included); a round-5 training line also carries train_roofline."""
    import glob
    import json
    import os
    good = {"ms_per_step": 0.566, "config": {"workload": "config2: ..."},
            "spread": {"ms_per_step_min": 0.56, "ms_per_step_max": 0.57},
            "roofline": {"kernel_us": 35.2, "launches_per_step": 16, "frac": 0.4184, "achieved": 65.82, "peak": 157.3}}
    assert bench.line_consistency_errors(good) == []
    bad = json.loads(json.dumps(good))
    bad["ms_per_step"] = 0.866
    bad["spread"] = {"ms_per_step_min": 0.86, "ms_per_step_max": 0.87}
    bad["roofline"]["kernel_us"] = 78.81
    assert any("kernel_us" in e for e in bench.line_consistency_errors(bad))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = 0
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r3*")) + glob.glob(os.path.join(root, "profiles", "r4*")) +
                    glob.glob(os.path.join(root, "profiles", "r5*"))):
        for line in open(f, errors="replace"):
            if line.startswith('{"metric"'):
                d = json.loads(line)
                assert bench.line_consistency_errors(d) == [], (f, bench.line_consistency_errors(d))
                for wl, e in (d.get("secondary_workloads") or {}).items():
                    assert e.get("consistency") in ([], None), (f, wl, e.get("consistency"))   # (None: round-3 lines had no per-child check)
                if "_train" in d["config"]["workload"].split(":")[0] and os.path.basename(f).startswith("r5"):
                    tr = d["train_roofline"]
                    assert abs(tr["frac"] - tr["achieved"] / tr["peak"]) <= 2e-3
                    assert abs(tr["achieved"] - tr["algorithmic_flops_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12) <= 2e-3 * tr["achieved"]
                seen += 1
    assert seen >= 20


def test_bench_gpus_n_without_devices_prints_an_error_line_and_fails():
    """`python bench.py --gpus 2` with no launcher and fewer than 2 devices: ONE JSON line with an `error` field and a
    non-zero exit code (not a hang in init_process_group, not a traceback)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and "error" in d and d["devices_visible"] == 0


def test_secondary_entry_keeps_the_trainer_fields():
    """The default run's secondary_workloads: a forward child keeps roofline frac / kernel_us / traffic (and kernel_a only if
    its flow launches k_aggregate); a training child also carries train_roofline (frac, flops per step, the dominant
    kernel's counter traffic) - row f4's driver-timed evidence (round 6).  Checked on the committed round-5 lines."""
    fwd = json.loads(open(os.path.join(ROOT, "profiles", "r5z_bench_config5.json")).read().strip().splitlines()[-1])
    e = bench.secondary_entry(fwd)
    assert e["frac"] == fwd["roofline"]["frac"] and e["kernel_a"]["kernel"].startswith("k_aggregate") and "train_frac" not in e
    trn = json.loads(open(os.path.join(ROOT, "profiles", "r5z_bench_config2_train.json")).read().strip().splitlines()[-1])
    e = bench.secondary_entry(trn)
    assert e["train_frac"] == trn["train_roofline"]["frac"] and e["ms_per_step"] == trn["ms_per_step"]
    assert e["train_roofline"]["algorithmic_flops_per_step"] == trn["train_roofline"]["algorithmic_flops_per_step"]
    assert abs(e["train_frac"] - e["train_roofline"]["algorithmic_flops_per_step"] / (e["ms_per_step"] * 1e-3) / 1e12 / bench.PEAK_FP32_MATRIX_TFLOPS) < 2e-3
    attn = dict(fwd, kernel_a=None)           # an attention workload's line (round 6): no kernel_a at all
    assert "kernel_a" not in bench.secondary_entry(attn)
    # every committed default-run line of round 6 carries both trainer children
    import glob
    for f in glob.glob(os.path.join(ROOT, "profiles", "r6*_bench_config2_default_run.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        sw = d["secondary_workloads"]
        for wl in ("config2_train", "data_default_flags_train"):
            assert "error" not in sw[wl], (f, wl, sw[wl])
            assert 0.0 < sw[wl]["train_frac"] < 1.0 and sw[wl]["consistency"] == []
        for wl in ("config2_attn", "default_flags", "data_default_flags"):
            assert "kernel_a" not in sw[wl], (f, wl)

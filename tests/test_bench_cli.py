"""bench.py's launch contract: a rank count that was not run never produces a line."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_more_gpus_than_the_node_has_is_an_error_not_a_one_gpu_line():
    """`python bench.py --gpus 64` outside a launcher starts the ranks itself -- after checking that the
    node has that many GPUs.  Round 1 printed an n_gpus = 1 line here."""
    r = run_bench(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "GPU(s)" in r.stderr
    assert r.stdout.strip() == ""


def test_rank_count_from_the_launcher_must_match():
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 2 but 1 rank(s)" in r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.gpu
def test_two_ranks_on_a_one_gpu_box_fail_loudly():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and r.stdout.strip() == ""


@pytest.mark.gpu
def test_default_line_is_strong_scaling_and_verified():
    """A small matrix through the default workload: the line names the whole-matrix (strong) split and carries
    the self-check of the last step's results."""
    r = run_bench(["--variants", "20000", "--samples", "50000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["verified"] is True
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_all_configs_ride_the_default_line():
    """--configs all on a small matrix: BASELINE configs 2-5 next to the freq headline, each verified, and the
    table-function section."""
    r = run_bench(["--variants", "30000", "--samples", "20000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                   "--configs", "all", "--config-steps", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"] == "plink_freq genotypes/s" and line["verified"] is True
    cfg = line["configs"]
    assert set(cfg) == {"fused", "unpack", "score16", "score1", "pca"}
    assert all(c["verified"] is True and c["frac"] > 0 and c["kernel_ms_avg"] > 0 for c in cfg.values())
    assert cfg["unpack"]["roofline"]["store_ceiling"]["GB/s"] > 0
    assert "algorithmic_flops_frac_of_fp64_peak" in cfg["score16"]["roofline"]
    sql = line["sql"]
    assert sql.pop("verified") is True
    assert len(sql) == 4 and all(v["rows"] in (30000, 20000) and v["scan_ms"] > 0 for v in sql.values())


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["freq", "score", "pca"])
def test_two_rank_run_rehearsed_on_one_gpu(workload):
    """The N-rank logic of bench.py (shard ranges, the reduction of per-sample partials, max-over-ranks timing, the
    verification vote, rank 0's line) under the driver's own launcher with two ranks -- both on GPU 0 and over gloo,
    since this box has one GPU and RCCL refuses two ranks on one device.  The line says it is a rehearsal."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PGH_BENCH_ONE_GPU_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    shape = {"freq": ["--variants", "40000", "--samples", "30000"],
             "score": ["--workload", "score", "--variants", "20000", "--samples", "20000"],
             "pca": ["--workload", "pca", "--variants", "6000", "--samples", "5000"]}[workload]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--cpu-seconds", "0"] + shape, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True and line["value"] > 0
    assert "REHEARSAL" in line["config"]["workload"]
    assert line["cpu_baseline"] is None and "configs" not in line  # N = 1 only

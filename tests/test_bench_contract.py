"""bench.py's output contract (one JSON line on stdout with the driver's keys plus `roofline` and
`cpu_baseline`) and its behaviour without a GPU (fails loudly: there is no CPU fallback)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                          text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = run_bench("--steps", "2", "--warmup", "1", "--scene", "killeroos", "--spp", "1", "--cpu-sample", "20000",
                    "--cpu-passes", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["unit"] == "Mray/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0
    rf = r["roofline"]
    # the record names what binds the kernel (vector-instruction issue, from committed PMC passes where this
    # configuration has them) and never carries a fraction above 1; the SURVEY §8d algorithmic-bytes figure
    # rides beside it as alg_hbm
    assert rf["bound"].startswith("valu_issue") and rf["unit"] == "G wave-instr/s" and "traffic" in rf
    assert rf["frac"] is None or (0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-2)
    ah = rf["alg_hbm"]
    assert ah["unit"] == "GB/s" and abs(ah["frac"] - ah["achieved"] / ah["peak"]) < 1e-3
    assert rf["avg_launch_ms"] > 0
    assert r["dependent_step"]["ms_per_step"] > 0 and set(r["per_class_mrays"]) >= {"primary_closest", "bounce2_closest"}
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["matches_gpu"] is True


def test_bench_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = run_bench("--steps", "1", "--warmup", "0", "--scene", "killeroos", "--spp", "1", timeout=300)
    assert out.returncode != 0
    assert out.stdout.strip() == "", "no result line may be printed without a GPU"


def test_bench_gpus_n_starts_its_own_ranks_and_joins_them():
    """`python bench.py --gpus 2` without a launcher must spawn one rank per GPU itself (before any
    GPU call) and exit with the children's status.  NNBVH_BENCH_DRYRUN=1 stops each rank after the
    rendezvous, so this runs on CPU: two gloo ranks meet, rank 0 prints ONE JSON line."""
    env = dict(os.environ, NNBVH_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r == {"dry_run": True, "n_gpus": 2, "rank_sum": 3.0, "steps": 3, "warmup": 1}


def test_bench_exits_nonzero_when_ranks_and_gpus_disagree():
    env = dict(os.environ, NNBVH_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 2 and out.stdout.strip() == ""


def test_bench_failing_rank_fails_the_launcher():
    """no GPU here: the spawned ranks exit non-zero and so must the parent (no result line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NNBVH_BENCH_DRYRUN")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--scene", "killeroos", "--spp", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_on_one_gpu_prints_one_line_with_the_film_gather():
    """`bench.py --gpus 2` self-launches; with NNBVH_BENCH_BACKEND=gloo both ranks share the one GPU
    of this box (rehearsal of the N>1 code path: tile shards, film accumulation, film all-gather)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["NNBVH_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--scene", "killeroos", "--spp", "1", "--sample-sets", "2"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["film_allgather_ms"] > 0 and r["film_allgather_bytes_per_rank"] > 0
    assert r["film"]["weight_sum"] > 0 and "cpu_baseline" not in r

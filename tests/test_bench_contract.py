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
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
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

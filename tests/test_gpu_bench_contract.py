"""bench.py is the driver's measurement contract: one JSON line on stdout with the agreed keys, whatever else changes.
Runs the real script (tiny K / W, side measurements off) on the GPU and checks the line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--kernel-reps", "1",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "quadruplets/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]              # whole-job throughput = B / step time
    cfg = d["config"]
    assert "workload" in cfg and "dropout 0.1" in cfg["workload"] and "model" not in cfg
    assert cfg["global_batch"] == 64 and cfg["seq_len"] == 128 and cfg["parallelism"] == "dp1"
    ro = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_side"):
        assert k in ro, k
    assert ro["bound"] in ("hbm", "mfma") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert d["step_without_dropout"]["value"] >= d["value"] * 0.9

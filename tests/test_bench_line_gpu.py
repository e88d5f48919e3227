"""The driver's contract with bench.py, on a real GPU: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON line on stdout with the
agreed keys — the metric BASELINE.json names, whole-job throughput, the `roofline` of the dominant kernel and (unless switched off) the
`cpu_baseline` — and the numbers hang together (value = B K / time; roofline.frac = achieved / peak)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", *flags],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must hold the result line and nothing else: %r" % lines[:3]
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys_and_consistent_numbers():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    r = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["metric"] == base["metric"] and r["unit"] == "samples/s" and r["n_gpus"] == 1 and r["steps"] == 8 and r["warmup"] == 2
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert "workload" in r["config"] and "model" not in r["config"]
    B = r["config"]["global_batch"]
    assert abs(r["value"] - B / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]
    assert 1e5 < r["value"] < 1e7 and 0.1 < r["ms_per_step"] < 2.0  # (a batch-256 step of this engine on an MI355X: 0.25 ms)
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9 and 0 < rf["frac"] < 1
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["unit"] == r["unit"]


def test_graph_and_launch_give_the_same_kind_of_line():
    a, b = _run("--no-cpu-baseline", "--steps-only", "--graph"), _run("--no-cpu-baseline", "--steps-only", "--no-graph")
    assert a["config"]["graph"] is True and b["config"]["graph"] is False
    assert "roofline" not in a and "cpu_baseline" not in a  # (--steps-only: profiler passes)
    assert 0.5 < a["value"] / b["value"] < 2.0

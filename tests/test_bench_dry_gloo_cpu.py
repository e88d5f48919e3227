"""De-risking the first multi-GPU bench run without hardware: the driver's own launch line
`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W`
and the self-spawning form `python bench.py --gpus 2` run HERE, on the CPU, as a dry run: gloo instead of RCCL and a stand-in engine
(tests/_dry_engine.py) behind the same DataParallelStep, the same step loop, fences, max-over-ranks timing and result line.  Asserted: ONE
JSON line on stdout, `n_gpus` 2, the exchange step taken with a world of 2, value = global samples / time."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRY = os.path.join(ROOT, "tests", "_dry_engine.py") + ":make"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NASREC_BENCH_DRY_ENGINE=DRY, NASREC_BENCH_DRY_BATCH="4", OMP_NUM_THREADS="2")
    return env


def _check(stdout, steps, warmup):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry the result line and nothing else: %r" % lines
    r = json.loads(lines[0])
    assert r["metric"] == "supernet samples/sec at batch 256 (Criteo-shape), 1/2/4/8 MI355X"
    assert r["n_gpus"] == 2 and r["steps"] == steps and r["warmup"] == warmup and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["config"]["dry_run"] is True and "DRY RUN" in r["data"]
    assert r["config"]["global_batch"] == 2 * r["config"]["per_gpu_batch"]
    ex = r["config"]["dp_exchange"]
    assert ex["world_size"] == 2 and ex["backend"].startswith("gloo")
    assert ex["allgather_MB"]["row_gradients"] > 0 and len(ex["pieces"]) >= 1
    # whole-job throughput: the samples of BOTH ranks over the slowest rank's time
    assert abs(r["value"] - r["config"]["global_batch"] * steps / (r["ms_per_step"] * 1e-3 * steps)) <= 1e-6 * r["value"]
    assert r["final_loss"] == r["final_loss"]
    return r


@pytest.mark.timeout(900)
def test_the_drivers_torchrun_line_prints_one_result_line_for_two_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    _check(p.stdout, 3, 1)


@pytest.mark.timeout(900)
def test_bench_gpus_2_spawns_its_own_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    _check(p.stdout, 2, 1)

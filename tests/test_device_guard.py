"""Every engine entry point that launches on the GPU runs with the ENGINE's device current (engine._on_device): raw launches, graph
replays and event creation act on the current device, and a single-process run with `--gpu N` never calls torch.cuda.set_device
(utils/dist.init_from_env does so only for world > 1).  Round 5 lost the guard on train_step when a method was inserted between the
decorator and the def; the CPU test pins the set of guarded entry points, the GPU test (two devices) runs a step from the other device."""
import inspect

import pytest

GUARDED = ("forward", "train_step", "run_forward", "run_backward", "forward_backward")


def test_gpu_entry_points_are_wrapped_by_the_device_guard():
    import nasrec_amd
    from nasrec_amd import engine
    if nasrec_amd.host_modules_compiled().get("engine"):
        pytest.skip("compiled engine module: functools.wraps attributes are not introspectable; tests/test_host_modules_cpu.py pins "
                    "that it is built from this source")
    for name in GUARDED:
        fn = getattr(engine.SupernetEngine, name)
        assert getattr(fn, "__wrapped__", None) is not None, "SupernetEngine.%s lost its @_on_device guard" % name
    # prefers_graph touches no GPU: no guard (and above all it must not have taken train_step's)
    assert getattr(engine.SupernetEngine.prefers_graph, "__wrapped__", None) is None
    src = inspect.getsource(engine.SupernetEngine)
    assert "@_on_device\n    def train_step" in src


def test_guard_is_pinned_in_the_source_even_when_the_module_is_compiled():
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(here, "nasrec_amd", "engine.py")).read()
    for name in GUARDED:
        assert "@_on_device\n    def %s(" % name in src, name
    assert "@_on_device\n    def prefers_graph" not in src


@pytest.mark.gpu
def test_train_step_runs_on_the_engines_device_whatever_device_is_current():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import os
    import numpy as np
    from nasrec_amd import plan as P
    from nasrec_amd.engine import SupernetEngine
    from nasrec_amd.search_space import ops_config_lib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ca = json.load(open(os.path.join(here, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
    choice = {"macro": ca["macro"], "micro": ca["micro"]}
    rows = [1000] * 26
    cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
    eng = SupernetEngine(cfg, 13, 26, rows, device="cuda:1", warm_choice=choice)
    rng = np.random.RandomState(0)
    B = 64
    int_x = torch.tensor(rng.rand(B, 13).astype(np.float32), device="cuda:1")
    cat_x = torch.tensor(rng.randint(0, 1000, size=(B, 26)), device="cuda:1")
    y = torch.tensor(rng.randint(0, 2, size=(B,)).astype(np.float32), device="cuda:1")
    torch.cuda.set_device(0)
    loss = eng.train_step(int_x, cat_x, y, lr=1e-3)
    torch.cuda.synchronize(1)
    assert torch.cuda.current_device() == 0
    assert loss.device.index == 1 and bool(torch.isfinite(loss).all())

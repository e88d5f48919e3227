"""CPU tests of the drop-in nn.Module surface: module tree / state_dict keys / parameters() order against what the real
reference produced (stored in the golden fixtures), sampling order against the reference's traces, C-ABI symbols."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, golden_files, load_golden
from nasrec_amd import _lib as L
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.supernet.supernet import SuperNet

NPZ = golden_files("fixed_*.npz") + golden_files("supernet_*.npz")


def build(meta, Fs):
    fixed = meta["mode"] == "fixed"
    return SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                    activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=Fs,
                    path_sampling_strategy="fixed-path" if fixed else "full-path", fixed=fixed,
                    fixed_choice=meta["choice"] if fixed else None, last_n_blocks_out=meta.get("last_n_blocks_out", 1))


@pytest.mark.parametrize("path", NPZ, ids=[os.path.basename(p)[:-4] for p in NPZ])
def test_state_dict_keys_and_parameter_order_match_reference(path):
    z, meta = load_golden(path)
    m = build(meta, z["cat_x"].shape[1])
    m._materialize(z["int_x"].shape[1])
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["param_shapes"]
    assert list(sd.keys()) == list(meta["param_shapes"].keys())  # state_dict order
    assert [n for n, _ in m.named_parameters()] == meta["param_order"]  # optimizer / clip order
    # init_weights dispatches on exact types (train_utils.py:76-87)
    kinds = {type(x) for x in m.modules()}
    assert torch.nn.Linear in kinds and torch.nn.Embedding in kinds
    assert not any(isinstance(x, torch.nn.LazyLinear) for x in m.modules())


def test_sampling_order_matches_reference_traces():
    traces = json.load(open(os.path.join(GOLDEN, "samplers.json")))
    for tr in traces:
        m = SuperNet(num_blocks=tr["num_blocks"], ops_config=ops_config_lib[tr["space"]], use_layernorm=True,
                     num_embeddings=[50] * 10, sparse_input_size=10, path_sampling_strategy="full-path", fixed=False,
                     anypath_choice=tr["anypath_choice"], supernet_training_steps=tr["supernet_training_steps"],
                     candidate_choices=tr.get("candidate_choices"))
        for _ in range(tr["warmup_forwards"]):
            m._resolve_choice(None)  # the full-path warm-up forward advances every counter
        m.configure_path_sampling_strategy(tr["strategy"])
        np.random.seed(tr["seed"])
        for want in tr["choices"]:
            got = m._resolve_choice(None)
            got = json.loads(json.dumps(got, default=lambda o: o.tolist() if hasattr(o, "tolist") else int(o)))
            assert got == want, (tr["strategy"], tr["anypath_choice"], tr["supernet_training_steps"])


def test_cpu_forward_fails_loudly():
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_xlarge.npz"))
    m = build(meta, 26)
    with pytest.raises(L.EngineError):
        m(torch.tensor(z["int_x"]), torch.tensor(z["cat_x"]))


def test_c_abi_exports_every_declared_symbol_and_struct_layouts():
    lib = L.load()  # verifies symbols + sizeof of every descriptor against the header's structs
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "nasrec_hip.h")).read()
    import re
    declared = set(re.findall(r"\n(?:int|int64_t|const char\*)\s+(nasrec_\w+)\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s)
    assert lib.nasrec_abi_version() == 17
    assert lib.nasrec_launch(None, None) != 0  # null descriptor: error code + message, no crash
    assert b"null descriptor" in lib.nasrec_last_error()


def test_deepcopy_and_modes():
    import copy
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_kdd_autoctr.npz"))
    m = build(meta, 10)
    m._materialize(3)
    m2 = copy.deepcopy(m)
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys())
    m.set_mode_to_finelune_last_only()
    assert all(p.requires_grad == n.startswith("_final") for n, p in m.named_parameters())
    m.set_mode_to_normal_mode()
    assert all(p.requires_grad for p in m.parameters())
    assert len(m.get_sparse_parameters()) == 10 and len(m.get_dense_parameters()) == len(list(m.parameters())) - 10


def test_reference_import_paths_resolve_to_the_engine():
    """the import statements of the reference's callers (supernet.py:34-51, train_utils.py:27-33, main_train.py:71)"""
    from nasrec.supernet.modules import (CleverMaskGenerator, CleverZeroTensorGenerator, DotProduct, ElasticLinear,  # noqa: F401
                                         ElasticLinear3D, FactorizationMachine3D, SigmoidGating, Sum, Transformer, Zeros2D,
                                         Zeros3D)
    from nasrec.supernet.supernet import SuperNet as S2, ops_config_lib as lib2  # noqa: F401
    from nasrec.supernet.utils import anypath_choice_fn, assert_valid_ops_config  # noqa: F401
    from nasrec.utils.config import NUM_EMBEDDINGS_CRITEO
    assert S2 is SuperNet and len(NUM_EMBEDDINGS_CRITEO) == 26
    m = CleverMaskGenerator()(8, 3)
    assert m.tolist() == [1, 1, 1, 0, 0, 0, 0, 0]


def test_lr_schedulers_match_reference_traces():
    """nasrec_amd.utils.lr_schedule (what the harness and bench.py step) against sequences recorded from the reference's own
    schedulers (tests/golden/lr.json), incl. the `step(epoch=e)` jumps eval_subnet_from_supernet.py:179 makes"""
    import json
    import os
    from helpers import GOLDEN
    from nasrec_amd.utils.lr_schedule import ConstantWithWarmup, CosineAnnealingWarmupRestarts
    lr = json.load(open(os.path.join(GOLDEN, "lr.json")))
    c = lr["cosine_restarts"]
    s = CosineAnnealingWarmupRestarts(c["first_cycle_steps"], c["cycle_mult"], c["max_lr"], c["min_lr"], c["warmup_steps"], c["gamma"])
    for want in c["lrs"]:
        assert abs(s.get_lr() - want) <= 1e-15 + 1e-12 * abs(want)
        s.step()
    c = lr["constant_warmup"]
    s = ConstantWithWarmup(c["base_lr"], c["num_warmup_steps"])
    for want in c["lrs"]:
        assert abs(s.get_lr() - want) <= 1e-15 + 1e-12 * abs(want)
        s.step()
    for c in lr["cosine_epoch_jumps"]:
        s = CosineAnnealingWarmupRestarts(c["first_cycle_steps"], c["cycle_mult"], c["max_lr"], c["min_lr"], c["warmup_steps"], c["gamma"])
        for e, a, b in c["seq"]:
            s.step(epoch=e)
            assert abs(s.get_lr() - a) <= 1e-15, (e, s.get_lr(), a)
            s.step()
            assert abs(s.get_lr() - b) <= 1e-15, (e, "next", s.get_lr(), b)
        st = s.state_dict()
        s2 = CosineAnnealingWarmupRestarts(c["first_cycle_steps"], c["cycle_mult"], c["max_lr"], c["min_lr"], c["warmup_steps"], c["gamma"])
        s2.load_state_dict(st)
        assert s2.step() == s.step()


def test_row_sharded_module_keeps_the_reference_state_dict_in_a_single_process():
    """SuperNet(table_sharding="row") with world size 1: this rank's row range is the whole table, so the module tree, the
    state_dict keys / shapes and the parameter order are the reference's; load_state_dict takes whole tables"""
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_kdd_autoctr.npz"))
    m = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                 activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=z["cat_x"].shape[1],
                 path_sampling_strategy="fixed-path", fixed=True, fixed_choice=meta["choice"], table_sharding="row")
    assert SuperNet._shard_rows(7) == (0, 7, 7)
    m._materialize(z["int_x"].shape[1])
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["param_shapes"]
    assert [n for n, _ in m.named_parameters()] == meta["param_order"]
    m.load_state_dict({k: torch.zeros(v.shape) for k, v in sd.items()}, strict=True)

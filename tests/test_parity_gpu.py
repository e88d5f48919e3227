"""End-to-end parity of the HIP engine (through the C-ABI) against the golden vectors produced by the real reference
and against the fp64 oracle, for every committed sub-network / supernet case.  Run with `-m gpu` on an MI355X.

Tolerances (BASELINE.json: "fp32 logits within 1e-5"): |logit_hip - logit_ref_fp64| <= 1e-5 * max(1, max|logit|) AND
<= max(1e-5, 1.25 x the reference's own |fp32 - fp64| on the same inputs), i.e. absolute 1e-5 wherever the reference's fp32 run meets it;
gradient checksums within 2e-5 of the gradient norm; three Adagrad steps keep parameters within 2e-5."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import check_trajectory, golden_files, load_golden, oracle_cfg, oracle_params, proj_checksum
from nasrec_amd import _lib as L
from nasrec_amd import plan as P
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from oracle import nasrec_oracle as O

pytestmark = pytest.mark.gpu
# 3-step clipped-Adagrad trajectories (fixtures at lr 1e-3, where the reference itself stays bounded on all 13 networks):
# parameters within 1e-4 of their norm, updates within 1e-2 of the update scale (the reference's own fp32 run drifts from
# its fp64 run by 9e-6 / 1.8e-3 on these scales — `ref_fp32_drift` in each fixture), losses / gradient norms within 1e-3
TRAJ_REL_PARAMS, TRAJ_REL_DELTA, TRAJ_REL_LOSS = 1e-4, 1e-2, 1e-3
NPZ = golden_files("fixed_*.npz") + golden_files("supernet_*.npz")
IDS = [os.path.basename(p)[:-4] for p in NPZ]


def _report_logit_err(name, err, scale, z):
    """achieved max |logit - logit_ref_fp64| per golden network -> test log + gpurun_out/golden_logit_err.json (scratch; BASELINE.md
    quotes it).  The bar is 1e-5 * max(1, max |logit|): the reference's OWN fp32 evaluation differs from its fp64 evaluation by
    `ref32` on the same inputs (stored in the fixture), which is the noise floor any fp32 implementation lives on."""
    import json
    ref32 = float(np.abs(z["logits_f32"].astype(np.float64) - z["logits_f64"]).max()) if "logits_f32" in z.files else None
    rec = dict(err=err, scale=scale, bar=1e-5 * scale, ref_fp32_vs_fp64=ref32)
    print("golden logit error", name, json.dumps(rec))
    try:
        f = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "golden_logit_err.json")
        os.makedirs(os.path.dirname(f), exist_ok=True)
        cur = json.load(open(f)) if os.path.exists(f) else {}
        cur[name] = rec
        json.dump(cur, open(f, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def build_engine(z, meta):
    cfg = P.NetConfig(meta["num_blocks"], ops_config_lib[meta["config"]], meta["use_layernorm"], meta["activation"],
                      fixed=(meta["mode"] == "fixed"), last_n_blocks_out=meta.get("last_n_blocks_out", 1))
    Fd, Fs = z["int_x"].shape[1], z["cat_x"].shape[1]
    eng = SupernetEngine(cfg, Fd, Fs, meta["tables"], warm_choice=meta["choice"] if cfg.fixed else None)
    assert {k: list(v) for k, v in eng.shapes.items()} == meta["param_shapes"]
    missing = eng.load_params({k: O.seeded_param(k, shp) for k, shp in meta["param_shapes"].items()})
    assert missing == []
    return eng


@pytest.mark.parametrize("path", NPZ, ids=IDS)
def test_logits_match_reference(path):
    z, meta = load_golden(path)
    eng = build_engine(z, meta)
    int_x, cat_x = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda()
    out = eng.forward(int_x, cat_x, meta["choice"])
    eng.check_indices()
    ref = z["logits_f64"]
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(out.cpu().numpy().astype(np.float64) - ref).max())
    _report_logit_err(os.path.basename(path)[:-4], err, scale, z)
    assert err <= 1e-5 * scale, "logit err %.3e (scale %.2f)" % (err, scale)
    # ... and tied to MEASURED noise, not only to the logit scale: the fixture holds the reference's own fp32 and fp64 logits on
    # these inputs; the engine may not be further from fp64 than 1.25 x the reference's fp32 run is (or 1e-5 absolute, whichever is
    # larger) — an engine that drifted to several times the reference's own rounding noise on a large-logit network fails here
    ref32 = float(np.abs(z["logits_f32"].astype(np.float64) - ref).max())
    assert err <= max(1e-5, 1.25 * ref32), "logit err %.3e against the reference's own fp32-vs-fp64 distance %.3e" % (err, ref32)
    if scale <= 2.0:
        assert err <= 1e-5, "absolute 1e-5 wherever max|logit| <= 2 (err %.3e)" % err
    # hipGraph replay gives bit-identical logits
    out2 = eng.forward(int_x, cat_x, meta["choice"], graph=True).clone()
    out3 = eng.forward(int_x, cat_x, meta["choice"], graph=True)
    assert torch.equal(out2, out3)
    assert float((out2 - out).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("path", NPZ, ids=IDS)
def test_gradients_match_reference(path):
    z, meta = load_golden(path)
    eng = build_engine(z, meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    cp = eng.forward_backward(int_x, cat_x, y, meta["choice"])
    torch.cuda.synchronize()
    assert abs(float(cp.loss.item()) - float(z["loss_f64"])) <= 1e-5 * max(1.0, abs(float(z["loss_f64"])))
    written = set(cp.ctx.grad_params) | {"_final.weight", "_final.bias"}
    bad = []
    for k, (dot, nrm) in meta["grads"].items():
        if k.startswith("_embedding."):
            continue
        assert k in written, "no gradient produced for %s" % k
        d, n = proj_checksum(k, eng.grads[k])
        tol = 2e-5 * max(nrm, 1e-6)
        if abs(n - nrm) > tol or abs(d - dot) > tol * max(1.0, np.sqrt(eng.grads[k].numel()) / 8):
            bad.append((k, "norm %.6g vs %.6g" % (n, nrm), "dot %.6g vs %.6g" % (d, dot)))
    assert not bad, "%d/%d parameter gradients differ: %s" % (len(bad), len(meta["grads"]), bad[:10])
    for k in meta["grad_none"]:
        if not k.startswith("_embedding."):
            assert k not in written, "gradient written for %s but autograd leaves it None" % k
    # embedding gradients: dense scatter of the engine's [B,Fs,16] gradient == reference dense table gradient
    g = cp.sparse0.grad_tensor().view(int_x.shape[0], -1, 16).cpu().double()
    for f in range(cat_x.shape[1]):
        k = "_embedding.%d.weight" % f
        dense = torch.zeros(meta["tables"][f], 16, dtype=torch.float64).index_add_(0, torch.tensor(z["cat_x"][:, f]), g[:, f])
        d, n = proj_checksum(k, dense)
        dot, nrm = meta["grads"][k]
        assert abs(n - nrm) <= 2e-5 * max(nrm, 1e-6) and abs(d - dot) <= 2e-5 * max(nrm, 1e-6), k


@pytest.mark.parametrize("path", NPZ, ids=IDS)
@pytest.mark.parametrize("graph", [False, True])
def test_three_adagrad_steps(path, graph):
    z, meta = load_golden(path)
    if graph and meta["mode"] != "fixed":
        pytest.skip("graph capture is used for the fixed sub-network step")
    eng = build_engine(z, meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    losses, norms = [], []
    for _ in range(meta["n_steps"]):
        loss = eng.train_step(int_x, cat_x, y, lr=meta["lr"], choice=meta["choice"], clip=5.0, eps=1e-2, graph=graph)
        torch.cuda.synchronize()
        losses.append(float(loss.item()))
        norms.append(float(eng.clip_out[1].item()))
    # every one of the 13 reference trajectories is checked, every step, every parameter (checksum AND norm)
    check_trajectory(z, meta, losses, norms, eng.state_dict(), rel_params=TRAJ_REL_PARAMS, rel_delta=TRAJ_REL_DELTA, rel_loss=TRAJ_REL_LOSS)


# ---------------------------------------------------------------------------------------------------------------
# the drop-in nn.Module path: reference-style harness (torch BCE loss, clip_grad_norm_, torch.optim.Adagrad) on top of
# nasrec_amd.supernet.SuperNet, compared with what the real reference produced with the same harness
# ---------------------------------------------------------------------------------------------------------------
MODULE_CASES = ["fixed_criteo_xlarge", "fixed_avazu_xlarge", "supernet_xlarge_any", "supernet_autoctr_single", "supernet_xlarge_any_last2",
                "fixed_criteo_xlarge_last2"]


@pytest.mark.parametrize("case", MODULE_CASES)
def test_module_dropin_with_unchanged_torch_harness(case):
    from nasrec_amd.supernet.supernet import SuperNet
    from helpers import GOLDEN
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    fixed = meta["mode"] == "fixed"
    Fs = z["cat_x"].shape[1]
    model = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                     activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=Fs,
                     path_sampling_strategy="fixed-path" if fixed else "full-path", fixed=fixed,
                     fixed_choice=meta["choice"] if fixed else None, last_n_blocks_out=meta.get("last_n_blocks_out", 1)).to("cuda")
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    with torch.no_grad():
        model(int_x, cat_x)  # warm-up (train_utils.py:392-433)
    if not fixed:  # pin the path like eval_subnet_from_supernet.py:101-103 / searcher_utils.py:71
        model.configure_path_sampling_strategy("fixed-path")
        model.configure_choice(meta["choice"])
    assert [n for n, _ in model.named_parameters()] == meta["param_order"]
    model.load_state_dict({k: torch.tensor(O.seeded_param(k, s), dtype=torch.float32) for k, s in meta["param_shapes"].items()}, strict=True)
    model.train()
    out = model(int_x, cat_x)
    ref = z["logits_f64"]
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(out.detach().cpu().numpy().astype(np.float64) - ref).max()) <= 1e-5 * scale
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, y)
    loss.backward()
    none = sorted(n for n, p in model.named_parameters() if p.grad is None)
    assert none == sorted(meta["grad_none"])
    for k, (dot, nrm) in meta["grads"].items():
        d, n = proj_checksum(k, dict(model.named_parameters())[k].grad)
        assert abs(n - nrm) <= 2e-5 * max(nrm, 1e-6), (k, n, nrm)
    # unchanged harness: zero_grad -> forward -> loss -> backward -> clip -> Adagrad
    opt = torch.optim.Adagrad(model.parameters(), lr=meta["lr"], eps=1e-2)
    losses, norms = [], []
    for _ in range(meta["n_steps"]):
        opt.zero_grad()
        l = torch.nn.functional.binary_cross_entropy_with_logits(model(int_x, cat_x), y)
        l.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)))
        opt.step()
        losses.append(float(l))
    sd = model.state_dict()
    check_trajectory(z, meta, losses, norms, sd, rel_params=TRAJ_REL_PARAMS, rel_delta=TRAJ_REL_DELTA, rel_loss=TRAJ_REL_LOSS)
    # the engine's fused step (row-sparse tables) lands on the same parameters as the dense-gradient harness
    model2 = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                      activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=Fs,
                      path_sampling_strategy="fixed-path" if fixed else "full-path", fixed=fixed,
                      fixed_choice=meta["choice"] if fixed else None, last_n_blocks_out=meta.get("last_n_blocks_out", 1)).to("cuda")
    with torch.no_grad():
        model2(int_x, cat_x)
    if not fixed:
        model2.configure_path_sampling_strategy("fixed-path")
        model2.configure_choice(meta["choice"])
    model2.load_state_dict({k: torch.tensor(O.seeded_param(k, s), dtype=torch.float32) for k, s in meta["param_shapes"].items()}, strict=True)
    for _ in range(meta["n_steps"]):
        model2.engine_train_step(int_x, cat_x, y.view(-1), lr=meta["lr"])
    torch.cuda.synchronize()
    sd2 = model2.state_dict()
    for k in sd:
        a, b = sd[k].double().cpu(), sd2[k].double().cpu()
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(a.abs().max())), k


@pytest.mark.parametrize("case", ["fixed_criteo_xlarge", "supernet_autoctr_single"])
def test_model_survives_jit_trace(case):
    """the unchanged reference CLIs trace the model: `writer.add_graph(model, (int_x, cat_x))` (main_train.py:137,
    train_supernet.py:192) is torch.jit.trace(model, args, strict=False) underneath, and fvcore's FlopCountAnalysis
    (train_utils.py:444-452) is torch.jit._get_trace_graph.  Both must go through (the network shows up as one opaque node), return
    the eager logits, and leave the model usable for training afterwards."""
    from nasrec_amd.supernet.supernet import SuperNet
    from helpers import GOLDEN
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    fixed = meta["mode"] == "fixed"
    model = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                     activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=z["cat_x"].shape[1],
                     path_sampling_strategy="fixed-path" if fixed else "full-path", fixed=fixed,
                     fixed_choice=meta["choice"] if fixed else None).to("cuda")
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    with torch.no_grad():
        model(int_x, cat_x)
    if not fixed:
        model.configure_path_sampling_strategy("fixed-path")
        model.configure_choice(meta["choice"])
    model.load_state_dict({k: torch.tensor(O.seeded_param(k, shp)) for k, shp in meta["param_shapes"].items()})
    with torch.no_grad():
        eager = model(int_x, cat_x).clone()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # TracerWarnings about Python values: expected, the reference's own forward raises them too
        traced = torch.jit.trace(model, (int_x, cat_x), strict=False)
        graph, out = torch.jit._get_trace_graph(model, (int_x, cat_x))
    kinds = [n.kind() for n in traced.graph.nodes()]
    assert any("PythonOp" in k for k in kinds), kinds
    assert torch.equal(out, eager)
    assert float(np.abs(eager.cpu().numpy().astype(np.float64) - z["logits_f64"]).max()) <= 1e-5 * max(1.0, float(np.abs(z["logits_f64"]).max()))
    # training still works after the trace
    loss = torch.nn.functional.binary_cross_entropy_with_logits(model(int_x, cat_x), y)
    loss.backward()
    assert abs(float(loss) - float(z["loss_f64"])) <= 1e-5 * max(1.0, abs(float(z["loss_f64"])))
    assert model._final.weight.grad is not None


def test_data_parallel_code_path_single_rank():
    """The N>1 step (all-reduce of the flat gradient arena, all-gather of ids + embedding-row gradients, optimizer over
    the gathered global batch) run in a single-rank RCCL group must land on the same parameters as the plain step."""
    import torch.distributed as dist
    from helpers import GOLDEN
    from nasrec_amd.parallel import DataParallelStep
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_xlarge.npz"))
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1)
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        # (plain step; exchange forced on with a one-rank gather as a copy kernel; exchange forced on with RCCL's all-gather / all-reduce
        # kernels inside the captured step — what N ranks run, as far as one rank can)
        for force, real in ((False, False), (True, False), (True, True)):
            eng = build_engine(z, meta)
            dp = DataParallelStep(eng, dict(meta["choice"]), int_x.shape[0], clip=5.0, eps=1e-2, graph=True, force_exchange=force, real_collectives=real)
            assert dp.exchange == force
            for _ in range(3):
                # an EQUAL choice rebuilt every step must hit the same plan (and the same captured graph): plans are keyed on content
                dp.step(int_x, cat_x, y, lr=0.05, choice=json.loads(json.dumps(meta["choice"])) if force else None)
            torch.cuda.synchronize()
            outs.append(eng.state_dict())
            if force:
                assert len(dp._plans) == 1
                plan = dp._last[1]
                assert isinstance(plan.step_graph, torch.cuda.CUDAGraph), "the exchange step should be captured as one graph"
                assert dp.ids_half is not None, "a global batch this small takes the two-halves row dedup (its id half is a launch of the exchange step: NASREC_DP_IDS_MODE)"
                # the last pieces' dense gradients ride in the row-gradient all-gather: the pieces in front of them are all-reduced
                assert dp.tail_n == sum(n for _, rg in plan.segments[plan.first_packed:] for _, n in rg) and 0 < dp.tail_n <= 65536
                assert 1 <= plan.first_packed < len(plan.segments)
                # the pieces' all-reduce ranges: disjoint, exactly the parameters some launch writes, each final at its cut
                names = sorted(eng.offsets, key=lambda n: eng.offsets[n])
                sent = torch.zeros(eng.flat_numel, dtype=torch.int32)
                assert len(plan.segments) >= 2 and plan.cuts == sorted(plan.cuts)
                for (_, ranges), end in zip(plan.segments, plan.cuts):
                    for off, n in ranges:
                        sent[off:off + n] += 1
                        for nm in names:
                            if eng.offsets[nm] < off + n and off < eng.offsets[nm] + eng.params[nm].numel():
                                # (a range may run over gradients NO launch writes — zero on every rank — never over one that is still to come)
                                assert nm not in plan.ready or plan.ready[nm] < end, (nm, end)
                assert int(sent.max()) == 1
                for nm, idx in plan.ready.items():
                    o = eng.offsets[nm]
                    assert int(sent[o:o + eng.params[nm].numel()].min()) == 1, nm
                never = [nm for nm in names if nm not in plan.ready]
                for nm in never:  # grad None in the reference: zero here, not sent
                    assert float(eng.grads[nm].abs().max()) == 0.0, nm
                # ... and the eager form of the same exchange step (work handles instead of a captured graph) lands on the same bits
                eng2 = build_engine(z, meta)
                dp2 = DataParallelStep(eng2, meta["choice"], int_x.shape[0], clip=5.0, eps=1e-2, graph=False, force_exchange=True)
                for _ in range(3):
                    dp2.step(int_x, cat_x, y, lr=0.05)
                torch.cuda.synchronize()
                sd2 = eng2.state_dict()
                for k in sd2:
                    assert torch.equal(sd2[k], outs[-1][k]), k
        for k in outs[0]:
            assert torch.allclose(outs[0][k], outs[1][k], rtol=0, atol=1e-6), k
            assert torch.equal(outs[1][k], outs[2][k]), k  # the library's one-rank collectives move the same bits as the copy
    finally:
        if own_pg:
            dist.destroy_process_group()


def test_exchange_step_falls_back_to_eager_when_the_capture_is_refused(monkeypatch):
    """The N-rank batch-256 step is captured into ONE graph with RCCL's collectives inside (parallel._capture); whether RCCL on N
    ranks accepts that capture has never been observed (no multi-GPU box in six rounds).  If the platform refuses, the step must
    run eagerly — work handles, the same sequence — and land on the SAME bits.  Here the refusal is forced: torch.cuda.graph raises
    inside the capture, with the real collectives on."""
    import warnings
    import torch.distributed as dist
    from helpers import GOLDEN
    from nasrec_amd import parallel
    from nasrec_amd.parallel import DataParallelStep
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_xlarge.npz"))
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1)
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29543", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        for refuse in (False, True):
            eng = build_engine(z, meta)
            if refuse:
                class Refused:
                    def __init__(self, *a, **kw):
                        pass

                    def __enter__(self):
                        raise RuntimeError("capture refused (forced by the test)")

                    def __exit__(self, *a):
                        return False
                monkeypatch.setattr(parallel.torch.cuda, "graph", Refused)
            dp = DataParallelStep(eng, dict(meta["choice"]), int_x.shape[0], clip=5.0, eps=1e-2, graph=True, force_exchange=True, real_collectives=True)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                for _ in range(3):
                    dp.step(int_x, cat_x, y, lr=0.05)
            torch.cuda.synchronize()
            plan = dp._last[1]
            if refuse:
                monkeypatch.undo()
                assert plan.step_graph is False, "the refused capture must leave the plan on the eager exchange"
                assert any("capture" in str(w.message) for w in caught), "the fallback must be announced, not silent"
            else:
                assert isinstance(plan.step_graph, torch.cuda.CUDAGraph)
            eng.check_indices()
            outs.append(eng.state_dict())
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), "eager fallback differs from the captured step: %s" % k
    finally:
        if own_pg:
            dist.destroy_process_group()


@pytest.mark.parametrize("case", ["fixed_criteo_xlarge", "fixed_kdd_autoctr", "fixed_criteo_xlarge_ln", "fixed_avazu_xlarge"])
def test_level_scheduled_step_is_bit_identical_to_one_launch_per_operator(case):
    """nasrec_amd/schedule.py + NASREC_OP_WORKLIST: a fixed sub-network's step with its independent operators packed into
    heterogeneous launches (dense node || sparse node, weight gradients and split-K second passes beside the backward chain) must
    reproduce the one-launch-per-operator program BIT FOR BIT — logits, loss, every gradient, parameters after two steps, eager
    and graph — because the same bodies run on the same operands in the same summation order; only kernel boundaries move."""
    z, meta = load_golden(os.path.join(os.path.dirname(NPZ[0]), case + ".npz"))
    int_x, cat_x = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda()
    y = torch.tensor(z["y"]).cuda().view(-1)
    res = []
    for scheduled in (False, True):
        for graph in (False, True):
            eng = build_engine(z, meta)
            eng.level_schedule = scheduled
            eng.park_weight_grads = False  # the baseline runs the SAME operator list one launch each (no parked zmode batches)
            eng.mha_bwd_form = 4           # ... and the Transformer-backward form the worklist launches run
            logits = eng.forward(int_x, cat_x, meta["choice"]).clone()
            cp = eng.forward_backward(int_x, cat_x, y, meta["choice"])
            torch.cuda.synchronize()
            grads, sg, loss = eng.flat_g.clone(), cp.sparse0.grad_tensor().clone(), float(cp.loss.item())
            nwl = sum(isinstance(d, L.WorklistDesc) for d in (cp.fb.descs if cp.fb is not None else cp.fwd.descs + cp.bwd.descs))
            for _ in range(2):
                eng.train_step(int_x, cat_x, y, 0.05, meta["choice"], graph=graph)
            torch.cuda.synchronize()
            res.append((scheduled, nwl, logits, grads, sg, loss, eng.flat_p.clone(), [t.clone() for t in eng.tables]))
    base = res[0]
    assert base[1] == 0 and res[2][1] >= 8, "the scheduled plan should consist of worklist launches"
    for r in res[1:]:
        assert torch.equal(r[2], base[2]), "logits"
        assert torch.equal(r[3], base[3]) and torch.equal(r[4], base[4]) and r[5] == base[5], "gradients / loss"
        assert torch.equal(r[6], base[6]), "dense parameters after two steps"
        for a, b in zip(r[7], base[7]):
            assert torch.equal(a, b)


def test_level_schedule_respects_every_dependency():
    """the scheduler's footprint model against a brute-force check: running the levels of the Criteo best-1shot step in REVERSED
    order inside each level (any order inside a level must be legal) gives the same bits as program order"""
    from nasrec_amd import schedule as S
    z, meta = load_golden(os.path.join(os.path.dirname(NPZ[0]), "fixed_criteo_xlarge.npz"))
    int_x, cat_x = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda()
    y = torch.tensor(z["y"]).cuda().view(-1)
    eng = build_engine(z, meta)
    eng.level_schedule = False
    cp = eng.forward_backward(int_x, cat_x, y, meta["choice"])
    torch.cuda.synchronize()
    want = (cp.logits.clone(), eng.flat_g.clone(), cp.sparse0.grad_tensor().clone())
    from nasrec_amd.engine import Program
    descs = cp.fwd.descs + cp.bwd.descs
    nodes = S.expand(descs)  # (split-K launches stay whole here: every node is a launchable descriptor)
    nodes = [S.Node(d) for d in descs]
    for i, n in enumerate(nodes):
        n.index = i
    nl = S.assign_levels(nodes)
    assert nl < len(nodes) // 2, "the step is far from a chain"
    order = []
    for lv in range(nl):
        order += [n.desc for n in reversed([n for n in nodes if n.level == lv])]
    eng.flat_g.zero_()
    sp = eng._sp()
    eng._stage_inputs(sp, cp, int_x, cat_x, y)
    Program(order).run(sp)
    torch.cuda.synchronize()
    assert torch.equal(cp.logits, want[0]) and torch.equal(eng.flat_g, want[1]) and torch.equal(cp.sparse0.grad_tensor(), want[2])


def test_level_launch_items_are_ordered_by_duration_and_the_order_changes_no_bits(monkeypatch):
    """schedule.pack: within a worklist launch the items go by decreasing estimated duration (schedule._cost: the DotProduct
    backward before the small products, a Transformer body first), and that order is only a dispatch order — the step under
    NASREC_WL_COST=size (the previous order) gives the same logits and gradients bit for bit"""
    from nasrec_amd import schedule as S, _lib as L
    z, meta = load_golden(os.path.join(os.path.dirname(NPZ[0]), "fixed_criteo_xlarge.npz"))
    int_x, cat_x = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda()
    y = torch.tensor(z["y"]).cuda().view(-1)
    out = {}
    for model in ("time", "size"):
        monkeypatch.setattr(S, "_COST_MODEL", model)
        eng = build_engine(z, meta)
        cp = eng.forward_backward(int_x, cat_x, y, meta["choice"])
        torch.cuda.synchronize()
        out[model] = (cp.logits.clone(), eng.flat_g.clone(), cp.sparse0.grad_tensor().clone())
        prog = cp.fb if getattr(cp, "fb", None) is not None else None
        descs = prog.descs if prog is not None else cp.fwd.descs + cp.bwd.descs
        lists = [d for d in descs if isinstance(d, L.WorklistDesc)]
        assert lists, "the level-scheduled step has worklist launches"
        seen_dot_first = False
        for d in lists:
            costs = [S._cost(n) for n in d.nodes]
            assert costs == sorted(costs, reverse=True), (model, costs)
            kinds = [n.desc.kind for n in d.nodes]
            if model == "time" and L.OP_DOT_TRI_BWD in kinds and any(k == L.OP_GEMM for k in kinds):
                small = [i for i, n in enumerate(d.nodes) if n.desc.kind == L.OP_GEMM and S._cost(n) < 10000]
                seen_dot_first = seen_dot_first or (small and kinds.index(L.OP_DOT_TRI_BWD) < min(small))
        if model == "time":
            assert seen_dot_first, "a DotProduct backward is dispatched before the small products of its level"
    for a, b in zip(out["time"], out["size"]):
        assert torch.equal(a, b)


def test_finetune_last_layer_mode_runs_only_the_final_backward():
    """SuperNet.set_mode_to_finelune_last_only (the searcher's candidate evaluation, eval_subnet_from_supernet.py): the gradient
    of _final equals the full backward's, every other parameter keeps grad None, and only the final-logit backward is launched."""
    from nasrec_amd.supernet.supernet import SuperNet
    z, meta = load_golden(os.path.join(os.path.dirname(NPZ[0]), "fixed_criteo_autoctr.npz"))
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1, 1)
    torch.manual_seed(0)
    m = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                 activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=cat_x.shape[1],
                 path_sampling_strategy="fixed-path", fixed=True, fixed_choice=meta["choice"]).cuda()
    with torch.no_grad():
        m(int_x, cat_x)
    loss_fn = torch.nn.BCEWithLogitsLoss()
    loss_fn(m(int_x, cat_x), y).backward()
    full = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    m.set_mode_to_finelune_last_only()
    loss_fn(m(int_x, cat_x), y).backward()
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert set(got) == {"_final.weight", "_final.bias"}
    for k in got:
        assert torch.equal(got[k], full[k]), k
    cp = m._engine.compile(m.choice, int_x.shape[0], train=True)
    assert len(cp.bwd_final_only.descs) == 1
    m.set_mode_to_normal_mode()
    m.zero_grad(set_to_none=True)
    loss_fn(m(int_x, cat_x), y).backward()
    assert len([1 for p in m.parameters() if p.grad is not None]) == len(full)


def test_place_embedding_on_cpu_mode_matches_the_reference_vectors():
    """SuperNet(place_embedding_on_cpu=True) (supernet.py:231,418-428,826-840): tables stay in host memory, ids go to the host for the
    lookup, rows come to the device; logits and every gradient — the dense table gradients on the host included — equal the
    reference's."""
    from nasrec_amd.supernet.supernet import SuperNet
    from helpers import GOLDEN
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_kdd_xlarge.npz"))
    Fs = z["cat_x"].shape[1]
    model = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=meta["use_layernorm"],
                     activation=meta["activation"], num_embeddings=meta["tables"], sparse_input_size=Fs, path_sampling_strategy="fixed-path",
                     fixed=True, fixed_choice=meta["choice"], place_embedding_on_cpu=True).to("cuda")
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    with torch.no_grad():
        model(int_x, cat_x)
    assert all(e.weight.device.type == "cpu" for e in model._embedding) and model._final.weight.is_cuda
    assert [n for n, _ in model.named_parameters()] == meta["param_order"]
    model.load_state_dict({k: torch.tensor(O.seeded_param(k, s), dtype=torch.float32) for k, s in meta["param_shapes"].items()}, strict=True)
    assert all(e.weight.device.type == "cpu" for e in model._embedding)
    model.train()
    out = model(int_x, cat_x)
    ref = z["logits_f64"]
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(out.detach().cpu().numpy().astype(np.float64) - ref).max()) <= 1e-5 * scale
    torch.nn.functional.binary_cross_entropy_with_logits(out, y).backward()
    named = dict(model.named_parameters())
    assert sorted(n for n, p in named.items() if p.grad is None) == sorted(meta["grad_none"])
    for k, (dot, nrm) in meta["grads"].items():
        g = named[k].grad
        assert g.device == named[k].device
        d, n = proj_checksum(k, g)
        assert abs(n - nrm) <= 2e-5 * max(nrm, 1e-6) and abs(d - dot) <= 2e-5 * max(nrm, 1e-6), k
    with pytest.raises(L.EngineError):
        model.engine_train_step(int_x, cat_x, y.view(-1), lr=0.01)


def test_dead_forward_elimination_changes_nothing_observable():
    """opt-in NASREC_DCE / SupernetEngine.dead_code_elimination: forward operators nobody reads (a block no later block selects, with
    last_n_blocks_out = 1) are dropped — logits, loss, every gradient and the parameters after two steps stay bit-identical, and the
    Criteo best-1shot plan does drop launches (its block 5 is computed by the reference and thrown away)"""
    import json as _json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ca = _json.load(open(os.path.join(root, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
    choice = {"macro": ca["macro"], "micro": ca["micro"]}
    tables = [1000] * 26
    int_x, cat_x, y = O.synthetic_batch(64, 13, tables, seed=5)
    int_x, cat_x, y = int_x.cuda(), cat_x.cuda(), y.view(-1).cuda()
    res = []
    for dce in (False, True):
        cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
        eng = SupernetEngine(cfg, 13, 26, tables, warm_choice=choice)
        eng.init_weights(seed=3)
        eng.dead_code_elimination = dce
        logits = eng.forward(int_x, cat_x, choice).clone()
        cp = eng.forward_backward(int_x, cat_x, y, choice)
        torch.cuda.synchronize()
        g, sg, loss, ndead = eng.flat_g.clone(), cp.sparse0.grad_tensor().clone(), float(cp.loss.item()), len(cp.dead_forward)
        for _ in range(2):
            eng.train_step(int_x, cat_x, y, 0.05, choice, graph=True)
        torch.cuda.synchronize()
        res.append((logits, g, sg, loss, eng.flat_p.clone(), ndead))
    a, b = res
    assert a[5] == 0 and b[5] >= 5, "block 5 of the Criteo best-1shot sub-network is dead code"
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3] and torch.equal(a[4], b[4])

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the real reference
(/root/reference, read-only, present only in the build container).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference never travels: only inputs + expected outputs are stored (np.savez_compressed), weights are
re-created on both sides from ``oracle.nasrec_oracle.seeded_param(state_dict_key, shape)``.

Fixture families:
  fixed_<cfg>.npz      G2  the six configs/*.json sub-nets (tables capped), B=8: logits fp32/fp64, loss,
                           per-parameter grad checksums, 3 Adagrad steps (torch.optim.Adagrad + clip_grad_norm_)
  supernet_<case>.npz  G3  weight-sharing supernet (LN on) under explicit choices
  samplers.json        G4  np.random-seeded choice traces per strategy
  lr.json              G6  LR scheduler sequences
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.setup()  # /root/reference first, the repository root (and its `nasrec/` shim) OFF sys.path

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nasrec.supernet.supernet import SuperNet, ops_config_lib  # noqa: E402  the REAL reference
from nasrec.utils.lr_schedule import ConstantWithWarmup, CosineAnnealingWarmupRestarts  # noqa: E402

_refimport.assert_reference_modules()
O = _refimport.load_oracle()  # by file path: name-seeded weights / synthetic batches / path samplers

torch.set_num_threads(1)

DATASETS = {
    "criteo": dict(Fd=13, Fs=26, tables=[1461, 584, 10131227, 2202609, 306, 25, 12518, 634, 4, 93146, 5684, 8351593, 3195, 28,
                                         14993, 5461307, 11, 5653, 2174, 5, 7046548, 19, 16, 286182, 106, 142573]),
    "avazu": dict(Fd=1, Fs=23, tables=[10000, 241, 8, 8, 4738, 7746, 27, 8553, 560, 37, 2686409, 6729487, 8252, 6, 5, 2627, 9,
                                       10, 436, 5, 69, 173, 61]),
    "kdd": dict(Fd=3, Fs=10, tables=[26274, 641708, 14848, 22122011, 1188090, 3735797, 2934102, 20004011, 4, 8]),
}
CAP = 997
OUT = os.environ.get("GOLDEN_OUT", HERE)  # where the fixtures are written (default: next to this script)
LR = 1e-3  # step size of the 3-step trajectories: the name-seeded weights are not a trained optimum, and at the recipe's
# lr 0.16 the REFERENCE itself diverges on them within 3 steps (losses 6.9 -> 15 368 -> 93 804 on fixed_criteo_xlarge), which
# leaves nothing to compare; at 1e-3 all 13 reference trajectories stay bounded (largest loss 113) and every step is checked


def batch(ds, B, seed):
    d = DATASETS[ds]
    tables = [min(n, CAP) for n in d["tables"]]
    int_x, cat_x, y = O.synthetic_batch(B, d["Fd"], tables, seed=seed, zero_dense=(ds == "avazu"))
    # exercise index edge cases: 0 ("missing"), n-1, and duplicates inside a field
    cat_x[0, :] = 0
    cat_x[1, :] = torch.tensor(tables) - 1
    cat_x[2, :] = cat_x[3, :]
    return tables, int_x, cat_x, y


def load_seeded(model, dtype):
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.tensor(O.seeded_param(k, v.shape), dtype=dtype))
    return {k: list(v.shape) for k, v in sd.items()}


def proj_checksum(name, t):
    """<t, r_name> with r_name a name-seeded N(0,1) vector, plus the L2 norm."""
    a = t.detach().double().numpy().reshape(-1)
    r = O.seeded_array("chk:" + name, a.shape)
    return [float(np.dot(a, r)), float(np.linalg.norm(a))]


def clean_choice(c):
    def cv(v):
        if isinstance(v, dict):
            return {k: cv(x) for k, x in v.items()}
        if isinstance(v, (list, tuple, np.ndarray)):
            return [cv(x) for x in np.asarray(v).tolist()] if not isinstance(v, list) else [cv(x) for x in v]
        if isinstance(v, (np.integer,)):
            return int(v)
        return v
    return cv(c)


def run_case(model_fn, ds, B, seed, out_name, keep_blocks=False, n_steps=3, lr=None, extra_meta=None):
    lr = LR if lr is None else lr
    tables, int_x, cat_x, y = batch(ds, B, seed)
    res = {}
    traj = {}
    shapes = None
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        model = model_fn(tables)
        with torch.no_grad():
            model(int_x, cat_x)  # warm-up: materialise lazies, delete unused projections
        if dtype == torch.float64:
            model = model.double()
        shapes = load_seeded(model, dtype)
        model.train()
        blocks = {}
        hooks = []
        if keep_blocks and tag == "f64":
            for i, blk in enumerate(model._blocks):
                hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: blocks.__setitem__(i, (o[0].detach().clone(), o[1].detach().clone()))))
        xi = int_x.to(dtype)
        logits = model(xi, cat_x)
        for h in hooks:
            h.remove()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y.to(dtype))
        res["logits_" + tag] = logits.detach().numpy().copy()
        res["loss_" + tag] = np.array(float(loss.detach()))
        if tag == "f64":
            model.zero_grad()
            loss.backward()
            grads, none = {}, []
            for n, p in model.named_parameters():
                if p.grad is None:
                    none.append(n)
                else:
                    grads[n] = proj_checksum(n, p.grad)
            res_meta_grads = grads
            res_meta_none = none
            for i, (d, s) in blocks.items():
                res["block%d_dense" % i] = d.numpy().astype(np.float64)
                res["block%d_sparse" % i] = s.numpy().astype(np.float64)
        # n_steps real optimizer steps (train_utils.py:262-286, main_train.py:152), in BOTH precisions: the fp64 run is the
        # expected value, the reference's own fp32 run measures how far fp32 arithmetic drifts from it (the noise floor the
        # engine's tolerance is set against, stored as *_f32)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        opt = torch.optim.Adagrad(model.parameters(), lr=lr, eps=1e-2)
        losses, norms = [], []
        for _ in range(n_steps):
            opt.zero_grad()
            out = model(xi, cat_x)
            l = torch.nn.functional.binary_cross_entropy_with_logits(out, y.to(dtype))
            l.backward()
            tn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
            opt.step()
            losses.append(float(l.detach()))
            norms.append(float(tn))
        suffix = "" if tag == "f64" else "_f32"
        res["step_losses" + suffix] = np.array(losses)
        res["step_gradnorms" + suffix] = np.array(norms)
        traj[tag] = dict(after={n: proj_checksum(n, p) for n, p in model.named_parameters()},
                         delta={n: proj_checksum(n, p.detach() - before[n]) for n, p in model.named_parameters()})
        if tag == "f64":
            with torch.no_grad():
                res["logits_after_f64"] = model(xi, cat_x).numpy().copy()
    after, delta = traj["f64"]["after"], traj["f64"]["delta"]
    # drift of the reference's own fp32 trajectory from its fp64 one, relative to each parameter's norm / update norm
    drift_p = max(max(abs(traj["f32"]["after"][n][i] - after[n][i]) for i in (0, 1)) / max(after[n][1], 1e-12) for n in after)
    # updates are compared on the scale max(|update|, 1 % of a full-size Adagrad update lr * n_steps * sqrt(numel)): parameters
    # whose true gradient is ~0 (e.g. a bias in front of a LayerNorm) move by rounding noise only
    def dscale(n):
        numel = int(np.prod(shapes[n])) if n in shapes else 1
        return max(delta[n][1], 1e-2 * lr * n_steps * np.sqrt(numel))
    drift_d = max(max(abs(traj["f32"]["delta"][n][i] - delta[n][i]) for i in (0, 1)) / dscale(n) for n in delta)
    meta = dict(dataset=ds, B=B, seed=seed, tables=tables, param_shapes=shapes, grads=res_meta_grads, grad_none=res_meta_none,
                params_after=after, params_delta=delta, ref_fp32_drift=dict(params=drift_p, delta=drift_d), lr=lr, n_steps=n_steps, param_order=[n for n, _ in model.named_parameters()])
    meta.update(extra_meta or {})
    np.savez_compressed(os.path.join(OUT, out_name), int_x=int_x.numpy(), cat_x=cat_x.numpy(), y=y.numpy(),
                        meta=np.array(json.dumps(meta)), **res)
    print("wrote", out_name, "logits f64[:3]", res["logits_f64"][:3, 0], "max|f32-f64|",
          float(np.abs(res["logits_f32"] - res["logits_f64"]).max()), "losses", res["step_losses"], "fp32 drift: params %.2e delta %.2e" % (drift_p, drift_d))


def fixed_cases():
    cfg_dir = "/root/reference/nasrec/configs"
    for ds in ("criteo", "avazu", "kdd"):
        for space in ("xlarge", "autoctr"):
            path = "%s/%s/ea_%s_kaggle_%s_best_1shot.json" % (cfg_dir, ds, ds, space)
            ch = json.load(open(path))
            for use_ln in ((False, True) if (ds == "criteo" and space == "xlarge") else (False,)):
                def mk(tables, ch=ch, use_ln=use_ln, ds=ds):
                    # main_train.py:258-269 (use_layernorm hard-coded False there; True kept as an extra case)
                    return SuperNet(num_blocks=ch["num_blocks"], ops_config=ops_config_lib[ch["config"]], use_layernorm=use_ln,
                                    activation="relu", num_embeddings=tables, sparse_input_size=DATASETS[ds]["Fs"],
                                    path_sampling_strategy="fixed-path", fixed=True, fixed_choice=ch)
                name = "fixed_%s_%s%s.npz" % (ds, space, "_ln" if use_ln else "")
                run_case(mk, ds, 8, 100 + len(name), name, keep_blocks=(ds == "criteo" and space == "xlarge"),
                         extra_meta=dict(mode="fixed", use_layernorm=use_ln, activation="relu", choice=clean_choice(
                             {"macro": ch["macro"], "micro": ch["micro"]}), num_blocks=ch["num_blocks"], config=ch["config"]))


def supernet_cases():
    cases = [
        ("xlarge_full", "criteo", "xlarge", 3, "full", "relu"),
        ("xlarge_single", "criteo", "xlarge", 3, "single", "relu"),
        ("xlarge_any", "criteo", "xlarge", 3, "any", "silu"),
        ("xlargezeros_any", "avazu", "xlarge-zeros", 3, "any", "relu"),
        ("autoctr_any", "kdd", "autoctr", 7, "any", "relu"),
        ("autoctr_single", "kdd", "autoctr", 4, "single", "relu"),
    ]
    for idx, (tag, ds, space, nb, kind, act) in enumerate(cases):
        ops = ops_config_lib[space]
        cfg = O.NetCfg(nb, O.ops_config_lib[space], True, act)
        np.random.seed(7 + idx)
        if kind == "full":
            choice = O.full_path_choice(cfg)
        else:
            strat = {"single": "single-path", "any": "any-path"}[kind]
            choice = O.PathSampler(cfg, strat, "binomial-0.5" if idx % 2 else "uniform").sample()
        choice = clean_choice(choice)

        def mk(tables, ops=ops, nb=nb, ds=ds, act=act, choice=choice):
            # train_supernet.py:242-254 then warm-up under full-path (train_utils.py:413-433)
            m = SuperNet(num_blocks=nb, ops_config=ops, use_layernorm=True, activation=act, num_embeddings=tables,
                         sparse_input_size=DATASETS[ds]["Fs"], path_sampling_strategy="full-path", fixed=False)
            d = DATASETS[ds]
            with torch.no_grad():
                m(torch.zeros(2, d["Fd"]), torch.zeros(2, d["Fs"], dtype=torch.long))
            # pin the path: eval_subnet_from_supernet.py:101-103 / searcher_utils.py:71
            m.configure_path_sampling_strategy("fixed-path")
            m.configure_choice(choice)
            return m
        run_case(mk, ds, 4, 200 + idx, "supernet_%s.npz" % tag,
                 extra_meta=dict(mode="supernet", use_layernorm=True, activation=act, choice=choice, num_blocks=nb, config=space))


def last_n_cases():
    """last_n_blocks_out = 2 (supernet.py:592-598 / 657-664: the final layer reads the last TWO blocks; their sparse outputs are
    concatenated on the last dim, which interleaves them per token and needs equal token counts):
      supernet_xlarge_any_last2   weight-sharing supernet (every block's sparse output has the space's maximum token count)
      fixed_criteo_xlarge_last2   the Criteo xlarge best-1shot sub-network with the last-but-one block's sparse node set to the last
                                  block's token count (the published choice has 16 vs 48 tokens there and fails in torch.cat)"""
    ops = ops_config_lib["xlarge"]
    cfg = O.NetCfg(3, O.ops_config_lib["xlarge"], True, "silu")
    np.random.seed(31)
    choice = clean_choice(O.PathSampler(cfg, "any-path", "uniform").sample())

    def mk(tables, choice=choice):
        m = SuperNet(num_blocks=3, ops_config=ops, use_layernorm=True, activation="silu", num_embeddings=tables,
                     sparse_input_size=DATASETS["criteo"]["Fs"], path_sampling_strategy="full-path", fixed=False, last_n_blocks_out=2)
        with torch.no_grad():
            m(torch.zeros(2, 13), torch.zeros(2, 26, dtype=torch.long))
        m.configure_path_sampling_strategy("fixed-path")
        m.configure_choice(choice)
        return m
    run_case(mk, "criteo", 4, 301, "supernet_xlarge_any_last2.npz",
             extra_meta=dict(mode="supernet", use_layernorm=True, activation="silu", choice=choice, num_blocks=3, config="xlarge", last_n_blocks_out=2))
    ch = json.load(open("/root/reference/nasrec/configs/criteo/ea_criteo_kaggle_xlarge_best_1shot.json"))
    ch["micro"][-2]["sparse_in_dims"] = ch["micro"][-1]["sparse_in_dims"]
    ch["micro"][-2]["dense_sparse_interact"] = ch["micro"][-1]["dense_sparse_interact"]

    def mkf(tables, ch=ch):
        return SuperNet(num_blocks=ch["num_blocks"], ops_config=ops_config_lib[ch["config"]], use_layernorm=False, activation="relu",
                        num_embeddings=tables, sparse_input_size=26, path_sampling_strategy="fixed-path", fixed=True, fixed_choice=ch,
                        last_n_blocks_out=2)
    run_case(mkf, "criteo", 8, 302, "fixed_criteo_xlarge_last2.npz",
             extra_meta=dict(mode="fixed", use_layernorm=False, activation="relu", choice=clean_choice({"macro": ch["macro"], "micro": ch["micro"]}),
                             num_blocks=ch["num_blocks"], config=ch["config"], last_n_blocks_out=2))


def sampler_traces():
    """G4: the reference's own sampling sequence, read back from model.choice after each forward."""
    out = []
    d = DATASETS["kdd"]
    tables = [min(n, 50) for n in d["tables"]]
    int_x, cat_x, _ = O.synthetic_batch(2, d["Fd"], tables, seed=5)
    for space, nb in (("autoctr", 4), ("xlarge-zeros", 2)):
        for strategy in ("default", "single-path", "any-path", "full-path"):
            for anypath in ("uniform", "binomial-0.5"):
                for steps in (0, 6):
                    m = SuperNet(num_blocks=nb, ops_config=ops_config_lib[space], use_layernorm=True, num_embeddings=tables,
                                 sparse_input_size=d["Fs"], path_sampling_strategy="full-path", fixed=False,
                                 anypath_choice=anypath, supernet_training_steps=steps)
                    with torch.no_grad():
                        m(int_x, cat_x)  # full-path warm-up (train_utils.py:431-432); counters advance to 0
                        m.configure_path_sampling_strategy(strategy)
                        seed = 11 + len(out)
                        np.random.seed(seed)
                        seq = []
                        for _ in range(10):
                            m(int_x, cat_x)
                            seq.append(clean_choice(m.choice))
                    out.append(dict(space=space, num_blocks=nb, strategy=strategy, anypath_choice=anypath,
                                    supernet_training_steps=steps, seed=seed, warmup_forwards=1, choices=seq))
    # evo-2shot-path (supernet.py:492-500): every forward draws one of the candidate architectures with np.random.randint
    cands = []
    np.random.seed(99)
    probe = SuperNet(num_blocks=3, ops_config=ops_config_lib["autoctr"], use_layernorm=True, num_embeddings=tables, sparse_input_size=d["Fs"],
                     path_sampling_strategy="single-path", fixed=False)
    with torch.no_grad():
        for _ in range(5):
            probe(int_x, cat_x)
            cands.append({"choice": clean_choice(probe.choice)})
    m = SuperNet(num_blocks=3, ops_config=ops_config_lib["autoctr"], use_layernorm=True, num_embeddings=tables, sparse_input_size=d["Fs"],
                 path_sampling_strategy="full-path", fixed=False, candidate_choices=cands)
    with torch.no_grad():
        m(int_x, cat_x)
        m.configure_path_sampling_strategy("evo-2shot-path")
        np.random.seed(123)
        seq = []
        for _ in range(10):
            m(int_x, cat_x)
            seq.append(clean_choice(m.choice))
    out.append(dict(space="autoctr", num_blocks=3, strategy="evo-2shot-path", anypath_choice="uniform", supernet_training_steps=0, seed=123,
                    warmup_forwards=1, choices=seq, candidate_choices=cands))
    json.dump(out, open(os.path.join(OUT, "samplers.json"), "w"))
    print("wrote samplers.json", len(out), "traces")


def lr_traces():
    out = {}
    p = torch.nn.Parameter(torch.zeros(1))
    # cfg-1 recipe: scripts/eval_best_model/eval_criteo_xlarge_best_1shot.sh → lr 0.16, B 256, 1 epoch
    steps_per_epoch = 36672495 // 256
    warm = steps_per_epoch // 10 // 1
    opt = torch.optim.Adagrad([p], lr=0.16, eps=1e-2)
    sch = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=steps_per_epoch, warmup_steps=warm, max_lr=0.16, min_lr=1e-8)
    probe = sorted(set(list(range(0, 12)) + [warm - 1, warm, warm + 1, 50000, steps_per_epoch - 1]))
    seq = {}
    for t in range(steps_per_epoch):
        if t in probe:
            seq[t] = opt.param_groups[0]["lr"]
        opt.step()
        sch.step()
    out["cosine_cfg1"] = dict(first_cycle_steps=steps_per_epoch, warmup_steps=warm, max_lr=0.16, min_lr=1e-8, lrs=seq)
    # short cycle with restarts
    opt = torch.optim.Adagrad([p], lr=0.1)
    sch = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=10, warmup_steps=3, max_lr=0.1, min_lr=1e-3, cycle_mult=2.0, gamma=0.5)
    seq = []
    for t in range(45):
        seq.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    out["cosine_restarts"] = dict(first_cycle_steps=10, warmup_steps=3, max_lr=0.1, min_lr=1e-3, cycle_mult=2.0, gamma=0.5, lrs=seq)
    opt = torch.optim.Adagrad([p], lr=0.12)
    sch = ConstantWithWarmup(opt, num_warmup_steps=8)
    seq = []
    for t in range(15):
        seq.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    out["constant_warmup"] = dict(base_lr=0.12, num_warmup_steps=8, lrs=seq)
    # step(epoch=e) jumps followed by a plain step (eval_subnet_from_supernet.py:179 calls lr_scheduler.step(epoch=-1))
    jumps = []
    for mult in (1.0, 2.0):
        opt = torch.optim.Adagrad([p], lr=0.1)
        sch = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=10, warmup_steps=3, max_lr=0.1, min_lr=1e-3, cycle_mult=mult, gamma=0.5)
        seq = []
        for e in (-1, 0, 2, 5, 9, 10, 17, 33, 71):
            sch.step(epoch=e)
            a = opt.param_groups[0]["lr"]
            sch.step()
            seq.append([e, a, opt.param_groups[0]["lr"]])
        jumps.append(dict(first_cycle_steps=10, warmup_steps=3, max_lr=0.1, min_lr=1e-3, cycle_mult=mult, gamma=0.5, seq=seq))
    out["cosine_epoch_jumps"] = jumps
    json.dump(out, open(os.path.join(OUT, "lr.json"), "w"))
    print("wrote lr.json")


if __name__ == "__main__":
    which = sys.argv[1:] or ["fixed", "supernet", "samplers", "lr"]
    if "fixed" in which:
        fixed_cases()
    if "supernet" in which:
        supernet_cases()
    if "last_n" in which or not sys.argv[1:]:
        last_n_cases()
    if "samplers" in which:
        sampler_traces()
    if "lr" in which:
        lr_traces()

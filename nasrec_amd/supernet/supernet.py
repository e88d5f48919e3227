"""SuperNet / SuperNetBlock — the reference's model API (nasrec/supernet/supernet.py) on top of the HIP engine.

Drop-in surface: constructor signature, `forward(int_feats, cat_feats, choices=None) -> logits [B,1]`,
`.choice`, path-sampling configuration and the global-`np.random` sampling order, mode setters, parameter getters,
`state_dict()` keys and `parameters()` order, lazy shapes fixed by the first forward (warm-up).  What differs is
*how* a forward runs: the whole choice is compiled into one launch plan (nasrec_amd/plan.py) and executed by
hand-written HIP kernels through the C-ABI; autograd sees one node for the whole network.

Gradient semantics of this `nn.Module` path are the reference's (dense `.grad` for every `nn.Embedding`, so the
unchanged `clip_grad_norm_` + `torch.optim.Adagrad` harness works).  The fast path used by `bench.py` is
`SuperNet.engine_train_step`, whose row-sparse table update is mathematically identical for weight_decay == 0.
"""
import copy
import json
from itertools import combinations
from typing import Any, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from .. import plan as P
from ..search_space import ops_config_lib, path_sampling_strategy_lib  # noqa: F401  (re-exported, supernet.py:134-207)
from ..utils.config import NUM_EMBEDDINGS_CRITEO
from .modules import (CleverMaskGenerator, CleverZeroTensorGenerator, DotProduct, ElasticLinear, ElasticLinear3D,  # noqa: F401
                      FactorizationMachine3D, SigmoidGating, Sum, Transformer, Zeros2D, Zeros3D)
from .utils import anypath_choice_fn, assert_valid_ops_config

_dense_unary_nodes = ["linear-2d", "zeros-2d"]
_dense_binary_nodes = ["sum", "sigmoid-gating"]
_dense_sparse_nodes = ["dot-product"]
_sparse_nodes = ["zeros-3d", "transformer", "linear-3d"]

assert_valid_ops_config(ops_config_lib)

DS_INTERACT_NUM_SPLITS = 8  # supernet.py:882


def _make_node(name, use_layernorm, dims, activation, embedding_dim, fixed):
    """supernet.py:53-113"""
    if name == "linear-2d":
        return ElasticLinear(use_layernorm=use_layernorm, max_dims_or_dims=dims, activation=activation, fixed=fixed)
    if name == "zeros-2d":
        return Zeros2D(use_layernorm=use_layernorm, max_dims_or_dims=dims, activation=activation, fixed=fixed)
    if name == "sigmoid-gating":
        return SigmoidGating(use_layernorm=use_layernorm, max_dims_or_dims=dims, activation=activation, fixed=fixed)
    if name == "sum":
        return Sum(use_layernorm=use_layernorm, max_dims_or_dims=dims, activation=activation, fixed=fixed)
    if name == "dot-product":
        return DotProduct(use_layernorm=use_layernorm, max_dims_or_dims=dims, embedding_dim=embedding_dim, fixed=fixed)
    if name == "zeros-3d":
        return Zeros3D(use_layernorm=use_layernorm, max_dims_or_dims=dims, fixed=fixed, embedding_dim=embedding_dim, activation=activation)
    if name == "transformer":
        return Transformer(use_layernorm=use_layernorm, max_dims_or_dims=dims, fixed=fixed, embedding_dim=embedding_dim, activation=activation)
    if name == "linear-3d":
        return ElasticLinear3D(use_layernorm=use_layernorm, max_dims_or_dims=dims, activation=activation, embedding_dim=embedding_dim, fixed=fixed)
    raise NotImplementedError("Block name {} is not supported in supernet!".format(name))


class _tracing_paused:
    """torch.jit.trace records every torch call it sees, also inside an autograd.Function's forward; the engine's host work (plan
    buffers carved out of byte arenas with `view(dtype)`, zero fills, copies into static buffers) is not part of the model's graph
    and trips the tracer's alias analysis.  Inside this block nothing is recorded; the Function itself stays the graph's one node."""

    def __enter__(self):
        self.state = torch._C._get_tracing_state()
        if self.state is not None:
            torch._C._set_tracing_state(None)

    def __exit__(self, *exc):
        if self.state is not None:
            torch._C._set_tracing_state(self.state)
        return False


class _SupernetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = forward program, backward = backward program."""

    @staticmethod
    def forward(ctx, model, choice, int_x, cat_x, rows, *params):
        """rows: None, or (place_embedding_on_cpu) the looked-up embedding rows [B, Fs, 16] — then a differentiable input whose
        gradient torch carries back to the host tables"""
        eng = model._engine
        B = int(int_x.shape[0])
        with _tracing_paused():
            cp = eng.compile(choice, B, train=True)
            eng.run_forward(cp, int_x, cat_x, rows=rows)
        out = cp.logits.view(B, 1).clone()  # (outside the pause: a result the tracer has not seen cannot become the node's output)
        ctx.has_rows = rows is not None
        cp.generation = getattr(cp, "generation", 0) + 1
        ctx.model, ctx.cp, ctx.cat_x, ctx.generation = model, cp, cat_x, cp.generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        model, cp = ctx.model, ctx.cp
        eng = model._engine
        if getattr(cp, "evicted", False):
            raise RuntimeError("the launch plan of this forward was recycled for another sampled path before its backward ran (the "
                               "engine keeps 4 supernet plans): run backward before sampling further paths")
        if getattr(cp, "generation", 0) != ctx.generation:
            # the activations a backward needs live in the plan's static buffers (one plan per (choice, batch size)): a second
            # grad-enabled forward through the same plan has overwritten them
            raise RuntimeError("SuperNet.forward ran again with the same choice and batch size before the backward of an earlier "
                               "call: the saved activations of that call are gone; run backward (or sum the losses of separately "
                               "shaped calls) before the next forward")
        # only the last layer trains (set_mode_to_finelune_last_only): skip the network's backward altogether
        final_only = all(name.startswith("_final.") or not p.requires_grad for name, p in model._param_names)
        eng.run_backward(cp, dlogits, final_only=final_only)
        written = {"_final.weight", "_final.bias"} if final_only else set(cp.ctx.grad_params) | {"_final.weight", "_final.bias"}
        grads = []
        sg = cp.sparse0.grad_tensor().view(dlogits.shape[0], eng.Fs, 16) if (cp.sparse0.grad_written and not final_only) else None
        for name, p in model._param_names:
            if not p.requires_grad:
                grads.append(None)
            elif name.startswith("_embedding."):
                f = int(name.split(".")[1])
                if sg is None:
                    grads.append(None)
                else:  # dense gradient, exactly like nn.Embedding(sparse=False) (supernet.py:407)
                    grads.append(torch.zeros_like(p).index_add_(0, ctx.cat_x[:, f], sg[:, f]))
            elif name in written:
                grads.append(eng.grads[name].clone())
            else:
                grads.append(None)  # dead branch or unused parameter: autograd leaves .grad = None
        drows = sg.clone() if (ctx.has_rows and sg is not None) else None
        return (None, None, None, None, drows) + tuple(grads)


class ShardEmbedding(nn.Embedding):
    """This rank's rows [row_lo, row_hi) of a `whole_rows`-row table (SuperNet(table_sharding="row")).  Its values are always rows of a
    WHOLE table drawn the way the unsharded module draws it — one `normal_` of [whole_rows, dim] from the default generator of the
    weight's device, at the same point of the initialisation sequence — of which the rank keeps its range: with the ranks seeded alike
    (utils/dist.init_from_env) a sharded model starts, at any world size, from exactly the unsharded model's tables (and, the random
    stream having advanced by the same amount, from its dense weights too).  A subclass, so that the type-exact `init_weights` of the
    harness does not initialise a shard by itself (its fan would be the shard's: std sqrt(world) times too large, the same values on
    every rank); SuperNet.apply routes that call here."""

    def __init__(self, rows, dim, whole_rows, row_lo, row_hi, field):
        self.whole_rows, self.row_lo, self.row_hi, self.field = int(whole_rows), int(row_lo), int(row_hi), int(field)
        super().__init__(rows, dim)

    def _draw_whole_(self, std: float):
        with torch.no_grad():
            whole = torch.empty(self.whole_rows, self.weight.shape[1], dtype=self.weight.dtype, device=self.weight.device).normal_(0.0, std)
            self.weight.zero_()
            k = self.row_hi - self.row_lo
            if k > 0:
                self.weight[:k] = whole[self.row_lo:self.row_hi]

    def reset_parameters(self):  # nn.Embedding's constructor default: N(0, 1)
        if hasattr(self, "whole_rows"):
            self._draw_whole_(1.0)

    def xavier_normal_whole_(self):
        """torch.nn.init.xavier_normal_ of the WHOLE table: std = sqrt(2 / (whole_rows + dim))"""
        self._draw_whole_((2.0 / (self.whole_rows + self.weight.shape[1])) ** 0.5)


class SuperNet(nn.Module):
    """Top-level supernet (supernet.py:210-880)."""

    def __init__(self, num_blocks: int, ops_config: Any, use_layernorm: bool, activation: str = "relu",
                 num_embeddings: List[int] = NUM_EMBEDDINGS_CRITEO, sparse_input_size: int = 26, embedding_dim: int = 16,
                 last_n_blocks_out: int = 1, path_sampling_strategy: str = "default", fixed: bool = False, fixed_choice: Any = None,
                 place_embedding_on_cpu: bool = False, anypath_choice: str = "uniform", supernet_training_steps: int = 0,
                 candidate_choices: Optional[List] = None, use_final_sigmoid: bool = False, table_sharding: Optional[str] = None):
        """table_sharding (not a reference argument): None = every process holds whole tables (the reference's layout); "row" = the
        rows of every table are split over the ranks of the process group (nasrec_amd/sharded_tables.py) — for tables that outgrow
        one GPU.  `_embedding[f]` then holds THIS rank's row range; `state_dict()` returns whole tables (a collective when world > 1);
        training goes through the fused engine step (Adagrad, weight decay 0), evaluation through the ordinary no-grad forward."""
        super().__init__()
        assert num_blocks >= 1, ValueError("Supernet must contain a minimum of 1 block, but found {}!".format(num_blocks))
        self._num_blocks = num_blocks
        self._ops_config = ops_config
        self._use_layernorm = use_layernorm
        self._activation = activation
        self._last_n_blocks_out = last_n_blocks_out
        self._sparse_input_size = sparse_input_size
        self._num_embeddings = num_embeddings
        self._embedding_dim = embedding_dim
        self._path_sampling_strategy = path_sampling_strategy
        self._macro_path_sampling_strategy = path_sampling_strategy_lib[path_sampling_strategy]["macro"]
        self._candidate_choices = candidate_choices
        self._fixed = fixed
        assert table_sharding in (None, "none", "row"), "table_sharding must be None or 'row'"
        self._table_sharding = "row" if table_sharding == "row" else None
        self._sharded = self._sharded_ops = None
        self._embedding = self._embedding_layers(sparse_input_size, num_embeddings, embedding_dim)
        self._final = nn.LazyLinear(1)
        self._final_sigmoid = nn.Sigmoid() if use_final_sigmoid else None
        self._place_embedding_on_cpu = place_embedding_on_cpu
        self._supernet_training_steps = supernet_training_steps
        self._anypath_choice_fn = anypath_choice_fn[anypath_choice]
        self._supernet_train_steps_counter = -1
        self._device_args = None
        if self._fixed and fixed_choice is not None:
            self.choice = fixed_choice
            self.macro_last_choice = fixed_choice["macro"]
        else:
            self.choice = []
            self.macro_last_choice = None
        if self._fixed:
            assert self._macro_path_sampling_strategy == "fixed-path", ValueError(
                "'fixed_path_strategy' should be explicitly specified when 'fixed' option is True.")
        blocks = []
        for idx in range(num_blocks):
            oc = ops_config[idx] if isinstance(ops_config, list) else ops_config
            blocks.append(SuperNetBlock(
                oc, use_layernorm, int(max(oc["dense_node_dims"])), int(max(oc["sparse_node_dims"])), embedding_dim, activation,
                path_sampling_strategy=path_sampling_strategy_lib[path_sampling_strategy]["micro"], fixed=fixed,
                fixed_micro_choice=None if (fixed_choice is None) or (not fixed) else fixed_choice["micro"][idx],
                anypath_choice=anypath_choice, supernet_training_steps=supernet_training_steps, sparse_input_size=sparse_input_size))
        self._blocks = nn.ModuleList(blocks)
        # engine state (not part of the reference API)
        self._engine = None
        self._materialized = False
        self._param_names = []

    # ------------------------------------------------------------------------------------------------ parameters
    def get_dense_parameters(self):
        return list(self._blocks.parameters()) + list(self._final.parameters())

    def get_sparse_parameters(self):
        return list(self._embedding.parameters())

    def load_embeddings_from_dlrm(self, dlrm_ckpt_path=None):
        """supernet.py:368-383"""
        if dlrm_ckpt_path is not None:
            checkpoint = torch.load(dlrm_ckpt_path, map_location=torch.device("cpu"))
            assert "model_state_dict" in checkpoint.keys(), "Please use the DLRM checkpoint to load!"
            checkpoint = checkpoint["model_state_dict"]
            with torch.no_grad():
                for idx in range(len(self._embedding)):
                    w = self._embedding[idx].weight
                    w.copy_(checkpoint["embedding_layers.{}.weight".format(idx)].to(w.device))

    def _embedding_layers(self, sparse_input_size, num_embeddings, embedding_dim):
        if self._table_sharding == "row":
            out = []
            for i in range(sparse_input_size):
                lo, hi, rows = self._shard_rows(num_embeddings[i])
                out.append(ShardEmbedding(rows, embedding_dim, whole_rows=num_embeddings[i], row_lo=lo, row_hi=hi, field=i))
            return nn.ModuleList(out)
        return nn.ModuleList([nn.Embedding(num_embeddings[i], embedding_dim) for i in range(sparse_input_size)])

    def apply(self, fn):
        """nn.Module.apply; with row-sharded tables, the reference's `init_weights` (train_utils.py:70-89: xavier_normal_ on every
        nn.Embedding, type-exact) initialises the shards too — each as rows [lo, hi) of a WHOLE table drawn with the whole table's fan
        at the shard's place in the traversal (ShardEmbedding), so the random stream is consumed exactly as by the unsharded model."""
        if self._table_sharding == "row" and getattr(fn, "__name__", "") == "init_weights":
            def routed(m, _fn=fn):
                if type(m) is ShardEmbedding:
                    m.xavier_normal_whole_()
                else:
                    _fn(m)
            return super().apply(routed)
        return super().apply(fn)

    @staticmethod
    def _shard_rows(n):
        """(first row, end row, rows held) of this rank's range of an n-row table — RowShardedTables' placement: ceil(n / world)
        rows per rank; an empty range keeps one unused row so that every pointer is valid"""
        import math
        import torch.distributed as dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
        rp = max(1, math.ceil(int(n) / world))
        lo, hi = min(int(n), rank * rp), min(int(n), (rank + 1) * rp)
        return lo, hi, max(hi - lo, 1)

    def state_dict(self, *args, **kwargs):
        """row-sharded tables: `_embedding.f.weight` is returned WHOLE (all-gather of the shards — a collective when world > 1: every
        rank must call it), so checkpoints keep the reference's keys and shapes whatever the placement"""
        sd = super().state_dict(*args, **kwargs)
        if self._table_sharding == "row" and self._sharded is not None and self._sharded.world > 1:
            prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
            for f in range(self._sparse_input_size):
                sd["%s_embedding.%d.weight" % (prefix, f)] = self._sharded.whole_table(f)
        return sd

    def load_state_dict(self, state_dict, *args, **kwargs):
        if self._table_sharding == "row":  # whole tables in, this rank's rows kept
            state_dict = dict(state_dict)
            for f in range(self._sparse_input_size):
                k = "_embedding.%d.weight" % f
                if k in state_dict and int(state_dict[k].shape[0]) == int(self._num_embeddings[f]):
                    lo, hi, rows = self._shard_rows(self._num_embeddings[f])
                    t = state_dict[k][lo:hi]
                    if hi - lo < rows:
                        t = torch.cat([t, torch.zeros(rows - (hi - lo), t.shape[1], dtype=t.dtype, device=t.device)])
                    state_dict[k] = t
        return super().load_state_dict(state_dict, *args, **kwargs)

    # ------------------------------------------------------------------------------------------------ sampling
    def configure_path_sampling_strategy(self, strategy):
        assert strategy in ["full-path", "single-path", "any-path", "fixed-path", "evo-2shot-path", "default"], \
            "Strategy {} is not found!".format(strategy)
        self.__dict__.pop("_fixed_resolved", None)
        self._path_sampling_strategy = strategy
        self._macro_path_sampling_strategy = path_sampling_strategy_lib[strategy]["macro"]
        for block in self._blocks:
            block._micro_path_sampling_strategy = path_sampling_strategy_lib[strategy]["micro"]

    def _thresh(self):
        c, n = self._supernet_train_steps_counter, self._supernet_training_steps
        return 1.0 - c / (n + 1e-10) if (c < n and c > 0) else 0  # supernet.py:446-453

    def _get_choice(self):
        """supernet.py:432-511 — draws from the global np.random stream in the reference's order."""
        nb = self._num_blocks
        s = self._macro_path_sampling_strategy
        thresh = self._thresh()
        if s == "single-path":
            choice = ([self._get_full_path_choice(1 + i) for i in range(nb)] if np.random.random() < thresh
                      else [self._get_single_path_choice(1 + i) for i in range(nb)])
        elif s == "full-path":
            choice = [self._get_full_path_choice(1 + i) for i in range(nb)]
        elif s == "any-path":
            choice = ([self._get_full_path_choice(1 + i) for i in range(nb)] if np.random.random() < thresh
                      else [self._get_any_path_choice(1 + i) for i in range(nb)])
        elif s == "fixed-path" and self.macro_last_choice is None:
            if getattr(self, "_fixed_path_called", False):
                raise ValueError("Error! fixed-path choice should be generated only once for each supernet!")
            self._fixed_path_called = True
            choice = [self._get_fixed_path_choice(1 + i) for i in range(nb)]
        elif s == "fixed-path":
            choice = self.macro_last_choice
        elif s == "evo-2shot-path":
            assert self._candidate_choices is not None, "You must specify self._candidate_choices before using 'evo-2shot-path'!"
            cand = self._candidate_choices[np.random.randint(len(self._candidate_choices))]["choice"]
            for i in range(nb):
                self._blocks[i].configure_choice(cand["micro"][i])
            choice = cand["macro"]
        else:
            raise NotImplementedError("Path strategy {} is not supported!".format(s))
        if s != "full-path":
            self.macro_last_choice = choice
        return choice

    def _get_single_path_choice(self, n: int):
        """supernet.py:723-736 (the pair of gating inputs is drawn first)"""
        bi = np.random.choice(n, 2)
        return {"dense_idx": [np.random.choice(n)], "sparse_idx": [np.random.choice(n)],
                "dense_left_idx": [bi[0]], "dense_right_idx": [bi[1]]}

    def _any_like(self, n: int, fn):
        nd, ns = fn(n), fn(n)
        bi = np.random.choice(n, 2)
        return {"dense_idx": np.random.choice(n, nd, replace=False).reshape(-1).tolist(),
                "sparse_idx": np.random.choice(n, ns, replace=False).reshape(-1).tolist(),
                "dense_left_idx": bi[:1].reshape(-1).tolist(), "dense_right_idx": bi[1:].reshape(-1).tolist()}

    def _get_any_path_choice(self, n: int):
        """supernet.py:738-770"""
        return self._any_like(n, self._anypath_choice_fn)

    def _get_fixed_path_choice(self, n: int):
        """supernet.py:772-812 (always the uniform count sampler)"""
        return self._any_like(n, anypath_choice_fn["uniform"])

    def _get_full_path_choice(self, n: int):
        """supernet.py:814-824"""
        return {k: np.arange(n) for k in ("dense_idx", "sparse_idx", "dense_left_idx", "dense_right_idx")}

    def get_all_subnet_macro_choices(self, block_idx: int):
        """supernet.py:670-712"""
        m = 1 + block_idx
        out = {"dense_left_idx": [], "dense_right_idx": [], "dense_idx": [], "sparse_idx": []}
        for k in range(1, m + 1):
            out["dense_idx"] += list(combinations(list(range(m)), k))
        for k in range(1, m + 1):
            out["sparse_idx"] += list(combinations(list(range(m)), k))
        for k in range(1, min(2, m + 1)):
            out["dense_left_idx"] += list(combinations(list(range(m)), k))
            out["dense_right_idx"] += list(combinations(list(range(m)), k))
        return out

    def get_all_subnet_choices(self):
        """supernet.py:714-721"""
        out = {"macro": [], "micro": []}
        for i in range(self._num_blocks):
            out["macro"].append(self.get_all_subnet_macro_choices(i))
            out["micro"].append(self._blocks[i].get_all_subnet_micro_choices())
        return out

    def configure_choice(self, choice: Any):
        """supernet.py:842-848"""
        self.__dict__.pop("_fixed_resolved", None)
        self.choice = copy.deepcopy(choice)
        self.macro_last_choice = copy.deepcopy(choice["macro"])
        for idx in range(self._num_blocks):
            self._blocks[idx].configure_choice(choice["micro"][idx])

    # ------------------------------------------------------------------------------------------------ modes
    def set_mode_to_finelune_last_only(self):
        self._embedding.requires_grad_(False)
        self._blocks.requires_grad_(False)
        self._final.requires_grad_(True)

    def set_mode_to_normal_mode(self):
        self._embedding.requires_grad_(True)
        self._blocks.requires_grad_(True)
        self._final.requires_grad_(True)

    def set_mode_to_layernorm_calibrate(self):
        self._embedding.requires_grad_(False)
        self._blocks.requires_grad_(False)
        self._final.requires_grad_(False)
        for _, m in self._blocks.named_modules():
            if isinstance(m, nn.LayerNorm):
                m.requires_grad_(True)

    def set_mode_to_finetune_no_embedding(self):
        self._embedding.requires_grad_(False)
        self._blocks.requires_grad_(True)
        self._final.requires_grad_(True)

    # ------------------------------------------------------------------------------------------------ engine glue
    def _net_config(self):
        return P.NetConfig(self._num_blocks, self._ops_config, self._use_layernorm, self._activation, self._embedding_dim,
                           fixed=self._fixed, last_n_blocks_out=self._last_n_blocks_out,
                           use_final_sigmoid=self._final_sigmoid is not None)

    def _warm_choice(self):
        if self._fixed:
            return {"macro": self.macro_last_choice, "micro": [b._fixed_micro_choice for b in self._blocks]}
        return P.full_path_choice(self._net_config())

    def _materialize(self, Fd: int):
        """Fix every lazy shape exactly as the reference's warm-up forward does (train_utils.py:392-433): LazyLinear ->
        nn.Linear with the inferred in_features (created in forward order, so the RNG stream is consumed in the
        reference's order), projections whose input already has the target width are dropped."""
        if self._materialized:
            return
        cfg = self._net_config()
        shapes = P.infer_param_shapes(cfg, self._warm_choice(), Fd, self._sparse_input_size, self._num_embeddings)
        self._shapes = shapes
        from ..opexec import materialize_lazies
        materialize_lazies(self, shapes, delete_unused=True, device=self._embedding[0].weight.device)
        got = {k: tuple(v.shape) for k, v in nn.Module.state_dict(self).items()}
        want = {k: tuple(v) for k, v in shapes.items()}
        if self._table_sharding == "row":  # this rank's row ranges
            for f in range(self._sparse_input_size):
                want["_embedding.%d.weight" % f] = (self._shard_rows(self._num_embeddings[f])[2], self._embedding_dim)
        assert got == want, "materialised module tree does not match the inferred parameter set"
        self._materialized = True

    def _bind_engine(self, device):
        from ..engine import SupernetEngine
        params = dict(self.named_parameters())
        Fd = self._Fd
        host = bool(self._place_embedding_on_cpu) or self._table_sharding == "row"
        eng = SupernetEngine(self._net_config(), Fd, self._sparse_input_size, self._num_embeddings, device=device,
                             warm_choice=self._warm_choice(), host_embedding=host,
                             tables=None if host else [params["_embedding.%d.weight" % f].data for f in range(self._sparse_input_size)])
        eng.load_params({k: v.data for k, v in params.items() if not k.startswith("_embedding.")})
        # re-point every dense nn.Parameter at the engine's flat arena (optimizers update in place; kernels read the
        # same bytes each step — SURVEY §8b "no engine-side copy of weights may go stale")
        for name, p in params.items():
            if not name.startswith("_embedding."):
                p.data = eng.params[name]
        self._engine = eng
        self._sharded = self._sharded_ops = None
        self.__dict__.pop("_sharded_step", None)
        if self._table_sharding == "row":
            # the engine holds no table (host_embedding mode: the looked-up rows come with the batch); this rank's row ranges are the
            # nn.Embedding weights themselves, adopted by RowShardedTables; gather / dedup / row-Adagrad run on them with the
            # engine's kernels (EngineShardedOps)
            from ..sharded_tables import EngineShardedOps, RowShardedTables
            self._sharded = RowShardedTables(self._num_embeddings, device,
                                             shards=[params["_embedding.%d.weight" % f].data for f in range(self._sparse_input_size)])
            self._sharded_ops = EngineShardedOps(eng)
        self._param_names = [(n, p) for n, p in self.named_parameters() if not (host and n.startswith("_embedding."))]
        st = self.__dict__.pop("_stashed_opt_state", None)
        if st is not None:  # accumulators of the engine this one replaces
            eng._ensure_table_state()
            for n, t in st.items():
                tgt = eng.table_state[int(n.split(".")[1])] if n.startswith("_embedding.") else eng.state.get(n)
                if tgt is not None and tuple(tgt.shape) == tuple(t.shape):
                    tgt.copy_(t.to(tgt.device))

    def _ensure_engine(self, int_feats):
        if not self._materialized:
            self._Fd = int(int_feats.shape[1])
            self._materialize(self._Fd)
        dev = int_feats.device
        if dev.type != "cuda":
            from .._lib import EngineError
            raise EngineError("SuperNet.forward runs on the HIP engine: move the model and its inputs to a GPU "
                              "(there is no CPU fallback)")
        if self._engine is None or self._engine.device != dev:
            if self._final.weight.device != dev:
                self.to(dev)
            self._bind_engine(dev)

    def __deepcopy__(self, memo):
        eng, names, sh = self._engine, self._param_names, (self._sharded, self._sharded_ops, self.__dict__.pop("_sharded_step", None))
        self._engine, self._param_names, self._sharded, self._sharded_ops = None, [], None, None
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            self._engine, self._param_names, self._sharded, self._sharded_ops = eng, names, sh[0], sh[1]
            if sh[2] is not None:
                self.__dict__["_sharded_step"] = sh[2]
        return new  # the copy re-binds its own engine at its next forward

    def to(self, *args, **kwargs):
        if self._place_embedding_on_cpu:
            # supernet.py:826-840: every layer but the embeddings follows the device
            self._engine = None
            for name, child in self.named_children():
                if name != "_embedding":
                    child.to(*args, **kwargs)
            self._device_args = args
            return self
        eng = self._engine
        before = [p.data_ptr() for p in self.parameters() if not isinstance(p, nn.parameter.UninitializedParameter)] if eng is not None else None
        out = super().to(*args, **kwargs)
        if eng is not None:
            after = [p.data_ptr() for p in self.parameters() if not isinstance(p, nn.parameter.UninitializedParameter)]
            if before != after:  # storage really moved (device / dtype change): re-bind lazily
                self._stash_engine_state()
                self._engine = None
        return out

    def _stash_engine_state(self):
        """keep the fused path's Adagrad accumulators across a re-bind (they live only in the engine otherwise)"""
        eng = self._engine
        if eng is None:
            return
        st = {n: t.detach().clone() for n, t in eng.state.items()}
        if eng.table_state is not None:
            for f, t in enumerate(eng.table_state):
                st["_embedding.%d.weight" % f] = t.detach().clone()
        self._stashed_opt_state = st

    # ------------------------------------------------------------------------------------------------ forward
    def _resolve_choice(self, choices):
        """choice bookkeeping of supernet.py:513-529,585 / 605-618,650 without running anything"""
        d = self.__dict__
        if self._fixed and choices is None:
            # a fixed sub-network resolves to the same choice every time: after the first call the step hands back the SAME object
            # (the engine recognises it by identity and skips the plan-cache key), without walking the seven blocks and without
            # nn.Module.__setattr__ (2 us per assignment) — the harness calls this once per 0.3 ms training step
            hit = d.get("_fixed_resolved")
            if hit is not None and hit[0] is self.macro_last_choice and self._macro_path_sampling_strategy == "fixed-path":
                d["choice"] = hit[1]
                return hit[1]
        if not self._fixed:
            self._supernet_train_steps_counter += 1
        self.choice = {"micro": [], "macro": []}
        macro = self._get_choice() if choices is None else choices["macro"]
        self.choice["macro"] = macro
        for i, blk in enumerate(self._blocks):
            # the reference hands the *whole* micro list to every block (supernet.py:574,583), which cannot work;
            # an explicit `choices` is honoured per block here
            self.choice["micro"].append(blk._resolve_choice(None if choices is None else choices["micro"][i]))
        if self._fixed and choices is None and self._macro_path_sampling_strategy == "fixed-path" and self.macro_last_choice is macro:
            d["_fixed_resolved"] = (macro, self.choice)
        return self.choice

    def forward(self, int_feats: torch.Tensor, cat_feats: torch.Tensor, choices=None):
        choice = self._resolve_choice(choices)  # host-side bookkeeping first: a fixed net may sample its path here
        self._ensure_engine(int_feats)
        eng = self._engine
        rows = None
        if self._table_sharding == "row":
            if torch.is_grad_enabled() and any(p.requires_grad for _, p in self._param_names) and not torch.jit.is_tracing():
                from .._lib import EngineError
                raise EngineError("row-sharded tables: train through the fused engine step (engine_train_step: Adagrad, weight decay 0 — "
                                  "the route train_and_test_one_epoch takes for the published recipes); a differentiable forward "
                                  "would need the row gradients routed back to their owners by torch.autograd, which is not built. "
                                  "Evaluate under torch.no_grad().")
            rows, _ = self._sharded.lookup(cat_feats, lambda idx: self._sharded_ops.gather(self._sharded, idx))
            out = eng.forward(int_feats, cat_feats, choice, rows=rows).clone()
            return self._final_sigmoid(out) if self._final_sigmoid is not None else out
        if self._place_embedding_on_cpu:
            # supernet.py:418-428: ids to the host, lookup in host memory, rows to the device ("10~100x slow down", :253-254) — the
            # one place where the reference itself runs the stem on the CPU; everything downstream stays on the engine
            ids = cat_feats.cpu()
            rows = torch.stack([emb(ids[:, f]) for f, emb in enumerate(self._embedding)], dim=1).to(int_feats.device)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for _, p in self._param_names)
        # torch.jit.trace (TensorBoard's writer.add_graph, main_train.py:137 / train_supernet.py:192; fvcore's FLOP counter,
        # train_utils.py:444-452): the network traces as ONE opaque node (prim::PythonOp) that consumes the inputs and every
        # parameter — the no-grad route would hand the tracer a result that depends on nothing it has seen
        if needs_grad or torch.jit.is_tracing():
            out = _SupernetFunction.apply(self, choice, int_feats, cat_feats, rows, *[p for _, p in self._param_names])
        else:
            out = eng.forward(int_feats, cat_feats, choice, rows=rows).clone()
        if self._final_sigmoid is not None:
            out = self._final_sigmoid(out)
        return out

    def fixed_forward(self, int_feats: torch.Tensor, cat_feats: torch.Tensor, choices: Any):
        return self.forward(int_feats, cat_feats, choices)

    def engine_train_step(self, int_feats, cat_feats, y, lr: float, clip: Optional[float] = 5.0, eps: float = 1e-2, graph=None):
        """Fused step on the engine (forward, BCE, backward, clip_grad_norm_, Adagrad with row-sparse table update):
        the counterpart of train_utils.py:262-286 for optimizer == Adagrad, weight_decay == 0."""
        if self._place_embedding_on_cpu:
            from .._lib import EngineError
            raise EngineError("engine_train_step needs the tables on the device; with place_embedding_on_cpu use forward / backward "
                              "and a torch optimizer")
        choice = self._resolve_choice(None)
        self._ensure_engine(int_feats)
        # (graph=None: a fixed sub-network's step is replayed as a graph.  Launching the level-scheduled program instead is ~6 us faster per
        # step when the host does nothing else — bench.py, engine.prefers_graph — but this harness's host has a data pipe and a Python
        # loop to run: measured on TSV shards 832 k rows/s launched against 896 k replayed)
        graph = self._fixed if graph is None else graph
        d = self.__dict__  # (plain bookkeeping: nn.Module.__setattr__ costs 2 us per assignment)
        d["_engine_steps"] = d.get("_engine_steps", 0) + 1
        d["_last_step_batch"] = int(int_feats.shape[0])
        d["_last_step_key"] = (choice, clip, eps, graph)
        if self._table_sharding == "row":
            from ..sharded_tables import ShardedTableStep
            st = self.__dict__.get("_sharded_step")
            key = (id(self._engine), int(int_feats.shape[0]), clip, eps)
            if st is None or st[0] != key:
                st = (key, ShardedTableStep(self._sharded_ops, self._sharded, int(int_feats.shape[0]), clip=clip, eps=eps))
                self.__dict__["_sharded_step"] = st
            return st[1].step(int_feats, cat_feats, y, lr, choice=choice)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # one process per GPU (utils/dist.py): the batch is this rank's share of the global batch; same path on every rank
            from ..parallel import DataParallelStep
            dp = self.__dict__.get("_dp_step")
            key = (id(self._engine), int(int_feats.shape[0]), clip, eps, graph)
            if dp is None or dp[0] != key:
                dp = (key, DataParallelStep(self._engine, choice if self._fixed else None, int(int_feats.shape[0]), clip=clip, eps=eps, graph=graph))
                self.__dict__["_dp_step"] = dp
            return dp[1].step(int_feats, cat_feats, y, lr, choice=None if self._fixed else choice)
        return self._engine.train_step(int_feats, cat_feats, y, lr, choice, clip, eps, graph=graph)

    def engine_last_logits(self):
        """logits [B, 1] of the most recent engine_train_step (what `model(int_x, cat_x)` returned inside that step)"""
        choice, clip, eps, graph = self._last_step_key
        B = self._last_step_batch
        if self._table_sharding == "row":
            return self._sharded_ops.cp.logits.view(B, 1)
        dp = self.__dict__.get("_dp_step")
        if dp is not None and dp[1].exchange:
            return dp[1].last_plan().logits.view(B, 1)
        cp = self._engine.compile(choice, B, train=True, clip=clip, eps=eps, graph=graph)
        return cp.logits.view(B, 1)

    def engine_bind_optimizer(self, optimizer):
        """Share the Adagrad accumulators between a torch.optim.Adagrad and the engine: existing `sum` state (a resumed
        checkpoint) is copied into the engine's arenas, then `optimizer.state[p]["sum"]` aliases them, so
        `optimizer.state_dict()` stays a faithful checkpoint while the fused step does the updates."""
        eng = self._engine
        assert eng is not None, "run one forward first (lazy shapes)"
        eng._ensure_table_state()
        for name, p in self.named_parameters():
            if name.startswith("_embedding.") and self._table_sharding == "row":
                tgt = self._sharded.state[int(name.split(".")[1])]
            elif name.startswith("_embedding."):
                tgt = eng.table_state[int(name.split(".")[1])]
            elif name in eng.state:
                tgt = eng.state[name]
            else:
                continue
            st = optimizer.state[p]
            if "sum" in st and st["sum"].data_ptr() != tgt.data_ptr():
                tgt.copy_(st["sum"].to(tgt.device).view_as(tgt))
            st["sum"] = tgt
            st.setdefault("step", torch.tensor(0.0))
        self._bound_optimizer_steps = getattr(self, "_engine_steps", 0)

    def engine_sync_optimizer_steps(self, optimizer):
        """add the fused steps taken since engine_bind_optimizer to the optimizer's per-parameter step counters"""
        done = getattr(self, "_engine_steps", 0) - getattr(self, "_bound_optimizer_steps", 0)
        if done:
            for p in self.parameters():
                st = optimizer.state.get(p)
                if st is not None and "step" in st:
                    st["step"] = st["step"] + float(done)
            self._bound_optimizer_steps = getattr(self, "_engine_steps", 0)


class SuperNetBlock(nn.Module):
    """One choice block (supernet.py:884-1381)."""

    def __init__(self, ops_config: Any, use_layernorm: bool, max_dims_or_dims_dense: int, max_dims_or_dims_sparse: int,
                 embedding_dim: int, activation: str = "relu", path_sampling_strategy: str = "single-path", fixed: bool = False,
                 fixed_micro_choice=None, anypath_choice: str = "uniform", supernet_training_steps: int = 0,
                 sparse_input_size: int = 26):
        super().__init__()
        self._ops_config = ops_config
        self._num_nodes = ops_config["num_nodes"]
        self._dense_nodes = ops_config["dense_nodes"]
        self._sparse_nodes = ops_config["sparse_nodes"]
        self._node_names = ops_config["node_names"]
        self._dense_node_dims = ops_config["dense_node_dims"]
        self._sparse_node_dims = ops_config["sparse_node_dims"]
        self._zero_nodes = ops_config["zero_nodes"]
        self._sparse_input_size = sparse_input_size
        self._use_layernorm = use_layernorm
        self._max_dims_or_dims_dense = max_dims_or_dims_dense
        self._max_dims_or_dims_sparse = max_dims_or_dims_sparse
        self._embedding_dim = embedding_dim
        self._activation = activation
        self._micro_path_sampling_strategy = path_sampling_strategy
        self._fixed = fixed
        self._fixed_micro_choice = fixed_micro_choice
        self._anypath_choice_fn = anypath_choice_fn[anypath_choice]
        self._supernet_training_steps = supernet_training_steps
        self._device_args = None
        self._supernet_train_steps_counter = -1
        self._nodes = nn.ModuleList()
        self.micro_last_choice = self._fixed_micro_choice if self._fixed else None
        if self._fixed:
            choice = self._get_choice()
            self._fixed_micro_choice = choice
            choice_nodes = choice["active_nodes"]
        else:
            choice, choice_nodes = None, list(range(self._num_nodes))
        for i in range(self._num_nodes):
            if i not in choice_nodes:
                self._nodes.append(nn.ModuleList([]))
                continue
            name = self._node_names[i]
            if name in _dense_binary_nodes + _dense_unary_nodes + _dense_sparse_nodes:
                dims = choice["dense_in_dims"] if self._fixed else self._max_dims_or_dims_dense
            elif name in _sparse_nodes:
                dims = choice["sparse_in_dims"] if self._fixed else self._max_dims_or_dims_sparse
            else:
                raise NotImplementedError("Block name {} is not supported in supernet!".format(name))
            self._nodes.append(_make_node(name, use_layernorm, int(dims), activation, embedding_dim, fixed))
        if self._fixed and choice["dense_sparse_interact"] == 0:
            self.project_emb_dim, self.project_emb_dim_layernorm = None, None
        else:
            self.ds_interact_expanded_dim = DS_INTERACT_NUM_SPLITS * self._embedding_dim
            self.project_emb_dim = nn.LazyLinear(self.ds_interact_expanded_dim, bias=not use_layernorm)
            self.project_emb_dim_layernorm = nn.LayerNorm(self.ds_interact_expanded_dim, eps=1e-5) if use_layernorm else None
        if self._fixed and choice["deep_fm"] == 0:
            self.deep_fm, self.deep_fm_output_ln = None, None
        else:
            self.deep_fm_dims = max(self._dense_node_dims) if not self._fixed else choice["dense_in_dims"]
            self.deep_fm = FactorizationMachine3D(fixed=self._fixed, use_layernorm=self._use_layernorm, max_dims_or_dims=int(self.deep_fm_dims))
        self.choice = []

    def forward(self, tensors, choices=None):
        """supernet.py:1067-1162 / fixed_forward :1185-1242: tensors = [dense [B,D], sparse [B,N,16], dense_left, dense_right] ->
        (dense_out, sparse_out).  Inside a SuperNet the block is part of the network's launch plan; called on its own it compiles
        a single-block plan from the same emitter (plan.block_walk) and runs it on the HIP engine."""
        from .. import opexec
        choice = self._resolve_choice(choices)
        dense_t, sparse_t, left_t, right_t = tensors
        self.materialize([t.shape for t in tensors], choice, dense_t.device)
        key = json.dumps(choice, sort_keys=True, default=lambda o: o.tolist() if hasattr(o, "tolist") else o.item())
        dense_out, sparse_out = opexec.run(self, self._emitter(choice), [dense_t, sparse_t, left_t, right_t], key_extra=(key,))
        return dense_out, sparse_out

    def fixed_forward(self, tensors, choices):
        return self.forward(tensors, choices)

    def _emitter(self, ch):
        from .. import opexec
        cfg = opexec.OpConfig(self._use_layernorm, self._activation, self._fixed)
        ops = self._ops_config

        def emit(ctx, ins):
            d, s, l, r = ins
            dv, sv = P.block_walk(ctx, cfg, "op", ops, ch, [P.Seg(d, 0, d.width)], d.width, [P.Seg(s, 0, s.N)], s.N,
                                  [P.Seg(l, 0, l.width)], l.width, [P.Seg(r, 0, r.width)], r.width)
            return [dv, sv]
        return emit

    def materialize(self, in_shapes, choice=None, device=None):
        """What the reference's first forward does to the module tree (host logic only).  Lazy shapes come from the first
        inputs; a weight-sharing block materialises every node (the reference warms it up on the full path,
        train_utils.py:413-433), a fixed block only what its choice reaches."""
        from .. import opexec
        if opexec.has_lazies(self):
            warm = (choice if choice is not None else self._fixed_micro_choice) if self._fixed else \
                {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in self._get_full_path_choice().items()}
            opexec.materialize_lazies(self, opexec.infer_shapes(self._emitter(warm), in_shapes), device=device)

    def _get_choice(self):
        """supernet.py:1009-1061"""
        c, n = self._supernet_train_steps_counter, self._supernet_training_steps
        thresh = 1.0 - c / (n + 1e-10) if (c < n and c > 0) else 0
        s = self._micro_path_sampling_strategy
        if s == "single-path":
            choice = self._get_full_path_choice() if np.random.random() < thresh else self._get_single_path_choice()
        elif s == "full-path":
            choice = self._get_full_path_choice()
        elif s == "any-path":
            choice = self._get_full_path_choice() if np.random.random() < thresh else self._get_any_path_choice()
        elif s == "fixed-path" and self.micro_last_choice is None:
            if getattr(self, "_fixed_path_called", False):
                raise ValueError("Error! fixed-path choice should be generated only once for each supernet!")
            self._fixed_path_called = True
            choice = self._get_fixed_path_choice()
        elif s in ("fixed-path", "evo-2shot-path"):
            choice = self.micro_last_choice
        else:
            raise NotImplementedError("Path strategy {} is not supported!".format(s))
        if s != "full-path":
            self.micro_last_choice = choice
        return choice

    def _resolve_choice(self, choices):
        """supernet.py:1067-1076: draw (or take) the micro choice, then advance the block's own step counter"""
        choice = self._get_choice() if choices is None else choices
        self.choice = choice
        if not self._fixed:
            self._supernet_train_steps_counter += 1
        return choice

    def get_all_subnet_micro_choices(self):
        """supernet.py:1164-1183"""
        out = {"active_nodes": [], "dense_in_dims": [], "sparse_in_dims": [], "dense_sparse_interact": [0, 1]}
        for s in self._sparse_nodes:
            for d in self._dense_nodes:
                out["active_nodes"].append((d, s))
        out["dense_in_dims"] = [(x,) for x in self._dense_node_dims]
        out["sparse_in_dims"] = [(x,) for x in self._sparse_node_dims]
        return out

    def _get_single_path_choice(self):
        """supernet.py:1244-1263"""
        while True:
            choice = {"active_nodes": sorted([np.random.choice(self._dense_nodes)] + [np.random.choice(self._sparse_nodes)]),
                      "dense_in_dims": np.random.choice(self._dense_node_dims),
                      "sparse_in_dims": np.random.choice(self._sparse_node_dims),
                      "dense_sparse_interact": np.random.choice([0, 1]), "deep_fm": np.random.choice([0, 1])}
            if choice["active_nodes"] != self._zero_nodes:
                return choice

    def _get_full_path_choice(self):
        """supernet.py:1265-1276"""
        return {"active_nodes": np.arange(self._num_nodes), "dense_in_dims": np.max(self._dense_node_dims),
                "sparse_in_dims": np.max(self._sparse_node_dims), "dense_sparse_interact": 1, "deep_fm": 1}

    def _get_any_path_choice(self):
        """supernet.py:1278-1303"""
        while True:
            nd = self._anypath_choice_fn(len(self._dense_nodes))
            ns = self._anypath_choice_fn(len(self._sparse_nodes))
            dn = np.random.choice(self._dense_nodes, nd, replace=False).tolist()
            sn = np.random.choice(self._sparse_nodes, ns, replace=False).tolist()
            choice = {"active_nodes": sorted(dn + sn), "dense_in_dims": np.random.choice(self._dense_node_dims),
                      "sparse_in_dims": np.random.choice(self._sparse_node_dims),
                      "dense_sparse_interact": np.random.choice([0, 1]), "deep_fm": np.random.choice([0, 1])}
            if choice["active_nodes"] != self._zero_nodes:
                return choice

    def _get_fixed_path_choice(self):
        """supernet.py:1305-1313"""
        return self._get_single_path_choice()

    def configure_choice(self, choice):
        """supernet.py:1316-1318"""
        self.choice = copy.deepcopy(choice)
        self.micro_last_choice = copy.deepcopy(choice)

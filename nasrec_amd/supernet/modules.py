"""Choice-block operators of the NASRec search space — parameter containers with the reference's names.

Reference: nasrec/supernet/modules.py.  In this engine an operator does not run as a chain of ATen calls: the plan
compiler (nasrec_amd/plan.py) turns a whole choice into HIP launches, and `SuperNet.forward` is the entry point every
reference caller uses (main_train.py:233-272, train_supernet.py:242-256, searcher_utils.py:61-73).  The classes here
therefore hold exactly the sub-modules the reference holds — same attribute names, same registration order, same
lazy-shape and "delete the projection when the input already has the target width" behaviour — so that
`state_dict()` keys, `parameters()` order and `model.apply(init_weights)` (type-exact on nn.Embedding / nn.Linear /
nn.MultiheadAttention, train_utils.py:76-87) behave identically.  Calling an operator on its own raises.
"""
from math import sqrt
from typing import Optional, Union

import torch
import torch.nn as nn

NUM_MHA_HEADS = 8  # modules.py:26
LN_INIT = 0.17  # modules.py:598

_activation_names = ("relu", "silu", "identity")  # modules.py:28-32


def apply_activation_fn(x, activation):
    """modules.py:35-36 (plain torch; used by callers outside the hot path)."""
    if activation == "relu":
        return torch.nn.functional.relu(x, inplace=True)
    if activation == "silu":
        return torch.nn.functional.silu(x, inplace=True)
    if activation == "identity":
        return x
    raise KeyError(activation)


class FLAGS:
    """modules.py:41-51. DEBUG disables the mask / zero caches in the reference; the engine has no such caches
    (masks are prefix lengths, zero inputs are skipped segments), the flag is kept for API compatibility."""

    def __init__(self):
        self.DEBUG = False

    def config_debug(self, debug: bool = False):
        self.DEBUG = debug


flags = FLAGS()


class CleverMaskGenerator:
    """modules.py:57-96: cached 0/1 prefix masks (ones[:dims_in_use] ++ zeros)."""

    def __init__(self):
        self.cached_mask = {}

    def __call__(self, max_dims_or_dims: int, dims_in_use: int, device: Optional[Union[int, torch.device]] = None):
        assert max_dims_or_dims >= dims_in_use, \
            "'max_dims_or_dims' should be larger than 'dims_in_use' to successfully generate a mask."
        token = "{}_{}".format(max_dims_or_dims, dims_in_use)
        if token in self.cached_mask and not flags.DEBUG:
            return self.cached_mask[token]
        mask = torch.cat([torch.ones(dims_in_use), torch.zeros(max_dims_or_dims - dims_in_use)], dim=-1).to(device)
        mask.requires_grad = False
        self.cached_mask[token] = mask
        return mask


class CleverZeroTensorGenerator:
    """modules.py:99-127: cached zero tensors keyed by shape."""

    def __init__(self):
        self.cached_zeros = {}

    def __call__(self, size: torch.Size, device: Optional[Union[int, torch.device]] = None):
        token = "_".join([str(x) for x in size])
        if token in self.cached_zeros and not flags.DEBUG:
            return self.cached_zeros[token]
        zeros = torch.zeros(size, dtype=torch.float).to(device)
        zeros.requires_grad = False
        self.cached_zeros[token] = zeros
        return zeros


_mask_generator = CleverMaskGenerator()
_zeros_generator = CleverZeroTensorGenerator()


class _EngineOperator(nn.Module):
    """Base of the choice-block operators.  The classes hold exactly the sub-modules the reference holds (same attribute
    names, registration order, lazy shapes and deleted projections -> identical state_dict keys / parameters() order), and
    `forward` has the reference's node signature (supernet.py:1113-1122).  Inside a SuperNet the whole choice runs as one launch
    plan; called on its own an operator compiles a single-operator plan from the same emitter (nasrec_amd/opexec.py) — the
    same HIP kernels either way, never a chain of ATen calls."""

    _n_tensors = 1

    def forward(self, *args):
        assert len(args) == self._n_tensors + 1, "%s.forward takes %d tensor(s) and dims_in_use" % (type(self).__name__, self._n_tensors)
        tensors, dims_in_use = list(args[:-1]), int(args[-1])
        self._check_inputs(tensors, dims_in_use)
        from .. import opexec
        emit = self._emitter(dims_in_use)
        self.materialize([t.shape for t in tensors], dims_in_use, tensors[0].device)
        return opexec.run(self, emit, tensors, key_extra=(dims_in_use,))[0]

    def _emitter(self, dims_in_use):
        from .. import opexec
        cfg = opexec.OpConfig(bool(getattr(self, "_use_layernorm", False)), getattr(self, "_activation", "relu"), getattr(self, "_fixed", False))
        return lambda ctx, ins: self._emit(ctx, cfg, "op", ins, dims_in_use)

    def materialize(self, in_shapes, dims_in_use, device=None):
        """What the reference's FIRST forward does to the module tree for inputs of these shapes (host logic only): lazy
        Linears get their in_features, projections the input never needs are deleted with their LayerNorm siblings."""
        from .. import opexec
        if opexec.has_lazies(self):
            opexec.materialize_lazies(self, opexec.infer_shapes(self._emitter(dims_in_use), in_shapes), device=device)

    def _check_inputs(self, tensors, dims_in_use):
        pass

    def _check_dims(self, dims_in_use):
        # modules.py:164-169 etc.
        if not getattr(self, "_fixed", False):
            assert dims_in_use <= self._max_dims_or_dims, ValueError(
                "If not in fixed mode where supernet is trained, 'dims_in_use' should always be smaller than "
                "'max_dims_or_dims', but found {} vs {}! ".format(dims_in_use, self._max_dims_or_dims))

    # helpers for the emitters ------------------------------------------------------------------------------------
    @staticmethod
    def _dense_out(ctx, width):
        from .. import plan as P
        buf = ctx.buf(ctx.B * width)
        return P.DV(buf, 0, width, width)

    @staticmethod
    def _sparse_out(ctx, n):
        from .. import plan as P
        buf = ctx.buf(ctx.B * n * 16)
        return P.SV(buf, 0, n, n * 16)


class ElasticLinear(_EngineOperator):
    """modules.py:134-181"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._activation = kwargs["activation"]
        assert self._activation in _activation_names
        self._use_layernorm = kwargs["use_layernorm"]
        self._fixed = fixed
        self._linear = nn.LazyLinear(self._max_dims_or_dims, bias=not self._use_layernorm)
        self._layernorm = nn.LayerNorm([self._max_dims_or_dims]) if self._use_layernorm else None

    def _check_inputs(self, tensors, dims_in_use):
        self._check_dims(dims_in_use)

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        x = ins[0]
        out = self._dense_out(ctx, self._max_dims_or_dims)
        P.op_elastic_linear(ctx, cfg, pre, [P.Seg(x, 0, x.width)], x.width, self._max_dims_or_dims, dims, P.Target(out, 0))
        return [out]


class ElasticLinear3D(_EngineOperator):
    """modules.py:184-235"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._activation = kwargs["activation"]
        assert self._activation in _activation_names
        self._use_layernorm = kwargs["use_layernorm"]
        self._fixed = fixed
        self._linear = nn.LazyLinear(self._max_dims_or_dims, bias=not self._use_layernorm)
        self._layernorm = nn.LayerNorm([self._max_dims_or_dims]) if self._use_layernorm else None

    def _check_inputs(self, tensors, dims_in_use):
        assert len(tensors[0].size()) == 3, "Tensor should be 3D!"
        self._check_dims(dims_in_use)

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        x = ins[0]
        out = self._sparse_out(ctx, self._max_dims_or_dims)
        P.op_elastic_linear3d(ctx, cfg, pre, [P.Seg(x, 0, x.N)], x.N, self._max_dims_or_dims, dims, out)
        return [out]


class Zeros2D(_EngineOperator):
    """modules.py:238-270 (no arithmetic: a cached zero tensor)"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._fixed = fixed

    def forward(self, dense_t: torch.Tensor, dims_in_use: int):
        assert len(dense_t.size()) == 2, ValueError("Input tensor to 'Zeros2D' should have a 2D shape.")
        self._check_dims(dims_in_use)
        width = self._max_dims_or_dims if not self._fixed else dims_in_use
        return _zeros_generator(torch.Size((dense_t.size(0), width)), dense_t.device)


class DotProduct(_EngineOperator):
    """modules.py:273-401"""

    _n_tensors = 2

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._use_layernorm = kwargs["use_layernorm"]
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._embedding_dim = kwargs["embedding_dim"]
        self._fixed = fixed
        ln = self._use_layernorm
        self._dense_proj = nn.LazyLinear(self._embedding_dim, bias=not ln)
        self._sparse_proj = nn.LazyLinear(self._embedding_dim, bias=not ln)
        self.sparse_inp_proj_dim = round(sqrt(2 * self._max_dims_or_dims))
        self._sparse_inp_proj = nn.LazyLinear(self.sparse_inp_proj_dim, bias=not ln)
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=not ln)
        self._dense_layernorm = nn.LayerNorm(self._embedding_dim) if ln else None
        self._sparse_layernorm = nn.LayerNorm(self._embedding_dim) if ln else None
        self._sparse_inp_proj_layernorm = nn.LayerNorm(self.sparse_inp_proj_dim) if ln else None
        self._linear_layernorm = nn.LayerNorm(self._max_dims_or_dims) if ln else None

    def _check_inputs(self, tensors, dims_in_use):
        dense_t, sparse_t = tensors
        assert len(dense_t.size()) == 2, ValueError("Dense tensor should be 2D, but found size {}!".format(dense_t.size()))
        assert len(sparse_t.size()) == 3, ValueError("Sparse tensor should be 3D, but found size {}!".format(sparse_t.size()))
        self._check_dims(dims_in_use)

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        d, s = ins
        out = self._dense_out(ctx, self._max_dims_or_dims)
        P.op_dot_product(ctx, cfg, pre, [P.Seg(d, 0, d.width)], d.width, [P.Seg(s, 0, s.N)], s.N, self._max_dims_or_dims, dims, P.Target(out, 0))
        return [out]


class _Binary2D(_EngineOperator):
    _n_tensors = 2

    def _check_inputs(self, tensors, dims_in_use):
        left_2d, right_2d = tensors
        assert len(left_2d.size()) == 2, ValueError("Left tensor should have a shape of 2D, but had shape {}!".format(left_2d.size()))
        assert len(right_2d.size()) == 2, ValueError("Right tensor should have a shape of 2D, but had shape {}!".format(right_2d.size()))


class Sum(_Binary2D):
    """modules.py:432-501"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._use_layernorm = kwargs["use_layernorm"]
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=not self._use_layernorm)
        self._layernorm = nn.LayerNorm(self._max_dims_or_dims) if self._use_layernorm else None
        self._fixed = fixed

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        l, r = ins
        out = self._dense_out(ctx, self._max_dims_or_dims)
        P.op_sum(ctx, cfg, pre, [P.Seg(l, 0, l.width)], l.width, [P.Seg(r, 0, r.width)], r.width, self._max_dims_or_dims, dims, P.Target(out, 0))
        return [out]


class LazySelfLinear(_EngineOperator):
    """modules.py:504-519: D -> D linear whose D is only known at the first forward (SigmoidGating's gate)."""

    def __init__(self):
        super().__init__()
        self._linear = None
        self._linear_size: int = -1

    def forward(self, x):
        from .. import opexec
        emit = self._emitter(-1)
        self.materialize([x.shape], -1, x.device)
        assert x.size(-1) == self._linear_size, "'LazySelfLinear' inconsistent size: {} vs {}".format(self._linear_size, x.size(-1))
        return opexec.run(self, emit, [x])[0]

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        x = ins[0]
        out = self._dense_out(ctx, x.width)
        P.linear_dense(ctx, [P.Seg(x, 0, x.width)], x.width, pre + "._linear", x.width, True, out)
        return [out]


class SigmoidGating(_Binary2D):
    """modules.py:521-595"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._use_layernorm = kwargs["use_layernorm"]
        self._fixed = fixed
        self._left_self_linear = LazySelfLinear()
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=True)
        self._layernorm = nn.LayerNorm(self._max_dims_or_dims) if self._use_layernorm else None

    def _check_inputs(self, tensors, dims_in_use):
        super()._check_inputs(tensors, dims_in_use)
        self._check_dims(dims_in_use)

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        l, r = ins
        out = self._dense_out(ctx, self._max_dims_or_dims)
        P.op_sigmoid_gating(ctx, cfg, pre, [P.Seg(l, 0, l.width)], l.width, [P.Seg(r, 0, r.width)], r.width, self._max_dims_or_dims, dims,
                            P.Target(out, 0))
        return [out]


class Transformer(_EngineOperator):
    """modules.py:599-688"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._use_layernorm = kwargs["use_layernorm"]
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._activation = kwargs["activation"]
        self._embedding_dim = kwargs["embedding_dim"]
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=not self._use_layernorm)
        self._proj_ln = nn.LayerNorm(self._max_dims_or_dims) if self._use_layernorm else None
        self._mha = nn.MultiheadAttention(self._embedding_dim, num_heads=NUM_MHA_HEADS, batch_first=True)
        self._attn_ln = nn.LayerNorm(self._embedding_dim, eps=1e-5)
        self.attn_fc1 = nn.LazyLinear(self._embedding_dim)
        self.attn_fc2 = nn.LazyLinear(self._embedding_dim)
        self._attn_fc_ln = nn.LayerNorm(self._embedding_dim, eps=1e-5)
        self._dropout = kwargs["dropout"] if "dropout" in kwargs else 0.0
        self._fixed = fixed
        torch.nn.init.constant_(self._attn_ln.weight, LN_INIT)
        torch.nn.init.constant_(self._attn_fc_ln.weight, LN_INIT)

    def _check_inputs(self, tensors, dims_in_use):
        assert len(tensors[0].size()) == 3, ValueError("Input must have a shape of 3D, but had shape {}!".format(tensors[0].size()))

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        x = ins[0]
        out = self._sparse_out(ctx, self._max_dims_or_dims)
        P.op_transformer(ctx, cfg, pre, [P.Seg(x, 0, x.N)], x.N, self._max_dims_or_dims, dims, out)
        return [out]


class Zeros3D(_EngineOperator):
    """modules.py:691-718 (no arithmetic: a cached zero tensor)"""

    def __init__(self, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]

    def forward(self, sparse_t: torch.Tensor, dims_in_use: int):
        assert len(sparse_t.size()) == 3, ValueError("Input must have a shape of 3D, but had shape {}!".format(sparse_t.size()))
        assert dims_in_use <= self._max_dims_or_dims, ValueError(
            "If not in fixed mode where supernet is trained, 'dims_in_use' should always be smaller than "
            "'max_dims_or_dims', but found {} vs {}! ".format(dims_in_use, self._max_dims_or_dims))
        return _zeros_generator(torch.Size((sparse_t.size(0), self._max_dims_or_dims, sparse_t.size(2))), sparse_t.device)


class FactorizationMachine3D(_EngineOperator):
    """modules.py:720-750"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._use_layernorm = kwargs["use_layernorm"]
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=not self._use_layernorm)
        self._fixed = fixed
        if self._use_layernorm:
            self._linear_layernorm = nn.LayerNorm(self._max_dims_or_dims, eps=1e-5)

    def _check_inputs(self, tensors, dims_in_use):
        assert len(tensors[0].size()) == 3, "Tensor must be a sparse tensor!"

    def _emit(self, ctx, cfg, pre, ins, dims):
        from .. import plan as P
        x = ins[0]
        out = self._dense_out(ctx, self._max_dims_or_dims)
        if not ctx.shape_only:
            P.zero_fill(ctx, out)
        # the LayerNorm module outlives a dropped projection (modules.py:743 only clears the flag): keep declaring it
        cfg_fm = type(cfg)(hasattr(self, "_linear_layernorm") and self._linear_layernorm is not None, cfg.activation, cfg.fixed)
        P.op_fm(ctx, cfg_fm, pre, x, self._max_dims_or_dims, dims, out)
        return [out]

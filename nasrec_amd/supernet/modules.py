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
    """Base of the operator containers: `forward` is served by the network-level plan, not per operator."""

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            "%s is executed by the HIP engine as part of a SuperNet plan; call the SuperNet that owns it "
            "(nasrec_amd.supernet.supernet.SuperNet.forward)" % type(self).__name__)

    def _check_dims(self, dims_in_use):
        # modules.py:164-169 etc.
        assert dims_in_use <= self._max_dims_or_dims, ValueError(
            "If not in fixed mode where supernet is trained, 'dims_in_use' should always be smaller than "
            "'max_dims_or_dims', but found {} vs {}! ".format(dims_in_use, self._max_dims_or_dims))


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


class Zeros2D(_EngineOperator):
    """modules.py:238-270"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._fixed = fixed


class DotProduct(_EngineOperator):
    """modules.py:273-401"""

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


class Sum(_EngineOperator):
    """modules.py:432-501"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._use_layernorm = kwargs["use_layernorm"]
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=not self._use_layernorm)
        self._layernorm = nn.LayerNorm(self._max_dims_or_dims) if self._use_layernorm else None
        self._fixed = fixed


class LazySelfLinear(_EngineOperator):
    """modules.py:504-519: D -> D linear whose D is only known at the first forward."""

    def __init__(self):
        super().__init__()
        self._linear = None
        self._linear_size: int = -1


class SigmoidGating(_EngineOperator):
    """modules.py:521-595"""

    def __init__(self, fixed: bool = False, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]
        self._use_layernorm = kwargs["use_layernorm"]
        self._fixed = fixed
        self._left_self_linear = LazySelfLinear()
        self._linear_proj = nn.LazyLinear(self._max_dims_or_dims, bias=True)
        self._layernorm = nn.LayerNorm(self._max_dims_or_dims) if self._use_layernorm else None


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


class Zeros3D(_EngineOperator):
    """modules.py:691-718"""

    def __init__(self, **kwargs):
        super().__init__()
        self._max_dims_or_dims = kwargs["max_dims_or_dims"]


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

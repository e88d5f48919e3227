"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in the product path (nasrec_amd/) may import this.

A CPU restatement of the NASRec supernet hot path (reference: facebookresearch/NasRec, mounted at
/root/reference while the repo is being built; it does NOT exist on the GPU box).  The restatement is
*functional*: parameters live in a flat ``dict`` keyed by the reference's ``state_dict`` names, every
operator is spelled out as explicit tensor math (no ``nn.Linear``/``nn.LayerNorm``/
``nn.MultiheadAttention``), and gradients come from torch autograd over that math.  It runs in fp32 or
fp64 (``dtype=`` of the parameter dict decides); the parity bar for the HIP engine is the **fp64**
evaluation.

Parity status: **pinned** — ``tests/golden/make_golden.py`` imports the real reference in the build
container, runs it on seeded inputs with name-seeded weights and commits inputs + outputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against those vectors
(fp64 to 1e-9, fp32 to 2e-5) without needing the reference.

Allowed users: ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s ``cpu_baseline`` leg.

Each function cites the reference ``file:line`` it follows (paths relative to /root/reference/).
"""
from __future__ import annotations

import math
import zlib
from typing import Any, Dict, List, Optional

import numpy as np
import torch

NUM_MHA_HEADS = 8  # nasrec/supernet/modules.py:26
LN_INIT = 0.17  # nasrec/supernet/modules.py:598
DS_INTERACT_NUM_SPLITS = 8  # nasrec/supernet/supernet.py:882
LN_EPS = 1e-5  # torch.nn.LayerNorm default; explicit at modules.py:625,630,730, supernet.py:995

# nasrec/supernet/supernet.py:116-122
DENSE_UNARY = ("linear-2d", "zeros-2d")
DENSE_BINARY = ("sum", "sigmoid-gating")
DENSE_SPARSE = ("dot-product",)
SPARSE_NODES = ("zeros-3d", "transformer", "linear-3d")

# nasrec/supernet/supernet.py:134-178 (data, restated)
_DENSE_DIMS = [16, 32, 64, 128, 256, 512, 768, 1024]
_SPARSE_DIMS = [16, 32, 48, 64]
ops_config_lib = {
    "xlarge": dict(
        num_nodes=6,
        node_names=["linear-2d", "dot-product", "sigmoid-gating", "sum", "transformer", "linear-3d"],
        dense_node_dims=_DENSE_DIMS, sparse_node_dims=_SPARSE_DIMS,
        dense_nodes=[0, 1, 2, 3], sparse_nodes=[4, 5], zero_nodes=[]),
    "xlarge-zeros": dict(
        num_nodes=8,
        node_names=["linear-2d", "dot-product", "sigmoid-gating", "sum", "zeros-2d", "transformer",
                    "zeros-3d", "linear-3d"],
        dense_node_dims=_DENSE_DIMS, sparse_node_dims=_SPARSE_DIMS,
        dense_nodes=[0, 1, 2, 3, 4], sparse_nodes=[5, 6, 7], zero_nodes=[4, 6]),
    "autoctr": dict(
        num_nodes=3,
        node_names=["linear-2d", "dot-product", "linear-3d"],
        dense_node_dims=_DENSE_DIMS, sparse_node_dims=_SPARSE_DIMS,
        dense_nodes=[0, 1], sparse_nodes=[2], zero_nodes=[]),
}


# ----------------------------------------------------------------------------------------------
# Name-seeded parameters: weights are reproducible from their state-dict key alone, so
# golden fixtures carry no weight tensors.
# ----------------------------------------------------------------------------------------------
def seeded_array(name: str, shape, scale: float = 1.0, base: float = 0.0) -> np.ndarray:
    """float64 array ``base + scale * N(0,1)`` from ``default_rng(crc32(name))``."""
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return base + scale * rng.standard_normal(tuple(shape))


def seeded_param(name: str, shape) -> np.ndarray:
    """Deterministic, realistic-magnitude value for the parameter called ``name``."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    is_ln = ("_ln" in name) or ("layernorm" in name)
    if name.startswith("_embedding."):
        return seeded_array(name, shape, 0.1)
    if is_ln:
        if leaf == "weight":
            base = LN_INIT if ("_attn_ln" in name or "_attn_fc_ln" in name) else 1.0
            return base * (1.0 + 0.1 * seeded_array(name, shape))
        return seeded_array(name, shape, 0.05)
    if len(shape) == 2:
        return seeded_array(name, shape, math.sqrt(2.0 / (shape[0] + shape[1])))
    return seeded_array(name, shape, 0.05)


class Params(dict):
    """Flat parameter dict with *lazy* creation, mirroring ``nn.LazyLinear`` (modules.py:154 etc.):
    a parameter comes into existence the first time a forward pass touches it, with ``in_features``
    taken from the tensor it is applied to.  Projections that the reference deletes because the
    input already has the target width (modules.py:344-364,389,491,586,743; supernet.py:1144,1225)
    are simply never touched, so they never get a key — same key set as the reference state_dict."""

    def __init__(self, dtype=torch.float64, frozen: bool = False):
        super().__init__()
        self.dtype = dtype
        self.frozen = frozen

    def get_or_create(self, name: str, shape) -> torch.Tensor:
        if name not in self:
            if self.frozen:
                raise KeyError("parameter %s missing from frozen Params" % name)
            self[name] = torch.tensor(seeded_param(name, shape), dtype=self.dtype)
        t = self[name]
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
        return t

    def astype(self, dtype) -> "Params":
        out = Params(dtype, frozen=True)
        for k, v in self.items():
            out[k] = v.detach().to(dtype).clone()
        return out


# ----------------------------------------------------------------------------------------------
# Primitive math
# ----------------------------------------------------------------------------------------------
def _linear(P: Params, name: str, x: torch.Tensor, out_features: int, bias: bool) -> torch.Tensor:
    """``y = x Wᵀ (+ b)`` with W ``[out, in]`` (torch.nn.Linear layout)."""
    w = P.get_or_create(name + ".weight", (out_features, x.shape[-1]))
    y = torch.matmul(x, w.t())
    if bias:
        y = y + P.get_or_create(name + ".bias", (out_features,))
    return y


def _layernorm(P: Params, name: str, x: torch.Tensor, eps: float = LN_EPS) -> torch.Tensor:
    """LayerNorm over the last dim, biased variance, affine (torch.nn.LayerNorm semantics)."""
    d = x.shape[-1]
    w = P.get_or_create(name + ".weight", (d,))
    b = P.get_or_create(name + ".bias", (d,))
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _act(x: torch.Tensor, activation: str) -> torch.Tensor:
    """modules.py:28-36."""
    if activation == "relu":
        # F.relu as the reference calls it: the (sub)gradient AT exactly 0 is 0 (clamp_min would pass 1 there — a tie that real
        # batches do hit: an all-zero dense input (Avazu, data_pipes.py:181) through LayerNorm with bias 0 lands on 0 exactly)
        return torch.relu(x)
    if activation == "silu":
        return x * torch.sigmoid(x)
    if activation == "identity":
        return x
    raise KeyError(activation)


def _prefix_mask(max_dims: int, dims_in_use: int, like: torch.Tensor) -> torch.Tensor:
    """modules.py:57-96: ones[:dims_in_use] ++ zeros."""
    assert max_dims >= dims_in_use
    m = torch.zeros(max_dims, dtype=like.dtype)
    m[: int(dims_in_use)] = 1.0
    return m


def _pad2d(left: torch.Tensor, right: torch.Tensor):
    """modules.py:403-430: zero-pad the narrower operand on the right."""
    dl, dr = left.shape[-1], right.shape[-1]
    if dl == dr:
        return left, right
    z = torch.zeros(left.shape[0], abs(dl - dr), dtype=left.dtype)
    if dl < dr:
        return torch.cat([left, z], dim=1), right
    return left, torch.cat([right, z], dim=1)


# ----------------------------------------------------------------------------------------------
# Operators (modules.py)
# ----------------------------------------------------------------------------------------------
def elastic_linear(P, pre, x, dims_in_use, max_dims, use_ln, activation, fixed):
    """modules.py:134-181."""
    out = _linear(P, pre + "._linear", x, max_dims, bias=not use_ln)
    if use_ln:
        out = _layernorm(P, pre + "._layernorm", out)
    out = _act(out, activation)
    if not fixed:
        out = out * _prefix_mask(max_dims, dims_in_use, out)
    return out


def elastic_linear3d(P, pre, x, dims_in_use, max_dims, use_ln, activation, fixed):
    """modules.py:184-235: Linear over the token axis of [B,N,E]."""
    assert x.dim() == 3
    out = x.transpose(1, 2)
    out = _linear(P, pre + "._linear", out, max_dims, bias=not use_ln)
    if use_ln:
        out = _layernorm(P, pre + "._layernorm", out)
    out = _act(out, activation)
    if not fixed:
        out = out * _prefix_mask(max_dims, dims_in_use, out)
    return out.transpose(1, 2)


def tril_pairs(n: int):
    """Row-major strictly-lower-triangle index pairs == torch.tril_indices(n, n, offset=-1)
    (modules.py:375-379)."""
    li, lj = [], []
    for i in range(n):
        for j in range(i):
            li.append(i)
            lj.append(j)
    return li, lj


def dot_product(P, pre, dense, sparse, dims_in_use, max_dims, use_ln, emb_dim, fixed):
    """modules.py:273-401."""
    assert dense.dim() == 2 and sparse.dim() == 3
    if dense.shape[-1] != emb_dim:  # :339-345
        x = _linear(P, pre + "._dense_proj", dense, emb_dim, bias=not use_ln)
        if use_ln:
            x = _layernorm(P, pre + "._dense_layernorm", x)
    else:
        x = dense
    if sparse.shape[-1] != emb_dim:  # :348-354 (never taken: E is always emb_dim)
        y = _linear(P, pre + "._sparse_proj", sparse, emb_dim, bias=not use_ln)
        if use_ln:
            y = _layernorm(P, pre + "._sparse_layernorm", y)
    else:
        y = sparse
    k = round(math.sqrt(2 * max_dims))  # :298
    if y.shape[1] != k:  # :357-364
        y = y.transpose(1, 2)
        y = _linear(P, pre + "._sparse_inp_proj", y, k, bias=not use_ln)
        if use_ln:
            y = _layernorm(P, pre + "._sparse_inp_proj_layernorm", y)
        y = y.transpose(1, 2)
    T = torch.cat([x.unsqueeze(1), y], dim=1)  # :368
    Z = torch.matmul(T, T.transpose(1, 2))  # :370
    li, lj = tril_pairs(Z.shape[1])
    R = Z[:, li, lj]  # :378-383
    if R.shape[-1] != max_dims:  # :384-389
        out = _linear(P, pre + "._linear_proj", R, max_dims, bias=not use_ln)
    else:
        out = R
    if use_ln:  # :392
        out = _layernorm(P, pre + "._linear_layernorm", out)
    if not fixed:
        out = out * _prefix_mask(max_dims, dims_in_use, out)
    return out


def sum_node(P, pre, left, right, dims_in_use, max_dims, use_ln, fixed):
    """modules.py:432-501."""
    left, right = _pad2d(left, right)
    out = left + right
    if out.shape[-1] != max_dims:
        out = _linear(P, pre + "._linear_proj", out, max_dims, bias=not use_ln)
    if use_ln:
        out = _layernorm(P, pre + "._layernorm", out)
    if not fixed:
        out = out * _prefix_mask(max_dims, dims_in_use, out)
    return out


def sigmoid_gating(P, pre, left, right, dims_in_use, max_dims, use_ln, fixed):
    """modules.py:521-595 (+ LazySelfLinear :504-519, always biased; _linear_proj always biased :541)."""
    left, right = _pad2d(left, right)
    d = left.shape[-1]
    g = torch.sigmoid(_linear(P, pre + "._left_self_linear._linear", left, d, bias=True))
    out = g * right
    if out.shape[-1] != max_dims:
        out = _linear(P, pre + "._linear_proj", out, max_dims, bias=True)
    if use_ln:
        out = _layernorm(P, pre + "._layernorm", out)
    if not fixed:
        out = out * _prefix_mask(max_dims, dims_in_use, out)
    return out


def multihead_self_attention(P, pre, x, emb_dim, num_heads=NUM_MHA_HEADS):
    """nn.MultiheadAttention(embed, heads, batch_first=True)(x,x,x) spelled out (modules.py:624,664):
    packed in-projection [3E,E], per-head scaled dot-product softmax, out-projection."""
    B, N, E = x.shape
    hd = E // num_heads
    w_in = P.get_or_create(pre + ".in_proj_weight", (3 * E, E))
    b_in = P.get_or_create(pre + ".in_proj_bias", (3 * E,))
    w_out = P.get_or_create(pre + ".out_proj.weight", (E, E))
    b_out = P.get_or_create(pre + ".out_proj.bias", (E,))
    qkv = torch.matmul(x, w_in.t()) + b_in
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    q = q.reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    k = k.reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    v = v.reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, N, E)
    return torch.matmul(o, w_out.t()) + b_out


def transformer(P, pre, x, dims_in_use, max_dims, use_ln, emb_dim, fixed):
    """modules.py:599-688.  The two inner LayerNorms always exist (:625,:630)."""
    assert x.dim() == 3
    t = _linear(P, pre + "._linear_proj", x.transpose(1, 2), max_dims, bias=not use_ln)  # :648
    if use_ln:
        t = _layernorm(P, pre + "._proj_ln", t)
    if not fixed:  # :653-662 token mask (on the [B,E,N'] view)
        t = t * _prefix_mask(max_dims, dims_in_use, t)
    t = t.transpose(1, 2)  # [B,N',E]
    a = multihead_self_attention(P, pre + "._mha", t, emb_dim)  # :664
    a = _layernorm(P, pre + "._attn_ln", a + t)  # :666-668
    f = _linear(P, pre + ".attn_fc1", a, emb_dim, bias=True)  # :671
    f = _act(f, "relu")  # F.relu, modules.py:671 (gradient 0 at exactly 0)
    f = _linear(P, pre + ".attn_fc2", f, emb_dim, bias=True)  # :672
    out = _layernorm(P, pre + "._attn_fc_ln", a + f)  # :673-675
    if not fixed:  # :678-686
        out = (out.transpose(1, 2) * _prefix_mask(max_dims, dims_in_use, out)).transpose(1, 2)
    return out


def fm3d(P, pre, x, dims_in_use, max_dims, use_ln, fixed):
    """modules.py:720-750."""
    assert x.dim() == 3
    ix = x.sum(dim=1) ** 2 - (x ** 2).sum(dim=1)
    if ix.shape[-1] != max_dims:
        ix = _linear(P, pre + "._linear_proj", ix, max_dims, bias=not use_ln)
        if use_ln:
            ix = _layernorm(P, pre + "._linear_layernorm", ix)
    elif use_ln:
        # :743 drops only ``_linear_proj``; the LayerNorm built at :730 stays registered as an unused
        # (never-trained, grad=None) entry of the state_dict.  Keep the key set identical.
        P.get_or_create(pre + "._linear_layernorm.weight", (max_dims,))
        P.get_or_create(pre + "._linear_layernorm.bias", (max_dims,))
    if not fixed:
        ix = ix * _prefix_mask(max_dims, dims_in_use, ix)
    return ix


# ----------------------------------------------------------------------------------------------
# Network description
# ----------------------------------------------------------------------------------------------
class NetCfg:
    """Constructor arguments of the reference ``SuperNet`` that shape the math (supernet.py:214-236)."""

    def __init__(self, num_blocks: int, ops_config: Any, use_layernorm: bool, activation: str = "relu",
                 embedding_dim: int = 16, fixed: bool = False, last_n_blocks_out: int = 1,
                 use_final_sigmoid: bool = False):
        self.num_blocks = num_blocks
        self.ops_config = ops_config
        self.use_layernorm = bool(use_layernorm)
        self.activation = activation
        self.embedding_dim = embedding_dim
        self.fixed = fixed
        self.last_n_blocks_out = last_n_blocks_out
        self.use_final_sigmoid = use_final_sigmoid

    def block_ops(self, i: int) -> dict:
        return self.ops_config[i] if isinstance(self.ops_config, list) else self.ops_config


def embedding_stem(P: Params, cat_x: torch.Tensor, num_embeddings: Optional[List[int]] = None):
    """supernet.py:404-430: one table per field, one id per sample, stacked on dim 1."""
    outs = []
    for f in range(cat_x.shape[1]):
        name = "_embedding.%d.weight" % f
        if name not in P:
            assert num_embeddings is not None
            P.get_or_create(name, (num_embeddings[f], 16))
        outs.append(P[name][cat_x[:, f]])
    return torch.stack(outs, dim=1)


def block_forward(P, pre, cfg: NetCfg, ops: dict, tensors, choice: dict, record: Optional[dict] = None):
    """SuperNetBlock.forward supernet.py:1067-1162 / fixed_forward :1185-1242."""
    dense, sparse, left, right = tensors
    fixed = cfg.fixed
    use_ln = cfg.use_layernorm
    E = cfg.embedding_dim
    max_dense = int(choice["dense_in_dims"]) if fixed else int(max(ops["dense_node_dims"]))
    max_sparse = int(choice["sparse_in_dims"]) if fixed else int(max(ops["sparse_node_dims"]))
    dd, sd = int(choice["dense_in_dims"]), int(choice["sparse_in_dims"])
    active = [int(a) for a in choice["active_nodes"]]
    out2d, out3d = [], []
    for i in range(ops["num_nodes"]):
        name = ops["node_names"][i]
        npre = "%s._nodes.%d" % (pre, i)
        if i not in active:
            if fixed:
                continue  # :1194-1195
            if name in DENSE_SPARSE + DENSE_BINARY + DENSE_UNARY:  # :1084-1094
                out2d.append(torch.zeros(dense.shape[0], max_dense, dtype=dense.dtype))
            else:  # :1096-1111
                out3d.append(torch.zeros(sparse.shape[0], max_sparse, sparse.shape[2], dtype=sparse.dtype))
            continue
        if name == "linear-2d":
            out2d.append(elastic_linear(P, npre, dense, dd, max_dense, use_ln, cfg.activation, fixed))
        elif name == "zeros-2d":  # modules.py:238-270
            out2d.append(torch.zeros(dense.shape[0], dd if fixed else max_dense, dtype=dense.dtype))
        elif name == "dot-product":
            out2d.append(dot_product(P, npre, dense, sparse, dd, max_dense, use_ln, E, fixed))
        elif name == "sum":
            out2d.append(sum_node(P, npre, left, right, dd, max_dense, use_ln, fixed))
        elif name == "sigmoid-gating":
            out2d.append(sigmoid_gating(P, npre, left, right, dd, max_dense, use_ln, fixed))
        elif name == "transformer":
            out3d.append(transformer(P, npre, sparse, sd, max_sparse, use_ln, E, fixed))
        elif name == "linear-3d":
            out3d.append(elastic_linear3d(P, npre, sparse, sd, max_sparse, use_ln, cfg.activation, fixed))
        elif name == "zeros-3d":  # modules.py:691-718
            out3d.append(torch.zeros(sparse.shape[0], max_sparse, sparse.shape[2], dtype=sparse.dtype))
        else:
            raise NotImplementedError(name)
    dense_out = torch.stack(out2d, dim=-1).sum(dim=-1)  # :1133 / :1215
    sparse_out = torch.stack(out3d, dim=-1).sum(dim=-1)  # :1134 / :1216

    dsi = int(choice["dense_sparse_interact"])
    proj = None
    proj_aliases_dense = False
    if dsi == 1:  # :1137-1146 / :1218-1227
        if dense_out.shape[-1] != E * DS_INTERACT_NUM_SPLITS:
            proj = _linear(P, pre + ".project_emb_dim", dense_out, E * DS_INTERACT_NUM_SPLITS, bias=not use_ln)
            if use_ln:
                proj = _layernorm(P, pre + ".project_emb_dim_layernorm", proj)
        else:
            proj_aliases_dense = True  # no clone here (:1145-1146 / :1226-1227): a VIEW of dense_out, updated by the in-place FM add below
            proj = dense_out
        proj = proj.reshape(-1, DS_INTERACT_NUM_SPLITS, E)
    elif not fixed:  # :1147-1150 (supernet mode appends 8 zero tokens; fixed mode appends nothing :1241-1242)
        proj = torch.zeros(sparse.shape[0], DS_INTERACT_NUM_SPLITS, E, dtype=dense_out.dtype)

    if int(choice["deep_fm"]) == 1:  # :1154-1157 / :1233-1236 — uses sparse_out BEFORE the dsi concat
        fm_dims = max_dense if fixed else int(max(ops["dense_node_dims"]))  # :1001
        dense_out = dense_out + fm3d(P, pre + ".deep_fm", sparse_out, dd, fm_dims, use_ln, fixed)
        if proj_aliases_dense:  # `dense_t_2d_out += ...` is in place (:1157 / :1236): the 8 extra token rows see the post-FM value
            proj = dense_out.reshape(-1, DS_INTERACT_NUM_SPLITS, E)
    if proj is not None:
        sparse_out = torch.cat([sparse_out, proj], dim=1)  # :1161 / :1240
    return dense_out, sparse_out


def _as_set(v):
    return set(int(x) for x in np.asarray(v).reshape(-1).tolist())


def supernet_forward(P: Params, cfg: NetCfg, int_x: torch.Tensor, cat_x: torch.Tensor, choice: dict,
                     num_embeddings: Optional[List[int]] = None, record: Optional[dict] = None):
    """SuperNet.forward supernet.py:513-602 (supernet mode) / fixed_forward :604-668.
    ``choice`` = {"macro": [...], "micro": [...]} (schema supernet.py:434-441, :1011-1012)."""
    dense0 = int_x
    sparse0 = embedding_stem(P, cat_x, num_embeddings)
    dlist, slist = [dense0], [sparse0]
    for i in range(cfg.num_blocks):
        mac = choice["macro"][i]
        sel = {k: _as_set(mac[k]) for k in ("dense_idx", "sparse_idx", "dense_left_idx", "dense_right_idx")}
        d_in, s_in, l_in, r_in = [], [], [], []
        for j in range(len(dlist)):  # ascending j, :536-568 / :625-633
            for key, src, dst in (("dense_idx", dlist, d_in), ("sparse_idx", slist, s_in),
                                  ("dense_left_idx", dlist, l_in), ("dense_right_idx", dlist, r_in)):
                if j in sel[key]:
                    dst.append(src[j])
                elif not cfg.fixed:
                    dst.append(torch.zeros_like(src[j]))
        tensors = (torch.cat(d_in, dim=-1), torch.cat(s_in, dim=1), torch.cat(l_in, dim=-1), torch.cat(r_in, dim=-1))
        d_out, s_out = block_forward(P, "_blocks.%d" % i, cfg, cfg.block_ops(i), tensors, choice["micro"][i])
        if record is not None:
            record.setdefault("dense", []).append(d_out)
            record.setdefault("sparse", []).append(s_out)
        dlist.append(d_out)
        slist.append(s_out)
    n = cfg.last_n_blocks_out
    flat_d = torch.cat(dlist[-n:], dim=-1)  # :592 / :657
    flat_s = torch.flatten(torch.cat(slist[-n:], dim=-1), 1, -1)  # :594-596 / :659-661
    feats = torch.cat([flat_d, flat_s], dim=-1)
    out = _linear(P, "_final", feats, 1, bias=True)  # :283,:598 / :664
    if cfg.use_final_sigmoid:
        out = torch.sigmoid(out)
    return out


# ----------------------------------------------------------------------------------------------
# Path samplers (supernet.py:432-511, :723-824, :1009-1061, :1244-1313; supernet/utils.py:21-43).
# They draw from the GLOBAL ``np.random`` stream in the reference's call order.
# ----------------------------------------------------------------------------------------------
def anypath_uniform(num_items, max_items=4):
    return np.random.choice(min(num_items, max_items)) + 1  # utils.py:21-28


def anypath_binomial(num_items, p=0.5, max_items=4):
    return 1 + np.random.binomial(min(num_items - 1, max_items - 1), p)  # utils.py:31-36


anypath_choice_fn = {"uniform": anypath_uniform, "binomial-0.5": anypath_binomial}

path_sampling_strategy_lib = {  # supernet.py:188-207
    "default": ("any-path", "single-path"),
    "single-path": ("single-path", "single-path"),
    "any-path": ("any-path", "any-path"),
    "full-path": ("full-path", "full-path"),
    "fixed-path": ("fixed-path", "fixed-path"),
    "evo-2shot-path": ("evo-2shot-path", "evo-2shot-path"),
}


def macro_full(n):  # :814-824
    return {k: list(range(n)) for k in ("dense_idx", "sparse_idx", "dense_left_idx", "dense_right_idx")}


def macro_single(n):  # :723-736 (bi-choices are drawn FIRST)
    bi = np.random.choice(n, 2)
    return {"dense_idx": [int(np.random.choice(n))], "sparse_idx": [int(np.random.choice(n))],
            "dense_left_idx": [int(bi[0])], "dense_right_idx": [int(bi[1])]}


def macro_any(n, fn):  # :738-770 / :772-812
    nd = fn(n)
    ns = fn(n)
    bi = np.random.choice(n, 2)
    return {"dense_idx": np.random.choice(n, nd, replace=False).reshape(-1).tolist(),
            "sparse_idx": np.random.choice(n, ns, replace=False).reshape(-1).tolist(),
            "dense_left_idx": [int(bi[0])], "dense_right_idx": [int(bi[1])]}


def micro_full(ops):  # :1265-1276
    return {"active_nodes": list(range(ops["num_nodes"])), "dense_in_dims": int(max(ops["dense_node_dims"])),
            "sparse_in_dims": int(max(ops["sparse_node_dims"])), "dense_sparse_interact": 1, "deep_fm": 1}


def micro_single(ops):  # :1244-1263
    while True:
        c = {"active_nodes": sorted([int(np.random.choice(ops["dense_nodes"]))] + [int(np.random.choice(ops["sparse_nodes"]))]),
             "dense_in_dims": int(np.random.choice(ops["dense_node_dims"])),
             "sparse_in_dims": int(np.random.choice(ops["sparse_node_dims"])),
             "dense_sparse_interact": int(np.random.choice([0, 1])),
             "deep_fm": int(np.random.choice([0, 1]))}
        if c["active_nodes"] != ops["zero_nodes"]:
            return c


def micro_any(ops, fn):  # :1278-1303
    while True:
        nd = fn(len(ops["dense_nodes"]))
        ns = fn(len(ops["sparse_nodes"]))
        dn = np.random.choice(ops["dense_nodes"], nd, replace=False).tolist()
        sn = np.random.choice(ops["sparse_nodes"], ns, replace=False).tolist()
        c = {"active_nodes": sorted(int(x) for x in dn + sn),
             "dense_in_dims": int(np.random.choice(ops["dense_node_dims"])),
             "sparse_in_dims": int(np.random.choice(ops["sparse_node_dims"])),
             "dense_sparse_interact": int(np.random.choice([0, 1])),
             "deep_fm": int(np.random.choice([0, 1]))}
        if c["active_nodes"] != ops["zero_nodes"]:
            return c


class PathSampler:
    """Reproduces the per-forward sampling order of the reference in supernet mode: the SuperNet-level
    macro choice for ALL blocks is drawn first (supernet.py:526), then each block draws its micro choice
    when it runs (:1069).  SuperNet and every block keep separate step counters (:292,:938) that are
    incremented at different points relative to the draw (:517 before the macro draw; :1076 after the
    micro draw)."""

    def __init__(self, cfg: NetCfg, strategy: str = "default", anypath_choice: str = "uniform",
                 supernet_training_steps: int = 0, candidate_choices: Optional[list] = None):
        self.cfg = cfg
        self.candidates = candidate_choices
        self.macro_strategy, self.micro_strategy = path_sampling_strategy_lib[strategy]
        self.fn = anypath_choice_fn[anypath_choice]
        self.steps = supernet_training_steps
        self.net_counter = -1
        self.block_counter = [-1] * cfg.num_blocks

    def _thresh(self, counter):  # :446-453 / :1014-1020
        if counter < self.steps and counter > 0:
            return 1.0 - counter / (self.steps + 1e-10)
        return 0

    def sample(self) -> dict:
        nb = self.cfg.num_blocks
        self.net_counter += 1
        th = self._thresh(self.net_counter)
        if self.macro_strategy == "single-path":
            macro = [macro_full(1 + i) for i in range(nb)] if np.random.random() < th else [macro_single(1 + i) for i in range(nb)]
        elif self.macro_strategy == "any-path":
            macro = [macro_full(1 + i) for i in range(nb)] if np.random.random() < th else [macro_any(1 + i, self.fn) for i in range(nb)]
        elif self.macro_strategy == "full-path":
            macro = [macro_full(1 + i) for i in range(nb)]
        elif self.macro_strategy == "evo-2shot-path":  # :492-500: one of the candidate architectures, macro and micro together
            cand = self.candidates[np.random.randint(len(self.candidates))]["choice"]
            self.block_counter = [c + 1 for c in self.block_counter]
            return {"macro": cand["macro"], "micro": list(cand["micro"])}
        else:
            raise NotImplementedError(self.macro_strategy)
        micro = []
        for i in range(nb):
            ops = self.cfg.block_ops(i)
            th = self._thresh(self.block_counter[i])
            if self.micro_strategy == "single-path":
                c = micro_full(ops) if np.random.random() < th else micro_single(ops)
            elif self.micro_strategy == "any-path":
                c = micro_full(ops) if np.random.random() < th else micro_any(ops, self.fn)
            elif self.micro_strategy == "full-path":
                c = micro_full(ops)
            else:
                raise NotImplementedError(self.micro_strategy)
            self.block_counter[i] += 1
            micro.append(c)
        return {"macro": macro, "micro": micro}


def full_path_choice(cfg: NetCfg) -> dict:
    return {"macro": [macro_full(1 + i) for i in range(cfg.num_blocks)],
            "micro": [micro_full(cfg.block_ops(i)) for i in range(cfg.num_blocks)]}


# ----------------------------------------------------------------------------------------------
# Step body (train_utils.py:255-287; main_train.py:122,152-154)
# ----------------------------------------------------------------------------------------------
def bce_with_logits_mean(z: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """torch.nn.BCEWithLogitsLoss() (main_train.py:122), numerically stable form: max(z, 0) - z y + log(1 + exp(-|z|)).  Written with
    softplus so that the gradient is sigmoid(z) - y everywhere, z = 0 included (clamp / abs would give 1 - y there)."""
    return (torch.nn.functional.softplus(z) - z * y).mean()


def clip_grad_norm_(grads: List[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (train_utils.py:285): L2 norm of the per-tensor
    L2 norms; coef = max_norm / (total + 1e-6), clamped to 1; every grad is multiplied in place."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adagrad_step_(p: torch.Tensor, g: torch.Tensor, state_sum: torch.Tensor, lr: float, eps: float = 1e-2):
    """torch.optim.Adagrad(lr, lr_decay=0, weight_decay=0, initial_accumulator_value=0, eps=1e-2)
    (main_train.py:152-154): sum += g²; p -= lr · g / (sqrt(sum) + eps)."""
    state_sum.addcmul_(g, g, value=1.0)
    p.sub_(lr * g / (state_sum.sqrt() + eps))


def train_step(P: Params, state: Dict[str, torch.Tensor], cfg: NetCfg, choice: dict, int_x, cat_x, y,
               lr: float, clip: Optional[float] = 5.0, eps: float = 1e-2):
    """zero_grad → forward → BCE → backward → clip_grad_norm_ → Adagrad (train_utils.py:262-286).
    Dense-gradient semantics for the tables, exactly like nn.Embedding(sparse=False) (supernet.py:407).
    Parameters untouched by this path keep grad=None and are skipped by Adagrad, as in torch."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    Pl = Params(P.dtype, frozen=True)
    Pl.update(leaves)
    logits = supernet_forward(Pl, cfg, int_x, cat_x, choice)
    loss = bce_with_logits_mean(logits, y)
    names = list(leaves.keys())
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    gd = {k: g for k, g in zip(names, grads) if g is not None}
    total = None
    if clip is not None:
        total = clip_grad_norm_(list(gd.values()), clip)
    with torch.no_grad():
        for k, g in gd.items():
            if k not in state:
                state[k] = torch.zeros_like(P[k])
            adagrad_step_(P[k], g, state[k], lr, eps)
    return logits.detach(), loss.detach(), total, gd


# ----------------------------------------------------------------------------------------------
# LR schedules (nasrec/utils/lr_schedule.py), as pure sequences: lr used by optimizer step t = seq[t]
# ----------------------------------------------------------------------------------------------
def cosine_warmup_restarts_lrs(n: int, first_cycle_steps: int, max_lr: float, min_lr: float, warmup_steps: int,
                               cycle_mult: float = 1.0, gamma: float = 1.0) -> List[float]:
    """CosineAnnealingWarmupRestarts lr_schedule.py:47-164.  The constructor runs step() once
    (_LRScheduler.__init__ → _initial_step) and then init_lr() overwrites the lr with min_lr (:88-95),
    so optimizer step 0 uses min_lr and step t≥1 uses get_lr() at step_in_cycle = t."""
    assert warmup_steps < first_cycle_steps
    out = [min_lr]
    cur_cycle_steps, cycle, step_in_cycle, mx = first_cycle_steps, 0, 0, max_lr
    for _ in range(1, n):
        step_in_cycle += 1
        if step_in_cycle >= cur_cycle_steps:
            cycle += 1
            step_in_cycle -= cur_cycle_steps
            cur_cycle_steps = int((cur_cycle_steps - warmup_steps) * cycle_mult) + warmup_steps
        mx = max_lr * (gamma ** cycle)
        if step_in_cycle < warmup_steps:
            lr = (mx - min_lr) * step_in_cycle / warmup_steps + min_lr
        else:
            lr = min_lr + (mx - min_lr) * (1 + math.cos(math.pi * (step_in_cycle - warmup_steps) / (cur_cycle_steps - warmup_steps))) / 2
        out.append(lr)
    return out


def constant_with_warmup_lrs(n: int, base_lr: float, num_warmup_steps: int) -> List[float]:
    """ConstantWithWarmup lr_schedule.py:21-43: get_lr() reads ``_step_count`` (1 after construction)."""
    out = []
    for t in range(n):
        sc = t + 1
        if sc <= num_warmup_steps:
            out.append(base_lr * (1.0 - (num_warmup_steps - sc) / num_warmup_steps))
        else:
            out.append(base_lr)
    return out


# ----------------------------------------------------------------------------------------------
# Synthetic Criteo/Avazu/KDD-shaped batches (data_pipes.py:137,141,164)
# ----------------------------------------------------------------------------------------------
def synthetic_batch(B: int, Fd: int, num_embeddings: List[int], seed: int = 1234, zero_dense: bool = False):
    g = torch.Generator().manual_seed(seed)
    if zero_dense:
        int_x = torch.zeros(B, Fd)
    else:
        int_x = torch.log(torch.randint(0, 1000, (B, Fd), generator=g).float() + 1.0)
    cat_x = torch.stack([torch.randint(0, int(n), (B,), generator=g) for n in num_embeddings], dim=1)
    y = (torch.rand(B, 1, generator=g) < 0.25).float()
    return int_x, cat_x, y

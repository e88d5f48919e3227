"""Stand-alone execution of ONE choice-block operator (or one SuperNetBlock) on the HIP engine.

The reference's operators are ordinary callables with the node signatures of supernet.py:1113-1122
(`op(tensor, dims_in_use)`, `op(dense, sparse, dims_in_use)`, `op(left, right, dims_in_use)`, `op(sparse, dims_in_use)`),
and `SuperNetBlock.forward(tensors, choices)` (supernet.py:1067).  Here such a call compiles a single-operator launch plan with
the same emitters the whole-network plan uses (nasrec_amd/plan.py `op_*` / `block_walk`), runs its forward program
through the C-ABI, and registers ONE autograd node whose backward is the plan's backward program — so an operator used on
its own produces the same values and gradients as inside a SuperNet, from the same kernels.

No CPU path: inputs must be CUDA tensors (EngineError otherwise).
"""
from typing import Dict, List

import torch
import torch.nn as nn

from . import _lib as L
from . import plan as P
from .engine import Program

E = 16

# LayerNorm siblings that disappear together with a dropped projection (modules.py:344-345,353-354,363-364; supernet.py:1144,1225)
LN_SIBLING = {"_dense_proj": "_dense_layernorm", "_sparse_proj": "_sparse_layernorm",
              "_sparse_inp_proj": "_sparse_inp_proj_layernorm", "project_emb_dim": "project_emb_dim_layernorm"}


class OpConfig:
    """the three constructor arguments an emitter reads (plan.NetConfig without the network-level fields)"""

    def __init__(self, use_layernorm, activation, fixed):
        self.use_layernorm = bool(use_layernorm)
        self.activation = activation
        self.fixed = bool(fixed)


def materialize_lazies(root: nn.Module, shapes: Dict[str, tuple], delete_unused: bool = True, device=None):
    """Fix the lazy shapes of `root` exactly as the reference's first forward does: every 2-D weight in `shapes` (keys relative
    to root, in FORWARD order — so nn.Linear construction consumes the torch RNG in the reference's order) whose module is still
    an nn.LazyLinear (or the not-yet-created LazySelfLinear._linear) becomes an nn.Linear; with delete_unused every projection the
    forward skipped is deleted together with its LayerNorm sibling (modules.py:344-364,389,491,586,743; supernet.py:1144,1225)."""
    dev = device
    if dev is None:
        for p in root.parameters():
            if not isinstance(p, nn.parameter.UninitializedParameter):
                dev = p.device
                break
    for name, shp in shapes.items():
        if not name.endswith(".weight") or len(shp) != 2 or name.startswith("_embedding."):
            continue
        path = name[:-len(".weight")].split(".")
        parent = root
        for a in path[:-1]:
            parent = getattr(parent, a) if not a.isdigit() else parent[int(a)]
        attr = path[-1]
        cur = getattr(parent, attr, None)
        if isinstance(cur, nn.LazyLinear) or cur is None:
            lin = nn.Linear(shp[1], shp[0], bias=(name[:-len("weight")] + "bias") in shapes)
            if dev is not None:
                lin = lin.to(dev)
            setattr(parent, attr, lin)
            if type(parent).__name__ == "LazySelfLinear":
                parent._linear_size = shp[1]
    if delete_unused:
        for mod in list(root.modules()):
            for attr, child in list(mod._modules.items()):
                if isinstance(child, nn.LazyLinear):
                    setattr(mod, attr, None)
                    sib = LN_SIBLING.get(attr)
                    if sib is not None and hasattr(mod, sib):
                        setattr(mod, sib, None)
                    if type(mod).__name__ == "FactorizationMachine3D":
                        mod._use_layernorm = None  # modules.py:743


def has_lazies(root: nn.Module) -> bool:
    for mod in root.modules():
        if isinstance(mod, nn.LazyLinear):
            return True
        if type(mod).__name__ == "LazySelfLinear" and mod._linear is None:
            return True
    return False


def _in_view(ctx, t, shape):
    """plan view over an input buffer: [B, D] -> DV, [B, N, 16] -> SV"""
    numel = 1
    for s in shape:
        numel *= int(s)
    buf = P.Buf(ctx, numel, need_grad=True, tensor=t if t is not None else P._FakeTensor())
    if len(shape) == 2:
        return P.DV(buf, 0, int(shape[1]), int(shape[1]))
    if len(shape) == 3:
        if int(shape[2]) != E:
            raise NotImplementedError("the engine is specialised for embedding_dim == 16 (supernet.py:224)")
        return P.SV(buf, 0, int(shape[1]), int(shape[1]) * E)
    raise ValueError("operator inputs are 2-D [B, D] or 3-D [B, N, 16] tensors, got shape %s" % (tuple(shape),))


def infer_shapes(emit, shapes_in, B=2) -> Dict[str, tuple]:
    """parameter names (prefix 'op.') and shapes the operator holds after a first forward on inputs of these shapes"""
    ctx = P.Ctx(B=B, device=None, params=None, grads=None, shape_only=True)
    emit(ctx, [_in_view(ctx, None, (B,) + tuple(s[1:])) for s in shapes_in])
    return {k[3:]: v for k, v in ctx.shapes.items()}


def _as_strided(buf_t, view, B):
    if isinstance(view, P.DV):
        return buf_t.as_strided((B, view.width), (view.ld, 1), view.off)
    return buf_t.as_strided((B, view.N, E), (view.ld, E, 1), view.off)


class OpPlan:
    """compiled single-operator plan: static input buffers, forward / backward programs, gradient buffers"""

    def __init__(self, emit, inputs: List[torch.Tensor], params: Dict[str, torch.Tensor], train: bool):
        dev = inputs[0].device
        self.B = B = int(inputs[0].shape[0])
        self.train = train
        self.generation = 0
        self.names = list(params.keys())
        ctx = P.Ctx(B, dev, {"op." + k: v for k, v in params.items()},
                    {"op." + k: torch.zeros_like(v) for k, v in params.items()} if train else {}, shape_only=False, train=train)
        self.ctx = ctx
        self.ins = [torch.empty(tuple(x.shape), dtype=torch.float32, device=dev) for x in inputs]
        self.in_views = [_in_view(ctx, t, t.shape) for t in self.ins]
        self.out_views = emit(ctx, self.in_views)
        self.fwd = Program(ctx.fwd)
        if train:
            roots = []
            for v in self.out_views:
                v.buf.grad_tensor()
                v.buf.mark(*v.cols())
                if all(v.buf is not r for r in roots):
                    roots.append(v.buf)
            self.roots = roots
            ctx.build_backward()
            self.bwd = Program(ctx.bwd)
            self.grad_names = [n[3:] for n in ctx.grad_params]
            self.grads = {k[3:]: v for k, v in ctx.grads.items()}

    def run_forward(self, inputs, stream_ptr):
        for dst, src in zip(self.ins, inputs):
            dst.copy_(src, non_blocking=True)
        self.generation += 1
        if self.fwd.n:
            self.fwd.run(stream_ptr)
        return [_as_strided(v.buf.t, v, self.B).clone() for v in self.out_views]

    def run_backward(self, douts, stream_ptr):
        for r in self.roots:
            r.g.zero_()
        for g in self.grads.values():  # a masked / partly covered weight gradient keeps zeros where nothing is written
            g.zero_()
        for v, d in zip(self.out_views, douts):
            if d is not None:
                _as_strided(v.buf.g, v, self.B).add_(d)  # two outputs may share storage (a block's dense output living in its sparse slab)
        if self.bwd.n:
            self.bwd.run(stream_ptr)
        din = []
        for v, t in zip(self.in_views, self.ins):
            din.append(v.buf.g.view(t.shape).clone() if (v.buf.g is not None and v.buf.grad_written) else None)
        return din


class _OpFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, n_in, *tensors):
        sp = torch.cuda.current_stream(tensors[0].device).cuda_stream
        outs = plan.run_forward(tensors[:n_in], sp)
        ctx.plan, ctx.n_in, ctx.generation = plan, n_in, plan.generation
        ctx.n_params = len(tensors) - n_in
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        plan = ctx.plan
        if plan.generation != ctx.generation:
            raise RuntimeError("this operator ran forward again (same shapes) before the backward of an earlier call: the plan's "
                               "saved activations were overwritten; call backward before the next forward")
        sp = torch.cuda.current_stream(plan.ins[0].device).cuda_stream
        din = plan.run_backward([d.contiguous() if d is not None else None for d in douts], sp)
        gp = [plan.grads[n].clone() if n in plan.grad_names else None for n in plan.names]
        return (None, None) + tuple(din) + tuple(gp)


def run(module: nn.Module, emit, inputs: List[torch.Tensor], key_extra=()):
    """Execute `emit` (a function (ctx, input views) -> [output views]) for `module` on `inputs`; returns the list of outputs.
    key_extra: whatever else shapes the plan (dims_in_use, the block's micro choice)."""
    L.load()
    for x in inputs:
        if not x.is_cuda:
            raise L.EngineError("%s runs on the HIP engine: move the module and its inputs to a GPU (there is no CPU fallback)"
                                % type(module).__name__)
    inputs = [x.to(torch.float32) for x in inputs]
    params = dict(module.named_parameters())
    for n, p in params.items():
        if p.device != inputs[0].device:
            raise L.EngineError("parameter %s lives on %s, the inputs on %s" % (n, p.device, inputs[0].device))
    train = torch.is_grad_enabled() and (any(p.requires_grad for p in params.values()) or any(x.requires_grad for x in inputs))
    key = (tuple(tuple(x.shape) for x in inputs), train, tuple(p.data_ptr() for p in params.values()), key_extra)
    cache = module.__dict__.setdefault("_op_plans", {})
    plan = cache.get(key)
    if plan is None:
        if len(cache) >= 8:
            cache.pop(next(iter(cache)))
        with torch.cuda.device(inputs[0].device):
            plan = OpPlan(emit, inputs, {n: p.data for n, p in params.items()}, train)
        cache[key] = plan
    with torch.cuda.device(inputs[0].device):
        if train:
            outs = _OpFunction.apply(plan, len(inputs), *inputs, *params.values())
        else:
            outs = plan.run_forward(inputs, torch.cuda.current_stream(inputs[0].device).cuda_stream)
    return list(outs)

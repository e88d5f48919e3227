"""Training / evaluation loops of the harness with the reference's signatures, logs and printouts
(nasrec/utils/train_utils.py:70-499), driving the HIP engine.

`train_and_test_one_epoch` keeps the reference's step order (H2D, zero_grad, forward, BCE + L2, backward,
clip_grad_norm_, optimizer.step, LR step after the optimizer, train_utils.py:255-287,386).  When the run is one the fused
engine step covers — the model exposes `engine_train_step`, the optimizer is torch.optim.Adagrad without weight/lr decay,
the L2 term is identically zero (`--wd 0`, the published recipes) and AMP is off — the whole step is ONE engine call
(captured in a hipGraph for fixed sub-networks); anything else takes the reference's operator-by-operator route through
`model(int_x, cat_x)` + autograd + the torch optimizer, which the module also supports.

Not available in this environment: fvcore and tensorboard.  FLOPs are counted analytically from the engine's launch plan
(multiply-accumulates of every Linear / attention product, what fvcore's FlopCountAnalysis reports for the supported
operators), the TensorBoard writer is optional (None disables it)."""
import copy
import time
from typing import Any, Optional, Union

import numpy as np
import sklearn.metrics
import torch
import torch.nn as nn

from ..supernet.supernet import SuperNet


def init_weights(m):
    """train_utils.py:70-89: type-exact dispatch (sub-classes keep their own init)"""
    if type(m) == nn.Embedding:
        torch.nn.init.xavier_normal_(m.weight)
    elif type(m) == nn.Linear:
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            torch.nn.init.zeros_(m.bias)
    elif type(m) == nn.MultiheadAttention:
        for p in m.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
            else:
                torch.nn.init.zeros_(p)


def get_l2_loss(model: nn.Module, reg: float, no_reg_param_name: Union[str, None] = None, gpu: Optional[int] = None):
    """reg * sum ||W||_2^2 over every parameter with >= 2 dimensions (tables included) whose name does not start with
    `no_reg_param_name`; exactly 0 for reg == 0 (train_utils.py:91-115)"""
    if reg == 0:
        return torch.tensor(0.0).to(gpu)
    total = None
    for name, p in model.named_parameters():
        if p.dim() == 1 or (no_reg_param_name is not None and name.startswith(no_reg_param_name)):
            continue
        term = torch.square(torch.norm(p, p=2)) * reg
        total = term if total is None else total + term
    return total


def accuracy(gt, pred):
    """fraction of predictions > 0.5 that equal the label (train_utils.py:118-126; callers pass probabilities)"""
    hit = (torch.gt(pred, 0.5).float() == gt).sum()
    return hit / pred.size(0)


def _auroc(y_true, y_prob):
    return sklearn.metrics.roc_auc_score(y_true.detach().cpu().numpy(), y_prob.detach().cpu().numpy())


def test_one_epoch(model, test_loader, loss_fn, gpu: Optional[int] = None, max_steps=-1, use_amp: bool = False):
    """-> (accuracy, AUROC, loss) over the concatenated predictions of the loader (train_utils.py:129-178)"""
    model.eval()
    preds, labels, n = [], [], 0
    with torch.no_grad():
        for int_x, cat_x, y in test_loader:
            int_x, cat_x, y = int_x.to(gpu), cat_x.to(gpu), y.to(gpu)
            with torch.autocast("cuda", enabled=use_amp):
                preds.append(model(int_x, cat_x).clone())
            labels.append(y)
            n += 1
            if n % 50 == 0:
                print("done {} batches!".format(n))
            if max_steps != -1 and n >= max_steps:
                break
        print("Done {} batches!".format(n))
        y_true, y_pred = torch.cat(labels).flatten(), torch.cat(preds).flatten()
        prob = torch.sigmoid(y_pred)
        auroc = _auroc(y_true, prob)
        acc = accuracy(y_true, prob)
        loss = loss_fn(y_pred, y_true)
    return acc.item(), float(auroc), loss.item()


test_one_epoch.__test__ = False  # not a pytest test


def _fused_step_applies(model, optimizer, l2_loss_fn, use_amp):
    if use_amp or not hasattr(model, "engine_train_step") or type(optimizer) is not torch.optim.Adagrad:
        return False
    if getattr(model, "_place_embedding_on_cpu", False):
        return False  # the tables stay on the host: torch's optimizer updates them there
    for g in optimizer.param_groups:
        if g.get("weight_decay", 0) != 0 or g.get("lr_decay", 0) != 0 or g.get("initial_accumulator_value", 0) != 0 or g.get("maximize", False):
            return False
    if len(optimizer.param_groups) != 1:
        return False
    # the fused step updates the WHOLE dense arena and every touched embedding row: it only stands in for optimizer.step() when
    # every parameter trains and the optimizer's single group covers all of them (set_mode_to_finelune_last_only /
    # layernorm_calibrate / finetune_no_embedding, or an optimizer over a parameter subset, take the torch route)
    params = list(model.parameters())
    if not all(p.requires_grad for p in params):
        return False
    if {id(p) for p in params} != {id(p) for p in optimizer.param_groups[0]["params"]}:
        return False
    with torch.no_grad():
        return float(l2_loss_fn(model)) == 0.0


def _agreed_batches(train_loader, train_batch_size: int, world: int, gpu):
    """The batches of one epoch.  Single process: the loader as it is (incl. its short last batch, which the reference evaluates
    without training on it).  Data parallel: every rank reads its own shards, which differ in length — the ranks agree one step
    ahead (utils/dist.StepAgreement, off the compute stream) on whether ALL of them hold a FULL batch for the next step, and the
    epoch ends for everybody at the first step some rank cannot take: no rank enters a collective the others skip."""
    if world <= 1:
        yield from train_loader
        return
    from .dist import StepAgreement
    agree = StepAgreement(gpu)
    it = iter(train_loader)

    def fetch():
        try:
            b = next(it)
        except StopIteration:
            return None
        return b

    nxt = fetch()
    agree.post(nxt is not None and len(nxt[2]) == train_batch_size)
    while True:
        ok = agree.take()
        if not ok:
            return
        cur = nxt
        nxt = fetch()
        agree.post(nxt is not None and len(nxt[2]) == train_batch_size)  # travels while the step on `cur` runs
        yield cur


def train_and_test_one_epoch(model, epoch: int, optimizer: Any, lr_scheduler, train_loader, test_loader, loss_fn, l2_loss_fn,
                             train_batch_size: int, gpu: Union[int, None], display_interval: int = 100, test_interval: int = 2000,
                             max_train_steps: int = -1, max_eval_steps: int = -1, test_only_at_last_step: bool = False,
                             grad_clip_value: float = None, tb_writer=None, use_amp: bool = False, use_engine_step: Optional[bool] = None):
    """One epoch of training with tests in between; returns the reference's log dict (train_utils.py:181-390).
    `use_engine_step`: None = fused engine step whenever it applies, False = always the torch route."""
    _peek(test_loader)  # the reference peeks one test batch here (train_utils.py:224-225)
    model.train()
    logs = {k: [] for k in ("train_loss", "train_AUROC", "train_Accuracy", "test_loss", "test_AUROC", "test_Accuracy", "epoch", "iters")}
    fused = _fused_step_applies(model, optimizer, l2_loss_fn, use_amp) if use_engine_step is None else bool(use_engine_step)
    bound = False
    scaler = torch.amp.GradScaler("cuda") if use_amp else None
    best_model, best_test_loss = None, 9999.99
    t_data0 = time.time()
    batch_num = -1
    from .dist import StepAgreement, allreduce_grads, any_rank, world_info
    world = world_info()[1]
    zero_l2 = None
    for batch_num, (int_x, cat_x, y) in enumerate(_agreed_batches(train_loader, train_batch_size, world, gpu)):
        t_data1 = time.time()
        on_dev = int_x.is_cuda and cat_x.is_cuda and y.is_cuda and (not isinstance(gpu, int) or int_x.device.index == gpu)
        if not on_dev:  # (device-staged batches are there already: three no-op .to() calls are 6 us of a 300 us step)
            int_x, cat_x, y = int_x.to(gpu, non_blocking=True), cat_x.to(gpu, non_blocking=True), y.to(gpu, non_blocking=True)
        t_gpu0 = time.time()
        full = len(y) == train_batch_size  # a short last batch is evaluated but not trained on
        if fused and full:
            if not bound:  # lazy shapes + engine, then share the Adagrad accumulators (a resumed optimizer keeps its state)
                model._ensure_engine(int_x)
                model.engine_bind_optimizer(optimizer)
                bound = True
            group = optimizer.param_groups[0]
            loss = model.engine_train_step(int_x, cat_x, y.view(-1), lr=float(group["lr"]), clip=grad_clip_value, eps=float(group["eps"]))
            # the step's logits and the (zero) L2 term are only looked at on display steps: fetched there, not 3 900 times per epoch
            # (`torch.zeros` is a fill launch on the GPU, `engine_last_logits` a plan lookup)
            res = None
            if zero_l2 is None:
                zero_l2 = torch.zeros((), device=y.device)
            l2_loss = zero_l2
        else:
            optimizer.zero_grad()
            with torch.autocast("cuda", enabled=use_amp):
                res = model(int_x, cat_x)
                loss = loss_fn(res, y)
                l2_loss = l2_loss_fn(model)
                total_loss = loss + l2_loss
            if full:
                if use_amp:
                    scaler.scale(total_loss).backward()
                    if world > 1:
                        # the DDP order: exchange the still-SCALED gradients, then unscale — every rank then sees the same inf / NaN
                        # (found_inf) and skips or takes the step together; unscaling first would let the overflowing rank skip
                        # alone while the others step on inf-averaged gradients
                        allreduce_grads(model)
                    scaler.unscale_(optimizer)
                    if grad_clip_value is not None:
                        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip_value)
                    scaler.step(optimizer)
                    scaler.update()
                else:
                    total_loss.backward()
                    if world > 1:  # the torch route has no exchange of its own: average the gradients over the ranks
                        allreduce_grads(model)
                    if grad_clip_value is not None:
                        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip_value)
                    optimizer.step()
        t_gpu1 = time.time()
        last = batch_num == max_train_steps - 1

        if batch_num % display_interval == 0 or last:
            if res is None:
                res = model.engine_last_logits()
            y_pred, y_true = res.detach(), y.detach()
            if any_rank(bool(torch.isnan(loss)), gpu):  # happens on KDD: report a diverged model (every rank leaves together)
                print("Loss NaN. Exiting...")
                logs["test_loss"].append(999.99)
                logs["test_AUROC"].append(-1)
                logs["test_Accuracy"].append(-1)
                return logs
            loss_v, l2_v = float(loss.detach()), float(l2_loss.detach())
            print(f"Epoch: {epoch:>d} L2: {l2_v:>7f} loss: {loss_v:>7f}  {batch_num}")
            lr_now = lr_scheduler.get_lr()
            print("Learning rate: {}".format(lr_now[0] if isinstance(lr_now, list) else lr_now))
            print("Data: {:.5f} (s), GPU: {:.5f} (s)".format(t_data1 - t_data0, t_gpu1 - t_gpu0))
            prob = torch.sigmoid(y_pred).view(-1)
            try:
                train_auroc = float(_auroc(y_true.view(-1), prob))
            except Exception:
                print("AUROC encountered issues. All training data has the same label.")
                train_auroc = 1.0
            train_acc = accuracy(y_true.view(-1), prob).item()
            print("Train Acc: {}, Train AUROC: {}".format(train_acc, train_auroc))
            if tb_writer is not None:
                x = batch_num * train_batch_size
                tb_writer.add_scalar("Loss/train/epoch{}".format(epoch), loss_v, x)
                tb_writer.add_scalar("Acc/train/epoch{}".format(epoch), train_acc, x)
                tb_writer.add_scalar("AUROC/train/epoch{}".format(epoch), train_auroc, x)
                tb_writer.add_scalar("iters/epoch{}".format(epoch), x, x)
            logs["train_loss"].append(loss_v)
            logs["train_AUROC"].append(train_auroc)
            logs["train_Accuracy"].append(train_acc)
            logs["epoch"].append(epoch)
            logs["iters"].append(batch_num)

        if batch_num % test_interval == 0 or last:
            if (not test_only_at_last_step) or last:
                model.eval()
                t0 = time.time()
                test_acc, test_auroc, test_loss = test_one_epoch(model, test_loader, loss_fn, gpu, max_steps=max_eval_steps, use_amp=use_amp)
                print("{:.4f} seconds elasped for testing!".format(time.time() - t0))
                print("Test Acc: {}, Test AUROC: {}, Test Loss: {}".format(test_acc, test_auroc, test_loss))
                logs["test_loss"].append(test_loss)
                logs["test_AUROC"].append(test_auroc)
                logs["test_Accuracy"].append(test_acc)
                if test_loss < best_test_loss:
                    best_model, best_test_loss = copy.deepcopy(model), test_loss
                if tb_writer is not None:
                    x = batch_num * train_batch_size
                    tb_writer.add_scalar("Loss/test/epoch{}".format(epoch), test_loss, x)
                    tb_writer.add_scalar("Acc/test/epoch{}".format(epoch), test_acc, x)
                    tb_writer.add_scalar("AUROC/test/epoch{}".format(epoch), test_auroc, x)
                    tb_writer.add_scalar("Best Loss/test/epoch{}".format(epoch), best_test_loss, x)
            model.train()
        if max_train_steps != -1 and batch_num >= max_train_steps - 1:
            break
        lr_scheduler.step()
        t_data0 = time.time()
    else:
        print("Batch counter total: {}".format(batch_num))
    if bound:
        model.engine_sync_optimizer_steps(optimizer)
    del best_model  # the reference rebinds a local name to a copy of it: no observable effect
    return logs


def _peek(loader):
    """first batch of a loader without leaving reader threads / device blocks behind (RoundRobinLoader.peek); any other iterable:
    the reference's `next(iter(loader))`"""
    peek = getattr(loader, "peek", None)
    return peek() if peek is not None else next(iter(loader))


def warmup_model(model: nn.Module, train_loader, gpu: Union[int, None]):
    """one forward pass: fixes every lazy shape (train_utils.py:392-410)"""
    model = model.to(gpu)
    int_x, cat_x, _ = _peek(train_loader)
    model(int_x.to(gpu), cat_x.to(gpu))
    return model


def warmup_supernet_model(model: nn.Module, train_loader, gpu, freeze_gc: bool = False):
    """full-path forward so that every candidate operator materialises (train_utils.py:413-433).  freeze_gc: see
    SupernetEngine.reserve — only for a process that keeps this one model for its whole life (the train_supernet CLI)"""
    assert isinstance(model, SuperNet), NotImplementedError("For 'warmup_supernet_model', the passed in model must be a 'SuperNet' object.")
    model = model.to(gpu)
    int_x, cat_x, _ = _peek(train_loader)
    model.configure_path_sampling_strategy("full-path")
    model(int_x.to(gpu), cat_x.to(gpu))
    eng = getattr(model, "_engine", None)
    if eng is not None and hasattr(eng, "reserve") and not getattr(eng, "host_embedding", False):
        eng.reserve(int(int_x.shape[0]), freeze_gc=freeze_gc)  # plan slots sized for the full path: sampled paths never grow an arena mid-step
    return model


def get_model_flops_and_params(model, train_loader, gpu):
    """(multiply-accumulates per sample of one forward pass, number of parameters).  The reference asks fvcore
    (train_utils.py:436-452), which counts one flop per MAC of Linear / matmul / bmm and ignores the rest; here the same
    quantity is read off the engine's launch plan for the model's current choice."""
    from .. import _lib as L
    from .. import plan as P
    model = model.to(gpu)
    int_x, cat_x, _ = _peek(train_loader)
    int_x, cat_x = int_x.to(gpu), cat_x.to(gpu)
    with torch.no_grad():
        model(int_x, cat_x)
    B = int(int_x.shape[0])
    eng = model._engine
    cp = eng.compile(model.choice, B, train=False)
    macs = 0
    for d in P.iter_ops(cp.fwd.descs):
        if isinstance(d, L.GemmDesc):
            macs += sum(d.seg[q].M * d.seg[q].N * d.seg[q].K for q in range(d.nseg) if d.seg[q].A)
        elif isinstance(d, L.MhaDesc):  # in/out projections + FFN (4 x 16x16 ... per token) and the two attention products
            macs += d.B * d.N * (3 * 256 + 256 + 256 + 256) + 2 * d.B * d.N * d.N * 16
        elif isinstance(d, L.DotTriDesc):
            macs += d.B * d.k1 * d.k1 * 16  # the reference's full bmm
        elif isinstance(d, L.FinalDesc):
            macs += d.B * sum(d.width[q] for q in range(d.nseg))
    params = sum(p.numel() for p in model.parameters())
    return macs / B, params


def get_model_latency(model, inputs, gpu, num_warmup_steps: int = 10, num_trials: int = 200):
    """(mean, std) seconds of an eval forward pass over the central 90 % of `num_trials` runs (train_utils.py:455-499)"""
    model = model.to(gpu)
    was_training = model.training
    model.eval()
    assert len(inputs) == 2, "Number of inputs should have exactly 2 items for RM models!"
    int_x, cat_x = inputs[0].to(gpu), inputs[1].to(gpu)
    samples = []
    with torch.no_grad():
        for i in range(num_warmup_steps + num_trials):
            t0 = time.time()
            model(int_x, cat_x)
            torch.cuda.synchronize(gpu)
            if i >= num_warmup_steps:
                samples.append(time.time() - t0)
    if was_training:
        model.train()
    samples = np.asarray(samples)
    lo, hi = np.percentile(samples, 5), np.percentile(samples, 95)
    kept = samples[(samples >= lo) & (samples <= hi)]
    return float(np.mean(kept)), float(np.std(kept))

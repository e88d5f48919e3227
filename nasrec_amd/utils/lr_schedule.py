"""Learning-rate schedules of the reference (nasrec/utils/lr_schedule.py) as closed-form functions of the optimizer
step index t (the lr the optimizer uses at its t-th step); the engine keeps lr in a device scalar, so no
torch.optim.lr_scheduler object is involved.  Both classes also accept the reference's call form with the optimizer as
first argument (`CosineAnnealingWarmupRestarts(optimizer, first_cycle_steps=..., ...)`,
`ConstantWithWarmup(optimizer, num_warmup_steps=...)`, main_train.py:163-174): every lr they compute is then written into
`optimizer.param_groups` as well, so an unchanged torch training loop sees the same schedule."""
import math


def _split_optimizer(args):
    """(optimizer or None, remaining positional args)"""
    if args and hasattr(args[0], "param_groups"):
        return args[0], args[1:]
    return None, args


def _publish(optimizer, lr):
    if optimizer is not None:
        for g in optimizer.param_groups:
            g["lr"] = lr


class CosineAnnealingWarmupRestarts:
    """lr_schedule.py:47-164.  Step 0 runs at min_lr (init_lr() overrides the constructor's first step, :88-95)."""

    def __init__(self, *args, **kw):
        self.optimizer, args = _split_optimizer(args)
        self._init(*args, **kw)
        _publish(self.optimizer, self.lr)

    def _init(self, first_cycle_steps, cycle_mult=1.0, max_lr=0.1, min_lr=0.001, warmup_steps=0, gamma=1.0):
        assert warmup_steps < first_cycle_steps
        self.first_cycle_steps, self.cycle_mult, self.base_max_lr = first_cycle_steps, cycle_mult, max_lr
        self.min_lr, self.warmup_steps, self.gamma = min_lr, warmup_steps, gamma
        self.cur_cycle_steps, self.cycle, self.step_in_cycle, self.max_lr = first_cycle_steps, 0, 0, max_lr
        self.lr = min_lr

    def get_lr(self):
        return self.lr

    def step(self, epoch=None):
        """lr_schedule.py:121-164; `epoch` = jump to that step index (eval_subnet_from_supernet.py calls step(epoch=-1))"""
        if epoch is None:
            self.step_in_cycle += 1
            if self.step_in_cycle >= self.cur_cycle_steps:
                self.cycle += 1
                self.step_in_cycle -= self.cur_cycle_steps
                self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps
        elif epoch >= self.first_cycle_steps:
            if self.cycle_mult == 1.0:
                self.step_in_cycle = epoch % self.first_cycle_steps
                self.cycle = epoch // self.first_cycle_steps
            else:
                n = int(math.log(epoch / self.first_cycle_steps * (self.cycle_mult - 1) + 1, self.cycle_mult))
                self.cycle = n
                self.step_in_cycle = epoch - int(self.first_cycle_steps * (self.cycle_mult ** n - 1) / (self.cycle_mult - 1))
                self.cur_cycle_steps = self.first_cycle_steps * self.cycle_mult ** n
        else:
            self.cur_cycle_steps = self.first_cycle_steps
            self.step_in_cycle = epoch
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        if self.step_in_cycle == -1:  # lr_schedule.py:97-98: base_lrs, which init_lr() set to min_lr
            self.lr = self.min_lr
        elif self.step_in_cycle < self.warmup_steps:
            self.lr = (self.max_lr - self.min_lr) * self.step_in_cycle / self.warmup_steps + self.min_lr
        else:
            self.lr = self.min_lr + (self.max_lr - self.min_lr) * (
                1 + math.cos(math.pi * (self.step_in_cycle - self.warmup_steps) / (self.cur_cycle_steps - self.warmup_steps))) / 2
        _publish(self.optimizer, self.lr)
        return self.lr


    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)
        _publish(self.optimizer, self.lr)


class ConstantWithWarmup:
    """lr_schedule.py:21-43: linear ramp over num_warmup_steps (first step at base/num_warmup_steps), then constant."""

    def __init__(self, *args, **kw):
        self.optimizer, args = _split_optimizer(args)
        if self.optimizer is not None:  # reference form: the base lr is the optimizer's
            args = (self.optimizer.param_groups[0]["lr"],) + tuple(args)
        self._init(*args, **kw)
        _publish(self.optimizer, self.lr)

    def _init(self, base_lr, num_warmup_steps):
        self.base_lr, self.num_warmup_steps, self.count = base_lr, num_warmup_steps, 1
        self.lr = self._at(1)

    def _at(self, c):
        if c <= self.num_warmup_steps:
            return self.base_lr * (1.0 - (self.num_warmup_steps - c) / self.num_warmup_steps)
        return self.base_lr

    def get_lr(self):
        return self.lr

    def step(self, epoch=None):
        """the reference's get_lr reads the scheduler's own call counter, whatever `epoch` says (lr_schedule.py:33-43)"""
        self.count += 1
        self.lr = self._at(self.count)
        _publish(self.optimizer, self.lr)
        return self.lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)
        _publish(self.optimizer, self.lr)

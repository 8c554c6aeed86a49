"""Learning-rate schedulers of the reference (lavis/common/optims.py:56-119), operating on anything with ``set_lr`` or on a
torch optimizer's param_groups."""
import math

from lavis.common.registry import registry


def _set_lr(optimizer, lr):
    if hasattr(optimizer, "set_lr"):
        optimizer.set_lr(lr)
    else:
        for g in optimizer.param_groups:
            g["lr"] = lr


def cosine_lr_schedule(optimizer, epoch, max_epoch, init_lr, min_lr):
    _set_lr(optimizer, (init_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * epoch / max_epoch)) + min_lr)


def warmup_lr_schedule(optimizer, step, max_step, init_lr, max_lr):
    _set_lr(optimizer, min(max_lr, init_lr + (max_lr - init_lr) * step / max(max_step, 1)))


def step_lr_schedule(optimizer, epoch, init_lr, min_lr, decay_rate):
    _set_lr(optimizer, max(min_lr, init_lr * (decay_rate ** epoch)))


@registry.register_lr_scheduler("linear_warmup_cosine_lr")
class LinearWarmupCosineLRScheduler:
    def __init__(self, optimizer, max_epoch, min_lr, init_lr, warmup_steps=0, warmup_start_lr=-1, **kwargs):
        self.optimizer, self.max_epoch, self.min_lr, self.init_lr = optimizer, max_epoch, min_lr, init_lr
        self.warmup_steps = warmup_steps
        self.warmup_start_lr = warmup_start_lr if warmup_start_lr >= 0 else init_lr
        self.max_iters_per_epoch = 0

    def step(self, cur_epoch, cur_step):
        if cur_step > self.max_iters_per_epoch:
            self.max_iters_per_epoch = cur_step
        g = cur_epoch * self.max_iters_per_epoch + cur_step
        if g < self.warmup_steps:
            warmup_lr_schedule(self.optimizer, g, self.warmup_steps, self.warmup_start_lr, self.init_lr)
        else:
            cosine_lr_schedule(self.optimizer, cur_epoch, self.max_epoch, self.init_lr, self.min_lr)


@registry.register_lr_scheduler("linear_warmup_step_lr")
class LinearWarmupStepLRScheduler:
    def __init__(self, optimizer, max_epoch, min_lr, init_lr, decay_rate=1, warmup_start_lr=-1, warmup_steps=0, **kwargs):
        self.optimizer, self.max_epoch, self.min_lr, self.init_lr, self.decay_rate = optimizer, max_epoch, min_lr, init_lr, decay_rate
        self.warmup_steps = warmup_steps
        self.warmup_start_lr = warmup_start_lr if warmup_start_lr >= 0 else init_lr

    def step(self, cur_epoch, cur_step):
        if cur_epoch == 0:
            warmup_lr_schedule(self.optimizer, cur_step, self.warmup_steps, self.warmup_start_lr, self.init_lr)
        else:
            step_lr_schedule(self.optimizer, cur_epoch, self.init_lr, self.min_lr, self.decay_rate)

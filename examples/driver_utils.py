"""Host-side conveniences of the example drivers that are NOT part of the accelerated path (SURVEY.md marks the
reference's counterparts out of scope): the --use_lr_schedule learning-rate schedule (utils.py:93-105, called at
run_grevnet.py:444) and checkpoint / resume of a GRevNetTrainer (the reference uses tf.train.Saver, run_grevnet.py:379,
449-453).  They only touch the trainer's public attributes."""
import ctypes as C
import math

import torch


def get_learning_rate(timestep, max_lr, ramp_up=1000, hold_steady=2000, const_multiple=3):
    """Linear warm-up to max_lr over `ramp_up` steps, max_lr up to and including step `hold_steady`, then
    max_lr * min(1, const_multiple) / sqrt(steps past hold_steady).  The warm-up test comes first, so with
    ramp_up > hold_steady the ramp runs to its end and the decay starts from there."""
    if timestep < ramp_up:
        return max_lr * timestep / ramp_up
    if timestep <= hold_steady:
        return max_lr
    inv = 1.0 / math.sqrt(timestep - hold_steady)
    return max_lr * min(inv, const_multiple * inv)


def trainer_state(tr):
    """Everything a resumed run needs: parameters, Adam moments, step counter, batch-norm moving statistics."""
    if tr.theta is None:
        raise RuntimeError("run a step (or loss_and_grads) first so that the variables exist")
    return {"theta": tr.theta.detach().cpu(), "m": tr.m.detach().cpu(), "v": tr.v.detach().cpu(),
            "global_step": tr.global_step,
            "bn_moving": [(b.moving_mean.detach().cpu(), b.moving_variance.detach().cpu()) for b in tr._bns]}


def load_trainer_state(tr, state):
    from gnf_amd import _abi
    if tr.theta is None:
        raise RuntimeError("connect the trainer first (run loss_and_grads on a batch)")
    if state["theta"].numel() != tr.theta.numel():
        raise ValueError(f"checkpoint has {state['theta'].numel()} parameters, the flow has {tr.theta.numel()}")
    tr.theta.copy_(state["theta"])
    tr.m.copy_(state["m"])
    tr.v.copy_(state["v"])
    tr.global_step = int(state["global_step"])
    for b, (mm, mv) in zip(tr._bns, state["bn_moving"]):
        b.moving_mean.copy_(mm)
        b.moving_variance.copy_(mv)
    dev = tr.theta.device
    with torch.cuda.device(dev):            # the matrix-core weight copies follow the restored parameters
        flow = tr.net._flow(tr.net.mlps("s")[0].layer_sizes[-1], dev)
        if tr.net.fused:
            _abi.check(_abi.lib().gnf_pack_flow(C.byref(flow), _abi.stream_ptr(dev)), "gnf_pack_flow")


def save_checkpoint(tr, path):
    torch.save(trainer_state(tr), path)


def load_checkpoint(tr, path):
    load_trainer_state(tr, torch.load(path, map_location="cpu"))

"""Host-side conveniences of the example drivers that are NOT part of the accelerated path (SURVEY.md marks the
reference's counterparts out of scope): the --use_lr_schedule learning-rate schedule (utils.py:93-105, called at
run_grevnet.py:444).  Checkpoint / resume of a GRevNetTrainer (the reference uses tf.train.Saver, run_grevnet.py:379,
449-453) moved into the package in round 6 (gnf_amd/train.py) and is re-exported at the bottom."""
import math


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


# checkpoint / resume of a GRevNetTrainer: in the package since round 6 (gnf_amd.train); re-exported for the drivers
from gnf_amd.train import load_checkpoint, load_trainer_state, save_checkpoint, trainer_state   # noqa: E402,F401

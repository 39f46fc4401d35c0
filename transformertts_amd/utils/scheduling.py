"""Host-side step schedules: what the trainer evaluates before every step and hands to
`model.set_constants(learning_rate=...)` (reference `train_tts.py:152-153` calling `utils/scheduling.py:5-48`).
A schedule is a table of `[step, value]` breakpoints.  Pure NumPy on the host; the value reaches the GPU as one
4-byte write into the device-resident learning-rate scalar that the fused Adam kernel reads.

Kept arithmetic (the outputs are held bit for bit to the reference's own, tests/test_reference_fixtures.py):
the interpolated value is `m * step + b` with `m = (y1 - y0) / (x1 - x0)` and `b = y0 - m * x0`, evaluated in
float64 and returned as float32 (`tf.cast(value, tf.float32)` in the reference)."""
import numpy as np


def _segment(step, xs: np.ndarray) -> int:
    """Index of the last breakpoint <= step, or -1 when the step lies before the first one."""
    return int(np.searchsorted(xs, step, side='right')) - 1


def piecewise_linear_schedule(step, schedule) -> np.float32:
    """Linear interpolation between breakpoints, constant before the first and after the last."""
    table = np.asarray(schedule)
    xs, ys = table[:, 0], table[:, 1]
    i = _segment(step, xs)
    if i < 0:
        return np.float32(ys[0])
    if i >= len(xs) - 1:
        return np.float32(ys[-1])
    slope = (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i])
    intercept = ys[i] - slope * xs[i]
    return np.float32(slope * step + intercept)


def reduction_schedule(step, schedule) -> int:
    """Step function: the value of the last breakpoint <= step.  Before the first breakpoint the reference
    returns the first breakpoint's STEP column (its loop starts from `schedule[0, 0]`); kept as is - every
    shipped table starts at step 0, where the two readings coincide."""
    table = np.asarray(schedule)
    i = _segment(step, table[:, 0])
    return int(table[0, 0] if i < 0 else table[i, 1])

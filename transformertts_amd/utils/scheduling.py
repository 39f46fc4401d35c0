"""Step schedules the trainer evaluates on the host before each step (reference utils/scheduling.py:5-48;
`train_tts.py:152-153`: `model.set_constants(learning_rate=piecewise_linear_schedule(model.step, ...))`).
Plain Python/NumPy like the reference; the value reaches the GPU through `set_constants` (one 4-byte write into
the device-resident learning-rate scalar the fused Adam kernel reads)."""
import numpy as np


def linear_function(x, x0, x1, y0, y1):
    m = (y1 - y0) / (x1 - x0)
    b = y0 - m * x0
    return m * x + b


def piecewise_linear(step, X, Y):
    """Piecewise linear function with values Y_i at breakpoints X_i, constant outside them."""
    assert len(X) == len(Y)
    X = np.array(X)
    if step < X[0]:
        return Y[0]
    idx = np.where(step >= X)[0][-1]
    if idx == (len(Y) - 1):
        return Y[-1]
    return linear_function(step, X[idx], X[idx + 1], Y[idx], Y[idx + 1])


def piecewise_linear_schedule(step, schedule) -> np.float32:
    """`schedule` = [[step, value], ...]; float32 like the reference's `tf.cast(value, tf.float32)`."""
    schedule = np.array(schedule)
    return np.float32(piecewise_linear(step, schedule[:, 0], schedule[:, 1]))


def reduction_schedule(step, schedule) -> int:
    schedule = np.array(schedule)
    r = schedule[0, 0]
    for i in range(schedule.shape[0]):
        if schedule[i, 0] <= step:
            r = schedule[i, 1]
        else:
            break
    return int(r)

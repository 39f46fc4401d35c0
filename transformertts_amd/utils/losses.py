"""Loss mirror of reference utils/losses.py:41-49,63-70 on the HIP L1 kernel."""
from __future__ import annotations

from .. import ops


def masked_mean_absolute_error(targets, logits, mask_value=0, mask=None):
    """Reference utils/losses.py:41-49.  On the ForwardTransformer path `mask` is never passed, so
    the loss is the plain mean over every element, padded frames included.  A
    non-None mask is not part of the path and is rejected rather than silently ignored."""
    if mask is not None:
        raise NotImplementedError('masked variant is not on the ForwardTransformer path')
    return ops.L1LossFn.apply(logits, targets)


def weighted_sum_losses(targets, pred, loss_functions, coeffs):
    """Reference utils/losses.py:63-70."""
    total_loss = 0
    loss_vals = []
    for i in range(len(loss_functions)):
        loss = loss_functions[i](targets[i], pred[i])
        loss_vals.append(loss)
        total_loss = total_loss + coeffs[i] * loss
    return total_loss, loss_vals

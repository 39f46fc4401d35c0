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


def weighted_sum_losses(targets, pred, loss_functions, coeffs, unit_seed=False, denominators=None):
    """Reference utils/losses.py:63-70: total = sum_i coeffs[i] * loss_i (accumulated from 0 in list order), and the
    list of the unweighted losses.  When every term is the unmasked MAE above (the ForwardTransformer's three losses)
    the whole sum is one fused op (ops.WeightedL1LossesFn: same values, same accumulation order; only the total carries
    a gradient); `unit_seed=True` is the train step's promise that it calls total.backward() with the default seed.
    `denominators` (not in the reference, which is single-device): per-term divisors that replace each mean's own element
    count - the GLOBAL padded-batch counts under batch data parallelism (transformertts_amd/dp.py), so that the sum over
    ranks is the reference's mean over the whole batch.  Only the fused path takes them.
    Note (differs from the reference's TF tensors): on the fused path the returned per-term `loss_vals` are detached
    device scalars - only `total` carries a gradient, which is all the train step differentiates (models.py:478-480)."""
    if (1 <= len(loss_functions) <= 8 and all(f is masked_mean_absolute_error for f in loss_functions)
            and all(getattr(p, 'is_cuda', False) for p in pred)):
        flat = [t for i in range(len(loss_functions)) for t in (pred[i], targets[i])]
        cs = tuple(float(c) for c in coeffs)
        if denominators is not None:
            cs = (cs, tuple(int(n) for n in denominators))
        total, *loss_vals = ops.WeightedL1LossesFn.apply(cs, unit_seed, *flat)
        return total, loss_vals
    if denominators is not None:
        raise NotImplementedError('global loss denominators need the fused L1 path (device tensors, unmasked MAE terms)')
    total_loss = 0
    loss_vals = []
    for i in range(len(loss_functions)):
        loss = loss_functions[i](targets[i], pred[i])
        loss_vals.append(loss)
        total_loss = total_loss + coeffs[i] * loss
    return total_loss, loss_vals

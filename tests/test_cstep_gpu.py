"""The train step issued from C++ off one descriptor (ttsmi_ft_train_step, transformertts_amd/step.py) against the per-layer
autograd path it replaces (reference model/models.py:464-482): the same launches with the same arguments on the same streams,
so losses, outputs, every gradient and the post-Adam weights must agree BIT FOR BIT - over several steps, with dropout on,
over changing batch shapes (buffer growth + re-binding), and with the chain kernels on."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(cfg_extra=None, lr=1e-3):
    from oracle import ft_oracle as fo
    from transformertts_amd.model.models import ForwardTransformer
    cfg = dict(dict(fo.make_config(), dropout_rate=0.1, predictors_dropout=0.1, seed=5, precision='bf16'), **(cfg_extra or {}))
    a = ForwardTransformer.from_config(dict(cfg, use_cstep=True))
    b = ForwardTransformer.from_config(dict(cfg, use_cstep=False))
    b.load_weights_dict(a.weights_dict())
    for m in (a, b):
        m._compile(learning_rate=lr)
    return fo, a, b


def _same(x, y, what):
    assert x.shape == y.shape, (what, x.shape, y.shape)
    assert torch.equal(x, y), f'{what}: max |diff| {float((x.float() - y.float()).abs().max())}'


def _check_step(a, b, batch, what):
    oa = a.train_step(*batch)
    ga = a.params.grad.clone()
    out_a = {k: oa[k].clone() for k in ('mel', 'duration', 'pitch', 'expanded_lengths', 'loss', 'expanded_mask')}
    la = {k: v.clone() for k, v in oa['losses'].items()}
    ob = b.train_step(*batch)
    torch.cuda.synchronize()
    for k, v in out_a.items():
        _same(v, ob[k], f'{what}: {k}')
    for k, v in la.items():
        _same(v, ob['losses'][k], f'{what}: losses[{k}]')
    _same(ga, b.params.grad, f'{what}: flat gradient')
    _same(a.params.data, b.params.data, f'{what}: weights after Adam')
    _same(a.shadow_set.flat_bf16, b.shadow_set.flat_bf16, f'{what}: bf16 shadows')
    assert int(a.step_dev) == int(b.step_dev) and a.step == b.step


def test_the_c_step_runs_and_is_the_default_for_the_benchmarked_architecture():
    fo, a, b = _models()
    assert a._cstep_ok() and not b._cstep_ok()
    batch = fo.synthetic_batch(4, 60, 300, seed=3, ragged=True)
    a.train_step(*batch)
    assert a._cstep is not None and a._cstep.shape[:3] == (4, 60, 300)


def test_c_step_equals_the_per_layer_path_bit_for_bit_over_changing_shapes():
    fo, a, b = _models()
    shapes = [(4, 60, 300), (4, 60, 300), (6, 80, 420), (3, 40, 200), (6, 80, 420), (8, 120, 500)]
    for i, (B, Tp, Tm) in enumerate(shapes):
        batch = fo.synthetic_batch(B, Tp, Tm, seed=11 + i, ragged=True)
        _check_step(a, b, batch, f'step {i} {B}x{Tp}x{Tm}')


def test_c_step_with_the_chain_kernels_and_without_dropout(monkeypatch):
    from transformertts_amd import ops
    monkeypatch.setattr(ops, 'CHAIN_MIN_ROWS', 0)
    fo, a, b = _models(dict(dropout_rate=0.0, predictors_dropout=0.0))
    # (shapes change under the chains: their weight streams are packed ahead by the previous step's optimiser phase and stay
    # valid across a re-targeted shape; a growing batch re-binds and packs at the step's start)
    for i, (B, Tp, Tm) in enumerate([(4, 100, 480), (4, 100, 480), (4, 90, 420), (3, 100, 480), (5, 110, 520), (5, 110, 520)]):
        batch = fo.synthetic_batch(B, Tp, Tm, seed=21 + i, ragged=True)
        _check_step(a, b, batch, f'chained step {i}')
    assert all(pl.chain_on for pl in a._cstep.plans_d)


def test_switching_between_the_two_paths_on_one_model_keeps_training_consistent():
    """One model alternating between the C step and the per-layer path (attention maps requested every other step) follows the
    trajectory of a model that only ever used the per-layer path."""
    fo, a, b = _models()
    for i in range(4):
        batch = fo.synthetic_batch(4, 60, 300, seed=31 + i, ragged=True)
        a.return_attention = bool(i % 2)          # maps requested: the per-layer path
        b.return_attention = bool(i % 2)
        oa, ob = a.train_step(*batch), b.train_step(*batch)
        torch.cuda.synchronize()
        _same(oa['loss'], ob['loss'], f'step {i}: loss')
        _same(a.params.data, b.params.data, f'step {i}: weights')


def test_outputs_of_a_step_survive_the_next_step():
    fo, a, _ = _models()
    b0 = fo.synthetic_batch(4, 60, 300, seed=41, ragged=True)
    b1 = fo.synthetic_batch(4, 60, 300, seed=42, ragged=True)
    o0 = a.train_step(*b0)
    mel0, loss0 = o0['mel'].clone(), o0['loss'].clone()
    a.train_step(*b1)
    torch.cuda.synchronize()
    assert torch.equal(o0['mel'], mel0) and torch.equal(o0['loss'], loss0)

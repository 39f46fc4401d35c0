"""The oracle against the reference's OWN model source, executed (tests/golden/make_reference_source_run.py:
model/models.py + model/layers.py + utils/losses.py imported from /root/reference and run over the
torch-float64 stand-in for TensorFlow in tests/_tf_shim.py).  Forward outputs, attention maps, the losses and the
gradient of EVERY variable of the reference's train step must equal the oracle's to fp64 round-off, for a
dense-block and a conv-block configuration on ragged batches; so must the reference's `predict` (predicted durations scaled by
1/speed_regulator and clamped per symbol drive the length regulator).  The frozen run is checked always; where /root/reference exists (the build
container) the reference source is also executed live."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))

import make_reference_source_run as gen  # noqa: E402
from oracle import ft_oracle as fo  # noqa: E402

TOL = 1e-10


@pytest.fixture(scope='module')
def frozen():
    with np.load(os.path.join(HERE, 'golden', 'reference_source_run.npz')) as z:
        return {k: z[k] for k in z.files}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _check_oracle_against(run, name):
    c = gen.CASES[name]
    cfg = fo.make_config(**c['cfg'])
    W = fo.init_weights(cfg, seed=c['wseed'], perturb=0.05)
    g = lambda k: run[f'{name}/{k}']
    batch = (g('tokens'), g('mel_target'), g('durations'), g('pitch'))
    want_batch = fo.synthetic_batch(c['B'], c['Tp'], c['Tm'], seed=c['bseed'], ragged=True)
    for a, b in zip(batch, want_batch):
        np.testing.assert_array_equal(a, b)                               # the fixture's inputs are the seeded ones
    ora = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    ora.learning_rate = 1e-3
    out = ora.train_step(*batch)
    assert abs(float(out['loss']) - float(g('loss'))) / float(g('loss')) < TOL
    np.testing.assert_allclose([float(out['losses'][k]) for k in ('mel', 'duration', 'pitch')], g('losses'), rtol=TOL)
    assert _rel(out['mel'].detach().numpy(), g('out_mel')) < TOL
    assert _rel(out['duration'].detach().numpy(), g('out_duration')) < TOL
    assert _rel(out['pitch'].detach().numpy(), g('out_pitch')) < TOL
    np.testing.assert_array_equal(out['expanded_mask'].detach().numpy(), g('out_expanded_mask'))
    enc_key, dec_key = str(g('enc_attn_key')), str(g('dec_attn_key'))
    assert _rel(out['encoder_attention'][enc_key].detach().numpy(), g('enc_attn')) < TOL
    assert _rel(out['decoder_attention'][dec_key].detach().numpy(), g('dec_attn')) < TOL
    gmax = max(float(np.abs(run[k]).max()) for k in run if k.startswith(f'{name}/grad::'))
    n = 0
    for k, gr in out['grads'].items():
        want = g(f'grad::{k}')
        assert want.shape == tuple(gr.shape), k
        assert np.abs(gr.detach().numpy() - want).max() < TOL * gmax, k    # every variable of the reference's step
        n += 1
    assert n == len([k for k in run if k.startswith(f'{name}/grad::')]) == len(fo.weight_spec(cfg))
    # inference through the reference's predict(): 1/speed scaling and the per-symbol duration clamps
    # (models.py:559-595), restated here as the masks its _make_max/min_duration_mask build
    row = g('pred_tokens')
    other = int(next(x for x in row if x != row[0]))
    W2 = dict(W)
    W2['dur.lin.b'] = W['dur.lin.b'] + float(g('pred_bias_shift'))
    mx = np.where(row == row[0], 2.0, np.inf)[None]
    mn = np.where(row == other, 4.0, 0.0)[None]
    with torch.no_grad():
        inf = fo.ForwardTransformerOracle(cfg, W2, torch.float64).call(
            torch.from_numpy(row)[None], training=False, durations_scalar=1.0 / 0.8,
            max_durations_mask=torch.from_numpy(mx), min_durations_mask=torch.from_numpy(mn))
    assert _rel(inf['duration'].numpy(), g('pred_duration')) < TOL
    assert tuple(inf['mel'][0].shape) == g('pred_mel').shape
    assert _rel(inf['mel'][0].numpy(), g('pred_mel')) < TOL


@pytest.mark.parametrize('name', list(gen.CASES))
def test_oracle_equals_the_frozen_run_of_the_reference_source(frozen, name):
    _check_oracle_against(frozen, name)


@pytest.mark.skipif(not os.path.isdir(gen.REF), reason='/root/reference not present (only in the build container)')
def test_oracle_equals_the_reference_source_executed_now(frozen):
    """Runs in a subprocess: the stand-in modules must not leak into this interpreter's sys.modules."""
    import subprocess
    import tempfile
    code = ('import sys, numpy as np; sys.dont_write_bytecode = True; sys.path.insert(0, %r); sys.path.insert(0, %r);'
            'import make_reference_source_run as g, _tf_shim; _tf_shim.install(); sys.path.insert(0, g.REF);'
            'out = {}; [out.update({n + "/" + k: v for k, v in g.run_case(n).items()}) for n in g.CASES];'
            'np.savez(sys.argv[1], **out)') % (HERE, os.path.join(HERE, 'golden'))
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'live.npz')
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH=os.path.dirname(HERE))
        r = subprocess.run([sys.executable, '-c', code, p], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        with np.load(p) as z:
            live = {k: z[k] for k in z.files}
    assert set(live) == set(frozen)
    for k in live:                                            # the committed fixture is what the reference computes
        if live[k].dtype.kind in 'fc':
            np.testing.assert_allclose(live[k], frozen[k], rtol=0, atol=1e-12 * max(1.0, float(np.abs(frozen[k]).max())))
        else:
            np.testing.assert_array_equal(live[k], frozen[k])
    for name in gen.CASES:
        _check_oracle_against(live, name)

"""Model factory mirror (reference model/factory.py:10-29).

`tts_custom(config_path, weights_path)` keeps the reference signature.  `tts_ljspeech` downloads a
released checkpoint from S3 in the reference (model/factory.py:10-19); there is no network here and
the release is a Keras HDF5 file, so it raises with instructions instead of pretending."""
from __future__ import annotations

from typing import Tuple

import yaml

from .models import ForwardTransformer


def tts_custom(config_path: str, weights_path: str, **overrides) -> Tuple[ForwardTransformer, dict]:
    with open(config_path, 'rb') as f:
        config = yaml.safe_load(f)
    for k in ('alphabet', 'step', 'git_hash', 'automatic'):
        config.pop(k, None)
    config.update(overrides)
    model = ForwardTransformer.from_config(config)
    model.build_model_weights()
    model.load_weights(weights_path)
    return model, config


def tts_ljspeech(step='95000'):
    raise RuntimeError('tts_ljspeech downloads bdf06b9_ljspeech_step_%s.zip (Keras HDF5) from S3 in the '
                       'reference; offline, convert the checkpoint to .npz with the reference variable '
                       'names and call tts_custom(config_path, weights_path)' % step)

"""Model factory mirror (reference model/factory.py:10-29).

`tts_custom(config_path, weights_path)` keeps the reference signature.  `tts_ljspeech` downloads a
released checkpoint from S3 in the reference (model/factory.py:10-19); downloading is out of scope, so it
raises and points at `tts_custom`, which loads the unpacked release (its Keras HDF5 weights) directly."""
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
    raise RuntimeError('tts_ljspeech downloads bdf06b9_ljspeech_step_%s.zip from S3 in the reference (downloads are '
                       'out of scope here): unpack the release yourself and call '
                       'tts_custom("<dir>/config.yaml", "<dir>/model_weights.hdf5") - the Keras HDF5 file loads as is'
                       % step)

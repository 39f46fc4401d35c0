"""Host-side constants of the hot path (reference model/transformer_utils.py:5-21).

The sinusoid table is an init-time constant: computed once in float64 numpy exactly as the
reference does (exponent 2*(i//2)/float32(model_dim), sin on even columns, cos on odd columns) and
cast to float32; the kernels only read it.  The padding masks of transformer_utils.py:24-32 are
device kernels (ttsmi_token_pad_mask / ttsmi_length_pad_mask)."""
import numpy as np


def get_angles(pos, i, model_dim):
    return pos * (1.0 / np.power(10000, (2 * (i // 2)) / np.float32(model_dim)))


def positional_encoding(position: int, model_dim: int) -> np.ndarray:
    """[position, model_dim] float32 (the reference adds a leading broadcast axis)."""
    ang = get_angles(np.arange(position)[:, None], np.arange(model_dim)[None, :], model_dim)
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return ang.astype(np.float32)

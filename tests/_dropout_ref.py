"""NumPy restatement of libttsmi's counter-based dropout decisions (transformertts_amd/csrc/common.h:
ttsmi_drop_key / ttsmi_row_base / ttsmi_pair_hash / ttsmi_keep_of), written from the comment block that documents them,
so that parity tests with dropout ON can be held to an fp64 reference that applies the SAME keep mask (the reference's
own dropout - tf.keras.layers.Dropout, model/layers.py:92,97,150,191 - draws from TensorFlow's RNG, which no
re-implementation can reproduce; what must match is inverted dropout with rate p on the same elements).

    keep(seed, step, site, row, col) -> bool          row = flat row of the tensor ((b*H + h)*T + q for attention
                                                       weights), col = column (key index for attention weights)
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z: int) -> int:
    z &= 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def _mix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h = h ^ (h >> np.uint32(16))
    h = h * np.uint32(0x7feb352d)
    h = h ^ (h >> np.uint32(15))
    h = h * np.uint32(0x846ca68b)
    h = h ^ (h >> np.uint32(16))
    return h


def drop_key(seed: int, step, site: int) -> int:
    s = seed if step is None else (seed + 0xA0761D6478BD642F * int(step)) & 0xFFFFFFFFFFFFFFFF
    return _mix64((s + 0x9E3779B97F4A7C15 * (site + 1)) & 0xFFFFFFFFFFFFFFFF)


def threshold(p: float) -> int:
    t = float(np.float32(p)) * 65536.0 + 0.5
    return int(min(max(t, 0.0), 65535.0))


def keep_mask(seed: int, step, site: int, rows: np.ndarray, cols: np.ndarray, p: float) -> np.ndarray:
    """Boolean keep decisions for the outer product rows x cols (rows, cols: 1-D integer arrays)."""
    with np.errstate(over='ignore'):
        key = drop_key(seed, step, site)
        rb = _mix32(np.asarray(rows, np.uint32) ^ np.uint32(key & 0xFFFFFFFF)) + np.uint32(key >> 32)
        cols = np.asarray(cols, np.uint32)
        h = _mix32(rb[:, None] + (cols >> np.uint32(1))[None, :] * np.uint32(0x85EBCA6B))
        u = np.where((cols & np.uint32(1))[None, :] != 0, h >> np.uint32(16), h & np.uint32(0xFFFF))
        return u >= np.uint32(threshold(p))

"""Batch producer for the ForwardTransformer train loop: the bucket-by-mel-length batching and zero
padding of the reference's `data/datasets.py:238-284` (a `tf.data` pipeline there:
`from_generator -> bucket_by_sequence_length -> repeat`), rebuilt as a plain producer thread that
ends in pinned host buffers and an asynchronous copy to the GPU.

Semantics kept from `tf.data.experimental.bucket_by_sequence_length` as the reference calls it:
  * `len(bucket_batch_sizes) == len(bucket_boundaries) + 1`; a sample of length L goes to bucket
    `i = #{b in boundaries : b <= L}` (so `L < boundaries[0]` is bucket 0 and `L >= boundaries[-1]`
    the last one);
  * samples stream through in (shuffled) order; a bucket emits a batch the moment it holds
    `bucket_batch_sizes[i]` samples; at the end of a pass the partially filled buckets are emitted in
    bucket order unless `drop_remainder`;
  * every component is zero-padded (or `padding_values[c]`) along its first axis to the longest
    sample of the batch; scalar / string components are stacked;
  * `next_batch()` repeats for ever, drawing a fresh shuffle from the seeded RNG at every pass
    (`datasets.py:276-284`); `all_batches()` is one pass.
The trainer's unpacking `mel, phonemes, durations, pitch, fname = dataset.next_batch()`
(`train_tts.py:151`) works unchanged with `TTSPreprocessor`.

With `device=` the numeric components of each batch are staged in pinned memory and copied with
`non_blocking=True` on a dedicated copy stream by the producer thread, `prefetch` batches ahead - the
train step's inputs are resident in HBM before the step needs them (the hot path never waits on PCIe).
One shape per bucket keeps the set of distinct batch shapes small (at most one per bucket and pass,
plus remainders), which is what a hipGraph-per-shape cache wants.
"""
from __future__ import annotations

import queue
import threading
from pathlib import Path
from random import Random
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

try:                                    # torch is only needed for the pinned / device staging
    import torch
except Exception:                       # pragma: no cover
    torch = None


def bucket_index(length: int, boundaries: Sequence[int]) -> int:
    """Bucket of a sample of `length`: the number of boundaries <= length."""
    i = 0
    for b in boundaries:
        if length >= b:
            i += 1
        else:
            break
    return i


def pad_and_stack(components: List, pad_value=0):
    """Stack one component of a batch: arrays are padded along axis 0 to the longest one."""
    first = components[0]
    if isinstance(first, (str, bytes)) or np.ndim(first) == 0:
        if isinstance(first, (str, bytes)):
            return list(components)
        return np.asarray(components)
    arrs = [np.asarray(c) for c in components]
    tmax = max(a.shape[0] for a in arrs)
    out = np.full((len(arrs), tmax) + arrs[0].shape[1:], pad_value, dtype=arrs[0].dtype)
    for i, a in enumerate(arrs):
        out[i, :a.shape[0]] = a
    return out


class Dataset:
    """Model-digestible dataset (reference `data/datasets.py:238-284`).  `samples` are opaque keys,
    `preprocessor(key)` returns the tuple of components of one sample, `len_function(*components)`
    its bucketing length."""

    def __init__(self, samples: list, preprocessor: Callable, len_function: Callable,
                 bucket_boundaries: list, bucket_batch_sizes: list, padded_shapes: tuple = None,
                 output_types: tuple = None, padding_values: tuple = None, shuffle: bool = True,
                 drop_remainder: bool = True, seed: int = 42, device=None, prefetch: int = 2):
        if len(bucket_batch_sizes) != len(bucket_boundaries) + 1:
            raise ValueError('bucket_batch_sizes must have len(bucket_boundaries) + 1 entries')
        if list(bucket_boundaries) != sorted(bucket_boundaries):
            raise ValueError('bucket_boundaries must be increasing')
        self._random = Random(seed)
        self._samples = samples[:]
        self.preprocessor = preprocessor
        self.len_function = len_function
        self.bucket_boundaries = list(bucket_boundaries)
        self.bucket_batch_sizes = list(bucket_batch_sizes)
        self.padding_values = padding_values
        self.shuffle = shuffle
        self.drop_remainder = drop_remainder
        self.output_types = output_types
        self.device = device
        self.prefetch = max(1, int(prefetch))
        self._q: Optional[queue.Queue] = None
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()

    # ------------------------------------------------------------------ one pass
    def _datagen(self, shuffle: bool):
        samples = self._samples[:]
        if shuffle:
            self._random.shuffle(samples)                     # shuffle once per pass (datasets.py:276-284)
        return (self.preprocessor(s) for s in samples)

    def _collate(self, items: List[tuple]) -> tuple:
        n = len(items[0])
        pv = self.padding_values or (0,) * n
        return tuple(pad_and_stack([it[c] for it in items], pv[c] if pv[c] is not None else 0) for c in range(n))

    def _one_pass(self, shuffle: bool) -> Iterator[tuple]:
        buckets: List[List[tuple]] = [[] for _ in self.bucket_batch_sizes]
        for item in self._datagen(shuffle):
            i = bucket_index(int(self.len_function(*item)), self.bucket_boundaries)
            buckets[i].append(item)
            if len(buckets[i]) == self.bucket_batch_sizes[i]:
                yield self._collate(buckets[i])
                buckets[i] = []
        if not self.drop_remainder:
            for b in buckets:
                if b:
                    yield self._collate(b)

    def all_batches(self) -> Iterator[tuple]:
        """One pass over the data (validation: `train_tts.py:48-55`)."""
        for batch in self._one_pass(self.shuffle):
            yield self._to_device(batch)

    # ------------------------------------------------------------------ endless, prefetched
    def _to_device(self, batch: tuple) -> tuple:
        """Numeric components -> device tensors through a RING of persistent pinned staging buffers (one flat pinned
        allocation per slot, grown when a batch needs more).  Round 4: `tensor.pin_memory()` per component and batch - a
        pinned allocation each time - ran the bucketed LJ-dist workload at 26 ms per step beside a 3.8 ms train step
        (bench.py --workload lj-dist, --lj-preload for the step alone): hipHostMalloc holds runtime locks the launching
        thread needs.  A slot is reused only after the copy that read it has completed (its event)."""
        if self.device is None or torch is None:
            return batch
        if not torch.cuda.is_available():
            return tuple(torch.from_numpy(np.ascontiguousarray(c)).to(self.device)
                         if isinstance(c, np.ndarray) and c.dtype.kind in 'fiu' else c for c in batch)
        numeric = [(i, np.ascontiguousarray(c)) for i, c in enumerate(batch) if isinstance(c, np.ndarray) and c.dtype.kind in 'fiu']
        need = sum((a.nbytes + 255) // 256 * 256 for _, a in numeric)
        # one ring PER CALLING THREAD (the producer thread of next_batch and the caller's thread of all_batches may both
        # be here on one Dataset: a shared slot counter let them pick the same slot and overwrite pinned memory an
        # in-flight copy was still reading)
        rings = self.__dict__.setdefault('_stage_rings', {})
        st = rings.get(threading.get_ident())
        if st is None:
            st = rings.setdefault(threading.get_ident(), {'ring': [], 'next': 0})
        ring = st['ring']
        nslots = self.prefetch + 2                          # in the queue + being consumed + being filled
        k = st['next']
        st['next'] = (k + 1) % nslots
        while len(ring) < nslots:
            ring.append({'buf': None, 'ev': None})
        slot = ring[k]
        if slot['ev'] is not None:
            slot['ev'].synchronize()                        # the copy that last read this slot (prefetch + 2 batches ago)
        if slot['buf'] is None or slot['buf'].numel() < need:
            slot['buf'] = torch.empty(int(need * 5 // 4) + 4096, dtype=torch.uint8).pin_memory()
        out = list(batch)
        off = 0
        dev_t = torch.device(self.device)
        # (under the Dataset's own device: the copies run on - and the event is recorded on - THAT device's current
        # stream, whatever device the calling thread has current)
        with torch.cuda.device(dev_t if dev_t.index is not None else torch.cuda.current_device()):
            for i, a in numeric:
                stage = slot['buf'][off:off + a.nbytes].view(torch.from_numpy(a).dtype).reshape(a.shape)
                stage.numpy()[...] = a                      # one memcpy into pinned memory (numpy releases the GIL for it)
                dev = torch.empty(a.shape, dtype=stage.dtype, device=self.device)
                dev.copy_(stage, non_blocking=True)
                out[i] = dev
                off += (a.nbytes + 255) // 256 * 256
            ev = torch.cuda.Event()
            ev.record()
        slot['ev'] = ev
        return tuple(out)

    def _producer(self):
        copy_stream = None
        if self.device is not None and torch is not None and torch.cuda.is_available():
            copy_stream = torch.cuda.Stream(device=self.device)
        try:
            while not self._stop.is_set():
                produced = False
                for batch in self._one_pass(self.shuffle):
                    produced = True
                    if copy_stream is not None:
                        with torch.cuda.stream(copy_stream):
                            staged = self._to_device(batch)
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                        item = (staged, ev)
                    else:
                        item = (self._to_device(batch), None)
                    while not self._stop.is_set():
                        try:
                            self._q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if self._stop.is_set():
                        return
                if not produced:
                    raise RuntimeError('the dataset produced no batch in a full pass '
                                       '(every bucket smaller than its batch size with drop_remainder=True?)')
        except BaseException as e:                            # surface producer errors in next_batch()
            self._q.put((e, None))

    def next_batch(self) -> tuple:
        """Next batch of the endless, reshuffled-every-pass stream (`train_tts.py:151`)."""
        if self._thread is None:
            self._q = queue.Queue(maxsize=self.prefetch)
            self._stop.clear()
            self._thread = threading.Thread(target=self._producer, name='ttsmi-batch-producer', daemon=True)
            self._thread.start()
        batch, ev = self._q.get()
        if isinstance(batch, BaseException):
            raise batch
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)                                # order the step after the H2D copies
            for comp in batch:                                # allocated on the copy stream, used on this one
                if torch.is_tensor(comp):
                    comp.record_stream(cur)
        return batch

    def close(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
            self._thread = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TTSPreprocessor:
    """Reference `data/datasets.py:153-169`: (text, mel, durations, pitch, name) -> the trainer's tuple
    `(mel f32 [T,80], phonemes i32 [Tp], durations i32 [Tp], pitch f32 [Tp], name)`."""

    def __init__(self, mel_channels: int, tokenizer):
        self.mel_channels = mel_channels
        self.tokenizer = tokenizer
        self.output_types = ('float32', 'int32', 'int32', 'float32', 'str')
        self.padded_shapes = ([None, mel_channels], [None], [None], [None], [])

    def __call__(self, text, mel, durations, pitch, sample_name):
        encoded_phonemes = np.asarray(self.tokenizer(text), dtype=np.int32)
        return (np.asarray(mel, dtype=np.float32), encoded_phonemes, np.asarray(durations, dtype=np.int32),
                np.asarray(pitch, dtype=np.float32), sample_name)

    def get_sample_length(self, mel, encoded_phonemes, durations, pitch, sample_name):
        return int(np.shape(mel)[0])


class TTSDataset:
    """Reference `data/datasets.py:172-235`: reads `<mel_dir>/<name>.npy`, `<duration_dir>/<name>.npy`,
    `<pitch_per_char_dir>/<name>.npy` and the phonemized text of each sample.  `text_dict` maps sample
    name -> phoneme string (the reference gets it from its metadata reader, which is out of scope)."""

    def __init__(self, text_dict: dict, preprocessor: TTSPreprocessor, mel_directory: str,
                 duration_directory: str, pitch_per_char_directory: str, filenames: Iterable[str] = None):
        self.text_dict = dict(text_dict)
        self.filenames = list(filenames) if filenames is not None else list(self.text_dict.keys())
        self.preprocessor = preprocessor
        self.mel_directory = Path(mel_directory)
        self.duration_directory = Path(duration_directory)
        self.pitch_per_char_directory = Path(pitch_per_char_directory)

    def _read_sample(self, sample_name: str) -> Tuple:
        text = self.text_dict[sample_name]
        mel = np.load((self.mel_directory / sample_name).with_suffix('.npy').as_posix())
        durations = np.load((self.duration_directory / sample_name).with_suffix('.npy').as_posix())
        pitch = np.load((self.pitch_per_char_directory / sample_name).with_suffix('.npy').as_posix())
        return mel, text, durations, pitch

    def _process_sample(self, sample_name: str):
        mel, text, durations, pitch = self._read_sample(sample_name)
        return self.preprocessor(mel=mel, text=text, durations=durations, pitch=pitch, sample_name=sample_name)

    def get_dataset(self, bucket_batch_sizes, bucket_boundaries, shuffle=True, drop_remainder=False,
                    device=None, prefetch: int = 2, seed: int = 42) -> Dataset:
        return Dataset(samples=self.filenames, preprocessor=self._process_sample,
                       len_function=self.preprocessor.get_sample_length,
                       output_types=self.preprocessor.output_types,
                       padded_shapes=self.preprocessor.padded_shapes, shuffle=shuffle,
                       drop_remainder=drop_remainder, bucket_batch_sizes=bucket_batch_sizes,
                       bucket_boundaries=bucket_boundaries, device=device, prefetch=prefetch, seed=seed)

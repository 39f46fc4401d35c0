"""Test infrastructure: a torch-float64 stand-in for the ~45 TensorFlow / Keras names the reference's
ForwardTransformer path uses (`grep -o 'tf\\.[A-Za-z_.]*' model/layers.py model/models.py
model/transformer_utils.py utils/losses.py`), so that the reference's OWN model source - constructors, `call`
bodies, the train / val step, the loss wiring - can be imported from /root/reference and EXECUTED in this
container, where TensorFlow cannot be installed.

What this does and does not establish.  The graph that runs is the reference's, statement by statement: which
tensor is added to which, where the masks enter, `Dense(concat([q_in, ctx]))`, the predictors' order of
conv / activation / norm / dropout, the (un)masked MAE, the attribute order that decides Keras' weight order.
The PRIMITIVES under it are restated here from the public TF / Keras documentation, one small function
each, independently of oracle/ft_oracle.py: Dense = x.W + b, Conv1D('same') = correlation with (k-1)//2 left
padding, LayerNormalization = biased variance with epsilon inside the square root, softmax, Embedding =
row lookup, Dropout(rate 0 or training=False) = identity, RaggedTensor.to_tensor = zero padding,
MeanAbsoluteError = mean over all elements, `tf.math.round` = round-half-even.  So a disagreement between this
run and the oracle is a transcription error in one of them; agreement pins the oracle to the reference's
wiring, not to TensorFlow's floating point.

Tensors are torch.float64 (the oracle's "truth" dtype); `tf.cast(numpy, tf.float32)` rounds through float32
first because the reference builds its constants (the sinusoid table) that way.  Autograd: `tf.GradientTape`
is torch.autograd.  Never imported by the product."""
import sys
import types

import numpy as np
import torch

F64 = torch.float64


def _t(x):
    if isinstance(x, torch.Tensor):
        return x
    a = np.asarray(x)
    if a.dtype.kind == 'f':
        return torch.from_numpy(a.astype(np.float64))
    return torch.from_numpy(a)


def _int(v):
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


def _shape(dims):
    if isinstance(dims, (list, tuple)):
        return tuple(_int(d) for d in dims)
    return (_int(dims),)


# ------------------------------------------------------------------------------------------ tf.* functions
def cast(x, dtype=None, **kw):
    dtype = kw.get('dtype', dtype)
    if not isinstance(x, torch.Tensor):
        a = np.asarray(x)
        if dtype is F64 and a.dtype.kind == 'f':
            a = a.astype(np.float32)                     # the reference's float32 constants
        x = _t(a)
    return x.to(dtype)


def shape(x):
    return _t(x).shape


def expand_dims(x, axis):
    return _t(x).unsqueeze(axis)


def squeeze(x, axis=None):
    x = _t(x)
    if axis is None:
        return x.squeeze()
    if isinstance(axis, (list, tuple)):
        for a in sorted([a % x.dim() for a in axis], reverse=True):
            x = x.squeeze(a)
        return x
    return x.squeeze(axis)


def reshape(x, shape_):
    return _t(x).reshape(_shape(shape_))


def reduce_sum(x, axis=None):
    x = _t(x)
    return x.sum() if axis is None else x.sum(axis)


def reduce_mean(x, axis=None):
    x = _t(x)
    return x.mean() if axis is None else x.mean(axis)


def reduce_max(x, axis=None):
    x = _t(x)
    return x.max() if axis is None else x.max(axis).values


def concat(values, axis):
    return torch.cat([_t(v) for v in values], dim=axis)


def tile(x, multiples):
    return _t(x).repeat(*_shape(multiples))


def transpose(x, perm=None):
    x = _t(x)
    return x.permute(*perm) if perm is not None else x.permute(*reversed(range(x.dim())))


def matmul(a, b, transpose_b=False):
    b = _t(b)
    return torch.matmul(_t(a), b.transpose(-1, -2) if transpose_b else b)


def ones(shape_, dtype=F64):
    return torch.ones(_shape(shape_), dtype=dtype)


def zeros(shape_, dtype=F64):
    return torch.zeros(_shape(shape_), dtype=dtype)


def multiply(a, b):
    return _t(a) * _t(b)


def _round(x):
    x = _t(x)
    return x if not x.is_floating_point() else torch.round(x)      # half to even, like tf.math.round


def convert_to_tensor(x, dtype=None):
    return cast(x, dtype) if dtype is not None else _t(x)


class RaggedTensor:
    def __init__(self, values, row_lengths):
        self.values, self.row_lengths = values, [_int(n) for n in row_lengths]

    @classmethod
    def from_row_lengths(cls, values, row_lengths):
        return cls(_t(values), _t(row_lengths))

    def to_tensor(self):
        n, m = len(self.row_lengths), max(self.row_lengths + [0])
        out = torch.zeros((n, m) + tuple(self.values.shape[1:]), dtype=self.values.dtype)
        o = 0
        rows = []
        for i, L in enumerate(self.row_lengths):
            rows.append(torch.cat([self.values[o:o + L], out[i, L:]], 0))
            o += L
        return torch.stack(rows, 0) if rows else out


class GradientTape:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def gradient(self, target, sources):
        return list(torch.autograd.grad(target, list(sources), allow_unused=True, retain_graph=True))


def function(fn=None, input_signature=None, **kw):
    if fn is None:
        return lambda f: f
    return fn


class TensorSpec:
    def __init__(self, shape=None, dtype=None, name=None):
        self.shape, self.dtype = shape, dtype


def Variable(value, trainable=True, **kw):
    return torch.nn.Parameter(torch.as_tensor(np.asarray(value, dtype=np.float64)), requires_grad=bool(trainable))


# ------------------------------------------------------------------------------------------ Keras layers
class Layer:
    """Keras' tracking rule: a layer's weights = the variables assigned to it directly, then the weights of the
    layers assigned to it (also inside lists), in assignment order, recursively, each variable once."""

    def __init__(self, name=None, **kwargs):
        object.__setattr__(self, '_own', [])
        object.__setattr__(self, '_sub', [])
        object.__setattr__(self, 'name', name or type(self).__name__.lower())

    def __setattr__(self, key, value):
        if isinstance(value, torch.nn.Parameter):
            self._own.append(value)
        elif isinstance(value, Layer):
            self._sub.append(value)
        elif isinstance(value, (list, tuple)) and value and all(isinstance(v, Layer) for v in value):
            self._sub.extend(value)
        object.__setattr__(self, key, value)

    def add_weight(self, shape_, init=0.0):
        p = torch.nn.Parameter(torch.full(tuple(shape_), float(init), dtype=F64))
        self._own.append(p)
        return p

    def __call__(self, *args, **kwargs):
        return self.call(*args, **kwargs)

    @property
    def weights(self):
        out, seen = [], set()
        for w in self._own + [w for s in self._sub for w in s.weights]:
            if id(w) not in seen:
                seen.add(id(w))
                out.append(w)
        return out

    trainable_weights = trainable_variables = weights

    @property
    def layers(self):
        return list(self._sub)


class Model(Layer):
    def compile(self, loss=None, loss_weights=None, optimizer=None, **kw):
        self.loss, self.optimizer = loss, optimizer


class Dense(Layer):
    def __init__(self, units, activation=None, **kw):
        super().__init__(**kw)
        self.units, self.activation, self.kernel = units, activation, None

    def call(self, x):
        x = _t(x)
        if self.kernel is None:
            self.kernel = self.add_weight((x.shape[-1], self.units))
            self.bias = self.add_weight((self.units,))
        y = torch.matmul(x, self.kernel) + self.bias
        return _activation(self.activation)(y)


class Conv1D(Layer):
    def __init__(self, filters, kernel_size, padding='valid', **kw):
        super().__init__(**kw)
        assert padding == 'same'
        self.filters, self.k, self.kernel = filters, kernel_size, None

    def call(self, x):                       # x [B, T, Cin], kernel [k, Cin, Cout]; correlation, stride 1
        if self.kernel is None:
            self.kernel = self.add_weight((self.k, x.shape[-1], self.filters))
            self.bias = self.add_weight((self.filters,))
        left = (self.k - 1) // 2
        xp = torch.nn.functional.pad(x, (0, 0, left, self.k - 1 - left))
        T = x.shape[1]
        y = self.bias
        for j in range(self.k):
            y = y + torch.matmul(xp[:, j:j + T], self.kernel[j])
        return y


class LayerNormalization(Layer):
    def __init__(self, epsilon=1e-3, **kw):
        super().__init__(**kw)
        self.epsilon, self.gamma = epsilon, None

    def call(self, x):
        if self.gamma is None:
            self.gamma = self.add_weight((x.shape[-1],), 1.0)
            self.beta = self.add_weight((x.shape[-1],), 0.0)
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        return (x - mean) / torch.sqrt(var + self.epsilon) * self.gamma + self.beta


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, **kw):
        super().__init__(**kw)
        self.embeddings = self.add_weight((input_dim, output_dim))

    def call(self, ids):
        return self.embeddings[_t(ids).long()]


class Dropout(Layer):
    def __init__(self, rate=0.0, **kw):
        super().__init__(**kw)
        self.rate = rate

    def call(self, x, training=False):
        if training and float(self.rate) > 0:
            raise RuntimeError('the stand-in runs the reference deterministically: dropout must be 0 when training')
        return x


def _activation(name):
    if name in (None, 'linear'):
        return lambda x: x
    if name == 'relu':
        return torch.relu
    raise ValueError(name)


class Activation(Layer):
    def __init__(self, activation, **kw):
        super().__init__(**kw)
        self.fn = _activation(activation)

    def call(self, x):
        return self.fn(x)


class MeanAbsoluteError:
    def __call__(self, y_true, y_pred, sample_weight=None):
        assert sample_weight is None
        y_pred = _t(y_pred)
        return (y_pred - _t(y_true).to(y_pred.dtype)).abs().mean()


class RecordingOptimizer:
    """Receives what the reference's train step hands to `optimizer.apply_gradients` (models.py:481)."""

    def __init__(self):
        self.applied = None

    def apply_gradients(self, grads_and_vars):
        self.applied = list(grads_and_vars)


class _Unused:
    def __init__(self, *a, **k):
        pass


def install():
    """Put the stand-in (and empty modules for the other absent imports) into sys.modules."""
    tf = types.ModuleType('tensorflow')
    tf.float32, tf.int32, tf.string, tf.newaxis, tf.Tensor = F64, torch.int64, str, None, torch.Tensor
    for f in (cast, shape, expand_dims, squeeze, reshape, reduce_sum, reduce_mean, reduce_max, concat, tile,
              transpose, matmul, ones, zeros, multiply, convert_to_tensor, function):
        setattr(tf, f.__name__, f)
    tf.maximum, tf.abs, tf.argmax = torch.maximum, torch.abs, torch.argmax
    tf.RaggedTensor, tf.GradientTape, tf.TensorSpec, tf.Variable = RaggedTensor, GradientTape, TensorSpec, Variable
    tf.math = types.SimpleNamespace(
        equal=lambda a, b: torch.eq(_t(a), b), logical_not=torch.logical_not, sqrt=lambda x: torch.sqrt(_t(x).to(F64)),
        round=_round, minimum=lambda a, b: torch.minimum(_t(a).to(F64), _t(b).to(F64)),
        maximum=lambda a, b: torch.maximum(_t(a).to(F64), _t(b).to(F64)), abs=torch.abs, reduce_sum=reduce_sum,
        reduce_max=reduce_max)
    tf.nn = types.SimpleNamespace(softmax=lambda x, axis=-1: torch.softmax(x, dim=axis))
    tf.linalg = types.SimpleNamespace(band_part=None)
    layers = types.SimpleNamespace(Layer=Layer, Dense=Dense, Conv1D=Conv1D, LayerNormalization=LayerNormalization,
                                   Embedding=Embedding, Dropout=Dropout, Activation=Activation)
    losses = types.SimpleNamespace(MeanAbsoluteError=MeanAbsoluteError, MeanSquaredError=_Unused,
                                   SparseCategoricalCrossentropy=_Unused, BinaryCrossentropy=_Unused)
    tf.keras = types.SimpleNamespace(layers=layers, losses=losses, models=types.SimpleNamespace(Model=Model))
    sys.modules['tensorflow'] = tf
    for name in ('librosa', 'librosa.display', 'matplotlib', 'matplotlib.pyplot', 'soundfile', 'webrtcvad', 'pyworld',
                 'phonemizer', 'phonemizer.phonemize', 'ruamel', 'ruamel.yaml'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['phonemizer'].phonemize = sys.modules['phonemizer.phonemize']
    sys.modules['phonemizer.phonemize'].phonemize = None
    sys.modules['ruamel'].yaml = sys.modules['ruamel.yaml']
    sys.modules['ruamel.yaml'].YAML = _Unused
    return tf

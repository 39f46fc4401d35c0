"""Test infrastructure: the REAL libhdf5 (the C library under h5py, which Keras uses to write the
reference's `model_weights.hdf5`) driven through ctypes - h5py itself is not installed here, the shared
library happens to be (`/opt/conda/lib/libhdf5.so.103` = HDF5 1.10.6 in this image).  Used to
  * write Keras-layout weight files exactly the way h5py does (fixed-length NUL-padded string arrays
    for `layer_names` / `weight_names`, variable-length UTF-8 scalars for `backend` / `keras_version`,
    UTF-8 attribute names, contiguous float32 datasets created through intermediate groups), which
    pins `transformertts_amd.utils.hdf5_min`'s reader against the real writer, and
  * read back what `hdf5_min.Writer` produced, which pins the writer against the real reader.
Never imported by the product."""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

hid_t = ctypes.c_int64
_lib = None


def find():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get('TTSMI_LIBHDF5'), ctypes.util.find_library('hdf5')]
    cands += sorted(glob.glob('/opt/conda/lib/libhdf5.so*')) + sorted(glob.glob('/usr/lib/x86_64-linux-gnu/libhdf5*.so*'))
    for c in cands:
        if not c:
            continue
        try:
            lib = ctypes.CDLL(c)
            if lib.H5open() >= 0:
                _lib = lib
                break
        except OSError:
            continue
    if _lib is None:
        return None
    L = _lib
    for name, res, args in [
        ('H5Fcreate', hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
        ('H5Fopen', hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
        ('H5Fclose', ctypes.c_int, [hid_t]),
        ('H5Gcreate2', hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]),
        ('H5Gopen2', hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        ('H5Gclose', ctypes.c_int, [hid_t]),
        ('H5Screate', hid_t, [ctypes.c_int]),
        ('H5Screate_simple', hid_t, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
        ('H5Sclose', ctypes.c_int, [hid_t]),
        ('H5Sget_simple_extent_ndims', ctypes.c_int, [hid_t]),
        ('H5Sget_simple_extent_dims', ctypes.c_int, [hid_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
        ('H5Tcopy', hid_t, [hid_t]),
        ('H5Tset_size', ctypes.c_int, [hid_t, ctypes.c_size_t]),
        ('H5Tset_strpad', ctypes.c_int, [hid_t, ctypes.c_int]),
        ('H5Tset_cset', ctypes.c_int, [hid_t, ctypes.c_int]),
        ('H5Tget_size', ctypes.c_size_t, [hid_t]),
        ('H5Tget_class', ctypes.c_int, [hid_t]),
        ('H5Tis_variable_str', ctypes.c_int, [hid_t]),
        ('H5Tclose', ctypes.c_int, [hid_t]),
        ('H5Pcreate', hid_t, [hid_t]),
        ('H5Pset_char_encoding', ctypes.c_int, [hid_t, ctypes.c_int]),
        ('H5Pset_create_intermediate_group', ctypes.c_int, [hid_t, ctypes.c_uint]),
        ('H5Pset_libver_bounds', ctypes.c_int, [hid_t, ctypes.c_int, ctypes.c_int]),
        ('H5Pset_chunk', ctypes.c_int, [hid_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
        ('H5Pset_layout', ctypes.c_int, [hid_t, ctypes.c_int]),
        ('H5Pclose', ctypes.c_int, [hid_t]),
        ('H5Dcreate2', hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        ('H5Dopen2', hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        ('H5Dwrite', ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        ('H5Dread', ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        ('H5Dget_space', hid_t, [hid_t]),
        ('H5Dget_type', hid_t, [hid_t]),
        ('H5Dclose', ctypes.c_int, [hid_t]),
        ('H5Acreate2', hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        ('H5Aopen', hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        ('H5Awrite', ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]),
        ('H5Aread', ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]),
        ('H5Aget_space', hid_t, [hid_t]),
        ('H5Aget_type', hid_t, [hid_t]),
        ('H5Aclose', ctypes.c_int, [hid_t]),
        ('H5Eset_auto2', ctypes.c_int, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
    ]:
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L.H5Eset_auto2(0, None, None)                            # no error-stack spam on expected failures
    return L


def _g(name):
    return hid_t.in_dll(find(), name).value


def version():
    L = find()
    a, b, c = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    L.H5get_libversion(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return a.value, b.value, c.value


def _ok(v, what):
    if v < 0:
        raise RuntimeError(f'libhdf5: {what} failed ({v})')
    return v


def _acpl():
    L = find()
    p = _ok(L.H5Pcreate(_g('H5P_CLS_ATTRIBUTE_CREATE_ID_g')), 'H5Pcreate(acpl)')
    L.H5Pset_char_encoding(p, 1)                             # h5py names attributes in UTF-8
    return p


def _write_attr_fixed_strings(obj, name, values):
    """numpy 'S' array -> fixed-length NUL-padded strings (what h5py does for bytes arrays)."""
    L = find()
    a = np.asarray(values, dtype='S')
    t = L.H5Tcopy(_g('H5T_C_S1_g'))
    L.H5Tset_size(t, max(a.dtype.itemsize, 1))
    L.H5Tset_strpad(t, 1)                                    # H5T_STR_NULLPAD
    dims = (ctypes.c_uint64 * 1)(a.shape[0])
    s = L.H5Screate_simple(1, dims, None)
    acpl = _acpl()
    at = _ok(L.H5Acreate2(obj, name.encode(), t, s, acpl, 0), f'H5Acreate2({name})')
    buf = np.ascontiguousarray(a)
    _ok(L.H5Awrite(at, t, buf.ctypes.data_as(ctypes.c_void_p)), 'H5Awrite')
    L.H5Aclose(at), L.H5Pclose(acpl), L.H5Sclose(s), L.H5Tclose(t)


def _write_attr_vlen_str(obj, name, value: str):
    """python str -> scalar variable-length UTF-8 string (what h5py does for str)."""
    L = find()
    t = L.H5Tcopy(_g('H5T_C_S1_g'))
    L.H5Tset_size(t, ctypes.c_size_t(-1).value)              # H5T_VARIABLE
    L.H5Tset_cset(t, 1)
    s = L.H5Screate(0)                                       # H5S_SCALAR
    acpl = _acpl()
    at = _ok(L.H5Acreate2(obj, name.encode(), t, s, acpl, 0), f'H5Acreate2({name})')
    ptr = ctypes.c_char_p(value.encode('utf8'))
    _ok(L.H5Awrite(at, t, ctypes.byref(ptr)), 'H5Awrite')
    L.H5Aclose(at), L.H5Pclose(acpl), L.H5Sclose(s), L.H5Tclose(t)


def write_keras_weights(path, layers, libver_latest=False, chunked=False, keras_version='2.4.0', utf8_links=True):
    """`layers` = [(layer_name, [(weight_name, float32 array), ...]), ...] -> the file layout of
    Keras `save_weights_to_hdf5_group` (tensorflow/python/keras/saving/hdf5_format.py)."""
    L = find()
    fapl = 0
    if libver_latest:
        fapl = L.H5Pcreate(_g('H5P_CLS_FILE_ACCESS_ID_g'))
        _ok(L.H5Pset_libver_bounds(fapl, 2, 2) if version() >= (1, 10, 2) else L.H5Pset_libver_bounds(fapl, 1, 1),
            'H5Pset_libver_bounds')
    f = _ok(L.H5Fcreate(str(path).encode(), 2, 0, fapl), 'H5Fcreate')          # H5F_ACC_TRUNC
    _write_attr_fixed_strings(f, 'layer_names', [n.encode('utf8') for n, _ in layers])
    _write_attr_vlen_str(f, 'backend', 'tensorflow')
    _write_attr_vlen_str(f, 'keras_version', keras_version)
    lcpl = L.H5Pcreate(_g('H5P_CLS_LINK_CREATE_ID_g'))
    L.H5Pset_create_intermediate_group(lcpl, 1)
    if utf8_links:                                           # h5py's default; turns the groups new-style
        L.H5Pset_char_encoding(lcpl, 1)
    for lname, weights in layers:
        g = _ok(L.H5Gcreate2(f, lname.encode(), lcpl, 0, 0), f'H5Gcreate2({lname})')   # h5py: UTF-8 lcpl here too
        _write_attr_fixed_strings(g, 'weight_names', [n.encode('utf8') for n, _ in weights])
        for wname, arr in weights:
            a = np.asarray(arr, dtype=np.float32)
            a = a if a.ndim == 0 else np.ascontiguousarray(a)      # ascontiguousarray would make a scalar 1-d
            if a.ndim:
                dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
                s = L.H5Screate_simple(a.ndim, dims, None)
            else:
                s = L.H5Screate(0)
            dcpl = 0
            if chunked and a.ndim:
                dcpl = L.H5Pcreate(_g('H5P_CLS_DATASET_CREATE_ID_g'))
                cd = (ctypes.c_uint64 * a.ndim)(*[max(1, (d + 1) // 2) for d in a.shape])
                _ok(L.H5Pset_chunk(dcpl, a.ndim, cd), 'H5Pset_chunk')
            d = _ok(L.H5Dcreate2(g, wname.encode(), _g('H5T_IEEE_F32LE_g'), s, lcpl, dcpl, 0), f'H5Dcreate2({wname})')
            _ok(L.H5Dwrite(d, _g('H5T_NATIVE_FLOAT_g'), 0, 0, 0, a.ctypes.data_as(ctypes.c_void_p)), 'H5Dwrite')
            L.H5Dclose(d), L.H5Sclose(s)
            if dcpl:
                L.H5Pclose(dcpl)
        L.H5Gclose(g)
    L.H5Pclose(lcpl)
    _ok(L.H5Fclose(f), 'H5Fclose')
    if fapl:
        L.H5Pclose(fapl)


def _read_attr_strings(obj, name):
    L = find()
    at = _ok(L.H5Aopen(obj, name.encode(), 0), f'H5Aopen({name})')
    t, s = L.H5Aget_type(at), L.H5Aget_space(at)
    if L.H5Tis_variable_str(t) > 0:
        raise RuntimeError('variable-length attribute')
    size = L.H5Tget_size(t)
    nd = L.H5Sget_simple_extent_ndims(s)
    dims = (ctypes.c_uint64 * max(nd, 1))()
    if nd:
        L.H5Sget_simple_extent_dims(s, dims, None)
    n = int(np.prod([dims[i] for i in range(nd)])) if nd else 1
    buf = np.zeros(n, dtype=f'S{size}')
    _ok(L.H5Aread(at, t, buf.ctypes.data_as(ctypes.c_void_p)), 'H5Aread')
    L.H5Tclose(t), L.H5Sclose(s), L.H5Aclose(at)
    return [bytes(x) for x in buf] if nd else bytes(buf[0])


def _read_dataset_f32(obj, name):
    L = find()
    d = _ok(L.H5Dopen2(obj, name.encode(), 0), f'H5Dopen2({name})')
    s = L.H5Dget_space(d)
    nd = L.H5Sget_simple_extent_ndims(s)
    dims = (ctypes.c_uint64 * max(nd, 1))()
    if nd:
        L.H5Sget_simple_extent_dims(s, dims, None)
    shape = tuple(int(dims[i]) for i in range(nd))
    out = np.zeros(shape, dtype=np.float32)
    _ok(L.H5Dread(d, _g('H5T_NATIVE_FLOAT_g'), 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)), 'H5Dread')
    L.H5Sclose(s), L.H5Dclose(d)
    return out


def read_keras_weights(path):
    """The reading half of Keras `load_weights_from_hdf5_group`, done by the real library."""
    L = find()
    f = _ok(L.H5Fopen(str(path).encode(), 0, 0), 'H5Fopen')                     # H5F_ACC_RDONLY
    out = []
    meta = {k: _read_attr_strings(f, k) for k in ('backend', 'keras_version')}
    for lname in _read_attr_strings(f, 'layer_names'):
        g = _ok(L.H5Gopen2(f, lname, 0), f'H5Gopen2({lname})')
        ws = [(w.decode('utf8'), _read_dataset_f32(g, w.decode('utf8'))) for w in _read_attr_strings(g, 'weight_names')]
        out.append((lname.decode('utf8'), ws))
        L.H5Gclose(g)
    L.H5Fclose(f)
    return out, meta

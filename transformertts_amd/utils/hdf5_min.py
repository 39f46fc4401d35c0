"""A small, dependency-free reader and writer for the slice of HDF5 that Keras weight files use.

Why: the reference persists its models as `model_weights.hdf5` through Keras
(`model/models.py:600-638`: `save_weights` / `load_weights`, i.e. h5py on top of libhdf5) and h5py is
not installable here.  Checkpoint interchange is host-side byte shuffling that
happens once per run, so this is plain Python + numpy; nothing here touches the GPU.

Scope of the format that is understood (HDF5 File Format Specification v3 names):
  reader  superblock v0/v1 (what libhdf5 / h5py write with the default `libver='earliest'`) and v2/v3;
          object headers v1 (with continuation blocks) and v2 (`OHDR`/`OCHK`); old-style groups (symbol
          table message -> v1 B-tree of `SNOD`s + local heap) and new-style groups (link messages in
          the header, or in a fractal heap once a group has more than 8 links - what h5py files have); datasets with compact, contiguous or unfiltered chunked (v1 B-tree) layout; fixed
          point, floating point, fixed-length string and variable-length string (global heap)
          datatypes; attribute messages v1/v2/v3.  Anything else (dense attribute storage,
          filters, v4 chunk indexes, references, compounds) raises `Hdf5Error` naming what was met - there is
          no silent guess.
  writer  superblock v0, object headers v1, old-style groups, contiguous little-endian datasets,
          attributes of numeric / fixed-length-string type.  That is the conservative subset every
          libhdf5 since 1.6 opens; `tests/test_keras_hdf5.py` has the real libhdf5 read these files
          back whenever the shared library can be found.

The API mirrors the few h5py calls Keras' `hdf5_format` uses: `File(path)`, `group.attrs[name]`,
`group[name]`, `np.asarray(dataset)`; `Writer().root.create_group / create_dataset / attrs`.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(RuntimeError):
    pass


# ======================================================================================== reader
class _Buf:
    """The whole file in memory (weight files are tens of MB) with little-endian field readers."""

    def __init__(self, data: bytes):
        self.d = data
        self.O = 8       # size of offsets, fixed up from the superblock
        self.L = 8       # size of lengths

    def u(self, off: int, n: int) -> int:
        if off < 0 or off + n > len(self.d):
            raise Hdf5Error(f'read of {n} bytes at {off} is outside the file ({len(self.d)} bytes): truncated file?')
        return int.from_bytes(self.d[off:off + n], 'little')

    def off(self, p: int) -> int:
        v = self.u(p, self.O)
        return UNDEF if v == (1 << (8 * self.O)) - 1 else v

    def len_(self, p: int) -> int:
        return self.u(p, self.L)

    def raw(self, off: int, n: int) -> bytes:
        if off < 0 or off + n > len(self.d):
            raise Hdf5Error(f'read of {n} bytes at {off} is outside the file ({len(self.d)} bytes): truncated file?')
        return self.d[off:off + n]


class _Datatype:
    """Decoded datatype message: `np_dtype` for fixed-size classes, `vlen_str` for variable-length strings."""

    def __init__(self, np_dtype: Optional[np.dtype], size: int, vlen_str: bool = False, utf8: bool = False):
        self.np_dtype = np_dtype
        self.size = size
        self.vlen_str = vlen_str
        self.utf8 = utf8


def _parse_datatype(b: bytes) -> _Datatype:
    cls, ver = b[0] & 0x0F, b[0] >> 4
    bits = b[1] | (b[2] << 8) | (b[3] << 16)
    size = int.from_bytes(b[4:8], 'little')
    order = '>' if (bits & 1) else '<'
    if cls == 0:                                             # fixed point
        kind = 'i' if (bits & 0x08) else 'u'
        return _Datatype(np.dtype(f'{order}{kind}{size}'), size)
    if cls == 1:                                             # floating point
        if bits & 0x40:
            raise Hdf5Error('VAX-ordered floating point datatype')
        if size not in (2, 4, 8):
            raise Hdf5Error(f'floating point datatype of {size} bytes')
        return _Datatype(np.dtype(f'{order}f{size}'), size)
    if cls == 3:                                             # fixed-length string
        return _Datatype(np.dtype(f'S{size}'), size, utf8=bool((bits >> 4) & 0x0F))
    if cls == 9:                                             # variable length
        if (bits & 0x0F) != 1:
            raise Hdf5Error('variable-length sequence datatype (only variable-length strings are supported)')
        return _Datatype(None, size, vlen_str=True, utf8=bool((bits >> 8) & 0x0F))
    names = {2: 'time', 4: 'bitfield', 5: 'opaque', 6: 'compound', 7: 'reference', 8: 'enumerated', 10: 'array'}
    raise Hdf5Error(f'unsupported datatype class {cls} ({names.get(cls, "?")}), version {ver}')


def _parse_dataspace(buf: _Buf, b: bytes) -> Optional[Tuple[int, ...]]:
    """Shape tuple; () for a scalar space, None for a null space."""
    ver, rank, flags = b[0], b[1], b[2]
    if ver == 1:
        p = 8
    elif ver == 2:
        if b[3] == 2:
            return None
        p = 4
    else:
        raise Hdf5Error(f'dataspace message version {ver}')
    return tuple(int.from_bytes(b[p + i * buf.L:p + (i + 1) * buf.L], 'little') for i in range(rank))


class _Message:
    __slots__ = ('type', 'flags', 'data')

    def __init__(self, type_: int, flags: int, data: bytes):
        self.type, self.flags, self.data = type_, flags, data


def _read_object_header(buf: _Buf, addr: int) -> List[_Message]:
    msgs: List[_Message] = []
    if buf.raw(addr, 4) == b'OHDR':
        return _read_object_header_v2(buf, addr)
    ver = buf.u(addr, 1)
    if ver != 1:
        raise Hdf5Error(f'object header at {addr}: version {ver} is neither 1 nor an OHDR block')
    n_msgs = buf.u(addr + 2, 2)
    first = buf.u(addr + 8, 4)
    blocks = [(addr + 16, first)]                            # the prefix is padded to 8 bytes
    while blocks and len(msgs) < n_msgs:
        p, size = blocks.pop(0)
        end = p + size
        while p + 8 <= end and len(msgs) < n_msgs:
            mtype, msize, mflags = buf.u(p, 2), buf.u(p + 2, 2), buf.u(p + 4, 1)
            data = buf.raw(p + 8, msize)
            p += 8 + msize
            if mtype == 0x10:                                # continuation: more messages elsewhere
                blocks.append((int.from_bytes(data[:buf.O], 'little'),
                               int.from_bytes(data[buf.O:buf.O + buf.L], 'little')))
            msgs.append(_Message(mtype, mflags, data))
    return msgs


def _read_object_header_v2(buf: _Buf, addr: int) -> List[_Message]:
    msgs: List[_Message] = []
    ver, flags = buf.u(addr + 4, 1), buf.u(addr + 5, 1)
    if ver != 2:
        raise Hdf5Error(f'OHDR version {ver}')
    p = addr + 6
    if flags & 0x20:
        p += 16                                              # access / modification / change / birth times
    if flags & 0x10:
        p += 4                                               # max compact / min dense attribute counts
    csize_bytes = 1 << (flags & 0x03)
    chunk0 = buf.u(p, csize_bytes)
    p += csize_bytes
    track_order = bool(flags & 0x04)
    blocks = [(p, chunk0)]
    while blocks:
        p, size = blocks.pop(0)
        end = p + size                                       # messages, then a gap, then the 4-byte checksum
        hdr = 4 + (2 if track_order else 0)
        while p + hdr <= end:
            mtype, msize, mflags = buf.u(p, 1), buf.u(p + 1, 2), buf.u(p + 3, 1)
            data = buf.raw(p + hdr, msize)
            p += hdr + msize
            if mtype == 0x10:
                caddr = int.from_bytes(data[:buf.O], 'little')
                clen = int.from_bytes(data[buf.O:buf.O + buf.L], 'little')
                if buf.raw(caddr, 4) != b'OCHK':
                    raise Hdf5Error(f'object header continuation at {caddr} lacks the OCHK signature')
                blocks.append((caddr + 4, clen - 8))         # minus signature and checksum
            msgs.append(_Message(mtype, mflags, data))
    return msgs


def _read_global_heap_object(buf: _Buf, caddr: int, index: int) -> bytes:
    if buf.raw(caddr, 4) != b'GCOL':
        raise Hdf5Error(f'global heap collection at {caddr} lacks the GCOL signature')
    size = buf.len_(caddr + 8)
    p, end = caddr + 8 + buf.L, caddr + size
    while p + 8 + buf.L <= end:
        idx, osize = buf.u(p, 2), buf.len_(p + 8)
        if idx == index:
            return buf.raw(p + 8 + buf.L, osize)
        if idx == 0:
            break
        p += 8 + buf.L + (osize + 7) // 8 * 8
    raise Hdf5Error(f'global heap object {index} not found in the collection at {caddr}')


def _decode_elements(buf: _Buf, dt: _Datatype, shape, raw: bytes):
    n = int(np.prod(shape)) if shape else 1
    if dt.vlen_str:
        stride = 4 + buf.O + 4
        out = []
        for i in range(n):
            e = raw[i * stride:(i + 1) * stride]
            length = int.from_bytes(e[:4], 'little')
            caddr = int.from_bytes(e[4:4 + buf.O], 'little')
            index = int.from_bytes(e[4 + buf.O:], 'little')
            s = b'' if (length == 0 or caddr == 0) else _read_global_heap_object(buf, caddr, index)[:length]
            out.append(s.decode('utf8') if dt.utf8 else s)
        if shape == ():
            return out[0]
        return np.array(out, dtype=object).reshape(shape)
    a = np.frombuffer(raw, dtype=dt.np_dtype, count=n)
    if dt.np_dtype.kind == 'S':
        pass                                                 # numpy strips the trailing NUL padding itself
    elif dt.np_dtype.byteorder == '>':
        a = a.astype(dt.np_dtype.newbyteorder('<'))
    if shape == ():
        return a[0]
    return a.reshape(shape).copy()


def _parse_attribute(buf: _Buf, b: bytes):
    ver = b[0]
    name_size = int.from_bytes(b[2:4], 'little')
    dt_size = int.from_bytes(b[4:6], 'little')
    ds_size = int.from_bytes(b[6:8], 'little')
    if ver == 1:
        p, pad = 8, (lambda n: (n + 7) // 8 * 8)
    elif ver == 2:
        if b[1] & 0x03:
            raise Hdf5Error('attribute with a shared datatype / dataspace message')
        p, pad = 8, (lambda n: n)
    elif ver == 3:
        if b[1] & 0x03:
            raise Hdf5Error('attribute with a shared datatype / dataspace message')
        p, pad = 9, (lambda n: n)
    else:
        raise Hdf5Error(f'attribute message version {ver}')
    name = b[p:p + name_size].split(b'\x00', 1)[0].decode('utf8')
    p += pad(name_size)
    dt = _parse_datatype(b[p:p + dt_size])
    p += pad(dt_size)
    shape = _parse_dataspace(buf, b[p:p + ds_size])
    p += pad(ds_size)
    if shape is None:
        return name, None
    return name, _decode_elements(buf, dt, shape, b[p:])


def _parse_link(buf: _Buf, d: bytes, p: int):
    """One link message body at d[p:] -> (name, object header address or None for soft / external links, end)."""
    if d[p] != 1:
        raise Hdf5Error(f'link message version {d[p]}')
    flags = d[p + 1]
    p += 2
    ltype = 0
    if flags & 0x08:
        ltype = d[p]
        p += 1
    if flags & 0x04:
        p += 8                                               # creation order
    if flags & 0x10:
        p += 1                                               # link name character set
    nlen_bytes = 1 << (flags & 0x03)
    nlen = int.from_bytes(d[p:p + nlen_bytes], 'little')
    p += nlen_bytes
    lname = d[p:p + nlen].decode('utf8')
    p += nlen
    if ltype == 0:
        return lname, int.from_bytes(d[p:p + buf.O], 'little'), p + buf.O
    vlen = int.from_bytes(d[p:p + 2], 'little')              # soft link target / user-defined link data
    return lname, None, p + 2 + vlen


def _dense_links(buf: _Buf, fheap: int, gname: str) -> Dict[str, int]:
    """Links of a new-style group kept in a fractal heap (more than 8 links in a group; h5py names links
    in UTF-8, which makes libhdf5 use new-style groups even with libver='earliest').  The heap's direct
    blocks are walked in heap order and the link messages packed in them parsed one after the other; the
    count is checked against the heap header, so an unexpected layout fails instead of losing links."""
    O, L = buf.O, buf.L
    if buf.raw(fheap, 4) != b'FRHP':
        raise Hdf5Error(f'{gname}: fractal heap at {fheap} lacks the FRHP signature')
    p = fheap + 5
    filt_len, flags = buf.u(p + 2, 2), buf.u(p + 4, 1)
    if filt_len:
        raise Hdf5Error(f'{gname}: filtered fractal heap')
    p += 5 + 4                                               # id length, filter length, flags, max managed object size
    p += L + O + L + O                                       # next huge id, huge b-tree, free space, free-space manager
    p += 3 * L                                               # managed space, allocated managed space, iterator offset
    n_managed = buf.len_(p)
    p += L
    n_other = buf.len_(p + L) + buf.len_(p + 3 * L)          # number of huge, number of tiny objects
    p += 4 * L
    width = buf.u(p, 2)
    start = buf.len_(p + 2)
    max_direct = buf.len_(p + 2 + L)
    heap_bits = buf.u(p + 2 + 2 * L, 2)
    p += 2 + 2 * L + 2 + 2                                   # ..., starting rows of the root indirect block
    root = buf.off(p)
    nrows = buf.u(p + O, 2)
    if n_other:
        raise Hdf5Error(f'{gname}: {n_other} huge / tiny objects in the link heap')
    off_bytes = (heap_bits + 7) // 8
    dhdr = 5 + O + off_bytes + (4 if flags & 0x02 else 0)
    max_direct_rows = (max_direct.bit_length() - 1) - (start.bit_length() - 1) + 2

    blocks: List[Tuple[int, int]] = []                       # (address, size) of the direct blocks, heap order
    if root == UNDEF:
        pass
    elif nrows == 0:
        blocks.append((root, start))
    else:
        if buf.raw(root, 4) != b'FHIB':
            raise Hdf5Error(f'{gname}: fractal heap indirect block at {root} lacks the FHIB signature')
        if nrows > max_direct_rows:
            raise Hdf5Error(f'{gname}: link heap with nested indirect blocks (more than ~500 KB of links)')
        q = root + 5 + O + off_bytes
        for r in range(nrows):
            size = start if r < 2 else start << (r - 1)
            for _ in range(width):
                a = buf.off(q)
                q += O
                if a != UNDEF:
                    blocks.append((a, size))
    links: Dict[str, int] = {}
    n = 0
    for addr, size in blocks:
        if buf.raw(addr, 4) != b'FHDB':
            raise Hdf5Error(f'{gname}: fractal heap direct block at {addr} lacks the FHDB signature')
        d = buf.raw(addr, size)
        q = dhdr
        while q + 4 <= size and d[q] == 1 and n < n_managed:
            lname, target, q = _parse_link(buf, d, q)
            if target is not None:
                links[lname] = target
            n += 1
    if n != n_managed:
        raise Hdf5Error(f'{gname}: walked {n} links, the heap header counts {n_managed}')
    return links


class _Node:
    def __init__(self, buf: _Buf, addr: int, name: str):
        self._buf, self._addr, self.name = buf, addr, name
        self._msgs = _read_object_header(buf, addr)
        self._attrs: Optional[Dict[str, object]] = None

    @property
    def attrs(self) -> Dict[str, object]:
        if self._attrs is None:
            self._attrs = {}
            for m in self._msgs:
                if m.type == 0x0C:
                    k, v = _parse_attribute(self._buf, m.data)
                    self._attrs[k] = v
                elif m.type == 0x15:                         # attribute info: dense storage in a fractal heap
                    p = 2 + (2 if m.data[1] & 1 else 0)       # version, flags, [max creation index]
                    if int.from_bytes(m.data[p:p + self._buf.O], 'little') != (1 << (8 * self._buf.O)) - 1:
                        raise Hdf5Error(f'{self.name}: attributes in dense storage (file written with '
                                        f"libver='latest'); re-save with the default libver")
        return self._attrs


class Dataset(_Node):
    def __init__(self, buf, addr, name):
        super().__init__(buf, addr, name)
        self._dt = self._layout = None
        self.shape: Optional[Tuple[int, ...]] = None
        for m in self._msgs:
            if m.type == 0x01:
                self.shape = _parse_dataspace(buf, m.data)
            elif m.type == 0x03:
                self._dt = _parse_datatype(m.data)
            elif m.type == 0x08:
                self._layout = m.data
            elif m.type == 0x0B:
                raise Hdf5Error(f'{name}: filtered (compressed) dataset; Keras weight files are not compressed')
        if self._dt is None or self._layout is None:
            raise Hdf5Error(f'{name}: object is not a dataset (no datatype / layout message)')

    @property
    def dtype(self):
        return self._dt.np_dtype if self._dt.np_dtype is not None else np.dtype(object)

    def read(self):
        buf, lay, shape = self._buf, self._layout, self.shape
        if shape is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        nbytes = n * self._dt.size
        ver = lay[0]
        if ver == 3 or (ver == 4 and lay[1] in (0, 1)):      # v4 differs from v3 only for chunked storage
            cls = lay[1]
            if cls == 0:                                     # compact: data inside the header
                size = int.from_bytes(lay[2:4], 'little')
                raw = lay[4:4 + size]
            elif cls == 1:                                   # contiguous
                addr = int.from_bytes(lay[2:2 + buf.O], 'little')
                raw = bytes(nbytes) if addr == (1 << (8 * buf.O)) - 1 else buf.raw(addr, nbytes)
            elif cls == 2:
                rank = lay[2]
                baddr = int.from_bytes(lay[3:3 + buf.O], 'little')
                dims = [int.from_bytes(lay[3 + buf.O + 4 * i:7 + buf.O + 4 * i], 'little') for i in range(rank)]
                return self._read_chunked(baddr, dims[:-1])
            else:
                raise Hdf5Error(f'{self.name}: layout class {cls}')
        elif ver in (1, 2):
            rank, cls = lay[1], lay[2]
            p = 8
            addr = None
            if cls != 0:
                addr = int.from_bytes(lay[p:p + buf.O], 'little')
                p += buf.O
            dims = [int.from_bytes(lay[p + 4 * i:p + 4 * i + 4], 'little') for i in range(rank)]
            p += 4 * rank
            if cls == 1:
                raw = bytes(nbytes) if addr == (1 << (8 * buf.O)) - 1 else buf.raw(addr, nbytes)
            elif cls == 2:
                return self._read_chunked(addr, dims)
            else:
                size = int.from_bytes(lay[p:p + 4], 'little')
                raw = lay[p + 4:p + 4 + size]
        else:
            raise Hdf5Error(f"{self.name}: data layout message version {ver} (file written with libver='latest'); "
                            f're-save with the default libver')
        if len(raw) < nbytes:
            raise Hdf5Error(f'{self.name}: {len(raw)} bytes stored, {nbytes} expected')
        return _decode_elements(buf, self._dt, shape, raw)

    def _read_chunked(self, baddr: int, chunk: List[int]):
        if self._dt.vlen_str:
            raise Hdf5Error(f'{self.name}: chunked variable-length strings')
        buf, shape = self._buf, self.shape
        out = np.zeros(shape, dtype=self._dt.np_dtype)
        rank = len(shape)
        if baddr == (1 << (8 * buf.O)) - 1:
            return out
        csize = int(np.prod(chunk)) * self._dt.size

        def walk(addr):
            if buf.raw(addr, 4) != b'TREE' or buf.u(addr + 4, 1) != 1:
                raise Hdf5Error(f'{self.name}: chunk index at {addr} is not a v1 raw-data B-tree node')
            level, used = buf.u(addr + 5, 1), buf.u(addr + 6, 2)
            p = addr + 8 + 2 * buf.O
            ksize = 8 + 8 * (rank + 1)
            for _ in range(used):
                nbytes, mask = buf.u(p, 4), buf.u(p + 4, 4)
                offs = [buf.u(p + 8 + 8 * i, 8) for i in range(rank)]
                child = buf.off(p + ksize)
                p += ksize + buf.O
                if level > 0:
                    walk(child)
                    continue
                if mask or nbytes != csize:
                    raise Hdf5Error(f'{self.name}: filtered chunk')
                c = np.frombuffer(buf.raw(child, csize), dtype=self._dt.np_dtype).reshape(chunk)
                sl = tuple(slice(o, min(o + c_, s)) for o, c_, s in zip(offs, chunk, shape))
                out[sl] = c[tuple(slice(0, s.stop - s.start) for s in sl)]
        walk(baddr)
        return out.astype(out.dtype.newbyteorder('<')) if out.dtype.byteorder == '>' else out

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.read())
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        a = self.read()
        return a if key == () or key is Ellipsis else np.asarray(a)[key]


class Group(_Node):
    def __init__(self, buf, addr, name):
        super().__init__(buf, addr, name)
        self._links: Optional[Dict[str, int]] = None

    def _load_links(self) -> Dict[str, int]:
        if self._links is not None:
            return self._links
        buf, links = self._buf, {}
        for m in self._msgs:
            if m.type == 0x11:                               # symbol table: v1 B-tree + local heap
                btree = int.from_bytes(m.data[:buf.O], 'little')
                heap = int.from_bytes(m.data[buf.O:2 * buf.O], 'little')
                if buf.raw(heap, 4) != b'HEAP':
                    raise Hdf5Error(f'{self.name}: local heap at {heap} lacks the HEAP signature')
                hdata = buf.off(heap + 8 + 2 * buf.L)

                def name_at(o):
                    end = buf.d.find(b'\x00', hdata + o) if hdata + o < len(buf.d) else -1
                    if end < 0:
                        raise Hdf5Error(f'{self.name}: link name at {hdata + o} is outside the file '
                                        f'({len(buf.d)} bytes): truncated file?')
                    return buf.d[hdata + o:end].decode('utf8')

                def walk(addr):
                    if buf.raw(addr, 4) != b'TREE' or buf.u(addr + 4, 1) != 0:
                        raise Hdf5Error(f'{self.name}: group index at {addr} is not a v1 group B-tree node')
                    level, used = buf.u(addr + 5, 1), buf.u(addr + 6, 2)
                    p = addr + 8 + 2 * buf.O + buf.L         # skip key 0
                    for _ in range(used):
                        child = buf.off(p)
                        p += buf.O + buf.L
                        if level > 0:
                            walk(child)
                            continue
                        if buf.raw(child, 4) != b'SNOD':
                            raise Hdf5Error(f'{self.name}: symbol node at {child} lacks the SNOD signature')
                        nsym = buf.u(child + 6, 2)
                        q = child + 8
                        for _ in range(nsym):
                            links[name_at(buf.off(q))] = buf.off(q + buf.O)
                            q += 2 * buf.O + 8 + 16
                if btree != (1 << (8 * buf.O)) - 1:
                    walk(btree)
            elif m.type == 0x06:                             # link message (compact new-style group)
                lname, target, _ = _parse_link(buf, m.data, 0)
                if target is not None:
                    links[lname] = target
            elif m.type == 0x02:                             # link info: links in a fractal heap ("dense" storage)
                d = m.data
                p = 2 + (8 if d[1] & 1 else 0)
                fheap = int.from_bytes(d[p:p + buf.O], 'little')
                if fheap != (1 << (8 * buf.O)) - 1:
                    links.update(_dense_links(buf, fheap, self.name))
        self._links = links
        return links

    def keys(self) -> List[str]:
        return sorted(self._load_links())

    def __contains__(self, path: str) -> bool:
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path: str) -> Union['Group', Dataset]:
        node: Union[Group, Dataset] = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(f'{node.name} is a dataset, cannot descend to {part!r}')
            links = node._load_links()
            if part not in links:
                raise KeyError(f'{part!r} not found in group {node.name!r} (has {sorted(links)[:8]}...)')
            node = _open(self._buf, links[part], node.name.rstrip('/') + '/' + part)
        return node


def _open(buf: _Buf, addr: int, name: str) -> Union[Group, Dataset]:
    msgs = _read_object_header(buf, addr)
    if any(m.type == 0x08 for m in msgs):
        return Dataset(buf, addr, name)
    return Group(buf, addr, name)


class File(Group):
    """Read-only view of an HDF5 file; `File(path)['a/b']`, `.attrs`, `.keys()`."""

    def __init__(self, path):
        with open(path, 'rb') as f:
            data = f.read()
        base = 0
        while data[base:base + 8] != SIGNATURE:              # a user block may precede the superblock
            base = 512 if base == 0 else base * 2
            if base + 8 > len(data):
                raise Hdf5Error(f'{path}: not an HDF5 file (no superblock signature)')
        if base:
            data = data[base:]                               # the base address of such files is the user block size
        buf = _Buf(data)
        ver = data[8]
        if ver in (0, 1):
            buf.O, buf.L = data[13], data[14]
            p = 24 + (4 if ver == 1 else 0)
            p += 4 * buf.O                                   # base, free-space, end-of-file, driver-info addresses
            root = buf.off(p + buf.O)                        # symbol table entry: name offset, header address
        elif ver in (2, 3):
            buf.O, buf.L = data[9], data[10]
            root = buf.off(12 + 3 * buf.O)
        else:
            raise Hdf5Error(f'{path}: superblock version {ver}')
        if buf.O not in (4, 8) or buf.L not in (4, 8):
            raise Hdf5Error(f'{path}: offsets of {buf.O} / lengths of {buf.L} bytes')
        super().__init__(buf, root, '/')

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ======================================================================================== writer
def _pad8(b: bytes) -> bytes:
    return b + bytes(-len(b) % 8)


def _dtype_message(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind == 'f' and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            sign, prec, eloc, esize, msize, bias = 31, 32, 23, 8, 23, 127
        else:
            sign, prec, eloc, esize, msize, bias = 63, 64, 52, 11, 52, 1023
        return (bytes([0x11, 0x20, sign, 0x00]) + struct.pack('<I', dt.itemsize)
                + struct.pack('<HHBBBBI', 0, prec, eloc, esize, 0, msize, bias))
    if dt.kind in 'iu' and dt.itemsize in (1, 2, 4, 8):
        return (bytes([0x10, 0x08 if dt.kind == 'i' else 0x00, 0, 0]) + struct.pack('<I', dt.itemsize)
                + struct.pack('<HH', 0, 8 * dt.itemsize))
    if dt.kind == 'S':
        return bytes([0x13, 0x01, 0, 0]) + struct.pack('<I', dt.itemsize)      # NUL-padded, ASCII
    raise Hdf5Error(f'cannot write dtype {dt}')


def _dataspace_message(shape: Tuple[int, ...]) -> bytes:
    return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b''.join(struct.pack('<Q', s) for s in shape)


def _as_writable(value) -> np.ndarray:
    if isinstance(value, str):
        value = value.encode('utf8')
    if isinstance(value, (list, tuple)) and value and isinstance(value[0], str):
        value = [v.encode('utf8') for v in value]
    a = np.asarray(value)
    if a.dtype.kind == 'U':
        a = np.char.encode(a, 'utf8')
    if a.dtype.kind == 'S' and a.dtype.itemsize == 0:
        a = a.astype('S1')
    if a.dtype.kind in 'fiu' and a.dtype.byteorder == '>':
        a = a.astype(a.dtype.newbyteorder('<'))
    if a.dtype.kind == 'b':
        a = a.astype(np.uint8)
    if a.dtype.kind not in 'fiuS':
        raise Hdf5Error(f'cannot write values of dtype {a.dtype}')
    return a if a.ndim == 0 else np.ascontiguousarray(a)     # (ascontiguousarray would turn a scalar into shape (1,))


def _message(mtype: int, data: bytes) -> bytes:
    data = _pad8(data)
    if len(data) > 0xFFF8:
        raise Hdf5Error(f'header message of {len(data)} bytes exceeds the 64 KiB object header limit '
                        f'(split long attributes the way Keras does: name0, name1, ...)')
    return struct.pack('<HHB3x', mtype, len(data), 0) + data


def _attribute_message(name: str, value) -> bytes:
    a = _as_writable(value)
    nb = name.encode('utf8') + b'\x00'
    dtm, dsm = _dtype_message(a.dtype), _dataspace_message(a.shape)
    body = (struct.pack('<BBHHH', 1, 0, len(nb), len(dtm), len(dsm))
            + _pad8(nb) + _pad8(dtm) + _pad8(dsm) + a.tobytes())
    return _message(0x0C, body)


class WGroup:
    def __init__(self, name: str = '/'):
        self.name = name
        self.attrs: Dict[str, object] = {}
        self.children: Dict[str, Union['WGroup', 'WDataset']] = {}

    def create_group(self, path: str) -> 'WGroup':
        node = self
        for part in [p for p in path.split('/') if p]:
            nxt = node.children.get(part)
            if nxt is None:
                nxt = node.children[part] = WGroup(node.name.rstrip('/') + '/' + part)
            if not isinstance(nxt, WGroup):
                raise Hdf5Error(f'{nxt.name} exists and is a dataset')
            node = nxt
        return node

    def require_group(self, path: str) -> 'WGroup':
        return self.create_group(path)

    def create_dataset(self, path: str, data) -> 'WDataset':
        parts = [p for p in path.split('/') if p]
        parent = self.create_group('/'.join(parts[:-1])) if len(parts) > 1 else self
        if parts[-1] in parent.children:
            raise Hdf5Error(f'{parent.name}/{parts[-1]} already exists')
        ds = parent.children[parts[-1]] = WDataset(parent.name.rstrip('/') + '/' + parts[-1], data)
        return ds


class WDataset:
    def __init__(self, name: str, data):
        self.name = name
        self.data = _as_writable(data)
        self.attrs: Dict[str, object] = {}


class Writer:
    """Build a tree with `root.create_group / create_dataset / attrs`, then `save(path)`."""

    def __init__(self):
        self.root = WGroup('/')

    def save(self, path):
        groups: List[WGroup] = []
        datasets: List[WDataset] = []

        def collect(g: WGroup):
            groups.append(g)
            for k in sorted(g.children, key=lambda s: s.encode('utf8')):
                c = g.children[k]
                (collect if isinstance(c, WGroup) else datasets.append)(c)
        collect(self.root)

        # one symbol node per group: pick the leaf K of the file so that the largest group fits (2K entries)
        max_entries = max([len(g.children) for g in groups] + [1])
        leaf_k = max(4, (max_entries + 1) // 2)
        if leaf_k > 0x7FFF:
            raise Hdf5Error(f'group with {max_entries} entries')
        internal_k = 16
        snod_size = 8 + 2 * leaf_k * 40
        btree_size = 24 + 2 * internal_k * 8 + (2 * internal_k + 1) * 8

        pos = [96]                                           # superblock v0 with 8-byte offsets: 56 + 40-byte root entry

        def alloc(n):
            a = pos[0]
            pos[0] += (n + 7) // 8 * 8
            return a

        # ---- sizes and addresses first (symbol entries need the children's header addresses)
        layout: Dict[int, dict] = {}
        for g in groups:
            names = sorted(g.children, key=lambda s: s.encode('utf8'))
            heap_data = bytearray(8)                         # offset 0: the empty string (key 0 of the B-tree)
            name_off = {}
            for nme in names:
                name_off[nme] = len(heap_data)
                heap_data += _pad8(nme.encode('utf8') + b'\x00')
            heap_data += bytes(16)                           # room for the (empty) free list the library may want
            msgs = [_message(0x11, b'\x00' * 16)] + [_attribute_message(k, v) for k, v in g.attrs.items()]
            layout[id(g)] = dict(names=names, name_off=name_off, heap_data=bytes(heap_data), n_msgs=len(msgs),
                                 hdr_size=sum(len(m) for m in msgs), attr_msgs=msgs[1:])
        for d in datasets:
            lay_msg_len = 8 + 24
            msgs = [_message(0x01, _dataspace_message(d.data.shape)), _message(0x03, _dtype_message(d.data.dtype)),
                    _message(0x05, bytes([2, 2, 2, 0])),     # fill value v2: late allocation, write if set, undefined
                    _message(0x08, bytes(18))] + [_attribute_message(k, v) for k, v in d.attrs.items()]
            layout[id(d)] = dict(msgs=msgs, n_msgs=len(msgs), hdr_size=sum(len(m) for m in msgs))
        for g in groups:
            L = layout[id(g)]
            L['hdr'] = alloc(16 + L['hdr_size'])
            L['btree'] = alloc(btree_size)
            L['snod'] = alloc(snod_size)
            L['heap'] = alloc(32)
            L['heap_data_addr'] = alloc(len(L['heap_data']))
        for d in datasets:
            L = layout[id(d)]
            L['hdr'] = alloc(16 + L['hdr_size'])
        for d in datasets:
            layout[id(d)]['data'] = alloc(max(d.data.nbytes, 1)) if d.data.nbytes else UNDEF
        eof = pos[0]

        out = bytearray(eof)

        def put(addr, b):
            out[addr:addr + len(b)] = b

        def header(L, msgs):
            put(L['hdr'], struct.pack('<BBHII4x', 1, 0, len(msgs), 1, L['hdr_size']) + b''.join(msgs))

        for g in groups:
            L = layout[id(g)]
            stab = _message(0x11, struct.pack('<QQ', L['btree'], L['heap']))
            header(L, [stab] + L['attr_msgs'])
            names = L['names']
            # B-tree: one leaf-level node, one child (the symbol node); keys = heap offsets of '' and of the largest name
            bt = bytearray(btree_size)
            bt[:8] = b'TREE' + struct.pack('<BBH', 0, 0, 1 if names else 0)
            bt[8:24] = struct.pack('<QQ', UNDEF, UNDEF)
            if names:
                bt[24:48] = struct.pack('<QQQ', 0, L['snod'], L['name_off'][names[-1]])
            put(L['btree'], bytes(bt))
            sn = bytearray(snod_size)
            sn[:8] = b'SNOD' + struct.pack('<BBH', 1, 0, len(names))
            for i, nme in enumerate(names):
                c = g.children[nme]
                CL = layout[id(c)]
                if isinstance(c, WGroup):
                    e = struct.pack('<QQII', L['name_off'][nme], CL['hdr'], 1, 0) + struct.pack('<QQ', CL['btree'], CL['heap'])
                else:
                    e = struct.pack('<QQII', L['name_off'][nme], CL['hdr'], 0, 0) + bytes(16)
                sn[8 + 40 * i:8 + 40 * (i + 1)] = e
            put(L['snod'], bytes(sn))
            hd = bytearray(L['heap_data'])
            free_off = len(hd) - 16                          # one free block at the tail: (next = 1 = end of list, size)
            hd[free_off:] = struct.pack('<QQ', 1, 16)
            put(L['heap'], b'HEAP' + struct.pack('<B3xQQQ', 0, len(hd), free_off, L['heap_data_addr']))
            put(L['heap_data_addr'], bytes(hd))
        for d in datasets:
            L = layout[id(d)]
            msgs = list(L['msgs'])
            msgs[3] = _message(0x08, bytes([3, 1]) + struct.pack('<QQ', L['data'], d.data.nbytes))
            header(L, msgs)
            if d.data.nbytes:
                put(L['data'], d.data.tobytes())

        R = layout[id(self.root)]
        sb = (SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', leaf_k, internal_k, 0)
              + struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
              + struct.pack('<QQII', 0, R['hdr'], 1, 0) + struct.pack('<QQ', R['btree'], R['heap']))
        assert len(sb) == 96
        put(0, sb)
        with open(path, 'wb') as f:
            f.write(bytes(out))

"""A small pure-Python reader and writer for the subset of HDF5 that Keras weight files use (SURVEY 8f row 2:
`face_detector.h5` / `yolov3_base.h5`, reference face_detection.py:337, 394, 598, 630).  There is no HDF5 library in the main
interpreter of this image, so the container format is restated from the published "HDF5 File Format Specification" (v1.1/v2.0):

reader  superblock v0/v1 (and v2/v3), object headers v1 and v2 (with continuation blocks), old-style groups (symbol table
        message -> B-tree v1 -> symbol-table nodes + local heap) and compact new-style groups (link messages), datasets with
        contiguous, compact or unfiltered chunked layout, fixed-point / IEEE-float / fixed-length-string / variable-length-string
        element types, scalar and simple dataspaces, attribute messages v1-v3.  Anything else (dense link / attribute storage,
        filters such as gzip, references, compound types) raises NotImplementedError or, for attributes, is skipped.
writer  superblock v0, v1 object headers, old-style groups, contiguous datasets, fixed-size attributes -- the plainest form
        libhdf5 of any version reads (checked with h5py 3.3.0 / HDF5 1.10.6 in tests/test_hdf5_cpu.py when that interpreter is
        present).

Pinned by tests/golden/keras_layout_*.h5, files written by libhdf5 itself (tests/golden/make_h5_fixture.py)."""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


# ======================================================================================================================= reader
class Reader(object):
    def __init__(self, buf):
        self.b = memoryview(buf if isinstance(buf, (bytes, bytearray, memoryview)) else open(buf, 'rb').read())
        off = 0
        while bytes(self.b[off:off + 8]) != SIGNATURE:        # the superblock may sit at 0, 512, 1024, ...
            off = 512 if off == 0 else off * 2
            if off + 8 > len(self.b):
                raise H5Error('not an HDF5 file (no superblock signature)')
        ver = self.b[off + 8]
        if ver in (0, 1):
            self.O, self.L = self.b[off + 13], self.b[off + 14]
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.O)
            p += 4 * self.O
            # root symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self._u(p + self.O, self.O)
        elif ver in (2, 3):
            self.O, self.L = self.b[off + 9], self.b[off + 10]
            p = off + 12
            self.base = self._u(p, self.O)
            self.root = self._u(p + 3 * self.O, self.O)
        else:
            raise H5Error('unsupported superblock version %d' % ver)
        if self.O != 8 or self.L != 8:
            raise NotImplementedError('only 8-byte offsets / lengths are supported')
        self.base += 0 if ver in (2, 3) else 0
        self._gcol = {}

    # ---- primitives
    def _u(self, p, n):
        return int.from_bytes(self.b[p:p + n], 'little')

    def _addr(self, a):
        return self.base + a

    # ---- object headers -> list of (type, flags, payload memoryview)
    def messages(self, addr):
        p = self._addr(addr)
        out = []
        if bytes(self.b[p:p + 4]) == b'OHDR':
            return self._messages_v2(p)
        if self.b[p] != 1:
            raise H5Error('object header version %d at %d' % (self.b[p], addr))
        nmsg = self._u(p + 2, 2)
        size = self._u(p + 8, 4)
        blocks = [(p + 16, size)]
        while blocks and len(out) < nmsg:
            q, n = blocks.pop(0)
            end = q + n
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self._u(q, 2), self._u(q + 2, 2), self.b[q + 4]
                data = self.b[q + 8:q + 8 + msize]
                if mtype == 0x10:                                   # continuation
                    blocks.append((self._addr(self._u(q + 8, 8)), self._u(q + 16, 8)))
                out.append((mtype, flags, data))
                q += 8 + msize
        return out

    def _messages_v2(self, p):
        out = []
        flags = self.b[p + 5]
        q = p + 6
        if flags & 0x20:
            q += 16
        if flags & 0x10:
            q += 4
        nsz = 1 << (flags & 3)
        size = self._u(q, nsz)
        q += nsz
        order = bool(flags & 0x04)
        blocks = [(q, size)]
        while blocks:
            q, n = blocks.pop(0)
            end = q + n                                             # (the chunk's checksum follows `end`)
            while q + 4 + (2 if order else 0) <= end:
                mtype, msize, mflags = self.b[q], self._u(q + 1, 2), self.b[q + 3]
                q += 4 + (2 if order else 0)
                data = self.b[q:q + msize]
                if mtype == 0x10:
                    a, ln = self._addr(self._u(q, 8)), self._u(q + 8, 8)
                    blocks.append((a + 4, ln - 8))                  # 'OCHK' signature in front, checksum behind
                out.append((mtype, mflags, data))
                q += msize
        return out

    # ---- groups
    def _heap_string(self, heap_data, off):
        e = off
        while self.b[heap_data + e] != 0:
            e += 1
        return bytes(self.b[heap_data + off:heap_data + e]).decode('utf8')

    def links(self, addr):
        """{name: object header address} of a group."""
        out = {}
        for mtype, _f, d in self.messages(addr):
            if mtype == 0x11:                                       # symbol table: B-tree v1 + local heap
                bt, hp = self._u_mv(d, 0, 8), self._u_mv(d, 8, 8)
                h = self._addr(hp)
                if bytes(self.b[h:h + 4]) != b'HEAP':
                    raise H5Error('bad local heap')
                heap_data = self._addr(self._u(h + 8 + 2 * self.L, self.O))
                self._walk_group_btree(self._addr(bt), heap_data, out)
            elif mtype == 0x06:                                     # link message (compact new-style group)
                name, target = self._link_message(d)
                if target is not None:
                    out[name] = target
            elif mtype == 0x02:                                     # link info: dense storage?
                fl = d[1]
                q = 2 + (8 if fl & 1 else 0)
                if self._u_mv(d, q, 8) != UNDEF:
                    raise NotImplementedError('dense link storage (fractal heap) is not supported')
        return out

    @staticmethod
    def _u_mv(d, p, n):
        return int.from_bytes(d[p:p + n], 'little')

    def _link_message(self, d):
        fl = d[1]
        q = 2
        ltype = 0
        if fl & 0x08:
            ltype = d[q]; q += 1
        if fl & 0x04:
            q += 8
        if fl & 0x10:
            q += 1
        nsz = 1 << (fl & 3)
        nlen = self._u_mv(d, q, nsz); q += nsz
        name = bytes(d[q:q + nlen]).decode('utf8'); q += nlen
        if ltype != 0:
            return name, None                                       # soft / external links are not followed
        return name, self._u_mv(d, q, 8)

    def _walk_group_btree(self, p, heap_data, out):
        if bytes(self.b[p:p + 4]) != b'TREE' or self.b[p + 4] != 0:
            raise H5Error('bad group B-tree node')
        level, n = self.b[p + 5], self._u(p + 6, 2)
        q = p + 8 + 2 * self.O
        for i in range(n):
            child = self._addr(self._u(q + self.L + i * (self.L + self.O), self.O))
            if level > 0:
                self._walk_group_btree(child, heap_data, out)
            else:
                if bytes(self.b[child:child + 4]) != b'SNOD':
                    raise H5Error('bad symbol table node')
                m = self._u(child + 6, 2)
                e = child + 8
                for _ in range(m):
                    out[self._heap_string(heap_data, self._u(e, self.O))] = self._u(e + self.O, self.O)
                    e += 2 * self.O + 24

    # ---- types / spaces
    def _datatype(self, d):
        cls, ver = d[0] & 0x0F, d[0] >> 4
        bits = d[1] | (d[2] << 8) | (d[3] << 16)
        size = self._u_mv(d, 4, 4)
        if cls == 0:
            return np.dtype(('>' if bits & 1 else '<') + ('i' if bits & 8 else 'u') + str(size))
        if cls == 1:
            return np.dtype(('>' if bits & 1 else '<') + 'f' + str(size))
        if cls == 3:
            return np.dtype('S%d' % size)
        if cls == 9:
            if (bits & 0x0F) != 1:
                raise NotImplementedError('variable-length sequences are not supported')
            return 'vlen_str'
        raise NotImplementedError('HDF5 datatype class %d (version %d) is not supported' % (cls, ver))

    def _dataspace(self, d):
        ver, rank = d[0], d[1]
        if ver == 1:
            q = 8
        elif ver == 2:
            if d[3] == 2:
                return None                                         # null dataspace
            q = 4
        else:
            raise H5Error('dataspace version %d' % ver)
        return tuple(self._u_mv(d, q + 8 * i, 8) for i in range(rank))

    def _vlen_strings(self, raw, count):
        out = []
        for i in range(count):
            ln = int.from_bytes(raw[16 * i:16 * i + 4], 'little')
            addr = int.from_bytes(raw[16 * i + 4:16 * i + 12], 'little')
            idx = int.from_bytes(raw[16 * i + 12:16 * i + 16], 'little')
            out.append(self._global_heap_object(addr, idx)[:ln].decode('utf8'))
        return out

    def _global_heap_object(self, addr, idx):
        if addr not in self._gcol:
            p = self._addr(addr)
            if bytes(self.b[p:p + 4]) != b'GCOL':
                raise H5Error('bad global heap collection')
            size = self._u(p + 8, 8)
            objs = {}
            q = p + 16
            while q + 16 <= p + size:
                i, n = self._u(q, 2), self._u(q + 8, 8)
                if i == 0:
                    break
                objs[i] = bytes(self.b[q + 16:q + 16 + n])
                q += 16 + ((n + 7) & ~7)
            self._gcol[addr] = objs
        return self._gcol[addr][idx]

    def _decode(self, dtype, shape, raw):
        count = int(np.prod(shape)) if shape else 1
        if isinstance(dtype, str):                                  # variable-length strings
            vals = self._vlen_strings(bytes(raw), count)
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        a = np.frombuffer(bytes(raw[:count * dtype.itemsize]), dtype=dtype, count=count)
        a = a.astype(dtype.newbyteorder('='), copy=True)
        return a.reshape(shape) if shape else a.reshape(())[()]

    # ---- attributes
    def attrs(self, addr):
        out = {}
        for mtype, _f, d in self.messages(addr):
            if mtype != 0x0C:
                continue
            try:
                ver = d[0]
                nsz, tsz, ssz = self._u_mv(d, 2, 2), self._u_mv(d, 4, 2), self._u_mv(d, 6, 2)
                if ver == 1:
                    pad = lambda n: (n + 7) & ~7
                    q = 8
                elif ver in (2, 3):
                    if d[1] & 3:
                        continue                                    # shared datatype / dataspace
                    pad = lambda n: n
                    q = 8 + (1 if ver == 3 else 0)
                else:
                    continue
                name = bytes(d[q:q + nsz]).split(b'\0')[0].decode('utf8'); q += pad(nsz)
                dt = self._datatype(d[q:q + tsz]); q += pad(tsz)
                shape = self._dataspace(d[q:q + ssz]); q += pad(ssz)
                if shape is None:
                    continue
                out[name] = self._decode(dt, shape, d[q:])
            except NotImplementedError:
                continue
        return out

    # ---- datasets
    def is_dataset(self, addr):
        return any(m[0] == 0x08 for m in self.messages(addr))

    def dataset(self, addr):
        dt = shape = layout = None
        for mtype, _f, d in self.messages(addr):
            if mtype == 0x01:
                shape = self._dataspace(d)
            elif mtype == 0x03:
                dt = self._datatype(d)
            elif mtype == 0x08:
                layout = d
            elif mtype == 0x0B:
                raise NotImplementedError('filtered (compressed) datasets are not supported')
        if dt is None or shape is None or layout is None:
            raise H5Error('incomplete dataset header')
        if isinstance(dt, str):
            raise NotImplementedError('variable-length string datasets are not supported')
        count = int(np.prod(shape)) if shape else 1
        nbytes = count * dt.itemsize
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise NotImplementedError('data layout message version %d' % ver)
        if cls == 0:
            n = self._u_mv(layout, 2, 2)
            return self._decode(dt, shape, layout[4:4 + n])
        if cls == 1:
            a = self._u_mv(layout, 2, 8)
            if a == UNDEF:                                          # never written: fill value (zeros)
                return np.zeros(shape, dt.newbyteorder('='))
            return self._decode(dt, shape, self.b[self._addr(a):self._addr(a) + nbytes])
        if cls == 2:
            nd = layout[2]
            bt = self._u_mv(layout, 3, 8)
            cdims = tuple(self._u_mv(layout, 11 + 4 * i, 4) for i in range(nd - 1))
            out = np.zeros(shape, dt.newbyteorder('='))
            if bt != UNDEF:
                self._walk_chunk_btree(self._addr(bt), nd, cdims, dt, out)
            return out
        raise NotImplementedError('data layout class %d' % cls)

    def _walk_chunk_btree(self, p, nd, cdims, dt, out):
        if bytes(self.b[p:p + 4]) != b'TREE' or self.b[p + 4] != 1:
            raise H5Error('bad chunk B-tree node')
        level, n = self.b[p + 5], self._u(p + 6, 2)
        ksz = 8 + 8 * nd
        q = p + 8 + 2 * self.O
        for i in range(n):
            k = q + i * (ksz + self.O)
            nbytes, mask = self._u(k, 4), self._u(k + 4, 4)
            offs = tuple(self._u(k + 8 + 8 * j, 8) for j in range(nd - 1))
            child = self._addr(self._u(k + ksz, self.O))
            if level > 0:
                self._walk_chunk_btree(child, nd, cdims, dt, out)
                continue
            if mask:
                raise NotImplementedError('filtered chunks are not supported')
            chunk = np.frombuffer(bytes(self.b[child:child + nbytes]), dtype=dt).astype(dt.newbyteorder('=')).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

    # ---- whole file
    def walk(self):
        """-> (datasets {'/a/b': array}, attributes {'/': {...}, '/a': {...}, ...})."""
        data, attrs = {}, {}
        seen = set()

        def rec(addr, path):
            if addr in seen:
                return
            seen.add(addr)
            at = self.attrs(addr)
            if at:
                attrs[path or '/'] = at
            if self.is_dataset(addr):
                data[path] = self.dataset(addr)
                return
            for name, child in sorted(self.links(addr).items()):
                rec(child, path + '/' + name)
        rec(self.root, '')
        return data, attrs


def read_hdf5(path_or_bytes):
    """-> ({'/group/dataset': ndarray}, {'/group': {attribute: value}})."""
    return Reader(path_or_bytes).walk()


def is_hdf5(path):
    try:
        with open(path, 'rb') as f:
            return f.read(8) == SIGNATURE
    except OSError:
        return False


# ======================================================================================================================= writer
LEAF_K, NODE_K = 32, 16                    # symbol-table nodes of up to 64 entries, B-tree nodes of up to 32 children


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f' and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
        return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    if dt.kind in 'iu':
        return struct.pack('<BBBBIHH', 0x10, 0x08 if dt.kind == 'i' else 0, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, dt.itemsize)          # null-padded ASCII
    raise NotImplementedError('cannot write dtype %r' % dt)


def _dataspace_message(shape):
    return struct.pack('<BBBB4x', 1, len(shape), 0, 0) + b''.join(struct.pack('<Q', int(s)) for s in shape)


def _message(mtype, payload, flags=0):
    payload = _pad8(payload)
    return struct.pack('<HHB3x', mtype, len(payload), flags) + payload


def _attribute_message(name, value):
    a = np.asarray(value)
    if a.dtype.kind == 'U':
        a = np.char.encode(a, 'utf8')
    if a.dtype.kind == 'O':
        raise NotImplementedError('object arrays cannot be written as attributes')
    a = np.asarray(a, dtype=a.dtype.newbyteorder('<'), order='C')          # (ascontiguousarray would turn a scalar into shape (1,))
    nm = name.encode('utf8') + b'\0'
    dtm, spm = _dtype_message(a.dtype), _dataspace_message(a.shape)
    body = struct.pack('<BxHHH', 1, len(nm), len(dtm), len(spm)) + _pad8(nm) + _pad8(dtm) + _pad8(spm) + a.tobytes()
    if len(body) > 65000:
        raise NotImplementedError('attribute %r is too large for an object header message (64 KiB)' % name)
    return _message(0x0C, body)


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BxHII4x', 1, len(messages), 1, len(body)) + body


class _Node(object):
    def __init__(self):
        self.children = {}      # name -> _Node (group) or ndarray (dataset)
        self.attrs = {}


def _build(datasets, attrs):
    root = _Node()

    def group(path):
        node = root
        for part in [p for p in path.split('/') if p]:
            nxt = node.children.setdefault(part, _Node())
            if not isinstance(nxt, _Node):
                raise H5Error('%r is both a dataset and a group' % path)
            node = nxt
        return node
    for path, arr in datasets.items():
        parts = [p for p in path.split('/') if p]
        group('/'.join(parts[:-1])).children[parts[-1]] = np.asarray(arr)
    for path, at in (attrs or {}).items():
        parts = [p for p in path.split('/') if p]
        parent = group('/'.join(parts[:-1])) if parts else None
        target = root if not parts else parent.children.get(parts[-1])
        if target is None:
            target = group(path)
        if isinstance(target, _Node):
            target.attrs.update(at)
        else:
            root.__dict__.setdefault('_dattrs', {}).setdefault(path if path.startswith('/') else '/' + path, {}).update(at)
    return root


def write_hdf5(path, datasets, attrs=None):
    """datasets: {'/group/name': array}; attrs: {'/group' or '/group/name': {attribute: array / bytes / number}}.  Writes the
    plainest HDF5: superblock v0, old-style groups, contiguous datasets."""
    root = _build(datasets, attrs)
    dattrs = getattr(root, '_dattrs', {})
    out = bytearray(96)                                             # superblock v0 with 8-byte offsets = 96 bytes

    def alloc(data):
        while len(out) % 8:
            out.append(0)
        a = len(out)
        out.extend(data)
        return a

    def emit_dataset(arr, path):
        arr = np.asarray(arr, dtype=arr.dtype.newbyteorder('<') if arr.dtype.kind != 'S' else arr.dtype, order='C')
        raw = arr.tobytes()
        daddr = alloc(raw) if raw else UNDEF
        msgs = [_message(0x01, _dataspace_message(arr.shape)), _message(0x03, _dtype_message(arr.dtype), flags=1),
                _message(0x05, struct.pack('<BBBB', 2, 2, 0, 0)),              # fill value v2: allocate late-ish, undefined fill
                _message(0x08, struct.pack('<BBQQ', 3, 1, daddr, len(raw)))]
        msgs += [_attribute_message(k, v) for k, v in dattrs.get(path, {}).items()]
        return alloc(_object_header(msgs))

    def emit_group(node, path):
        entries = []
        for name in sorted(node.children, key=lambda s: s.encode('utf8')):
            child = node.children[name]
            cpath = path + '/' + name
            addr = emit_group(child, cpath)[0] if isinstance(child, _Node) else emit_dataset(child, cpath)
            entries.append((name, addr))
        # local heap: offset 0 holds the empty string (the first B-tree key), names follow, 8-byte aligned
        heap = bytearray(8)
        offs = []
        for name, _ in entries:
            offs.append(len(heap))
            heap.extend(_pad8(name.encode('utf8') + b'\0'))
        free = len(heap)
        heap.extend(struct.pack('<QQ', 1, 16))                      # one free block closing the segment (next = 1: last)
        heap_data = alloc(bytes(heap))
        heap_hdr = alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free, heap_data))
        if len(entries) > 2 * LEAF_K * 2 * NODE_K:
            raise NotImplementedError('group %r has too many entries for a one-level B-tree' % (path or '/'))
        snods, keys = [], [0]
        for i in range(0, len(entries), 2 * LEAF_K):
            part = list(zip(offs[i:i + 2 * LEAF_K], entries[i:i + 2 * LEAF_K]))
            body = b'SNOD' + struct.pack('<BxH', 1, len(part))
            for o, (_n, a) in part:
                body += struct.pack('<QQII16x', o, a, 0, 0)
            body += b'\0' * (8 + 2 * LEAF_K * 40 - len(body))
            snods.append(alloc(body))
            keys.append(part[-1][0])
        node_b = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(snods), UNDEF, UNDEF)
        for i, a in enumerate(snods):
            node_b += struct.pack('<QQ', keys[i], a)
        node_b += struct.pack('<Q', keys[len(snods)])
        node_b += b'\0' * (24 + (2 * NODE_K + 1) * 8 + 2 * NODE_K * 8 - len(node_b))
        btree = alloc(node_b)
        msgs = [_message(0x11, struct.pack('<QQ', btree, heap_hdr))]
        msgs += [_attribute_message(k, v) for k, v in node.attrs.items()]
        return alloc(_object_header(msgs)), btree, heap_hdr

    root_addr, root_bt, root_heap = emit_group(root, '')
    while len(out) % 8:
        out.append(0)
    sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, NODE_K, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, len(out), UNDEF)
    sb += struct.pack('<QQII', 0, root_addr, 1, 0) + struct.pack('<QQ', root_bt, root_heap)
    assert len(sb) == 96
    out[:96] = sb
    with open(path, 'wb') as f:
        f.write(bytes(out))

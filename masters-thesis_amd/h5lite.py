"""Minimal pure-Python HDF5 reader / writer for Keras weight files (no h5py in the deployment image).

Keras ``model.save_weights("x.h5")`` / ``ModelCheckpoint(save_weights_only=True)`` (AttemptFour/main.py:168-190) and
``model.load_weights(path, by_name=True, skip_mismatch=True)`` (eval.py:140) use this layout
(tf.keras hdf5_format.save_weights_to_hdf5_group):

    /                attrs  layer_names   fixed-length string array   (also backend, keras_version)
    /<layer>         attrs  weight_names  fixed-length string array   ("<layer>/<weight>:0", may contain "/")
    /<layer>/<weight_name>  one dataset per weight, nested groups along the "/" of the weight name

Supported subset of the HDF5 file format (what libhdf5 1.8-1.12 / h5py write with the default "earliest" bounds):
superblock v0/v1, version-1 object headers with continuation blocks, old-style groups (symbol-table message, v1 B-tree,
local heap, symbol-table nodes), dataspace v1/v2, datatype classes fixed-point / floating-point / string, contiguous
and compact layouts (v3), attribute messages v1-v3 with fixed-length data.  Chunked / compressed datasets, new-style
(link-message / fractal-heap) groups and variable-length strings raise ``H5Unsupported`` -- nothing is guessed.

Pinned against the real library: tests/golden/keras_weights_libhdf5.h5 is written by the HDF5 1.10 C library
(tests/golden/make_h5_fixture.c) and parsed here element by element; files written here were read back with h5ls /
h5dump 1.10.6 in the build container (tests/test_h5lite.py runs that check whenever the tools are present).
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Unsupported(NotImplementedError):
    pass


def _pad8(n):
    return (n + 7) & ~7


# =============================================================================================== reader
class H5File:
    """Read-only view: ``H5File(path).root`` is a tree of ``Group`` objects (``.attrs`` dict, ``.children`` dict of
    name -> Group | numpy array)."""

    class Group:
        def __init__(self):
            self.attrs, self.children = {}, {}

        def __getitem__(self, path):
            node = self
            for part in [p for p in path.split("/") if p]:
                node = node.children[part]
            return node

        def __contains__(self, path):
            try:
                self[path]
                return True
            except (KeyError, AttributeError):
                return False

    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        b = self.b
        base = 0
        while b[base:base + 8] != SIG:
            base = 512 if base == 0 else base * 2
            if base >= len(b):
                raise ValueError(f"{path}: not an HDF5 file")
        ver = b[base + 8]
        if ver not in (0, 1):
            raise H5Unsupported(f"superblock version {ver} (written with libver='latest'?) is not supported")
        self.O, self.L = b[base + 13], b[base + 14]
        if (self.O, self.L) != (8, 8):
            raise H5Unsupported("only 8-byte offsets / lengths are supported")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = self._u(p, 8)
        root_entry = p + 32
        hdr = self._u(root_entry + 8, 8)
        self.root = self._object(hdr)

    # ---- primitives
    def _u(self, off, n):
        return int.from_bytes(self.b[off:off + n], "little")

    def _messages(self, addr):
        b = self.b
        if b[addr:addr + 4] == b"OHDR":
            raise H5Unsupported("version-2 object headers (libver='latest') are not supported")
        if b[addr] != 1:
            raise ValueError(f"bad object header version {b[addr]} at {addr}")
        nmsg, size = self._u(addr + 2, 2), self._u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self._u(p, 2), self._u(p + 2, 2), b[p + 4]
                body = p + 8
                if mtype == 0x0010:
                    blocks.append((self._u(body, 8) + self.base_addr, self._u(body + 8, 8)))
                out.append((mtype, body, msize, flags))
                p = body + msize
        return out

    def _object(self, addr):
        addr += self.base_addr
        msgs = self._messages(addr)
        types = {m[0] for m in msgs}
        attrs = {}
        for mtype, body, msize, _ in msgs:
            if mtype == 0x000C:
                name, val = self._attribute(body)
                attrs[name] = val
        if 0x0011 in types:                                   # old-style group
            g = H5File.Group()
            g.attrs = attrs
            body = next(m[1] for m in msgs if m[0] == 0x0011)
            btree, heap = self._u(body, 8), self._u(body + 8, 8)
            for name, hdr in self._group_entries(btree + self.base_addr, heap + self.base_addr):
                g.children[name] = self._object(hdr)
            return g
        if 0x0002 in types or 0x0006 in types:
            raise H5Unsupported("new-style groups (link messages) are not supported")
        if 0x0008 in types:                                   # dataset
            return self._dataset(msgs)
        g = H5File.Group()                                    # an empty object
        g.attrs = attrs
        return g

    def _heap_name(self, heap, off):
        b = self.b
        if b[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        data = self._u(heap + 24, 8) + self.base_addr
        end = b.index(b"\0", data + off)
        return b[data + off:end].decode("utf8")

    def _group_entries(self, node, heap):
        b = self.b
        if b[node:node + 4] == b"SNOD":
            n = self._u(node + 6, 2)
            for i in range(n):
                e = node + 8 + i * 40
                yield self._heap_name(heap, self._u(e, 8)), self._u(e + 8, 8)
            return
        if b[node:node + 4] != b"TREE":
            raise ValueError("bad B-tree node signature")
        if b[node + 4] != 0:
            raise ValueError("not a group B-tree")
        used = self._u(node + 6, 2)
        p = node + 24
        for i in range(used):
            child = self._u(p + 8, 8) + self.base_addr        # key (8), child (8)
            yield from self._group_entries(child, heap)
            p += 16

    def _dtype(self, body):
        b = self.b
        cls, size = b[body] & 0x0F, self._u(body + 4, 4)
        bits0 = b[body + 1]
        order = ">" if (bits0 & 1) else "<"
        if cls == 1:
            return np.dtype(f"{order}f{size}")
        if cls == 0:
            return np.dtype(f"{order}{'i' if (bits0 & 8) else 'u'}{size}")
        if cls == 3:
            return np.dtype(f"S{size}")
        if cls == 9:
            raise H5Unsupported("variable-length datatypes are not supported")
        raise H5Unsupported(f"datatype class {cls} is not supported")

    def _dspace(self, body):
        b = self.b
        ver, rank = b[body], b[body + 1]
        if ver == 1:
            p = body + 8
        elif ver == 2:
            if b[body + 3] == 2:                              # null dataspace
                return None
            p = body + 4
        else:
            raise H5Unsupported(f"dataspace version {ver}")
        return tuple(self._u(p + 8 * i, 8) for i in range(rank))

    def _attribute(self, body):
        b = self.b
        ver = b[body]
        nsz, tsz, ssz = self._u(body + 2, 2), self._u(body + 4, 2), self._u(body + 6, 2)
        if ver == 1:
            p = body + 8
            name = b[p:p + nsz].split(b"\0")[0].decode("utf8"); p += _pad8(nsz)
            tb = p; p += _pad8(tsz)
            sb = p; p += _pad8(ssz)
        elif ver in (2, 3):
            p = body + 8 + (1 if ver == 3 else 0)
            name = b[p:p + nsz].split(b"\0")[0].decode("utf8"); p += nsz
            tb = p; p += tsz
            sb = p; p += ssz
        else:
            raise H5Unsupported(f"attribute message version {ver}")
        try:
            dt, shape = self._dtype(tb), self._dspace(sb)
        except H5Unsupported:
            return name, None                                 # e.g. a variable-length string: not needed here
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        if n == 0:                                            # h5py stores `weight_names = []` (a Dropout / InputLayer
            return name, []                                   # group) as an empty float64 attribute of shape (0,)
        arr = np.frombuffer(b, dtype=dt, count=n, offset=p).reshape(shape)
        if dt.kind == "S":
            vals = [bytes(x).split(b"\0")[0] for x in arr.reshape(-1)]
            return name, (vals[0] if not shape else vals)
        return name, (arr.copy() if shape else arr.reshape(-1)[0])

    def _dataset(self, msgs):
        b = self.b
        dt = shape = None
        data = None
        for mtype, body, msize, _ in msgs:
            if mtype == 0x0003:
                dt = self._dtype(body)
            elif mtype == 0x0001:
                shape = self._dspace(body)
            elif mtype == 0x000B:
                raise H5Unsupported("filtered (compressed) datasets are not supported")
        for mtype, body, msize, _ in msgs:
            if mtype != 0x0008:
                continue
            ver, cls = b[body], b[body + 1]
            if ver != 3:
                raise H5Unsupported(f"data layout version {ver}")
            n = int(np.prod(shape)) if shape else 1
            if cls == 1:
                addr = self._u(body + 2, 8)
                if addr == UNDEF:
                    data = np.zeros(shape, dt)
                else:
                    data = np.frombuffer(b, dtype=dt, count=n, offset=addr + self.base_addr).reshape(shape).copy()
            elif cls == 0:
                data = np.frombuffer(b, dtype=dt, count=n, offset=body + 4).reshape(shape).copy()
            else:
                raise H5Unsupported("chunked datasets are not supported (Keras writes its weights contiguous)")
        if data is None:
            raise ValueError("dataset without a layout message")
        return data


def read_keras_weights(path):
    """-> (layer_names, {layer: [(weight_name, array), ...]}) of a Keras weight file."""
    f = H5File(path)
    root = f.root

    def names(attrs, key):
        if key in attrs and attrs[key] is not None:
            v = attrs[key]
            if isinstance(v, np.ndarray):                     # a non-string attribute names nothing (empty float64 array
                return []                                     # of a layer without weights; also size-0 arrays)
            return [x.decode("utf8") for x in (v if isinstance(v, list) else [v]) if isinstance(x, (bytes, bytearray))]
        out, i = [], 0                                        # keras splits attributes > 64 KB: layer_names0, layer_names1, ...
        while f"{key}{i}" in attrs:
            out += [x.decode("utf8") for x in attrs[f"{key}{i}"]]
            i += 1
        return out
    layers = names(root.attrs, "layer_names")
    if not layers:
        raise ValueError(f"{path}: no layer_names attribute -- not a Keras weight file")
    out = {}
    for ln in layers:
        g = root[ln]
        out[ln] = [(wn, np.asarray(g[wn])) for wn in names(g.attrs, "weight_names")]
    return layers, out


# =============================================================================================== writer
class _Image:
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, n):
        off = len(self.buf)
        self.buf += b"\0" * _pad8(n)
        return off

    def put(self, off, data):
        self.buf[off:off + len(data)] = data


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        size = dt.itemsize
        if size == 4:
            props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            bits = bytes([0x20, 0x1F, 0x00])
        elif size == 8:
            props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            bits = bytes([0x20, 0x3F, 0x00])
        else:
            raise H5Unsupported("float16 datasets are not supported")
        return bytes([0x11]) + bits + struct.pack("<I", size) + props
    if dt.kind in "iu":
        bits = bytes([0x08 if dt.kind == "i" else 0x00, 0, 0])
        return bytes([0x10]) + bits + struct.pack("<I", dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", dt.itemsize)        # null-padded, ASCII
    raise H5Unsupported(f"dtype {dt}")


def _dspace_msg(shape):
    return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b"".join(struct.pack("<Q", d) for d in shape)


def _msg(mtype, body, flags=0):
    body = body + b"\0" * (_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attr_msg(name, value):
    if isinstance(value, (bytes, str)):
        v = value.encode("utf8") if isinstance(value, str) else value
        arr, shape = np.array(v, dtype=f"S{max(1, len(v))}"), ()
    else:
        vals = [x.encode("utf8") if isinstance(x, str) else x for x in value]
        w = max([1] + [len(x) for x in vals])
        arr = np.array(vals, dtype=f"S{w}")
        shape = (len(vals),)
    nm = name.encode("utf8") + b"\0"
    dtm, dsm = _dtype_msg(arr.dtype), _dspace_msg(shape)
    pad = lambda x: x + b"\0" * (_pad8(len(x)) - len(x))
    body = struct.pack("<BxHHH", 1, len(nm), len(dtm), len(dsm)) + pad(nm) + pad(dtm) + pad(dsm) + arr.tobytes()
    if len(body) > 65000:
        raise ValueError("attribute too large for one header message")
    return _msg(0x000C, body)


def _object_header(img, msgs):
    body = b"".join(msgs)
    off = img.alloc(16 + len(body))
    img.put(off, struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body)
    return off


def _write_dataset(img, arr):
    arr = np.ascontiguousarray(arr)
    data = img.alloc(max(arr.nbytes, 1))
    img.put(data, arr.tobytes())
    msgs = [_msg(0x0001, _dspace_msg(arr.shape)), _msg(0x0003, _dtype_msg(arr.dtype), flags=1),
            _msg(0x0005, bytes([2, 2, 2, 1]) + struct.pack("<I", 0)),
            _msg(0x0008, bytes([3, 1]) + struct.pack("<QQ", data, arr.nbytes))]
    return _object_header(img, msgs)


def _write_group(img, children, attrs, leaf_k, internal_k=16):
    """children: dict name -> object-header address (already written).  One symbol-table node, one B-tree node."""
    names = sorted(children, key=lambda s: s.encode("utf8"))
    heap_data, offs = bytearray(b"\0" * 8), {}
    for n in names:
        offs[n] = len(heap_data)
        e = n.encode("utf8") + b"\0"
        heap_data += e + b"\0" * (_pad8(len(e)) - len(e))
    seg = img.alloc(len(heap_data))
    img.put(seg, bytes(heap_data))
    heap = img.alloc(32)
    img.put(heap, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, seg))
    snod = img.alloc(8 + 2 * leaf_k * 40)
    ent = b"".join(struct.pack("<QQII16x", offs[n], children[n], 0, 0) for n in names)
    img.put(snod, b"SNOD" + struct.pack("<BxH", 1, len(names)) + ent)
    tree = img.alloc(24 + 2 * internal_k * 16 + 8)
    keys = struct.pack("<QQQ", 0, snod, offs[names[-1]]) if names else b""
    img.put(tree, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF) + keys)
    msgs = [_msg(0x0011, struct.pack("<QQ", tree, heap))]
    for k, v in attrs.items():
        msgs.append(_attr_msg(k, v))
    return _object_header(img, msgs), tree, heap


def write_h5(path, tree, attrs=None):
    """tree: nested dict name -> (dict | ndarray | (dict, attrs)); attrs: root attributes (str / bytes / list of str)."""
    def count(node):
        d = node[0] if isinstance(node, tuple) else node
        if not isinstance(d, dict):
            return 0
        return max([len(d)] + [count(v) for v in d.values()])
    leaf_k = max(4, (count(tree) + 1) // 2 + 1)
    if leaf_k > 32000:
        raise ValueError("too many entries in one group")
    img = _Image()
    img.alloc(96)                                             # superblock v0 (56 bytes + 40-byte root entry)

    def emit(node):
        a = {}
        if isinstance(node, tuple):
            node, a = node
        if not isinstance(node, dict):
            return _write_dataset(img, node), None, None
        kids = {k: emit(v)[0] for k, v in node.items()}
        return _write_group(img, kids, a, leaf_k)
    root_hdr, root_tree, root_heap = emit((tree, attrs or {}))
    eof = len(img.buf)
    sb = SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", leaf_k, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_tree, root_heap)
    img.put(0, sb)
    with open(path, "wb") as f:
        f.write(bytes(img.buf))


def write_keras_weights(path, layers, keras_version="2.4.0", backend="tensorflow"):
    """layers: ordered dict layer_name -> [(weight_name, array), ...]; weight names as Keras spells them
    ("<layer>/<weight>:0")."""
    tree = {}
    gattrs = {}                                               # id(dict) -> attributes of that group

    def descend(node, parts):
        for p in parts:
            node = node.setdefault(p, {})
        return node
    for ln, ws in layers.items():
        grp = descend(tree, ln.split("/"))                    # a "/" in a layer name nests groups (f["a/b"] resolves it)
        gattrs[id(grp)] = {"weight_names": [wn for wn, _ in ws]}
        for wn, arr in ws:
            parts = wn.split("/")
            descend(grp, parts[:-1])[parts[-1]] = np.asarray(arr, dtype=np.float32)

    def attach(node):
        if not isinstance(node, dict):
            return node
        kids = {k: attach(v) for k, v in node.items()}
        return (kids, gattrs[id(node)]) if id(node) in gattrs else kids
    tree = attach(tree)
    attrs = {"backend": backend, "keras_version": keras_version}
    names = list(layers)
    blob = sum(len(n) for n in names)
    if blob < 40000:
        attrs["layer_names"] = names
    else:                                                     # keras' own chunking rule for large attributes
        per = max(1, len(names) * 40000 // blob)
        for i in range(0, len(names), per):
            attrs[f"layer_names{i // per}"] = names[i:i + per]
    write_h5(path, tree, attrs)

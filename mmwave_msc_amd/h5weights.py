"""Read the weights of a Keras `.h5` model file without h5py.

The reference saves its keypoint model with `keypoint_model.save(".../MARS.h5")` (train.py:252) and loads it with
`keras.models.load_model(const.P_MODEL_PATH)` (offline_main.py:33).  Neither keras nor h5py exists on the target
image, so this module reads the one thing the inference path needs -- the weight tensors -- straight from the HDF5
container: `MarsCNN.from_h5(path)` / `load_keras_h5(path)` return what `MarsCNN.from_keras_weights` takes.

Scope of the reader (what Keras 2.x + h5py with default settings write; anything else raises `H5FormatError`
naming the unsupported feature instead of guessing):
  * superblock version 0 or 1, 8-byte offsets and lengths;
  * version-1 object headers, continuation blocks included;
  * old-style groups (symbol-table message -> v1 B-tree of SNOD leaves + local heap), and compact new-style groups
    (link messages in the object header);
  * datasets with contiguous or compact layout, IEEE little-endian float32 / float64, any rank.
Not read: attributes (the layer order is derived from the Keras auto-names, see `_layer_order`), chunked or
compressed datasets, version-2 object headers ("OHDR"), external links, dense link storage.

HDF5 file format specification version 1.1/2.0, sections II.A (superblock), III.A (B-trees), III.D (local heap),
IV.A (object headers and messages 0x0001 dataspace, 0x0003 datatype, 0x0006 link, 0x0008 layout, 0x0010
continuation, 0x0011 symbol table).
"""
from __future__ import annotations

import re
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(ValueError):
    pass


class _File:
    def __init__(self, data: bytes):
        self.d = data
        # the superblock may sit at 0, 512, 1024, ... (a user block in front of it)
        off = 0
        while True:
            if data[off:off + 8] == _SIG:
                break
            off = 512 if off == 0 else off * 2
            if off + 8 > len(data):
                raise H5FormatError("not an HDF5 file (no superblock signature)")
        self.sb = off
        ver = data[off + 8]
        if ver not in (0, 1):
            raise H5FormatError(f"superblock version {ver} (written with libver='latest'?) is not supported; "
                                "re-save with h5py defaults or convert to .npz (INTEGRATION.md)")
        so, sl = data[off + 13], data[off + 14]
        if so != 8 or sl != 8:
            raise H5FormatError(f"offset/length sizes {so}/{sl} are not supported (8/8 expected)")
        p = off + 24 + (4 if ver == 1 else 0)
        self.base, _fs, _eof, _drv = struct.unpack_from("<4Q", data, p)
        p += 32
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        _name, self.root, cache = struct.unpack_from("<QQI", data, p)

    # -- primitives --------------------------------------------------------------------------------------
    def u(self, fmt: str, off: int):
        return struct.unpack_from("<" + fmt, self.d, off)

    def addr(self, a: int) -> int:
        if a == _UNDEF:
            raise H5FormatError("undefined address")
        return self.base + a

    # -- object headers ----------------------------------------------------------------------------------
    def messages(self, hdr_addr: int):
        """[(type, flags, bytes)] of a version-1 object header, continuation blocks followed."""
        p = self.addr(hdr_addr)
        if self.d[p:p + 4] == b"OHDR":
            raise H5FormatError("version-2 object headers (libver='latest') are not supported")
        ver, _r, nmsg, _ref, hsize = self.u("BBHII", p)
        if ver != 1:
            raise H5FormatError(f"object header version {ver} is not supported")
        blocks = [(p + 16, hsize)]  # the first message starts on the next 8-byte boundary after the 12-byte prefix
        out = []
        while blocks and len(out) < nmsg:
            q, size = blocks.pop(0)
            end = q + size
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = self.u("HHB", q)
                body = self.d[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x0010:  # continuation: offset, length
                    coff, clen = struct.unpack("<QQ", body[:16])
                    blocks.append((self.addr(coff), clen))
                out.append((mtype, mflags, body))
        return out

    # -- groups ------------------------------------------------------------------------------------------
    def _heap_name(self, heap_data: int, off: int) -> str:
        e = self.d.index(b"\0", heap_data + off)
        return self.d[heap_data + off:e].decode("utf-8")

    def _btree_entries(self, node_addr: int, heap_data: int, out: dict):
        p = self.addr(node_addr)
        sig = self.d[p:p + 4]
        if sig == b"SNOD":
            _ver, _r, nsym = self.u("BBH", p + 4)
            q = p + 8
            for _ in range(nsym):
                name_off, hdr = self.u("QQ", q)
                out[self._heap_name(heap_data, name_off)] = hdr
                q += 40
            return
        if sig != b"TREE":
            raise H5FormatError("bad group B-tree node signature")
        ntype, _level, used = self.u("BBH", p + 4)
        if ntype != 0:
            raise H5FormatError("B-tree node of a non-group type inside a group")
        q = p + 8 + 16  # skip left / right sibling addresses
        for i in range(used):
            child = self.u("Q", q + 8 + i * 16)[0]  # key_i (8) child_i (8) ... key_used (8)
            self._btree_entries(child, heap_data, out)

    def children(self, hdr_addr: int) -> dict:
        """name -> object header address of the members of a group (empty for a non-group)."""
        out: dict = {}
        for mtype, _f, body in self.messages(hdr_addr):
            if mtype == 0x0011:  # symbol table: B-tree address, local heap address
                bt, heap = struct.unpack("<QQ", body[:16])
                hp = self.addr(heap)
                if self.d[hp:hp + 4] != b"HEAP":
                    raise H5FormatError("bad local heap signature")
                _dsize, _free, dseg = self.u("QQQ", hp + 8)
                self._btree_entries(bt, self.addr(dseg), out)
            elif mtype == 0x0006:  # link message (compact new-style group)
                ver, flags = body[0], body[1]
                q = 2
                ltype = 0
                if flags & 0x08:
                    ltype = body[q]; q += 1
                if flags & 0x04:
                    q += 8
                if flags & 0x10:
                    q += 1
                lsz = 1 << (flags & 3)
                nlen = int.from_bytes(body[q:q + lsz], "little"); q += lsz
                name = body[q:q + nlen].decode("utf-8"); q += nlen
                if ver != 1 or ltype != 0:
                    raise H5FormatError(f"link '{name}': only hard links are supported")
                out[name] = struct.unpack("<Q", body[q:q + 8])[0]
            elif mtype == 0x0002:
                # link info: fine as long as the links themselves are stored compactly (messages above)
                fl = body[1]
                q = 2 + (8 if fl & 1 else 0)
                fheap = struct.unpack("<Q", body[q:q + 8])[0]
                if fheap != _UNDEF:
                    raise H5FormatError("dense link storage (fractal heap) is not supported")
        return out

    # -- datasets ----------------------------------------------------------------------------------------
    def dataset(self, hdr_addr: int):
        """numpy array of a dataset object, or None when the object is not a dataset."""
        shape = dtype = None
        layout = None
        for mtype, _f, body in self.messages(hdr_addr):
            if mtype == 0x0001:  # dataspace
                ver, rank, flags = body[0], body[1], body[2]
                q = 8 if ver == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", body, q) if rank else ()
            elif mtype == 0x0003:  # datatype
                cls, ver = body[0] & 0x0F, body[0] >> 4
                bits0 = body[1]
                size = struct.unpack_from("<I", body, 4)[0]
                if cls != 1:
                    return None  # not a floating-point dataset (e.g. the optimizer's int64 step counter): not a weight
                if (bits0 & 1) != 0 or size not in (4, 8):
                    raise H5FormatError(f"floating-point datatype of size {size}, byte-order bit {bits0 & 1}: only little-endian "
                                        "float32/float64 are supported")
                dtype = np.dtype("<f4" if size == 4 else "<f8")
            elif mtype == 0x0008:  # data layout
                ver = body[0]
                if ver != 3:
                    raise H5FormatError(f"data layout message version {ver} is not supported")
                lcls = body[1]
                if lcls == 1:
                    a, n = struct.unpack_from("<QQ", body, 2)
                    layout = ("contiguous", a, n)
                elif lcls == 0:
                    n = struct.unpack_from("<H", body, 2)[0]
                    layout = ("compact", bytes(body[4:4 + n]), n)
                else:
                    raise H5FormatError("chunked (compressed?) datasets are not supported; Keras writes contiguous ones")
        if shape is None or dtype is None or layout is None:
            return None
        count = int(np.prod(shape)) if len(shape) else 1
        if layout[0] == "compact":
            raw = layout[1]
        elif layout[1] == _UNDEF:  # never written: fill value zero
            raw = bytes(count * dtype.itemsize)
        else:
            p = self.addr(layout[1])
            raw = self.d[p:p + count * dtype.itemsize]
        if len(raw) < count * dtype.itemsize:
            raise H5FormatError("dataset extends past the end of the file")
        return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape).copy()


def read_h5_datasets(path: str) -> dict:
    """{"/group/.../name": ndarray} for every float dataset of an HDF5 file (see the module docstring for scope)."""
    with open(path, "rb") as f:
        F = _File(f.read())
    out: dict = {}
    seen = set()

    def walk(hdr: int, prefix: str):
        if hdr in seen:  # hard links may form cycles
            return
        seen.add(hdr)
        kids = F.children(hdr)
        if kids:
            for name, h in kids.items():
                walk(h, prefix + "/" + name)
            return
        arr = F.dataset(hdr)
        if arr is not None:
            out[prefix] = arr

    walk(F.root, "")
    return out


def _suffix(name: str) -> int:
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def _layer_order(layers, kind: str):
    """Keras auto-names `<kind>`, `<kind>_1`, ... in creation order (the counter is per class and keeps running when
    a script builds the model repeatedly, as train.py does: `conv3d_18`, `conv3d_19` are still first and second)."""
    own = [l for l in layers if re.fullmatch(kind + r"(_\d+)?", l)]
    return sorted(own, key=_suffix)


def load_keras_h5(path: str) -> dict:
    """The MARS model's tensors from a Keras `.h5` (`model.save` or `model.save_weights`), keyed as
    `mars.MarsCNN.from_keras_weights` expects: conv1/conv2 kernels and biases, two BatchNormalization layers
    (gamma, beta, moving mean, moving variance), dense1/dense2 kernels and biases -- all in Keras layouts."""
    ds = read_h5_datasets(path)
    # /model_weights/<layer>/<layer>/<var>:0 (model.save) or /<layer>/<layer>/<var>:0 (save_weights);
    # optimizer state lives under /optimizer_weights and is ignored
    per_layer: dict = {}
    for full, arr in ds.items():
        parts = [p for p in full.split("/") if p]
        if parts and parts[0] == "optimizer_weights":
            continue
        if parts and parts[0] == "model_weights":
            parts = parts[1:]
        if len(parts) < 2:
            continue
        layer, var = parts[0], parts[-1].split(":")[0]
        per_layer.setdefault(layer, {})[var] = arr
    layers = list(per_layer)
    convs = _layer_order(layers, "conv3d") or _layer_order(layers, "conv2d")
    bns = _layer_order(layers, "batch_normalization")
    denses = _layer_order(layers, "dense")
    if len(convs) != 2 or len(bns) != 2 or len(denses) != 2:
        raise H5FormatError(f"{path}: expected 2 conv, 2 batch-normalization and 2 dense layers (train.py:33-106), found "
                            f"{convs} / {bns} / {denses}")

    def g(layer, var):
        try:
            return np.asarray(per_layer[layer][var], dtype=np.float32)
        except KeyError:
            raise H5FormatError(f"{path}: layer '{layer}' has no variable '{var}'") from None

    return {
        "conv1_w": g(convs[0], "kernel"), "conv1_b": g(convs[0], "bias"),
        "conv2_w": g(convs[1], "kernel"), "conv2_b": g(convs[1], "bias"),
        "bn1_gamma": g(bns[0], "gamma"), "bn1_beta": g(bns[0], "beta"),
        "bn1_mean": g(bns[0], "moving_mean"), "bn1_var": g(bns[0], "moving_variance"),
        "dense1_w": g(denses[0], "kernel"), "dense1_b": g(denses[0], "bias"),
        "bn2_gamma": g(bns[1], "gamma"), "bn2_beta": g(bns[1], "beta"),
        "bn2_mean": g(bns[1], "moving_mean"), "bn2_var": g(bns[1], "moving_variance"),
        "dense2_w": g(denses[1], "kernel"), "dense2_b": g(denses[1], "bias"),
    }

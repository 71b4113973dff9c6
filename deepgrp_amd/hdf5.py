"""Minimal pure-Python HDF5 reader (and writer) for the files Keras' ``model.save("*.hdf5")``
produces (deepgrp/__main__.py:351 writes them, :264-269 loads them).

h5py / libhdf5 are not available next to PyTorch-ROCm on the target machines, and the model
files are tiny (~50 k floats), so the subset HDF5 1.8's default ("earliest") format needs is
implemented here:

* superblock version 0 (and 2/3 for robustness), object headers version 1 (and 2),
* old-style groups: symbol-table message -> v1 B-tree -> SNOD nodes + local heap,
  and compact new-style groups (link messages),
* datasets with contiguous or compact layout (chunked, unfiltered is supported for a single
  chunk B-tree walk), little-endian floats / integers,
* attributes with fixed-length or variable-length (global heap) strings, scalar or 1-D.

The writer emits the same old-style structures (one B-tree leaf + one SNOD per group, contiguous
datasets, fixed-length string attributes), enough for Keras-layout model files; files it writes
are read back by h5py (checked in the build container by tests/golden/make_h5_fixtures.py).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Optional, Tuple, Union

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(ValueError):
    pass


# ============================================================================================
# reader
# ============================================================================================
class _Datatype:
    def __init__(self, cls: int, size: int, np_dtype=None, vlen_str: bool = False, base=None, pad: int = 0):
        self.cls, self.size, self.np_dtype, self.vlen_str, self.base, self.pad = cls, size, np_dtype, vlen_str, base, pad


class Node:
    """A group or a dataset."""

    def __init__(self, f: "File", addr: int, name: str):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = f._read_object_header(addr)
        self.attrs: Dict[str, object] = {}
        for t, body in self._msgs:
            if t == 0x000C:
                k, v = f._parse_attribute(body)
                self.attrs[k] = v
        self._links: Optional[Dict[str, int]] = None

    # ---- group interface ------------------------------------------------------------------
    @property
    def is_group(self) -> bool:
        return any(t in (0x0011, 0x0002, 0x0006) for t, _ in self._msgs) and not self.is_dataset

    @property
    def is_dataset(self) -> bool:
        return any(t == 0x0008 for t, _ in self._msgs)

    def _load_links(self) -> Dict[str, int]:
        if self._links is None:
            links: Dict[str, int] = {}
            for t, body in self._msgs:
                if t == 0x0011:                                   # symbol table message
                    btree, heap = struct.unpack_from("<QQ", body, 0)
                    links.update(self._f._walk_group_btree(btree, heap))
                elif t == 0x0006:                                 # link message (compact new-style group)
                    name, addr = self._f._parse_link(body)
                    if addr is not None:
                        links[name] = addr
                elif t == 0x0002:
                    fheap = struct.unpack_from("<Q", body, 2 + (8 if body[1] & 1 else 0))[0]
                    if fheap != UNDEF:
                        raise HDF5Error("dense link storage (fractal heap) is not supported; re-save the model "
                                        "with the default HDF5 format")
            self._links = links
        return self._links

    def keys(self) -> List[str]:
        return sorted(self._load_links())

    def __contains__(self, key: str) -> bool:
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, path: str) -> "Node":
        node = self
        for part in [p for p in path.split("/") if p]:
            links = node._load_links()
            if part not in links:
                raise KeyError(f"{part!r} not found in {node.name!r}")
            node = Node(self._f, links[part], (node.name.rstrip("/") + "/" + part))
        return node

    def walk(self) -> Iterator["Node"]:
        for k in self.keys():
            child = self[k]
            yield child
            if not child.is_dataset:
                yield from child.walk()

    # ---- dataset interface ----------------------------------------------------------------
    def read(self) -> np.ndarray:
        f = self._f
        dtype = shape = layout = None
        for t, body in self._msgs:
            if t == 0x0003:
                dtype = f._parse_datatype(body, 0)[0]
            elif t == 0x0001:
                shape = f._parse_dataspace(body)
            elif t == 0x0008:
                layout = body
            elif t == 0x000B:
                nfilters = body[1]
                if nfilters:
                    raise HDF5Error(f"dataset {self.name}: filtered (compressed) datasets are not supported")
        if dtype is None or shape is None or layout is None:
            raise HDF5Error(f"{self.name} is not a dataset")
        if dtype.np_dtype is None:
            raise HDF5Error(f"dataset {self.name}: unsupported datatype class {dtype.cls}")
        count = int(np.prod(shape)) if shape else 1
        nbytes = count * dtype.size
        version = layout[0]
        if version == 3:
            cls = layout[1]
            if cls == 0:
                size = struct.unpack_from("<H", layout, 2)[0]
                raw = layout[4:4 + size]
            elif cls == 1:
                addr, size = struct.unpack_from("<QQ", layout, 2)
                raw = b"" if addr == UNDEF else f._read(addr, min(size, nbytes))
                if addr == UNDEF:
                    raw = bytes(nbytes)
            elif cls == 2:
                raw = f._read_chunked(layout, shape, dtype.size)
            else:
                raise HDF5Error(f"dataset {self.name}: layout class {cls} not supported")
        elif version in (1, 2):
            rank = layout[1]
            cls = layout[2]
            off = 8
            if cls == 1:
                addr = struct.unpack_from("<Q", layout, off)[0]
                raw = f._read(addr, nbytes)
            elif cls == 0:
                off += 4 * rank
                size = struct.unpack_from("<I", layout, off)[0]
                raw = layout[off + 4:off + 4 + size]
            else:
                raise HDF5Error(f"dataset {self.name}: old chunked layout not supported")
        else:
            raise HDF5Error(f"dataset {self.name}: data layout version {version} not supported")
        return np.frombuffer(raw[:nbytes], dtype=dtype.np_dtype, count=count).reshape(shape).copy()


class File(Node):
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        self.path = path
        base = self._buf.find(_SIG)
        if base != 0:
            raise HDF5Error(f"{path}: not an HDF5 file (signature not at offset 0)")
        version = self._buf[8]
        if version in (0, 1):
            so, sl = self._buf[13], self._buf[14]
            if (so, sl) != (8, 8):
                raise HDF5Error("only 8-byte offsets/lengths are supported")
            off = 24 + (4 if version == 1 else 0)
            off += 32                                           # base, free-space, eof, driver addresses
            root_addr = struct.unpack_from("<Q", self._buf, off + 8)[0]   # symbol table entry: name off, header addr
        elif version in (2, 3):
            if (self._buf[9], self._buf[10]) != (8, 8):
                raise HDF5Error("only 8-byte offsets/lengths are supported")
            root_addr = struct.unpack_from("<Q", self._buf, 12 + 24)[0]
        else:
            raise HDF5Error(f"superblock version {version} not supported")
        super().__init__(self, root_addr, "/")

    def close(self):
        self._buf = b""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- low level ------------------------------------------------------------------------
    def _read(self, addr: int, n: int) -> bytes:
        if addr == UNDEF or addr + n > len(self._buf):
            raise HDF5Error(f"read of {n} bytes at {addr:#x} is outside the file")
        return self._buf[addr:addr + n]

    def _read_object_header(self, addr: int) -> List[Tuple[int, bytes]]:
        b = self._buf
        msgs: List[Tuple[int, bytes]] = []
        if b[addr:addr + 4] == b"OHDR":                          # version 2
            flags = b[addr + 5]
            off = addr + 6
            if flags & 0x20:
                off += 16
            if flags & 0x10:
                off += 4
            szlen = 1 << (flags & 3)
            chunk0 = int.from_bytes(b[off:off + szlen], "little")
            off += szlen
            blocks = [(off, chunk0)]
            track = bool(flags & 0x04)
            while blocks:
                start, size = blocks.pop(0)
                p, endp = start, start + size
                while p + 4 <= endp:
                    t = b[p]
                    sz = struct.unpack_from("<H", b, p + 1)[0]
                    p += 4 + (2 if track else 0)
                    body = b[p:p + sz]
                    p += sz
                    if t == 0x10:
                        caddr, clen = struct.unpack_from("<QQ", body, 0)
                        blocks.append((caddr + 4, clen - 8))      # skip OCHK signature, drop checksum
                    elif t != 0:
                        msgs.append((t, body))
            return msgs
        version = b[addr]
        if version != 1:
            raise HDF5Error(f"object header version {version} at {addr:#x} not supported")
        nmsgs = struct.unpack_from("<H", b, addr + 2)[0]
        hsize = struct.unpack_from("<I", b, addr + 8)[0]
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsgs + 64:
            start, size = blocks.pop(0)
            p, endp = start, start + size
            while p + 8 <= endp:
                t, sz = struct.unpack_from("<HH", b, p)
                body = b[p + 8:p + 8 + sz]
                p += 8 + sz
                if t == 0x0010:
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, clen))
                elif t != 0:
                    msgs.append((t, body))
        return msgs

    def _walk_group_btree(self, btree: int, heap: int) -> Dict[str, int]:
        b = self._buf
        if b[heap:heap + 4] != b"HEAP":
            raise HDF5Error("bad local heap signature")
        heap_data = struct.unpack_from("<Q", b, heap + 24)[0]
        out: Dict[str, int] = {}

        def name_at(off: int) -> str:
            s = heap_data + off
            e = b.index(b"\x00", s)
            return b[s:e].decode("utf-8")

        def visit(addr: int):
            if b[addr:addr + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, addr + 6)[0]
                for i in range(n):
                    e = addr + 8 + 40 * i
                    noff, ohdr = struct.unpack_from("<QQ", b, e)
                    out[name_at(noff)] = ohdr
                return
            if b[addr:addr + 4] != b"TREE":
                raise HDF5Error(f"bad B-tree node at {addr:#x}")
            used = struct.unpack_from("<H", b, addr + 6)[0]
            p = addr + 24
            for i in range(used):
                child = struct.unpack_from("<Q", b, p + 8)[0]     # key i (8), child i (8)
                visit(child)
                p += 16

        visit(btree)
        return out

    def _parse_link(self, body: bytes) -> Tuple[str, Optional[int]]:
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        ln = 1 << (flags & 3)
        nlen = int.from_bytes(body[p:p + ln], "little"); p += ln
        name = body[p:p + nlen].decode("utf-8"); p += nlen
        if ltype != 0:
            return name, None
        return name, struct.unpack_from("<Q", body, p)[0]

    def _parse_dataspace(self, body: bytes) -> Tuple[int, ...]:
        version, rank, flags = body[0], body[1], body[2]
        if version == 1:
            p = 8
        elif version == 2:
            if body[3] == 2:
                return (0,)
            p = 4
        else:
            raise HDF5Error(f"dataspace version {version} not supported")
        return tuple(struct.unpack_from("<Q", body, p + 8 * i)[0] for i in range(rank))

    def _parse_datatype(self, body: bytes, p: int) -> Tuple[_Datatype, int]:
        cv = body[p]
        cls, bits0 = cv & 0x0F, body[p + 1]
        size = struct.unpack_from("<I", body, p + 4)[0]
        q = p + 8
        if cls == 0:                                             # fixed point
            if bits0 & 1:
                raise HDF5Error("big-endian integers are not supported")
            signed = bool(bits0 & 0x08)
            return _Datatype(cls, size, np.dtype(f"<{'i' if signed else 'u'}{size}")), q + 4
        if cls == 1:                                             # float
            if bits0 & 1:
                raise HDF5Error("big-endian floats are not supported")
            return _Datatype(cls, size, np.dtype(f"<f{size}")), q + 12
        if cls == 3:                                             # fixed-length string
            return _Datatype(cls, size, np.dtype(f"S{size}"), pad=bits0 & 0x0F), q
        if cls == 9:                                             # variable length
            base, q2 = self._parse_datatype(body, q)
            is_str = (bits0 & 0x0F) == 1
            return _Datatype(cls, size, None, vlen_str=is_str, base=base), q2
        return _Datatype(cls, size, None), q

    def _global_heap_object(self, coll: int, index: int) -> bytes:
        b = self._buf
        if b[coll:coll + 4] != b"GCOL":
            raise HDF5Error("bad global heap signature")
        size = struct.unpack_from("<Q", b, coll + 8)[0]
        p, endp = coll + 16, coll + size
        while p + 16 <= endp:
            idx, _ref, _res, osz = struct.unpack_from("<HHIQ", b, p)
            if idx == 0:
                break
            if idx == index:
                return b[p + 16:p + 16 + osz]
            p += 16 + (osz + 7) // 8 * 8
        raise HDF5Error("global heap object not found")

    def _parse_attribute(self, body: bytes) -> Tuple[str, object]:
        version = body[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
        p = 8 + (1 if version == 3 else 0)
        pad = (lambda n: (n + 7) // 8 * 8) if version == 1 else (lambda n: n)
        name = body[p:p + nsz].split(b"\x00")[0].decode("utf-8")
        p += pad(nsz)
        dtype, _ = self._parse_datatype(body, p)
        p += pad(tsz)
        shape = self._parse_dataspace(body[p:p + ssz]) if ssz >= 2 else ()
        if ssz >= 4 and body[p] == 2 and body[p + 3] == 2:
            shape = (0,)
        p += pad(ssz)
        count = int(np.prod(shape)) if shape else 1
        data = body[p:]
        if dtype.cls == 3:
            vals = [data[i * dtype.size:(i + 1) * dtype.size].split(b"\x00")[0] for i in range(count)]
        elif dtype.cls == 9 and dtype.vlen_str:
            vals = []
            for i in range(count):
                ln, coll, idx = struct.unpack_from("<IQI", data, 16 * i)
                vals.append(self._global_heap_object(coll, idx)[:ln] if ln else b"")
        elif dtype.np_dtype is not None:
            arr = np.frombuffer(data[:count * dtype.size], dtype=dtype.np_dtype, count=count).reshape(shape).copy()
            return name, (arr if shape else arr.reshape(()).item())
        else:
            return name, None
        if not shape:
            return name, vals[0]
        return name, vals

    def _read_chunked(self, layout: bytes, shape, esize: int) -> bytes:
        rank = layout[2] - 1
        btree = struct.unpack_from("<Q", layout, 3)[0]
        cdims = struct.unpack_from("<" + "I" * rank, layout, 11)
        out = np.zeros(shape, dtype=np.uint8).reshape(-1)
        full = np.zeros(tuple(shape) + (esize,), np.uint8)
        b = self._buf

        def visit(addr):
            if b[addr:addr + 4] != b"TREE":
                raise HDF5Error("bad chunk B-tree")
            level = b[addr + 5]
            used = struct.unpack_from("<H", b, addr + 6)[0]
            p = addr + 24
            keysz = 8 + 8 * (rank + 1)
            for i in range(used):
                csize, mask = struct.unpack_from("<II", b, p)
                offs = struct.unpack_from("<" + "Q" * (rank + 1), b, p + 8)
                child = struct.unpack_from("<Q", b, p + keysz)[0]
                if level > 0:
                    visit(child)
                else:
                    if mask:
                        raise HDF5Error("filtered chunks are not supported")
                    chunk = np.frombuffer(b[child:child + csize], np.uint8).reshape(tuple(cdims) + (esize,))
                    sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[:rank], cdims, shape))
                    src = tuple(slice(0, s.stop - s.start) for s in sl)
                    full[sl] = chunk[src]
                p += keysz + 8
        del out
        if btree != UNDEF:
            visit(btree)
        return full.tobytes()


# ============================================================================================
# writer (old-style groups, contiguous datasets, fixed-length string attributes)
# ============================================================================================
AttrValue = Union[bytes, str, List[bytes], np.ndarray, int, float]


class _WGroup:
    def __init__(self):
        self.children: Dict[str, Union["_WGroup", np.ndarray]] = {}
        self.attrs: Dict[str, AttrValue] = {}


class Writer:
    """Build a tree with ``create_group`` / ``create_dataset`` / ``set_attr`` and ``save(path)``."""

    def __init__(self):
        self.root = _WGroup()

    def _group(self, path: str, create: bool = True) -> _WGroup:
        g = self.root
        for part in [p for p in path.split("/") if p]:
            if part not in g.children:
                if not create:
                    raise KeyError(path)
                g.children[part] = _WGroup()
            g = g.children[part]
            if not isinstance(g, _WGroup):
                raise HDF5Error(f"{path}: {part} is a dataset")
        return g

    def create_group(self, path: str) -> None:
        self._group(path)

    def create_dataset(self, path: str, data: np.ndarray) -> None:
        parts = [p for p in path.split("/") if p]
        g = self._group("/".join(parts[:-1]))
        g.children[parts[-1]] = np.ascontiguousarray(data)

    def set_attr(self, path: str, name: str, value: AttrValue) -> None:
        self._group(path).attrs[name] = value

    # ---- serialisation --------------------------------------------------------------------
    @staticmethod
    def _dt_msg(dtype: np.dtype) -> bytes:
        if dtype.kind == "f":
            size = dtype.itemsize
            exp_loc, exp_sz, man_sz, bias = {4: (23, 8, 23, 127), 8: (52, 11, 52, 1023), 2: (10, 5, 10, 15)}[size]
            return struct.pack("<BBBBI", 0x11, 0x20, size * 8 - 1, 0, size) + struct.pack("<HHBBBBI", 0, size * 8, exp_loc,
                                                                                       exp_sz, 0, man_sz, bias)
        if dtype.kind in "iu":
            return struct.pack("<BBBBI", 0x10, 0x08 if dtype.kind == "i" else 0, 0, 0, dtype.itemsize) + struct.pack(
                "<HH", 0, dtype.itemsize * 8)
        if dtype.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, dtype.itemsize)
        raise HDF5Error(f"dtype {dtype} not supported by the writer")

    @staticmethod
    def _ds_msg(shape) -> bytes:
        rank = len(shape)
        return struct.pack("<BBBB4x", 1, rank, 0, 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)

    def _attr_msg(self, name: str, value: AttrValue) -> bytes:
        if isinstance(value, str):
            value = value.encode("utf-8")
        if isinstance(value, bytes):
            arr = np.array(value if value else b"\x00", dtype=f"S{max(len(value), 1)}")
            shape: Tuple[int, ...] = ()
        elif isinstance(value, (list, tuple)):
            items = [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in value]
            width = max([len(v) for v in items] + [1])
            arr = np.array(items, dtype=f"S{width}")
            shape = (len(items),)
        else:
            arr = np.asarray(value)
            shape = arr.shape
        pad8 = lambda bts: bts + bytes(-len(bts) % 8)
        nm = name.encode("utf-8") + b"\x00"
        dt, ds = self._dt_msg(arr.dtype), self._ds_msg(shape)
        return struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds)) + pad8(nm) + pad8(dt) + pad8(ds) + arr.tobytes()

    def save(self, path: str) -> None:
        buf = bytearray(b"\x00" * 96)                            # superblock v0 + root entry, patched at the end

        def alloc(data: bytes) -> int:
            while len(buf) % 8:
                buf.append(0)
            addr = len(buf)
            buf.extend(data)
            return addr

        def header(msgs: List[Tuple[int, bytes]]) -> int:
            body = b""
            for t, m in msgs:
                m = m + bytes(-len(m) % 8)
                body += struct.pack("<HHB3x", t, len(m), 0) + m
            return alloc(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

        def write_dataset(arr: np.ndarray, attrs) -> int:
            data_addr = alloc(arr.tobytes()) if arr.size else UNDEF
            msgs = [(0x0001, self._ds_msg(arr.shape)), (0x0003, self._dt_msg(arr.dtype)),
                    (0x0008, struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes))]
            msgs += [(0x000C, self._attr_msg(k, v)) for k, v in attrs.items()]
            return header(msgs)

        def write_group(g: _WGroup) -> Tuple[int, int, int]:
            entries = []
            for name in sorted(g.children):
                child = g.children[name]
                if isinstance(child, _WGroup):
                    ohdr, bt, hp = write_group(child)
                    entries.append((name, ohdr, 1, bt, hp))
                else:
                    entries.append((name, write_dataset(child, {}), 0, 0, 0))
            # local heap: "" at offset 0, then the names
            heap = bytearray(b"\x00" * 8)
            offs = []
            for name, *_ in entries:
                offs.append(len(heap))
                nb = name.encode("utf-8") + b"\x00"
                heap.extend(nb + bytes(-len(nb) % 8))
            heap_data = alloc(bytes(heap))
            heap_addr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data))   # 1 = H5HL_FREE_NULL: empty free list
            if len(entries) > 2 * 16:
                raise HDF5Error("more than 32 children per group are not supported by the writer")
            snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(entries))
            for (name, ohdr, ctype, bt, hp), off in zip(entries, offs):
                snod += struct.pack("<QQII", off, ohdr, ctype, 0) + (struct.pack("<QQ", bt, hp) if ctype == 1 else bytes(16))
            snod += bytes(40 * (32 - len(entries)))
            snod_addr = alloc(snod)
            last = offs[-1] if offs else 0
            btree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if entries else 0, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod_addr, last)
            btree += bytes(8 * (2 * 32 + 1) - 24 + 8 * 0)          # room for the remaining keys/children (K = 16)
            bt_addr = alloc(btree)
            msgs = [(0x0011, struct.pack("<QQ", bt_addr, heap_addr))]
            msgs += [(0x000C, self._attr_msg(k, v)) for k, v in g.attrs.items()]
            return header(msgs), bt_addr, heap_addr

        root_hdr, root_bt, root_heap = write_group(self.root)
        eof = len(buf) + (-len(buf) % 8)
        sb = _SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 16, 16, 0)   # leaf K = 16: 32 entries per SNOD
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", root_bt, root_heap)
        buf[:len(sb)] = sb
        while len(buf) < eof:
            buf.append(0)
        with open(path, "wb") as fh:
            fh.write(bytes(buf))

"""TEST INFRASTRUCTURE: a minimal HDF5 WRITER, independent of coflux/hdf5_subset.py's reader (it shares no code with it), for the
subset of the HDF5 File Format Specification (version 3.0) that NetCDF-4 forcing files use.  Two encodings of the same content:

  style="classic"   superblock version 0, version-1 object headers (one message pushed into a continuation block), a
                    symbol-table root group (B-tree v1 type 0, one SNOD, local heap), dataspace / fill-value / filter-pipeline /
                    attribute messages in their version-1 forms
  style="new"       superblock version 2, version-2 object headers ("OHDR", with timestamps and tracked creation order, one
                    message pushed into an "OCHK" block), a compact new-style root group (Link Info + Link messages), the
                    version-2/3 message forms

Datasets are contiguous or chunked (data layout version 3, chunk B-tree v1 with a small fan-out so that a year of time levels
makes a tree of several levels), with shuffle / deflate / fletcher32 in the library's order; `layout4` writes the version-4 layout
with the single-chunk or implicit index.  Checksums of the version-2 structures are written as zero (the reader does not verify
them; libhdf5 would refuse such a file — this writer exists to test the reader against the specification, not to make files)."""
import struct
import zlib

import numpy as np

O = L = 8
UNDEF = b"\xff" * 8


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _dtype_msg(dt):
    dt = np.dtype(dt)
    order = 1 if dt.byteorder == ">" else 0
    if dt.kind == "f":
        bits = order | 0x20 | (((8 * dt.itemsize - 1) & 0xFF) << 8)      # mantissa normalisation: implied msb; sign bit position
        exp, mant, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[dt.itemsize]
        props = struct.pack("<HHBBBBI", 0, 8 * dt.itemsize, mant, exp, 0, mant, bias)
        return struct.pack("<BBBBI", 0x11, bits & 0xFF, (bits >> 8) & 0xFF, 0, dt.itemsize) + props
    if dt.kind in "iu":
        bits = order | (0x08 if dt.kind == "i" else 0)
        return struct.pack("<BBBBI", 0x10, bits, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0, 0, 0, dt.itemsize)
    raise ValueError(dt)


def _dataspace_msg(shape, version):
    if version == 1:
        return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", n) for n in shape)
    return struct.pack("<BBBB", 2, len(shape), 0, 1 if shape else 0) + b"".join(struct.pack("<Q", n) for n in shape)


def _attr_msg(name, value, version):
    if isinstance(value, str):
        raw = value.encode() + b"\0"
        dt, shape, data = _dtype_msg(f"S{len(raw)}"), (), raw
    else:
        a = np.atleast_1d(np.asarray(value))
        dt, shape, data = _dtype_msg(a.dtype), (() if np.ndim(value) == 0 else a.shape), a.tobytes()
    nm = name.encode() + b"\0"
    sp = _dataspace_msg(shape, 1 if version == 1 else 2)
    if version == 1:
        return struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + data
    return struct.pack("<BBHHHB", 3, 0, len(nm), len(dt), len(sp), 0) + nm + dt + sp + data


def _filters_msg(filters, version):
    if version == 1:
        out = struct.pack("<BB6x", 1, len(filters))
        for fid, cd in filters:
            out += struct.pack("<HHHH", fid, 0, 1, len(cd)) + b"".join(struct.pack("<I", v) for v in cd) + (b"\0" * 4 if len(cd) % 2 else b"")
        return out
    out = struct.pack("<BB", 2, len(filters))
    for fid, cd in filters:
        out += struct.pack("<HHH", fid, 1, len(cd)) + b"".join(struct.pack("<I", v) for v in cd)
    return out


class _File:
    def __init__(self):
        self.buf = bytearray()

    def alloc(self, data, align=8):
        self.buf += b"\0" * (-len(self.buf) % align)
        at = len(self.buf)
        self.buf += data
        return at

    def patch(self, at, data):
        self.buf[at:at + len(data)] = data


def _encode_chunk(block, filters):
    raw = np.ascontiguousarray(block).tobytes()
    for fid, cd in filters:
        if fid == 2:
            size = cd[0]
            n = len(raw) // size
            raw = np.frombuffer(raw, np.uint8)[:n * size].reshape(n, size).T.tobytes() + raw[n * size:]
        elif fid == 1:
            raw = zlib.compress(raw, cd[0])
        elif fid == 3:
            raw = raw + struct.pack("<I", zlib.adler32(raw) & 0xFFFFFFFF)     # (a stand-in value: the reader drops it unverified)
    return raw


def _chunk_btree(f, entries, rank, fanout):
    """entries: [(offset tuple, address, stored size)] sorted; returns the root node's address"""
    level, nodes = 0, [(e[0], e) for e in entries]       # (first key offset, payload)

    def key(off, size=0):
        return struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)

    while True:
        groups = [nodes[i:i + fanout] for i in range(0, len(nodes), fanout)] or [[]]
        new = []
        for g in groups:
            body = b""
            for first, payload in g:
                if level == 0:
                    off, addr, size = payload
                    body += key(off, size) + struct.pack("<Q", addr)
                else:
                    body += key(first) + struct.pack("<Q", payload)
            last = g[-1][0] if g else (0,) * rank
            body += key(tuple(o + 1 for o in last))         # the closing key: beyond the last chunk
            node = b"TREE" + struct.pack("<BBH", 1, level, len(g)) + UNDEF + UNDEF + body
            new.append((g[0][0] if g else (0,) * rank, f.alloc(node)))
        if len(new) == 1:
            return new[0][1]
        nodes, level = new, level + 1


def _dataset_messages(f, spec, style):
    data = np.asarray(spec["data"])
    v1 = style == "classic"
    msgs = [(0x01, _dataspace_msg(data.shape, 1 if v1 else 2)), (0x03, _dtype_msg(data.dtype))]
    fill = spec.get("fill")
    if v1:
        msgs.append((0x05, struct.pack("<BBBB", 2, 2, 2, 1 if fill is not None else 0) +
                     (struct.pack("<I", data.dtype.itemsize) + np.asarray(fill, data.dtype).tobytes() if fill is not None else b"")))
    else:
        msgs.append((0x05, struct.pack("<BB", 3, 0x20 | 0x0A if fill is not None else 0x0A) +
                     (struct.pack("<I", data.dtype.itemsize) + np.asarray(fill, data.dtype).tobytes() if fill is not None else b"")))
    chunks = spec.get("chunks")
    filters = []
    if spec.get("shuffle"):
        filters.append((2, [data.dtype.itemsize]))
    if spec.get("deflate") is not None:
        filters.append((1, [spec["deflate"]]))
    if spec.get("fletcher32"):
        filters.append((3, []))
    if chunks is None:
        addr = f.alloc(np.ascontiguousarray(data).tobytes())
        msgs.append((0x08, struct.pack("<BB", 3, 1) + struct.pack("<QQ", addr, data.nbytes)))
    else:
        rank = data.ndim
        grid = [-(-s // c) for s, c in zip(data.shape, chunks)]
        entries = []
        skip = set(map(tuple, spec.get("missing_chunks", ())))
        for idx in np.ndindex(*grid):
            off = tuple(i * c for i, c in zip(idx, chunks))
            if off in skip:
                continue
            block = np.zeros(chunks, data.dtype) if fill is None else np.full(chunks, fill, data.dtype)
            src = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, chunks, data.shape))
            block[tuple(slice(0, s.stop - s.start) for s in src)] = data[src]
            raw = _encode_chunk(block, filters)
            entries.append((off, f.alloc(raw), len(raw)))
        layout4 = spec.get("layout4")
        if layout4 is None:
            root = _chunk_btree(f, entries, rank, spec.get("fanout", 4))
            msgs.append((0x08, struct.pack("<BBB", 3, 2, rank + 1) + struct.pack("<Q", root) +
                         b"".join(struct.pack("<I", c) for c in tuple(chunks) + (data.dtype.itemsize,))))
        else:
            dims = b"".join(struct.pack("<I", c) for c in tuple(chunks) + (data.dtype.itemsize,))
            if layout4 == "single":
                assert len(entries) == 1
                flags = 0x02 if filters else 0
                info = struct.pack("<QI", entries[0][2], 0) if filters else b""
                msgs.append((0x08, struct.pack("<BBBBB", 4, 2, flags, rank + 1, 4) + dims + struct.pack("<B", 1) + info + struct.pack("<Q", entries[0][1])))
            else:   # implicit: unfiltered chunks back to back in index order
                assert not filters
                blob = b"".join(bytes(f.buf[a:a + n]) for _, a, n in entries)
                at = f.alloc(blob)
                msgs.append((0x08, struct.pack("<BBBBB", 4, 2, 0, rank + 1, 4) + dims + struct.pack("<B", 2) + struct.pack("<Q", at)))
        if filters:
            msgs.append((0x0B, _filters_msg(filters, 1 if v1 else 2)))
    for k, v in (spec.get("attrs") or {}).items():
        msgs.append((0x0C, _attr_msg(k, v, 1 if v1 else 3)))
    return msgs


def _object_header_v1(f, msgs):
    """version-1 header; the last message goes into a continuation block"""
    def enc(t, d):
        d = _pad8(d)
        return struct.pack("<HHB3x", t, len(d), 0) + d
    head, tail = msgs[:-1], msgs[-1:]
    cont = b"".join(enc(t, d) for t, d in tail) if len(msgs) > 2 else b""
    if not cont:
        head = msgs
    body = b"".join(enc(t, d) for t, d in head)
    if cont:
        cat = f.alloc(cont)
        body += enc(0x10, struct.pack("<QQ", cat, len(cont)))
    n = len(head) + (1 + len(tail) if cont else 0)
    return f.alloc(struct.pack("<BBHII4x", 1, 0, n, 1, len(body)) + body)


def _object_header_v2(f, msgs, tracked=True, times=True):
    flags = 0x02 | (0x04 if tracked else 0) | (0x20 if times else 0) | 0x10     # 4-byte chunk size; attribute phase-change values stored
    def enc(t, d, n):
        return struct.pack("<BHB", t, len(d), 0) + (struct.pack("<H", n) if tracked else b"") + d
    head, tail = (msgs[:-1], msgs[-1:]) if len(msgs) > 2 else (msgs, [])
    body = b"".join(enc(t, d, n) for n, (t, d) in enumerate(head))
    if tail:
        ochk = b"OCHK" + b"".join(enc(t, d, len(head) + n) for n, (t, d) in enumerate(tail)) + struct.pack("<I", 0)
        cat = f.alloc(ochk)
        body += enc(0x10, struct.pack("<QQ", cat, len(ochk)), 0)
    body += b"\0" * 3           # a gap too small for a message header
    pre = b"OHDR" + struct.pack("<BB", 2, flags) + (struct.pack("<IIII", 1, 2, 3, 4) if times else b"") + struct.pack("<HH", 8, 6)
    return f.alloc(pre + struct.pack("<I", len(body)) + body + struct.pack("<I", 0))


def write_hdf5(path, datasets, style="classic"):
    """datasets: {name: dict(data, chunks=None, shuffle=False, deflate=None, fletcher32=False, fill=None, attrs=None, fanout=4,
    missing_chunks=(), layout4=None)}"""
    f = _File()
    if style == "classic":
        f.alloc(b"\0" * (8 + 16 + 4 * O + 2 * O + 24))          # superblock v0, patched at the end
        headers = {name: _object_header_v1(f, _dataset_messages(f, spec, style)) for name, spec in datasets.items()}
        names = sorted(headers)
        heap_data, offsets = bytearray(b"\0" * 8), {}
        for n in names:
            offsets[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\0")
        heap_at = f.alloc(bytes(heap_data))
        heap = f.alloc(b"HEAP" + struct.pack("<B3x", 0) + struct.pack("<QQQ", len(heap_data), 0xFFFFFFFFFFFFFFFF, heap_at))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for n in names:
            snod += struct.pack("<QQII16x", offsets[n], headers[n], 0, 0)
        snod_at = f.alloc(snod)
        tree = b"TREE" + struct.pack("<BBH", 0, 0, 1) + UNDEF + UNDEF + struct.pack("<Q", 0) + struct.pack("<Q", snod_at) + struct.pack("<Q", offsets[names[-1]] if names else 0)
        tree_at = f.alloc(tree)
        root = _object_header_v1(f, [(0x11, struct.pack("<QQ", tree_at, heap))])
        sb = (b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, O, L, 0) + struct.pack("<HHI", 4, 16, 0) +
              struct.pack("<Q", 0) + UNDEF + struct.pack("<Q", len(f.buf)) + UNDEF +
              struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", tree_at, heap))
        f.patch(0, sb)
    else:
        f.alloc(b"\0" * (12 + 4 * O + 4))
        headers = {name: _object_header_v2(f, _dataset_messages(f, spec, style), tracked=(n % 2 == 0), times=(n % 2 == 1))
                   for n, (name, spec) in enumerate(datasets.items())}
        msgs = [(0x02, struct.pack("<BB", 0, 0) + UNDEF + UNDEF)]                                   # link info: compact storage
        for n, (name, addr) in enumerate(headers.items()):
            nm = name.encode()
            if n % 2:   # the short form …
                msgs.append((0x06, struct.pack("<BB", 1, 0) + struct.pack("<B", len(nm)) + nm + struct.pack("<Q", addr)))
            else:       # … and one with every optional field: link type, creation order, character set, 2-byte name length
                msgs.append((0x06, struct.pack("<BB", 1, 0x08 | 0x04 | 0x10 | 0x01) + struct.pack("<BQB", 0, n, 0) + struct.pack("<H", len(nm)) + nm + struct.pack("<Q", addr)))
        root = _object_header_v2(f, msgs, tracked=False, times=False)
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBB", 2, O, L, 0) + struct.pack("<Q", 0) + UNDEF + struct.pack("<Q", len(f.buf)) + struct.pack("<Q", root) + struct.pack("<I", 0)
        f.patch(0, sb)
    with open(path, "wb") as fh:
        fh.write(bytes(f.buf))

"""A reader for the subset of HDF5 that NetCDF-4 files of gridded forcing use — enough to take time planes out of the yearly
JRA55-do files the reference stages (`/root/reference/src/OMIPConfigurations/jra55_data_staging.jl:8,134`: `tas_….nc` etc.,
NetCDF-4 = HDF5, variables `[time, lat, lon]` chunked one time level per chunk, shuffle + deflate) without an HDF5 library
(this image has none: no h5py, no netCDF4).  Pure Python: struct + zlib + NumPy.  Host I/O beside the path, not part of it.

Written from the published HDF5 File Format Specification (version 3.0).  What is read:

  superblock        versions 0, 1 (classic) and 2, 3
  object headers    version 1 (with continuation blocks) and version 2 ("OHDR" / "OCHK", creation-order and time fields)
  groups            the ROOT group only: symbol-table groups (B-tree v1 type 0 + "SNOD" nodes + local heap) and compact
                    new-style groups (Link messages in the object header).  Dense link storage (fractal heap; groups with
                    more than eight links under the library's defaults) is refused with a message that says so
  dataspace         versions 1 and 2, simple extents
  datatype          fixed point (class 0) and IEEE floating point (class 1), 1–8 bytes, either byte order
  layout            version 3: compact, contiguous, chunked (B-tree v1 type 1, any depth); version 4: contiguous, compact,
                    chunked with the "single chunk" and "implicit" indexes (the other 1.10 chunk indexes are refused)
  filter pipeline   versions 1 and 2: deflate (1), shuffle (2), fletcher32 (3: the checksum is dropped, not verified)
  attributes        compact attribute messages (version 1–3) of scalars / short arrays of the numeric types above, and fixed
                    strings: enough for `scale_factor`, `add_offset`, `_FillValue`, `units`

Everything else raises HDF5Unsupported naming what was met.  Nothing here has been run against a file written by the HDF5
library in this image (there is none): tests/test_hdf5_subset.py reads files produced by an independent minimal WRITER of the same
specification (tests/hdf5_write.py, both the classic and the new-style encodings), which pins the reader to the specification
as two implementations agree on it — not to libhdf5.  `python -m coflux.hdf5_subset FILE [VARIABLE]` prints what a file holds."""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Unsupported(NotImplementedError):
    pass


class HDF5FormatError(ValueError):
    pass


class Dataset:
    """One dataset: shape, dtype, chunking, filters; read() / read_leading(k) return NumPy arrays."""

    def __init__(self, f, name, shape, dtype, layout, filters, fill, attrs):
        self.file, self.name, self.shape, self.dtype = f, name, tuple(shape), dtype
        self.layout, self.filters, self.fill, self.attrs = layout, filters, fill, attrs
        self._chunks = None

    @property
    def chunks(self):
        return self.layout.get("chunk")

    def _chunk_index(self):
        """{chunk offset tuple: (address, stored size, filter mask)}"""
        if self._chunks is None:
            lay = self.layout
            if lay["index"] == "btree1":
                self._chunks = {}
                if lay["address"] != UNDEF:
                    self.file._walk_chunk_btree(lay["address"], len(self.shape), self._chunks)
            elif lay["index"] == "single":
                self._chunks = {(0,) * len(self.shape): (lay["address"], lay.get("stored", self._chunk_bytes()), lay.get("mask", 0))}
            elif lay["index"] == "implicit":
                self._chunks, n = {}, 0
                grid = [-(-s // c) for s, c in zip(self.shape, lay["chunk"])]
                for idx in np.ndindex(*grid):
                    self._chunks[tuple(i * c for i, c in zip(idx, lay["chunk"]))] = (lay["address"] + n * self._chunk_bytes(), self._chunk_bytes(), 0)
                    n += 1
        return self._chunks

    def _chunk_bytes(self):
        return int(np.prod(self.layout["chunk"])) * self.dtype.itemsize

    def _read_chunk(self, offset):
        rec = self._chunk_index().get(tuple(offset))
        cshape = self.layout["chunk"]
        if rec is None:   # never written: the fill value
            return np.full(cshape, self.fill if self.fill is not None else 0, dtype=self.dtype)
        addr, stored, mask = rec
        raw = self.file._read(addr, stored)
        for n in range(len(self.filters) - 1, -1, -1):       # the pipeline is undone back to front
            fid, cd = self.filters[n]
            if mask & (1 << n):
                continue                                       # this filter was skipped for this chunk
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                size = cd[0] if cd else self.dtype.itemsize
                n_el = len(raw) // size
                raw = np.frombuffer(raw, np.uint8)[:n_el * size].reshape(size, n_el).T.tobytes() + raw[n_el * size:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise HDF5Unsupported(f"{self.name}: filter id {fid} (only deflate, shuffle, fletcher32 are read)")
        want = self._chunk_bytes()
        if len(raw) < want:
            raise HDF5FormatError(f"{self.name}: chunk at {offset} decodes to {len(raw)} bytes, expected {want}")
        return np.frombuffer(raw[:want], dtype=self.dtype).reshape(cshape)

    def read(self, selection=None):
        """The whole array, or a hyperslab: selection = tuple of slices / ints (step 1), one per dimension."""
        rank = len(self.shape)
        sel = list(selection) if selection is not None else []
        sel += [slice(None)] * (rank - len(sel))
        lo, hi, squeeze = [], [], []
        for d, s in enumerate(sel):
            if isinstance(s, (int, np.integer)):
                k = int(s) + (self.shape[d] if s < 0 else 0)
                if not 0 <= k < self.shape[d]:
                    raise IndexError(f"{self.name}: index {s} outside dimension {d} of length {self.shape[d]}")
                lo.append(k); hi.append(k + 1); squeeze.append(d)
            else:
                a, b, st = s.indices(self.shape[d])
                if st != 1:
                    raise HDF5Unsupported("strided selections")
                lo.append(a); hi.append(max(a, b))
        out = np.empty([h - l for l, h in zip(lo, hi)], dtype=self.dtype.newbyteorder("="))
        lay = self.layout
        if lay["class"] in ("compact", "contiguous"):
            if lay["class"] == "compact":
                raw = lay["data"]
            elif lay["address"] == UNDEF:
                raw = None
            else:
                raw = self.file._read(lay["address"], int(np.prod(self.shape)) * self.dtype.itemsize)
            full = (np.frombuffer(raw, dtype=self.dtype).reshape(self.shape) if raw is not None
                    else np.full(self.shape, self.fill if self.fill is not None else 0, dtype=self.dtype))
            out[...] = full[tuple(slice(l, h) for l, h in zip(lo, hi))]
        else:
            c = lay["chunk"]
            ranges = [range((l // cs) * cs, h, cs) for l, h, cs in zip(lo, hi, c)]
            for off in np.array(np.meshgrid(*ranges, indexing="ij")).reshape(rank, -1).T if rank else [()]:
                off = tuple(int(x) for x in off)
                block = self._read_chunk(off)
                src, dst = [], []
                for d in range(rank):
                    a, b = max(lo[d], off[d]), min(hi[d], off[d] + c[d], self.shape[d])
                    src.append(slice(a - off[d], b - off[d])); dst.append(slice(a - lo[d], b - lo[d]))
                out[tuple(dst)] = block[tuple(src)]
        return out.reshape([n for d, n in enumerate(out.shape) if d not in squeeze]) if squeeze else out

    def read_leading(self, k):
        """array[k] — one time level of a [time, lat, lon] variable."""
        return self.read((int(k),))


class HDF5File:
    def __init__(self, path):
        self.path = path
        self._fh = open(path, "rb")
        self._parse_superblock()
        self._root = None

    def close(self):
        if self._fh:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- raw access -------------------------------------------------------------------------------------------------------
    def _read(self, addr, n):
        self._fh.seek(self.base + addr)
        b = self._fh.read(n)
        if len(b) != n:
            raise HDF5FormatError(f"{self.path}: short read of {n} bytes at {addr}")
        return b

    def _uint(self, b, off, size):
        return int.from_bytes(b[off:off + size], "little")

    def _parse_superblock(self):
        self._fh.seek(0, 2)
        end = self._fh.tell()
        at = 0
        while True:     # the signature sits at 0, 512, 1024, 2048, … (a user block may precede it)
            self._fh.seek(at)
            if self._fh.read(8) == SIGNATURE:
                break
            at = 512 if at == 0 else at * 2
            if at >= end:
                raise HDF5FormatError(f"{self.path}: no HDF5 signature (a NetCDF classic file? use ClassicNetCDFFiles)")
        self.base = 0
        self._fh.seek(at)
        head = self._fh.read(128)
        ver = head[8]
        self.sb_version = ver
        if ver in (0, 1):
            self.O, self.L = head[13], head[14]
            p = 24 + (4 if ver == 1 else 0)
            base = self._uint(head, p, self.O)
            p += 4 * self.O                       # base, free-space info, end of file, driver info
            entry = head[p:p + 2 * self.O + 24]   # the root group's symbol table entry
            self.root_header = self._uint(entry, self.O, self.O)
            self.base = base if base != UNDEF else at
        elif ver in (2, 3):
            self.O, self.L = head[9], head[10]
            p = 12
            base = self._uint(head, p, self.O)
            self.root_header = self._uint(head, p + 3 * self.O, self.O)
            self.base = base
        else:
            raise HDF5Unsupported(f"{self.path}: superblock version {ver}")
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise HDF5Unsupported(f"{self.path}: {self.O}-byte offsets / {self.L}-byte lengths")

    # ---- object headers ---------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, data bytes)] of the object header at `addr`, continuation blocks followed."""
        out = []
        first = self._read(addr, 16)
        if first[:4] == b"OHDR":
            if first[4] != 2:
                raise HDF5Unsupported(f"object header version {first[4]}")
            flags = first[5]
            p = 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
            size_len = 1 << (flags & 0x3)
            pre = self._read(addr, p + size_len)
            chunk0 = self._uint(pre, p, size_len)
            blocks = [(addr + p + size_len, chunk0)]
            tracked = bool(flags & 0x04)
            while blocks:
                a, n = blocks.pop(0)
                b = self._read(a, n)
                q = 0
                while q + 4 <= n:
                    mtype, msize, mflags = b[q], self._uint(b, q + 1, 2), b[q + 3]
                    q += 4 + (2 if tracked else 0)
                    if q + msize > n:
                        break
                    data = b[q:q + msize]
                    q += msize
                    if mtype == 0x10:
                        ca, cl = self._uint(data, 0, self.O), self._uint(data, self.O, self.L)
                        if self._read(ca, 4) != b"OCHK":
                            raise HDF5FormatError("object header continuation without OCHK signature")
                        blocks.append((ca + 4, cl - 8))      # between the signature and the checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        # version 1
        if first[0] != 1:
            raise HDF5Unsupported(f"object header version {first[0]} at {addr}")
        nmsg = self._uint(first, 2, 2)
        size = self._uint(first, 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg + 64:
            a, n = blocks.pop(0)
            b = self._read(a, n)
            q = 0
            while q + 8 <= n:
                mtype, msize, mflags = self._uint(b, q, 2), self._uint(b, q + 2, 2), b[q + 4]
                data = b[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:
                    blocks.append((self._uint(data, 0, self.O), self._uint(data, self.O, self.L)))
                elif mtype != 0:
                    out.append((mtype, mflags, data))
        return out

    # ---- the root group ------------------------------------------------------------------------------------------------------
    def links(self):
        """{name: object header address} of the root group."""
        if self._root is not None:
            return self._root
        links = {}
        for mtype, _, d in self._messages(self.root_header):
            if mtype == 0x11:     # symbol table message: a classic group
                self._walk_group_btree(self._uint(d, 0, self.O), self._local_heap(self._uint(d, self.O, self.O)), links)
            elif mtype == 0x06:   # link message
                name, target = self._link_message(d)
                if target is not None:
                    links[name] = target
            elif mtype == 0x02:   # link info: dense storage?
                flags = d[1]
                p = 2 + (8 if flags & 1 else 0)
                heap = self._uint(d, p, self.O)
                if heap != UNDEF & ((1 << (8 * self.O)) - 1):
                    raise HDF5Unsupported(f"{self.path}: the root group stores its links densely (fractal heap: more than eight objects); "
                                          "not read by this subset — convert with `nccopy -k classic` or `-k cdf5`")
        self._root = links
        return links

    def _link_message(self, d):
        flags = d[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = d[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        ln = 1 << (flags & 0x3)
        n = self._uint(d, p, ln); p += ln
        name = d[p:p + n].decode("utf-8", "replace"); p += n
        return name, (self._uint(d, p, self.O) if ltype == 0 else None)     # hard links only

    def _local_heap(self, addr):
        h = self._read(addr, 8 + 2 * self.L + self.O)
        if h[:4] != b"HEAP":
            raise HDF5FormatError("local heap signature")
        size = self._uint(h, 8, self.L)
        data_addr = self._uint(h, 8 + 2 * self.L, self.O)
        return self._read(data_addr, size)

    def _walk_group_btree(self, addr, heap, links):
        h = self._read(addr, 8 + 2 * self.O)
        if h[:4] != b"TREE" or h[4] != 0:
            raise HDF5FormatError("group B-tree node")
        level, used = h[5], self._uint(h, 6, 2)
        body = self._read(addr + 8 + 2 * self.O, (2 * used + 1) * max(self.L, self.O))
        p = self.L    # key 0
        for _ in range(used):
            child = self._uint(body, p, self.O); p += self.O + self.L
            if level > 0:
                self._walk_group_btree(child, heap, links)
            else:
                s = self._read(child, 8)
                if s[:4] != b"SNOD":
                    raise HDF5FormatError("symbol table node signature")
                n = self._uint(s, 6, 2)
                ent = self._read(child + 8, n * (2 * self.O + 24))
                for k in range(n):
                    e = ent[k * (2 * self.O + 24):]
                    off, target = self._uint(e, 0, self.O), self._uint(e, self.O, self.O)
                    name = heap[off:heap.index(b"\0", off)].decode("utf-8", "replace")
                    links[name] = target

    # ---- datasets ---------------------------------------------------------------------------------------------------------------
    def dataset(self, name):
        links = self.links()
        if name not in links:
            raise KeyError(f"{self.path}: no object '{name}' in the root group (it holds {sorted(links)})")
        shape = dtype = layout = None
        filters, fill, attrs = [], None, {}
        for mtype, _, d in self._messages(links[name]):
            if mtype == 0x01:
                shape = self._dataspace(d)
            elif mtype == 0x03:
                dtype = self._datatype(d)[0]
            elif mtype == 0x08:
                layout = self._layout(d)
            elif mtype == 0x0B:
                filters = self._filters(d)
            elif mtype == 0x05 and dtype is not None:
                fill = self._fill_value(d, dtype)
            elif mtype == 0x0C:
                try:
                    k, v = self._attribute(d)
                    attrs[k] = v
                except (HDF5Unsupported, HDF5FormatError, IndexError, ValueError, struct.error):
                    pass       # an attribute of a type this subset does not read is skipped, not fatal
        if shape is None or dtype is None or layout is None:
            raise HDF5FormatError(f"{self.path}: '{name}' is not a dataset this subset understands (dataspace / datatype / layout message missing)")
        if layout["class"] == "chunked":
            layout["chunk"] = tuple(layout["chunk"][:len(shape)])
        return Dataset(self, name, shape, dtype, layout, filters, fill, attrs)

    def _dataspace(self, d):
        ver, rank, flags = d[0], d[1], d[2]
        p = 8 if ver == 1 else 4
        if ver not in (1, 2):
            raise HDF5Unsupported(f"dataspace message version {ver}")
        return tuple(self._uint(d, p + k * self.L, self.L) for k in range(rank))

    def _datatype(self, d):
        cls, ver = d[0] & 0x0F, d[0] >> 4
        bits = d[1] | (d[2] << 8) | (d[3] << 16)
        size = self._uint(d, 4, 4)
        order = ">" if bits & 1 else "<"
        if cls == 0:
            signed = bool(bits & 0x08)
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}"), 8 + 4
        if cls == 1:
            if size not in (2, 4, 8):
                raise HDF5Unsupported(f"{size}-byte floating point")
            return np.dtype(f"{order}f{size}"), 8 + 12
        if cls == 3:
            return np.dtype(f"S{size}"), 8
        raise HDF5Unsupported(f"datatype class {cls}")

    def _layout(self, d):
        ver = d[0]
        if ver == 3:
            cls = d[1]
            if cls == 0:
                n = self._uint(d, 2, 2)
                return {"class": "compact", "data": bytes(d[4:4 + n])}
            if cls == 1:
                return {"class": "contiguous", "address": self._uint(d, 2, self.O)}
            if cls == 2:
                nd = d[2]
                addr = self._uint(d, 3, self.O)
                dims = [self._uint(d, 3 + self.O + 4 * k, 4) for k in range(nd)]
                return {"class": "chunked", "index": "btree1", "address": addr, "chunk": dims[:-1], "element": dims[-1]}
            raise HDF5Unsupported(f"layout class {cls}")
        if ver == 4:
            cls = d[1]
            if cls == 0:
                n = self._uint(d, 2, 2)
                return {"class": "compact", "data": bytes(d[4:4 + n])}
            if cls == 1:
                return {"class": "contiguous", "address": self._uint(d, 2, self.O)}
            if cls == 2:
                flags, nd, enc = d[2], d[3], d[4]
                dims = [self._uint(d, 5 + enc * k, enc) for k in range(nd)]
                p = 5 + enc * nd
                itype = d[p]; p += 1
                lay = {"class": "chunked", "chunk": dims[:-1], "element": dims[-1]}
                if itype == 1:
                    if flags & 0x02:   # filtered single chunk: its stored size and filter mask
                        lay["stored"] = self._uint(d, p, self.L); lay["mask"] = self._uint(d, p + self.L, 4); p += self.L + 4
                    lay.update(index="single", address=self._uint(d, p, self.O))
                elif itype == 2:
                    lay.update(index="implicit", address=self._uint(d, p, self.O))
                else:
                    raise HDF5Unsupported(f"version-4 chunk index type {itype} (fixed array / extensible array / B-tree v2: files written with "
                                          "the 1.10 'latest' format) — convert with `nccopy -k classic`, `-k cdf5`, or `h5repack --low=0 --high=1`")
                return lay
            raise HDF5Unsupported(f"layout class {cls}")
        raise HDF5Unsupported(f"data layout message version {ver}")

    def _filters(self, d):
        ver, n = d[0], d[1]
        out, p = [], 8 if ver == 1 else 2
        for _ in range(n):
            fid = self._uint(d, p, 2); p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = self._uint(d, p, 2); p += 2
            p += 2     # flags
            ncd = self._uint(d, p, 2); p += 2
            if ver == 1:
                nlen = (nlen + 7) // 8 * 8
            p += nlen
            cd = [self._uint(d, p + 4 * k, 4) for k in range(ncd)]
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _fill_value(self, d, dtype):
        ver = d[0]
        try:
            if ver in (1, 2):
                defined = d[3] if ver == 2 else 1
                if not defined:
                    return None
                n = self._uint(d, 4, 4)
                return np.frombuffer(d[8:8 + n], dtype=dtype)[0] if n == dtype.itemsize else None
            if ver == 3:
                if not d[1] & 0x20:
                    return None
                n = self._uint(d, 2, 4)
                return np.frombuffer(d[6:6 + n], dtype=dtype)[0] if n == dtype.itemsize else None
        except (IndexError, ValueError):
            pass
        return None

    def _attribute(self, d):
        ver = d[0]
        if ver not in (1, 2, 3):
            raise HDF5Unsupported("attribute version")
        nlen, tlen, slen = self._uint(d, 2, 2), self._uint(d, 4, 2), self._uint(d, 6, 2)
        p = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = d[p:p + nlen].split(b"\0")[0].decode("utf-8", "replace"); p += pad(nlen)
        dtype, _ = self._datatype(d[p:p + tlen]); p += pad(tlen)
        shape = self._dataspace(d[p:p + slen]) if slen >= 4 else (); p += pad(slen)
        n = int(np.prod(shape)) if shape else 1
        a = np.frombuffer(d[p:p + n * dtype.itemsize], dtype=dtype)
        if dtype.kind == "S":
            return name, a[0].split(b"\0")[0].decode("utf-8", "replace")
        return name, (a[0].item() if n == 1 else a.copy())

    # ---- chunk index -------------------------------------------------------------------------------------------------------------
    def _walk_chunk_btree(self, addr, rank, out):
        h = self._read(addr, 8 + 2 * self.O)
        if h[:4] != b"TREE" or h[4] != 1:
            raise HDF5FormatError("chunk B-tree node")
        level, used = h[5], self._uint(h, 6, 2)
        key = 8 + 8 * (rank + 1)
        body = self._read(addr + 8 + 2 * self.O, used * (key + self.O) + key)
        p = 0
        for _ in range(used):
            size, mask = self._uint(body, p, 4), self._uint(body, p + 4, 4)
            off = tuple(self._uint(body, p + 8 + 8 * k, 8) for k in range(rank))
            child = self._uint(body, p + key, self.O)
            p += key + self.O
            if level > 0:
                self._walk_chunk_btree(child, rank, out)
            else:
                out[off] = (child, size, mask)


def open_variable(path, name=None):
    """(HDF5File, Dataset) for `name`, or for the file's only dataset of rank ≥ 3 when the name is not there."""
    f = HDF5File(path)
    links = f.links()
    if name is not None and name in links:
        return f, f.dataset(name)
    cubes = []
    for k in links:
        try:
            ds = f.dataset(k)
        except (HDF5Unsupported, HDF5FormatError):
            continue
        if len(ds.shape) >= 3:
            cubes.append(ds)
    if len(cubes) != 1:
        f.close()
        raise KeyError(f"{path}: no variable '{name}' and {len(cubes)} datasets of rank >= 3 to choose from ({sorted(links)})")
    return f, cubes[0]


if __name__ == "__main__":
    import sys
    with HDF5File(sys.argv[1]) as f:
        print(f"superblock version {f.sb_version}, {f.O}-byte offsets; root group: {sorted(f.links())}")
        for k in ([sys.argv[2]] if len(sys.argv) > 2 else sorted(f.links())):
            try:
                ds = f.dataset(k)
                print(f"  {k}: shape {ds.shape} dtype {ds.dtype} layout {ds.layout['class']} chunk {ds.chunks} filters {ds.filters} attrs {ds.attrs}")
            except (HDF5Unsupported, HDF5FormatError) as exc:
                print(f"  {k}: {exc}")

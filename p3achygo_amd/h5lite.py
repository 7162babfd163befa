"""Read-only HDF5 subset in numpy — what a Keras `model.weights.h5` needs.

The reference stores checkpoints as `.keras` archives: a zip whose `model.weights.h5` holds one HDF5 dataset
per variable (python/model_utils.py:197-204; python/scripts/migrate_checkpoint.py:81-90 walks it with
h5py.visititems).  h5py is not importable by the interpreter this build runs under, so this module restates the
parts of the published HDF5 file format (File Format Specification, version 3.0) such files use:

  superblock versions 0-3; object headers versions 1 and 2 (continuation blocks included); groups stored as
  symbol tables (B-tree v1 + local heap: h5py's default, libver="earliest") or as compact link messages
  (libver="latest", up to eight links per group); datasets with contiguous, compact or chunked (B-tree v1)
  layout, little- or big-endian IEEE floats and integers, deflate / shuffle / fletcher32 filters.

Not covered (a clear NotImplementedError, never a silent wrong answer): dense link storage (fractal heaps),
version-4 chunk indexes other than the single-chunk one, compound / string / variable-length types, external
storage.  Pinned by files the HDF5 library itself wrote (tests/golden/h5/, generator beside them).
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(ValueError):
    pass


class _Dataset:
    def __init__(self):
        self.shape: Optional[Tuple[int, ...]] = None
        self.dtype: Optional[np.dtype] = None
        self.layout = None          # ("contiguous", addr, size) | ("compact", bytes) | ("chunked", btree, chunk dims, elem size) | ("single", addr, size, chunk dims)
        self.filters: List[Tuple[int, Tuple[int, ...]]] = []


class File:
    """`File(path_or_bytes)`; `datasets()` -> {"a/b/c": ndarray} in file order; `read(path)` for one."""

    def __init__(self, src):
        if isinstance(src, (bytes, bytearray, memoryview)):
            self.buf = bytes(src)
        else:
            with open(src, "rb") as f:
                self.buf = f.read()
        self._superblock()

    # ---- primitives ---------------------------------------------------------------------
    def _u(self, off: int, n: int) -> int:
        if off < 0 or off + n > len(self.buf):
            raise H5Error(f"read of {n} bytes at {off} beyond the end of the file ({len(self.buf)})")
        return int.from_bytes(self.buf[off:off + n], "little")

    def _addr(self, off: int) -> Optional[int]:
        v = self._u(off, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v + self.base

    def _superblock(self):
        pos = 0
        while True:
            if self.buf[pos:pos + 8] == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(self.buf):
                raise H5Error("not an HDF5 file (no signature)")
        ver = self.buf[pos + 8]
        self.base = 0
        if ver in (0, 1):
            self.O, self.L = self.buf[pos + 13], self.buf[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = self._u(p, self.O)
            p += 4 * self.O                    # base, free-space info, end of file, driver info
            self.root = self._addr(p + self.O)  # symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            self.O, self.L = self.buf[pos + 9], self.buf[pos + 10]
            p = pos + 12
            self.base = self._u(p, self.O)
            self.root = self._addr(p + 3 * self.O)
        else:
            raise H5Error(f"superblock version {ver}")
        if self.O not in (2, 4, 8) or self.L not in (2, 4, 8):
            raise H5Error("size of offsets / lengths")

    # ---- object headers -----------------------------------------------------------------
    def _messages(self, addr: int) -> Iterator[Tuple[int, int, int]]:
        """(type, data offset, data size) of every message of the object header at `addr`."""
        if self.buf[addr:addr + 4] == b"OHDR":
            yield from self._messages_v2(addr)
            return
        if self.buf[addr] != 1:
            raise H5Error(f"object header version {self.buf[addr]} at {addr}")
        nmsg = self._u(addr + 2, 2)
        blocks = [(addr + 16, self._u(addr + 8, 4))]
        seen = 0
        while blocks and seen < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and seen < nmsg:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                data = p + 8
                seen += 1
                if mtype == 0x10:
                    blocks.append((self._addr(data), self._u(data + self.O, self.L)))
                else:
                    yield mtype, data, msize
                p = data + msize   # version-1 message sizes are multiples of 8 already

    def _messages_v2(self, addr: int) -> Iterator[Tuple[int, int, int]]:
        flags = self.buf[addr + 5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        n = 1 << (flags & 3)
        size0 = self._u(p, n)
        p += n
        blocks = [(p, size0)]
        track = bool(flags & 0x04)
        while blocks:
            p, size = blocks.pop(0)
            end = p + size
            hdr = 4 + (2 if track else 0)
            while p + hdr <= end:
                mtype, msize = self.buf[p], self._u(p + 1, 2)
                data = p + hdr
                if mtype == 0x10:
                    caddr, clen = self._addr(data), self._u(data + self.O, self.L)
                    if self.buf[caddr:caddr + 4] != b"OCHK":
                        raise H5Error("continuation block signature")
                    blocks.append((caddr + 4, clen - 8))   # without signature and checksum
                elif mtype != 0:
                    yield mtype, data, msize
                p = data + msize

    # ---- groups -------------------------------------------------------------------------
    def _links(self, addr: int) -> Optional[List[Tuple[str, int]]]:
        """Children (name, object header address) if the object at `addr` is a group, else None."""
        links: List[Tuple[str, int]] = []
        is_group = False
        for mtype, d, size in self._messages(addr):
            if mtype == 0x11:       # symbol table
                is_group = True
                btree, heap = self._addr(d), self._addr(d + self.O)
                links += self._symbol_table(btree, heap)
            elif mtype == 0x02:     # link info
                is_group = True
                flags = self.buf[d + 1]
                p = d + 2 + (8 if flags & 1 else 0)
                if self._addr(p) is not None:
                    raise NotImplementedError("group with dense link storage (fractal heap)")
            elif mtype == 0x06:     # link
                is_group = True
                flags = self.buf[d + 1]
                p = d + 2
                ltype = 0
                if flags & 0x08:
                    ltype = self.buf[p]
                    p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                n = 1 << (flags & 3)
                ln = self._u(p, n)
                p += n
                name = self.buf[p:p + ln].decode("utf-8")
                p += ln
                if ltype == 0:
                    links.append((name, self._addr(p)))
                # soft / external links carry no data of their own
            elif mtype == 0x08 or mtype == 0x01:
                return None         # a dataset
        return links if is_group else None

    def _symbol_table(self, btree: int, heap: int) -> List[Tuple[str, int]]:
        if self.buf[heap:heap + 4] != b"HEAP":
            raise H5Error("local heap signature")
        heap_data = self._addr(heap + 8 + 2 * self.L)
        out: List[Tuple[str, int]] = []

        def node(addr: int):
            sig = self.buf[addr:addr + 4]
            if sig == b"TREE":
                if self.buf[addr + 4] != 0:
                    raise H5Error("group B-tree node type")
                used = self._u(addr + 6, 2)
                p = addr + 8 + 2 * self.O + self.L       # past the first key
                for _ in range(used):
                    node(self._addr(p))
                    p += self.O + self.L
            elif sig == b"SNOD":
                nsym = self._u(addr + 6, 2)
                p = addr + 8
                for _ in range(nsym):
                    name_off, ohdr = self._u(p, self.O), self._addr(p + self.O)
                    q = heap_data + name_off
                    e = self.buf.index(b"\0", q)
                    out.append((self.buf[q:e].decode("utf-8"), ohdr))
                    p += 2 * self.O + 24
            else:
                raise H5Error(f"group node signature {sig!r}")

        node(btree)
        return out

    # ---- datasets -----------------------------------------------------------------------
    def _dataset(self, addr: int) -> _Dataset:
        ds = _Dataset()
        for mtype, d, size in self._messages(addr):
            if mtype == 0x01:
                ver, rank, flags = self.buf[d], self.buf[d + 1], self.buf[d + 2]
                if ver == 1:
                    p = d + 8
                elif ver == 2:
                    p = d + 4
                    if self.buf[d + 3] == 2:
                        raise H5Error("null dataspace")
                else:
                    raise H5Error(f"dataspace version {ver}")
                ds.shape = tuple(self._u(p + i * self.L, self.L) for i in range(rank))
            elif mtype == 0x03:
                cls, bits0, tsize = self.buf[d] & 0x0f, self.buf[d + 1], self._u(d + 4, 4)
                order = ">" if bits0 & 1 else "<"
                if cls == 0:
                    kind = "i" if bits0 & 0x08 else "u"
                elif cls == 1:
                    kind = "f"
                else:
                    raise NotImplementedError(f"datatype class {cls}")
                if (kind == "f" and tsize not in (2, 4, 8)) or tsize not in (1, 2, 4, 8):
                    raise NotImplementedError(f"{kind}{tsize}")
                ds.dtype = np.dtype(f"{order}{kind}{tsize}")
            elif mtype == 0x08:
                ver, cls = self.buf[d], self.buf[d + 1]
                if ver == 3 or ver == 4:
                    if cls == 0:
                        n = self._u(d + 2, 2)
                        ds.layout = ("compact", self.buf[d + 4:d + 4 + n])
                    elif cls == 1:
                        ds.layout = ("contiguous", self._addr(d + 2), self._u(d + 2 + self.O, self.L))
                    elif cls == 2 and ver == 3:
                        nd = self.buf[d + 2]
                        bt = self._addr(d + 3)
                        dims = tuple(self._u(d + 3 + self.O + 4 * i, 4) for i in range(nd))
                        ds.layout = ("chunked", bt, dims[:-1], dims[-1])
                    elif cls == 2:
                        flags, nd, enc = self.buf[d + 2], self.buf[d + 3], self.buf[d + 4]
                        dims = tuple(self._u(d + 5 + enc * i, enc) for i in range(nd))
                        p = d + 5 + enc * nd
                        index = self.buf[p]
                        if index != 1:
                            raise NotImplementedError(f"chunk index type {index} (version-4 layout)")
                        p += 1
                        fsize = None
                        if flags & 0x02:
                            fsize = self._u(p, self.L)
                            p += self.L + 4
                        ds.layout = ("single", self._addr(p), fsize, dims[:-1])
                    else:
                        raise NotImplementedError(f"layout class {cls}")
                else:
                    raise NotImplementedError(f"data layout version {ver}")
            elif mtype == 0x0B:
                ver, nf = self.buf[d], self.buf[d + 1]
                p = d + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = self._u(p, 2)
                    p += 2
                    nlen = 0
                    if ver == 1 or fid >= 256:
                        nlen = self._u(p, 2)
                        p += 2
                    p += 2      # flags
                    ncd = self._u(p, 2)
                    p += 2
                    if ver == 1:
                        nlen = (nlen + 7) // 8 * 8
                    p += nlen
                    cd = tuple(self._u(p + 4 * i, 4) for i in range(ncd))
                    p += 4 * ncd
                    if ver == 1 and ncd % 2:
                        p += 4
                    ds.filters.append((fid, cd))
        if ds.shape is None or ds.dtype is None or ds.layout is None:
            raise H5Error("dataset without dataspace / datatype / layout")
        return ds

    def _unfilter(self, raw: bytes, ds: _Dataset, mask: int) -> bytes:
        for i in reversed(range(len(ds.filters))):
            if mask >> i & 1:
                continue
            fid, cd = ds.filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else ds.dtype.itemsize
                n = len(raw) // es
                a = np.frombuffer(raw[:n * es], np.uint8).reshape(es, n).T.tobytes()
                raw = a + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise NotImplementedError(f"filter {fid}")
        return raw

    def _read(self, ds: _Dataset) -> np.ndarray:
        count = int(np.prod(ds.shape, dtype=np.int64)) if ds.shape else 1
        nbytes = count * ds.dtype.itemsize
        kind = ds.layout[0]
        if kind == "compact":
            raw = ds.layout[1][:nbytes]
        elif kind == "contiguous":
            addr = ds.layout[1]
            if addr is None:        # never written: the fill value (zero)
                return np.zeros(ds.shape, ds.dtype.newbyteorder("="))
            raw = self.buf[addr:addr + nbytes]
        elif kind == "single":
            _, addr, fsize, cdims = ds.layout
            n = fsize if fsize is not None else int(np.prod(cdims, dtype=np.int64)) * ds.dtype.itemsize
            chunk = np.frombuffer(self._unfilter(self.buf[addr:addr + n], ds, 0), ds.dtype).reshape(cdims)
            return np.ascontiguousarray(chunk[tuple(slice(0, s) for s in ds.shape)]).astype(ds.dtype.newbyteorder("="))
        else:
            _, bt, cdims, esize = ds.layout
            out = np.zeros(ds.shape, ds.dtype)
            if bt is not None:
                self._chunks(bt, ds, cdims, out)
            return out.astype(ds.dtype.newbyteorder("="))
        if len(raw) != nbytes:
            raise H5Error("dataset data beyond the end of the file")
        return np.frombuffer(raw, ds.dtype).reshape(ds.shape).astype(ds.dtype.newbyteorder("="))

    def _chunks(self, addr: int, ds: _Dataset, cdims, out: np.ndarray):
        if self.buf[addr:addr + 4] != b"TREE" or self.buf[addr + 4] != 1:
            raise H5Error("chunk B-tree node")
        level, used = self.buf[addr + 5], self._u(addr + 6, 2)
        nd = len(cdims)
        keysize = 8 + 8 * (nd + 1)
        p = addr + 8 + 2 * self.O
        for _ in range(used):
            csize, mask = self._u(p, 4), self._u(p + 4, 4)
            offs = tuple(self._u(p + 8 + 8 * i, 8) for i in range(nd))
            child = self._addr(p + keysize)
            if level > 0:
                self._chunks(child, ds, cdims, out)
            else:
                raw = self._unfilter(self.buf[child:child + csize], ds, mask)
                chunk = np.frombuffer(raw, ds.dtype).reshape(cdims)
                sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, ds.shape))
                out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
            p += keysize + self.O

    # ---- public -------------------------------------------------------------------------
    def walk(self) -> Iterator[Tuple[str, int]]:
        """(path, object header address) of every dataset, depth first in stored link order."""
        seen = set()

        def rec(addr: int, prefix: str):
            if addr in seen:
                return
            seen.add(addr)
            links = self._links(addr)
            if links is None:
                yield prefix, addr
                return
            for name, child in links:
                if child is not None:
                    yield from rec(child, f"{prefix}/{name}" if prefix else name)

        yield from rec(self.root, "")

    def datasets(self) -> Dict[str, np.ndarray]:
        try:
            return {path: self._read(self._dataset(addr)) for path, addr in self.walk()}
        except (IndexError, struct.error, zlib.error, UnicodeDecodeError, TypeError) as e:   # a damaged file
            raise H5Error(f"malformed HDF5 file: {e}") from e
        except ValueError as e:
            if isinstance(e, H5Error):
                raise
            raise H5Error(f"malformed HDF5 file: {e}") from e

    def read(self, path: str) -> np.ndarray:
        addr = self.root
        for part in [p for p in path.split("/") if p]:
            links = self._links(addr)
            if links is None:
                raise KeyError(path)
            d = dict(links)
            if part not in d:
                raise KeyError(path)
            addr = d[part]
        return self._read(self._dataset(addr))

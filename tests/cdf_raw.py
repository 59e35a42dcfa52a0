"""A reader and a writer of NetCDF classic / 64-bit-offset files written from the format's published layout (header:
magic, numrecs, dimension list, global attributes, variable list with offsets; then the fixed-size variables and the
record variables, everything big-endian and padded to four bytes) -- and from nothing else: no library, and none of
this repository's own file code.  It exists so that the tests can look at the bytes mesh_io writes, and hand mesh_io
bytes it did not write.  Only what MPAS grid and restart files use: int32 and float64 variables, text and number
attributes, at most one record dimension."""
import struct

import numpy as np

NC_BYTE, NC_CHAR, NC_SHORT, NC_INT, NC_FLOAT, NC_DOUBLE = 1, 2, 3, 4, 5, 6
NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE = 10, 11, 12
_DT = {NC_INT: ">i4", NC_DOUBLE: ">f8", NC_FLOAT: ">f4", NC_SHORT: ">i2", NC_BYTE: "i1"}


def _pad(n):
    return (4 - n % 4) % 4


class _In:
    def __init__(self, raw):
        self.raw, self.o = raw, 0

    def i4(self):
        v = struct.unpack_from(">i", self.raw, self.o)[0]
        self.o += 4
        return v

    def i8(self):
        v = struct.unpack_from(">q", self.raw, self.o)[0]
        self.o += 8
        return v

    def name(self):
        n = self.i4()
        s = self.raw[self.o:self.o + n].decode()
        self.o += n + _pad(n)
        return s

    def atts(self):
        tag, n = self.i4(), self.i4()
        assert tag in (0, NC_ATTRIBUTE)
        out = {}
        for _ in range(n):
            k = self.name()
            t, cnt = self.i4(), self.i4()
            if t == NC_CHAR:
                v = self.raw[self.o:self.o + cnt].decode()
                nbytes = cnt
            else:
                v = np.frombuffer(self.raw, dtype=_DT[t], count=cnt, offset=self.o)
                nbytes = v.nbytes
                v = v[0] if cnt == 1 else v.copy()
            self.o += nbytes + _pad(nbytes)
            out[k] = v
        return out


def read(path):
    """(dimensions {name: length, the record dimension as None}, global attributes, variables {name: (dimension
    names, array)})"""
    raw = open(path, "rb").read()
    f = _In(raw)
    assert raw[:3] == b"CDF" and raw[3] in (1, 2), "not a classic or 64-bit-offset NetCDF file"
    big = raw[3] == 2
    f.o = 4
    numrecs = f.i4()
    tag, n = f.i4(), f.i4()
    assert tag in (0, NC_DIMENSION)
    dims = []
    for _ in range(n):
        k = f.name()
        ln = f.i4()
        dims.append((k, None if ln == 0 else ln))
    gatts = f.atts()
    tag, n = f.i4(), f.i4()
    assert tag in (0, NC_VARIABLE)
    metas = []
    for _ in range(n):
        k = f.name()
        nd = f.i4()
        ids = [f.i4() for _ in range(nd)]
        f.atts()
        t, vsize = f.i4(), f.i4()
        begin = f.i8() if big else f.i4()
        metas.append((k, ids, t, vsize, begin))
    recsize = sum(vs for _, ids, _, vs, _ in metas if ids and dims[ids[0]][1] is None)
    out = {}
    for k, ids, t, vsize, begin in metas:
        names = tuple(dims[i][0] for i in ids)
        is_rec = bool(ids) and dims[ids[0]][1] is None
        shape = tuple(dims[i][1] for i in (ids[1:] if is_rec else ids))
        cnt = int(np.prod(shape)) if shape else 1
        if is_rec:
            recs = [np.frombuffer(raw, dtype=_DT[t], count=cnt, offset=begin + r * recsize).reshape(shape)
                    for r in range(numrecs)]
            a = np.stack(recs) if recs else np.zeros((0,) + shape, dtype=_DT[t])
        else:
            a = np.frombuffer(raw, dtype=_DT[t], count=cnt, offset=begin).reshape(shape)
        out[k] = (names, a.astype(a.dtype.newbyteorder("=")))
    return dict(dims), gatts, out


def write(path, dims, gatts, variables, version=2):
    """dims: list of (name, length or None for the one record dimension); gatts: {name: str | float | int};
    variables: list of (name, dimension names, numpy array of int32 or float64 with the full shape; a record variable
    with its records along axis 0)."""
    names = [d for d, _ in dims]
    rec = [d for d, ln in dims if ln is None]
    numrecs = 0
    for _, dn, a in variables:
        if dn and dn[0] in rec:
            numrecs = max(numrecs, a.shape[0])

    def nm(s):
        b = s.encode()
        return struct.pack(">i", len(b)) + b + b"\0" * _pad(len(b))

    def att_list(d):
        if not d:
            return struct.pack(">ii", 0, 0)
        out = struct.pack(">ii", NC_ATTRIBUTE, len(d))
        for k, v in d.items():
            if isinstance(v, str):
                b = v.encode()
                out += nm(k) + struct.pack(">ii", NC_CHAR, len(b)) + b + b"\0" * _pad(len(b))
            elif isinstance(v, (int, np.integer)):
                out += nm(k) + struct.pack(">ii", NC_INT, 1) + struct.pack(">i", int(v))
            else:
                out += nm(k) + struct.pack(">ii", NC_DOUBLE, 1) + struct.pack(">d", float(v))
        return out
    head = b"CDF" + bytes([version]) + struct.pack(">i", numrecs)
    head += struct.pack(">ii", NC_DIMENSION, len(dims))
    for d, ln in dims:
        head += nm(d) + struct.pack(">i", 0 if ln is None else ln)
    head += att_list(gatts)
    # variable entries need the offsets, which need the header's length: lay the entries out first
    entries = []
    for k, dn, a in variables:
        t = NC_INT if a.dtype.kind == "i" else NC_DOUBLE
        is_rec = bool(dn) and dn[0] in rec
        per = a[0] if is_rec else a
        vsize = per.size * (4 if t == NC_INT else 8)
        vsize += _pad(vsize)
        entries.append((k, dn, a, t, is_rec, vsize))
    var_head_len = 8 + sum(len(nm(k)) + 4 + 4 * len(dn) + 8 + 8 + (8 if version == 2 else 4) for k, dn, *_ in entries)
    o = len(head) + var_head_len
    begins = {}
    for k, dn, a, t, is_rec, vsize in entries:
        if not is_rec:
            begins[k] = o
            o += vsize
    rec_start = o
    for k, dn, a, t, is_rec, vsize in entries:
        if is_rec:
            begins[k] = o
            o += vsize
    recsize = o - rec_start
    vh = struct.pack(">ii", NC_VARIABLE if entries else 0, len(entries))
    for k, dn, a, t, is_rec, vsize in entries:
        vh += nm(k) + struct.pack(">i", len(dn)) + b"".join(struct.pack(">i", names.index(d)) for d in dn)
        vh += struct.pack(">ii", 0, 0) + struct.pack(">ii", t, vsize)
        vh += struct.pack(">q" if version == 2 else ">i", begins[k])
    assert len(vh) == var_head_len
    with open(path, "wb") as f:
        f.write(head + vh)
        for k, dn, a, t, is_rec, vsize in entries:
            if not is_rec:
                b = np.ascontiguousarray(a, dtype=_DT[t]).tobytes()
                f.write(b + b"\0" * (vsize - len(b)))
        for r in range(numrecs):
            for k, dn, a, t, is_rec, vsize in entries:
                if is_rec:
                    b = np.ascontiguousarray(a[r], dtype=_DT[t]).tobytes()
                    f.write(b + b"\0" * (vsize - len(b)))
        assert recsize >= 0

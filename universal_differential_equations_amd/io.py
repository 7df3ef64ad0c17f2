"""Result files in the reference's own formats.

    save(joinpath(pwd(), "results", "..recovery_...jld2"), "X", Xn, "t", ts, "initial_parameters", p, ...)   scenario_1.jl:210-213
    @save "data/model.bson" pstar                                                                        Fisher-KPP-CNN.jl:243

`save_jld2` writes the HDF5 subset that JLD2.jl 0.4 itself writes for plain numeric data (the structure of the reference's
`LotkaVolterra/results/*.jld2`): 512-byte JLD2 banner, superblock v2, version-2 object headers with
lookup3 checksums, a root group of hard links, one dataset per entry (IEEE little-endian floats / integers, Julia's
column-major dims reversed, compact layout for small arrays and contiguous otherwise, scalar dataspaces for numbers).
Julia structs (ODESolution, Lux models) are not representable without Julia's type system: save their arrays.
`save_bson` writes BSON.jl's lowering of `Dict(:name => Array)`.  Host-side, numpy only; no GPU involved.
The files are read back by tools/jld2_reader.py (an independent walker written against the reference's files) in tests/.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
BANNER = b"HDF5-based Julia Data Format, version 0.1.1\x00 (udecore MI355X writer, 64-bit LE)\x00"


def lookup3(data, init=0):
    """Bob Jenkins' lookup3 hashlittle, the checksum of HDF5's version-2 metadata (H5_checksum_lookup3)"""
    M = 0xFFFFFFFF

    def rot(x, k):
        return ((x << k) | (x >> (32 - k))) & M

    n = len(data)
    a = b = c = (0xDEADBEEF + n + init) & M
    p = 0
    while n > 12:
        a = (a + int.from_bytes(data[p:p + 4], "little")) & M
        b = (b + int.from_bytes(data[p + 4:p + 8], "little")) & M
        c = (c + int.from_bytes(data[p + 8:p + 12], "little")) & M
        a = (a - c) & M; a ^= rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= rot(b, 4); b = (b + a) & M
        p += 12
        n -= 12
    tail = data[p:p + n] + b"\x00" * (12 - n)
    if n == 0:
        return c
    a = (a + int.from_bytes(tail[0:4], "little")) & M
    b = (b + int.from_bytes(tail[4:8], "little")) & M
    c = (c + int.from_bytes(tail[8:12], "little")) & M
    c ^= b; c = (c - rot(b, 14)) & M
    a ^= c; a = (a - rot(c, 11)) & M
    b ^= a; b = (b - rot(a, 25)) & M
    c ^= b; c = (c - rot(b, 16)) & M
    a ^= c; a = (a - rot(c, 4)) & M
    b ^= a; b = (b - rot(a, 14)) & M
    c ^= b; c = (c - rot(b, 24)) & M
    return c


def _msg(mtype, data, flags=0):
    return struct.pack("<BHB", mtype, len(data), flags) + data


def _ohdr(msgs):
    body = b"".join(msgs)
    assert len(body) < 65536
    head = b"OHDR" + bytes([2, 0x01]) + struct.pack("<H", len(body))   # version 2, flags: 2-byte chunk-0 size
    blob = head + body
    return blob + struct.pack("<I", lookup3(blob))


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        size = dt.itemsize
        ebits, mbits, bias = (11, 52, 1023) if size == 8 else (8, 23, 127)
        # class 1 (floating point), version 3; bit field: little-endian, mantissa normalisation implied, sign at the top bit
        return bytes([0x31, 0x20, 8 * size - 1, 0x00]) + struct.pack("<IHHBBBBI", size, 0, 8 * size, mbits, ebits, 0, mbits, bias)
    if dt.kind in "iu":
        size = dt.itemsize
        return bytes([0x30, 0x08 if dt.kind == "i" else 0x00, 0x00, 0x00]) + struct.pack("<IHH", size, 0, 8 * size)
    raise TypeError("save_jld2: only float32/float64/integer data, got %s" % dt)


def save_jld2(path, **entries):
    """save(path, "name", value, ...) for numbers and numeric arrays (numpy (ns, n) C order == Julia n x ns column-major
    is NOT assumed: an array is written with its numpy shape as Julia's reversed dims, i.e. pass X.T for a Julia n x ns matrix
    held as (ns, n)... simply: what you pass with shape (a, b) reads back in Julia as a b x a matrix, and in
    tools/jld2_reader.py as shape (a, b))."""
    base = 512
    chunks = []          # (relative address, bytes)
    pos = 48             # relative to base: the superblock occupies [0, 48)

    def place(blob):
        nonlocal pos
        addr = pos
        chunks.append((addr, blob))
        pos += len(blob)
        pos = (pos + 7) // 8 * 8
        return addr

    links = []
    for name, val in entries.items():
        arr = np.asarray(val)
        if arr.dtype.kind not in "fiu":
            raise TypeError("save_jld2: entry %r is not numeric" % name)
        arr = np.ascontiguousarray(arr.astype(arr.dtype.newbyteorder("<")))
        raw = arr.tobytes()
        if arr.ndim == 0:
            space = bytes([2, 0, 0, 0])                                   # version 2, rank 0, scalar
        else:
            space = bytes([2, arr.ndim, 0, 1]) + b"".join(struct.pack("<Q", d) for d in arr.shape)   # simple dataspace
        msgs = [_msg(0x05, bytes([3, 0x09])),                              # fill value: version 3, never written / undefined
                _msg(0x01, space), _msg(0x03, _dtype_msg(arr.dtype), flags=1)]
        if len(raw) <= 8192:
            msgs.append(_msg(0x08, bytes([4, 0]) + struct.pack("<H", len(raw)) + raw))               # layout v4, compact
            addr = place(_ohdr(msgs))
        else:
            daddr = place(raw)
            msgs.append(_msg(0x08, bytes([4, 1]) + struct.pack("<QQ", daddr, len(raw))))             # layout v4, contiguous
            addr = place(_ohdr(msgs))
        links.append((name, addr))
    gm = [_msg(0x02, bytes([0, 0]) + struct.pack("<QQ", UNDEF, UNDEF)),     # link info: no fractal heap / name index
          _msg(0x0A, bytes([0, 0]))]                                        # group info
    for name, addr in links:
        nb = name.encode()
        assert len(nb) < 256
        gm.append(_msg(0x06, bytes([1, 0x10, 1, len(nb)]) + nb + struct.pack("<Q", addr)))   # hard link, UTF-8 name
    root = place(_ohdr(gm))
    eof = pos
    sb = b"\x89HDF\r\n\x1a\n" + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", base, UNDEF, eof, root)
    sb += struct.pack("<I", lookup3(sb))
    out = bytearray(base + eof)
    out[:len(BANNER)] = BANNER
    out[base:base + 48] = sb
    for addr, blob in chunks:
        out[base + addr:base + addr + len(blob)] = blob
    with open(path, "wb") as fh:
        fh.write(bytes(out))


# ---- BSON (bsonspec.org 1.1) with BSON.jl's array lowering -------------------------------------------------------------
def _bson_doc(d):
    body = b""
    for k, v in d.items():
        kb = k.encode() + b"\x00"
        if isinstance(v, dict):
            body += b"\x03" + kb + _bson_doc(v)
        elif isinstance(v, (list, tuple)):
            body += b"\x04" + kb + _bson_doc({str(i): x for i, x in enumerate(v)})
        elif isinstance(v, str):
            sb = v.encode() + b"\x00"
            body += b"\x02" + kb + struct.pack("<i", len(sb)) + sb
        elif isinstance(v, (bytes, bytearray)):
            body += b"\x05" + kb + struct.pack("<i", len(v)) + b"\x00" + bytes(v)
        elif isinstance(v, bool):
            body += b"\x08" + kb + (b"\x01" if v else b"\x00")
        elif isinstance(v, (int, np.integer)):
            body += b"\x12" + kb + struct.pack("<q", int(v))
        elif isinstance(v, (float, np.floating)):
            body += b"\x01" + kb + struct.pack("<d", float(v))
        else:
            raise TypeError("bson: unsupported value %r" % type(v))
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


_JL = {"float64": "Float64", "float32": "Float32", "int64": "Int64", "int32": "Int32"}


def save_bson(path, **entries):
    """`@save path name1 name2 ...` for numeric arrays: BSON.jl lowers an Array to
    Dict(:tag => "array", :type => <datatype Core.T>, :size => [dims...], :data => bytes) (column-major data)."""
    doc = {}
    for name, val in entries.items():
        arr = np.asarray(val)
        jl = _JL.get(arr.dtype.name)
        if jl is None:
            raise TypeError("save_bson: dtype %s" % arr.dtype)
        doc[name] = {"tag": "array",
                     "type": {"tag": "datatype", "params": [], "name": ["Core", jl]},
                     "size": [int(d) for d in arr.shape[::-1]],       # numpy (a, b) C order == Julia b x a column-major
                     "data": np.ascontiguousarray(arr).tobytes()}
    with open(path, "wb") as fh:
        fh.write(_bson_doc(doc))


def load_bson(path):
    """inverse of save_bson (arrays only)"""
    b = open(path, "rb").read()

    def doc(p):
        n = struct.unpack_from("<i", b, p)[0]
        end = p + n - 1
        p += 4
        out = {}
        while p < end:
            t = b[p]
            e = b.index(b"\x00", p + 1)
            k = b[p + 1:e].decode()
            p = e + 1
            if t in (3, 4):
                v, p = doc(p)
                if t == 4:
                    v = [v[str(i)] for i in range(len(v))]
            elif t == 2:
                m = struct.unpack_from("<i", b, p)[0]
                v = b[p + 4:p + 4 + m - 1].decode()
                p += 4 + m
            elif t == 5:
                m = struct.unpack_from("<i", b, p)[0]
                v = b[p + 5:p + 5 + m]
                p += 5 + m
            elif t == 0x12:
                v = struct.unpack_from("<q", b, p)[0]
                p += 8
            elif t == 1:
                v = struct.unpack_from("<d", b, p)[0]
                p += 8
            elif t == 8:
                v = bool(b[p])
                p += 1
            else:
                raise ValueError("bson element type %d" % t)
            out[k] = v
        return out, end + 1

    top, _ = doc(0)
    res = {}
    rev = {v: k for k, v in _JL.items()}
    for k, v in top.items():
        dt = np.dtype(rev[v["type"]["name"][1]])
        res[k] = np.frombuffer(v["data"], dtype=dt).reshape(v["size"][::-1]).copy()
    return res

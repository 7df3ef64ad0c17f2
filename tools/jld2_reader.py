"""Minimal pure-Python reader for the JLD2 (HDF5 subset) result files shipped by the reference.

Build-container-only tooling: it reads /root/reference/LotkaVolterra/results/*.jld2 (golden
ODESolution dumps written by `JLD2.save` at scenario_1.jl:210-213 etc.) so that
tools/make_golden.py can turn them into small JSON fixtures under tests/golden/.
No h5py exists in this image; the format subset handled here is the one described in
SURVEY.md Appendix B (superblock v2, object header v2, compact/contiguous layouts,
compound/committed datatypes, object references).
"""
import struct
import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class DType:
    def __init__(self, cls, size, **kw):
        self.cls = cls
        self.size = size
        self.__dict__.update(kw)

    def __repr__(self):
        if self.cls == 6:
            return "compound(%d){%s}" % (self.size, ",".join(n for n, _, _ in self.members))
        return "dtype(cls=%d,size=%d)" % (self.cls, self.size)


class JLD2File:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        assert b[512:520] == b"\x89HDF\r\n\x1a\n", "not a JLD2/HDF5 file with 512-byte user block"
        ver = b[520]
        assert ver == 2 and b[521] == 8 and b[522] == 8
        self.base = struct.unpack_from("<Q", b, 524)[0]
        self.root = struct.unpack_from("<Q", b, 548)[0]
        self._dcache = {}
        self.types = {}
        self.links = self.read_group(self.root)

    # -- object headers ---------------------------------------------------------------
    def messages(self, rel):
        b = self.b
        off = rel + self.base
        assert b[off:off + 4] == b"OHDR", (hex(off), b[off:off + 4])
        flags = b[off + 5]
        p = off + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        w = 1 << (flags & 3)
        size = int.from_bytes(b[p:p + w], "little")
        p += w
        msgs = []
        chunks = [(p, p + size)]
        ci = 0
        while ci < len(chunks):
            p, end = chunks[ci]
            ci += 1
            while p + 4 <= end:
                mtype = b[p]
                msize = struct.unpack_from("<H", b, p + 1)[0]
                mflags = b[p + 3]
                p += 4
                if flags & 0x04:
                    p += 2
                data = b[p:p + msize]
                if mtype == 0x10:
                    coff, clen = struct.unpack_from("<QQ", data, 0)
                    a = coff + self.base
                    assert b[a:a + 4] == b"OCHK"
                    chunks.append((a + 4, a + clen - 4))
                elif mtype != 0:
                    msgs.append((mtype, mflags, data, p))
                p += msize
        return msgs

    def read_group(self, rel):
        out = {}
        for mtype, mflags, d, _ in self.messages(rel):
            if mtype == 0x06:
                lf = d[1]
                p = 2
                ltype = 0
                if lf & 0x08:
                    ltype = d[p]
                    p += 1
                if lf & 0x04:
                    p += 8
                if lf & 0x10:
                    p += 1
                w = 1 << (lf & 3)
                n = int.from_bytes(d[p:p + w], "little")
                p += w
                name = d[p:p + n].decode()
                p += n
                if ltype == 0:
                    out[name] = struct.unpack_from("<Q", d, p)[0]
        return out

    # -- datatypes ----------------------------------------------------------------------
    def parse_dtype(self, d, p=0):
        """returns (DType, bytes consumed)"""
        cv = d[p]
        cls = cv & 0x0F
        ver = cv >> 4
        bits = d[p + 1] | (d[p + 2] << 8) | (d[p + 3] << 16)
        size = struct.unpack_from("<I", d, p + 4)[0]
        q = p + 8
        if cls == 0:
            q += 4
            return DType(0, size, signed=bool(bits & 0x08)), q - p
        if cls == 1:
            q += 12
            return DType(1, size), q - p
        if cls == 3:
            return DType(3, size), q - p
        if cls == 4:
            q += 4
            return DType(4, size), q - p
        if cls == 5:
            taglen = bits & 0xFF
            q += (taglen + 7) // 8 * 8
            return DType(5, size), q - p
        if cls == 6:
            nm = bits & 0xFFFF
            members = []
            ow = 1 if size < 256 else 2 if size < 65536 else 4 if size < 2**32 else 8
            assert ver == 3, "compound version %d" % ver
            for _ in range(nm):
                e = d.index(b"\x00", q)
                name = d[q:e].decode()
                q = e + 1
                moff = int.from_bytes(d[q:q + ow], "little")
                q += ow
                mt, used = self.parse_dtype(d, q)
                q += used
                members.append((name, moff, mt))
            return DType(6, size, members=members), q - p
        if cls == 7:
            return DType(7, size), q - p
        if cls == 9:
            bt, used = self.parse_dtype(d, q)
            q += used
            return DType(9, size, base_type=bt, vtype=bits & 0x0F), q - p
        raise NotImplementedError("datatype class %d" % cls)

    def committed_dtype(self, rel):
        if rel in self.types:
            return self.types[rel]
        for mtype, mflags, d, _ in self.messages(rel):
            if mtype == 0x03:
                t, _ = self.parse_dtype(d)
                self.types[rel] = t
                return t
        raise ValueError("no datatype message at %x" % rel)

    # -- datasets -------------------------------------------------------------------------
    def dataset_info(self, rel):
        dims = None
        dt = None
        data = None
        for mtype, mflags, d, pos in self.messages(rel):
            if mtype == 0x01:
                rank = d[1]
                if d[0] == 2:
                    dims = [struct.unpack_from("<Q", d, 4 + 8 * i)[0] for i in range(rank)]
                    if d[3] == 2:      # null dataspace
                        dims = None
                        data = b""
                else:
                    dims = [struct.unpack_from("<Q", d, 8 + 8 * i)[0] for i in range(rank)]
            elif mtype == 0x03:
                if mflags & 0x02:
                    addr = struct.unpack_from("<Q", d, 2)[0]
                    dt = self.committed_dtype(addr)
                else:
                    dt, _ = self.parse_dtype(d)
            elif mtype == 0x08:
                ver, cls = d[0], d[1]
                assert ver in (3, 4)
                if cls == 0:
                    n = struct.unpack_from("<H", d, 2)[0]
                    data = (pos + 4, n)
                elif cls == 1:
                    addr, n = struct.unpack_from("<QQ", d, 2)
                    data = (None, 0) if addr == UNDEF else (addr + self.base, n)
                else:
                    raise NotImplementedError("chunked layout")
        return dims, dt, data

    def decode(self, dt, off, depth):
        b = self.b
        if dt.cls == 0:
            return int.from_bytes(b[off:off + dt.size], "little", signed=dt.signed)
        if dt.cls == 1:
            return struct.unpack_from("<d" if dt.size == 8 else "<f", b, off)[0]
        if dt.cls == 3:
            return b[off:off + dt.size].split(b"\x00")[0].decode(errors="replace")
        if dt.cls == 6:
            return {n: self.decode(mt, off + mo, depth) for n, mo, mt in dt.members}
        if dt.cls == 7:
            ref = struct.unpack_from("<Q", b, off)[0]
            if ref == 0 or ref == UNDEF:
                return None
            return self.read(ref, depth + 1)
        if dt.cls == 9:
            return "<vlen>"
        return None

    def read(self, rel, depth=0):
        if rel in self._dcache:
            return self._dcache[rel]
        if depth > 12:
            return "<deep>"
        try:
            dims, dt, data = self.dataset_info(rel)
        except Exception as e:  # committed type, group, or unsupported object
            return "<unreadable:%s>" % type(e).__name__
        if dt is None or data is None or data == b"":
            self._dcache[rel] = None
            return None
        off, n = data
        if off is None:
            self._dcache[rel] = None
            return None
        if dims is None or len(dims) == 0:
            val = self.decode(dt, off, depth)
        else:
            cnt = int(np.prod(dims)) if dims else 1
            if dt.cls == 1:
                val = np.frombuffer(self.b, dtype="<f8" if dt.size == 8 else "<f4", count=cnt, offset=off).reshape(dims)
            elif dt.cls == 0:
                val = np.frombuffer(self.b, dtype="<%s%d" % ("i" if dt.signed else "u", dt.size), count=cnt, offset=off).reshape(dims)
            else:
                val = [self.decode(dt, off + i * dt.size, depth) for i in range(cnt)]
        self._dcache[rel] = val
        return val

    def __getitem__(self, name):
        return self.read(self.links[name])

    def keys(self):
        return [k for k in self.links if k != "_types"]

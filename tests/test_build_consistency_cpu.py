"""Every kernel the host side of a built library registers exists in one of the gfx950 code objects embedded in that library.

Round 6 found `libudecore_ra2.so` linked from second-allocation objects that were compiled from the headers of five minutes earlier
(a header was edited while the minutes-long units were compiling; their mtimes were newer than the edit, so the incremental build
called them current): the host stubs of the newer `udecore.o` asked for `fwd_kernel<..., bool SORTED>` and the stale code objects
held the kernel without that parameter -- `hipLaunchKernel` aborted the process on the GPU box.  build.py now stamps an object with
the time its compile STARTED; this test is the check that needs no GPU: it reads the registered kernel names out of the host code
(`__hipRegisterFunction` receives the mangled device name as a string) and the symbol tables of the embedded code objects
(clang offload bundles in `.hip_fatbin`), for every library variant that has been built."""
import os
import re
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "universal_differential_equations_amd")
LIBS = ["libudecore.so", "libudecore_dbg.so", "libudecore_ra2.so", "libudecore_nrw.so"]
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def elf_sections(buf, base=0):
    """(name, offset, size, link, entsize) of every section of the ELF64 image that starts at buf[base]"""
    assert buf[base:base + 4] == b"\x7fELF" and buf[base + 4] == 2, "not an ELF64 image"
    shoff, = struct.unpack_from("<Q", buf, base + 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", buf, base + 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", buf, base + shoff + i * shentsize)
        raw.append((name, typ, off, size, link, entsize))
    stroff = raw[shstrndx][2]
    out = []
    for name, typ, off, size, link, entsize in raw:
        end = buf.index(b"\0", base + stroff + name)
        out.append((buf[base + stroff + name:end].decode(), typ, off, size, link, entsize))
    return out


def defined_functions(buf, base):
    """names of the defined FUNC symbols of the ELF64 image at buf[base] (.symtab, else .dynsym)"""
    secs = elf_sections(buf, base)
    names = set()
    for want in (2, 11):  # SHT_SYMTAB, SHT_DYNSYM
        for _name, typ, off, size, link, entsize in secs:
            if typ != want:
                continue
            stroff = secs[link][2]
            for i in range(size // entsize):
                st_name, st_info, _other, shndx, _value, _size = struct.unpack_from("<IBBHQQ", buf, base + off + i * entsize)
                if (st_info & 0xF) == 2 and shndx != 0:
                    end = buf.index(b"\0", base + stroff + st_name)
                    names.add(buf[base + stroff + st_name:end].decode())
        if names:
            break
    return names


def device_kernels(buf, off, size):
    """union of the function symbols of every gfx950 code object in the `.hip_fatbin` bytes buf[off : off + size]"""
    out = set()
    n_objects = 0
    pos = buf.find(MAGIC, off, off + size)
    while pos != -1:
        n, = struct.unpack_from("<Q", buf, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", buf, p)
            triple = buf[p + 24:p + 24 + tsize].decode()
            p += 24 + tsize
            if "gfx950" in triple and esize:
                out |= defined_functions(buf, pos + eoff)
                n_objects += 1
        pos = buf.find(MAGIC, pos + len(MAGIC), off + size)
    return out, n_objects


def registered_kernels(buf, skip):
    """the device names handed to __hipRegisterFunction: NUL-terminated mangled names of ude:: kernels in the host image (outside `skip`)"""
    names = set()
    for m in re.finditer(rb"_ZN3ude[0-9A-Za-z_]*_kernelI[0-9A-Za-z_]+\0", buf):
        if skip[0] <= m.start() < skip[1]:
            continue
        names.add(m.group()[:-1].decode())
    return {n for n in names if "__device_stub__" not in n}


@pytest.mark.parametrize("lib", LIBS)
def test_every_registered_kernel_has_a_code_object(lib):
    path = os.path.join(PKG, lib)
    if not os.path.exists(path):
        pytest.skip("%s has not been built" % lib)
    buf = open(path, "rb").read()
    fat = [s for s in elf_sections(buf) if s[0] == ".hip_fatbin"]
    assert len(fat) == 1
    _, _, off, size, _, _ = fat[0]
    dev, n_objects = device_kernels(buf, off, size)
    host = registered_kernels(buf, (off, off + size))
    assert n_objects >= 50 and len(host) >= 300, (n_objects, len(host))   # (the parser saw the library: ~100 translation units, ~600 kernels)
    missing = sorted(host - dev)
    assert not missing, "%s: %d registered kernel(s) without device code (stale object?), e.g. %s" % (lib, len(missing), missing[:3])

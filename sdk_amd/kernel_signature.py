"""Identity of a compiled device kernel: SHA-256 of its machine code as it sits in a built libspiral_hip.so.

Why: bench.py cannot collect PMC counters itself, so `roofline.traffic` is replayed from a record made with rocprofv3 on the
builder's box (profiles/*pmc_sweep_c2.json).  That is only honest while the kernel that was profiled is the kernel that runs.
The record stores this signature; bench.py compares it with the library it has loaded and refuses a record that does not match
(VERDICT r04 item 8).

The .so carries one clang offload bundle per translation unit in its `.hip_fatbin` section (uncompressed); each holds the gfx950
code object, an ELF whose .symtab names every kernel.  Branches inside a kernel are relative and the sweep kernels call nothing,
so the bytes of the function do not depend on where it was linked.  Pure Python, no GPU, no ROCm tools."""
import hashlib
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    pos = 0
    while True:
        i = blob.find(_MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", blob, i + len(_MAGIC))
        p = i + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode("ascii", "replace")
            p += 24 + tl
            if "amdgcn" in triple and size:
                yield triple, blob[i + off:i + off + size]
        pos = i + len(_MAGIC)


def _elf_functions(elf):
    """{symbol name: bytes} for the FUNC symbols of a 64-bit little-endian ELF"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2 or elf[5] != 1:
        return {}
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]
    out = {}
    for (_, typ, _, _, off, size, link, _, _, entsize) in secs:
        if typ != 2 or not entsize:      # SHT_SYMTAB
            continue
        stroff = secs[link][4]
        for k in range(size // entsize):
            name_i, info, _, shndx, value, sz = struct.unpack_from("<IBBHQQ", elf, off + k * entsize)
            if (info & 0xF) != 2 or not sz or shndx == 0 or shndx >= shnum:   # STT_FUNC, defined
                continue
            end = elf.index(b"\0", stroff + name_i)
            name = elf[stroff + name_i:end].decode("ascii", "replace")
            _, _, _, saddr, soff, _, _, _, _, _ = secs[shndx]
            start = soff + (value - saddr)
            out[name] = elf[start:start + sz]
    return out


def kernel_signature(so_path, name_fragment):
    """(signature, mangled name) of the one gfx950 kernel whose mangled name contains `name_fragment`; raises if there is none or
    more than one distinct body."""
    blob = open(so_path, "rb").read()
    found = {}
    for triple, elf in _code_objects(blob):
        if "gfx950" not in triple:
            continue
        for name, body in _elf_functions(elf).items():
            if name_fragment in name:
                found[name] = hashlib.sha256(body).hexdigest()[:32]
    if len(found) != 1:
        raise LookupError("%d gfx950 kernels match %r in %s: %s" % (len(found), name_fragment, so_path, sorted(found)))
    (name, sig), = found.items()
    return sig, name


def compiler_version():
    """first two lines of `hipcc --version` (HIP + clang build), or "" -- stored beside a record's signature: machine code compiled
    by another hipcc differs without the kernel's SOURCE having changed, which a reader of a refused record wants to know"""
    import subprocess
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
        return " | ".join(line.strip() for line in out.splitlines()[:2])
    except Exception:
        return ""


SWEEP_C2 = "k_sweep_packed_ringILi8E"   # the judged kernel: the ring-form PACKED sweep with buffers of 8 row pairs


if __name__ == "__main__":
    import os
    import sys
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspiral_hip.so")
    print(*kernel_signature(so, sys.argv[2] if len(sys.argv) > 2 else SWEEP_C2))

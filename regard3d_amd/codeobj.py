"""Machine-code fingerprints of the kernels inside a built library (bench tooling).

`kernel_code_hashes(path)` walks the clang offload bundles of an ELF shared object (one per translation unit, magic
`__CLANG_OFFLOAD_BUNDLE__`), opens each gfx950 code object and returns {demangled-ish kernel symbol: sha256[:16] of the bytes
of that function in .text}.  bench.py keys counter measurements taken in ANOTHER process (PMC traffic: rocprofv3 cannot run
inside the timed process) to the kernel they were taken on: an entry is reported only while the kernel's code is the code that
was profiled -- an edit elsewhere in the same source file does not stale it, a compiler or source change of the kernel does.
"""
import hashlib
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob: bytes):
    at = 0
    while True:
        at = blob.find(_MAGIC, at)
        if at < 0:
            return
        n = struct.unpack_from("<Q", blob, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode("ascii", "replace")
            p += 24 + tl
            if triple.startswith("hip") and "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at += 24


def _functions(elf: bytes):
    """(name, bytes) of every FUNC symbol of an ELF64 little-endian code object"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    for name_off, typ, flags, addr, off, size, link, info, align, entsize in secs:
        if typ != 2:           # SHT_SYMTAB
            continue
        stroff = secs[link][4]
        for k in range(size // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + k * 24)
            if (st_info & 0xF) != 2 or not st_size or st_shndx == 0 or st_shndx >= shnum:      # STT_FUNC, defined
                continue
            end = elf.index(b"\0", stroff + st_name)
            sym = elf[stroff + st_name:end].decode("ascii", "replace")
            s = secs[st_shndx]
            fo = s[4] + (st_value - s[3])
            yield sym, elf[fo:fo + st_size]


def kernel_code_hashes(path: str) -> dict:
    blob = open(path, "rb").read()
    out = {}
    for co in _code_objects(blob):
        for sym, code in _functions(co):
            out[sym] = hashlib.sha256(code).hexdigest()[:16]
    return out


def kernel_hash(path: str, *needles: str):
    """sha16 over the code of every kernel whose mangled name contains all `needles` (sorted by name); None if none does"""
    hs = sorted((k, v) for k, v in kernel_code_hashes(path).items() if all(n in k for n in needles))
    if not hs:
        return None
    return hashlib.sha256("".join(k + v for k, v in hs).encode()).hexdigest()[:16]


def mangled_needle(kernel: str) -> str:
    """'l2_knn2_mfma_kernel<16, 2, 4, 3, 2>' -> 'l2_knn2_mfma_kernelILi16ELi2ELi4ELi3ELi2EE' (integer template arguments only)"""
    base, _, targs = kernel.replace("r3dm::", "").partition("<")
    if not targs:
        return base
    args = [a.strip() for a in targs.rstrip(">").split(",")]
    return base + "I" + "".join(f"Li{a}E" if not a.startswith("-") else f"Lin{a[1:]}E" for a in args) + "E"


if __name__ == "__main__":
    import sys
    for k, v in sorted(kernel_code_hashes(sys.argv[1]).items()):
        if len(sys.argv) < 3 or all(n in k for n in sys.argv[2:]):
            print(v, k)

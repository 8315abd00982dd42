"""Disassembly of one gfx950 kernel of a built object or library (no GPU needed):
   python tools/kernel_disasm.py build/product/kernels_match_16bit.o 'l2_knn2_counts3_kernelILi9ELi9E' > /tmp/k.s
Walks the clang offload bundles like regard3d_amd/codeobj.py, runs llvm-objdump -d on the code object that holds a symbol containing
the given (mangled) needle and prints that symbol's instructions; a histogram of the mnemonics goes to stderr."""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regard3d_amd import codeobj

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    blob = open(path, "rb").read()
    for co in codeobj._code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        m = re.search(r"^[0-9a-f]+ <([^>]*" + re.escape(needle) + r"[^>]*)>:\n(.*?)(?=^\S|\Z)", txt, re.S | re.M)
        if not m:
            continue
        body = m.group(2)
        print(f"; {m.group(1)}")
        print(body)
        hist = collections.Counter(l.split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith(("//", ";")) and not l.rstrip().endswith(":"))
        tot = sum(hist.values())
        print(f"; {tot} instructions", file=sys.stderr)
        for k, v in hist.most_common(40):
            print(f";   {k:32s} {v}", file=sys.stderr)
        return
    sys.exit("no such kernel")


if __name__ == "__main__":
    main()

"""Static check of the built gfx950 code: every v_permlane32_swap_b32 must be two wait states away from a VALU write of either of its
operands (the instructions are issued from inline assembly, where the compiler's hazard recogniser cannot insert them: kernels_match_16bit.hip,
swap_lane_halves*).  Walks every kernel of an object / library; exit code 1 and a listing if a swap is too close to such a write.
   python tools/check_swap_hazard.py build/product/kernels_match_16bit.o"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regard3d_amd import codeobj

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def regs(op):
    """VGPR numbers named by an operand: v7, v[4:7]; anything else -> empty"""
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(path):
    blob = open(path, "rb").read()
    bad, n_swaps = [], 0
    for co in codeobj._code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        fn, ins = "?", []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                fn, ins = m.group(1), []
                continue
            t = line.split("//")[0].strip()
            if not t:
                continue
            parts = t.replace(",", " ").split()
            ins.append(parts)
            if parts[0].startswith("v_permlane32_swap"):
                n_swaps += 1
                mine = regs(parts[1]) | regs(parts[2])
                wait, j = 0, len(ins) - 2
                while wait < 2 and j >= 0:
                    p = ins[j]
                    if p[0] == "s_nop":
                        wait += int(p[1], 0) + 1
                    else:
                        if p[0].startswith("v_") and len(p) > 1:
                            dst = regs(p[1]) | (regs(p[2]) if p[0].startswith("v_permlane32_swap") and len(p) > 2 else set())
                            if dst & mine:
                                bad.append((fn, " ".join(p), " ".join(parts)))
                        wait += 1
                    j -= 1
    return n_swaps, bad


if __name__ == "__main__":
    n, bad = check(sys.argv[1])
    print(f"{n} swaps checked, {len(bad)} too close to a VALU write of an operand")
    for b in bad:
        print("  ", b)
    sys.exit(1 if bad else 0)

"""Register / LDS / scratch use of the gfx950 kernels inside a built object or library (no GPU needed):
   python tools/kernel_resources.py build/product/kernels_liop.o [name-filter]
Walks the clang offload bundles like regard3d_amd/codeobj.py and prints the AMDGPU metadata notes of every kernel
(llvm-readelf --notes): VGPRs (+ AGPRs), SGPRs, spilled registers, LDS bytes, scratch bytes."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regard3d_amd import codeobj

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blob = open(path, "rb").read()
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'lds':>7s} {'scratch':>7s}")
    for co in codeobj._code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            except Exception:
                pass
            name = re.sub(r"\(.*", "", name).replace("void r3dm::", "").replace("r3dm::", "")
            if flt and flt not in name:
                continue
            print(f"{name[:70]:70s} {g('vgpr_count'):>5s} {g('agpr_count'):>5s} {g('sgpr_count'):>5s} {g('vgpr_spill_count'):>6s} {g('sgpr_spill_count'):>6s} "
                  f"{g('group_segment_fixed_size'):>7s} {g('private_segment_fixed_size'):>7s}")


if __name__ == "__main__":
    main()

"""profiles/<tag>_pmc_<config>.txt (FETCH_SIZE / WRITE_SIZE passes of `python bench.py --config <config> --steps 1 --warmup 0
--no-cpu-baseline`, summarised by tools/pmc_summary.py) -> profiles/pmc_traffic.json, the file bench.py reads `roofline.traffic` from.
Every entry carries code_sha16 = the fingerprint of the kernel's MACHINE CODE in regard3d_amd/libr3dm.so at the time of the PMC run
(regard3d_amd/codeobj.py): bench.py reports an entry only while the library it runs holds that very code.
Usage: python tools/pmc_traffic_json.py profiles/r04_pmc_bench_c2.txt [config = c2] [kernel_base=algorithmic_bytes_per_launch ...]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from regard3d_amd.codeobj import kernel_hash, mangled_needle
LIB = os.path.join(ROOT, "regard3d_amd", "libr3dm.so")
KERNELS = "l2_knn2_mfma_kernel|l2_knn2_int_kernel|l2_knn2_int_lds_kernel|l2_knn2_split_kernel|l2_knn2_counts2_kernel|l2_knn2_counts_kernel|hamming_knn2_kernel|hamming_knn2_mfma_kernel"
config = sys.argv[2] if len(sys.argv) > 2 else "c2"
txt = open(sys.argv[1]).read()
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for m in re.finditer(r"(\S*(" + KERNELS + r")<[^>]*>) .*?\b" + ctr + r"=([0-9.e+]+)", txt):
        vals.setdefault(m.group(2), {"kernel": m.group(1).replace("r3dm::", "")})[ctr] = float(m.group(3))
ALG = {"c2": {"l2_knn2_mfma_kernel": (169548000000.0, "both f32 descriptor sets + 16 B of results per query, per pair (8.52 MB x 19,900)"),
              "l2_knn2_int_kernel": (85422899200.0, "opt-in integer path: both bf16 tile sets + norms + one result word per query, per pair; measured below it "
                                     "when consecutive pairs share their dataset image in L2 / Infinity Cache"),
              "l2_knn2_int_lds_kernel": (85422899200.0, "opt-in integer path (workgroup-shared tiles)")}}.get(config, {})
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ALG[k] = (float(v), "given on the command line (bench.py: roofline.algorithmic_bytes_per_launch of the same run)")
out = {"_comment": "HBM-side traffic of ONE launch of the dominant kernel per bench config, keyed '<config>:<kernel>'.  rocprofv3 reports FETCH_SIZE / "
       "WRITE_SIZE in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for wide coalesced "
       "16 B/lane streams -> read side doubled; WRITE_SIZE is uncalibrated on gfx950 and tiny here.  bench.py reports an entry only while "
       "code_sha16 equals the fingerprint of that kernel's machine code in the library it runs (regard3d_amd/codeobj.py)."}
for k, v in vals.items():
    if "FETCH_SIZE" not in v:
        continue
    rd = v["FETCH_SIZE"] * 1024 * 2; wr = v.get("WRITE_SIZE", 0.0) * 1024
    ent = {"kernel": v["kernel"], "from": os.path.relpath(sys.argv[1], ROOT), "code_sha16": kernel_hash(LIB, mangled_needle(v["kernel"])),
           "fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v.get("WRITE_SIZE", 0.0), "read_bytes_corrected": rd, "write_bytes": wr,
           "traffic_bytes_per_launch": rd + wr}
    if k in ALG:
        ent["algorithmic_bytes_per_launch"] = ALG[k][0]; ent["note"] = "algorithmic = " + ALG[k][1]
    out[f"{config}:{k}"] = ent
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
if os.path.exists(dst):                      # entries of other configs / kernels stay
    old = json.load(open(dst))
    old.update(out)
    out = old
json.dump(out, open(dst, "w"), indent=1)
print({k: (round(v["traffic_bytes_per_launch"] / 1e9, 1), v.get("code_sha16")) for k, v in out.items() if k.startswith(config + ":")})

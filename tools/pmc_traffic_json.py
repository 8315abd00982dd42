"""profiles/<tag>_pmc_bench_c2.txt (FETCH_SIZE / WRITE_SIZE passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`,
summarised by tools/pmc_summary.py) -> profiles/r02_pmc_traffic.json, the file bench.py reads `roofline.traffic` from.
Usage: python tools/pmc_traffic_json.py profiles/r02_x_pmc_bench_c2.txt"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "regard3d_amd", "csrc", "kernels_match.hip")
sha16 = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
txt = open(sys.argv[1]).read()
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for m in re.finditer(r"(\S*(l2_knn2_mfma_kernel|l2_knn2_int_kernel|l2_knn2_int_lds_kernel)<[^>]*>) " + ctr + r"=([0-9.e+]+)", txt):
        vals.setdefault(m.group(2), {"kernel": m.group(1).replace("r3dm::", "")})[ctr] = float(m.group(3))
ALG = {"l2_knn2_mfma_kernel": (169548000000.0, "both f32 descriptor sets + 16 B of results per query, per pair (8.52 MB x 19,900)"),
       "l2_knn2_int_kernel": (85422899200.0, "opt-in integer path: both bf16 tile sets + norms + one result word per query, per pair; measured below it "
                              "when consecutive pairs share their dataset image in L2 / Infinity Cache"),
       "l2_knn2_int_lds_kernel": (85422899200.0, "opt-in integer path (workgroup-shared tiles)")}
out = {"_comment": "HBM-side traffic of ONE launch of the dominant kernel per bench config, keyed '<config>:<kernel>'.  rocprofv3 reports FETCH_SIZE / "
       "WRITE_SIZE in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for wide coalesced "
       "16 B/lane streams -> read side doubled; WRITE_SIZE is uncalibrated on gfx950 and tiny here.  bench.py reports an entry only while "
       "source_sha16 matches the current kernel source."}
for k, v in vals.items():
    rd = v["FETCH_SIZE"] * 1024 * 2; wr = v.get("WRITE_SIZE", 0.0) * 1024
    out[f"c2:{k}"] = {"kernel": v["kernel"], "from": os.path.relpath(sys.argv[1], ROOT), "source": "kernels_match.hip", "source_sha16": sha16,
                      "fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v.get("WRITE_SIZE", 0.0), "read_bytes_corrected": rd, "write_bytes": wr,
                      "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": ALG[k][0], "note": "algorithmic = " + ALG[k][1]}
dst = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
if os.path.exists(dst):                      # entries of other configs / kernels (c5: kernels_ann.hip) stay
    old = json.load(open(dst))
    old.update(out)
    out = old
json.dump(out, open(dst, "w"), indent=1)
print({k: round(v["traffic_bytes_per_launch"] / 1e9, 1) for k, v in out.items() if k != "_comment" and "traffic_bytes_per_launch" in v})

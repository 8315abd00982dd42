"""Derived figures from a tools/pmc_summary.py digest (profiles/*_pmc_*.txt): per kernel, the ratios DESIGN.md quotes.
  python tools/pmc_derive.py profiles/r04_final_pmc_c2.txt
GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 -> cycles = GUI / 8; 256 CUs, 1,024 SIMDs, one TA per CU;
SQ_WAVE_CYCLES counts in units of 4 cycles (guide, profiling section)."""
import re, sys, collections
vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(?:void )?(?:r3dm::)?(\S+?)(?:\(|\s)", line)
    if not m or line.startswith("#"):
        continue
    name = line.split(" ")[1] if line.startswith("void ") else line.split(" ")[0]
    name = name.replace("r3dm::", "")
    if line.startswith("void "):
        name = line[5:line.index(">") + 1].replace("r3dm::", "") if ">" in line else name
    for c, v in re.findall(r"([A-Za-z_0-9]+)=([0-9.e+]+)\(n=", line):
        vals[name][c] = float(v)
for k, v in vals.items():
    if "GRBM_GUI_ACTIVE" not in v:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    out = [f"{k}: {cyc / 2.4e9 * 1e3:.2f} ms at 2.4 GHz"]
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES"): out.append(f"matrix pipe busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}")
    if v.get("SQ_BUSY_CU_CYCLES"): out.append(f"CUs busy {v['SQ_BUSY_CU_CYCLES'] / (256 * cyc):.3f}")
    if v.get("SQ_WAVE_CYCLES"): out.append(f"wavefronts resident per CU {v['SQ_WAVE_CYCLES'] * 4 / (256 * cyc):.2f}")
    if v.get("TA_TA_BUSY_sum"): out.append(f"TA busy {v['TA_TA_BUSY_sum'] / (256 * cyc):.3f}")
    if v.get("SQ_INSTS_VALU") and v.get("SQ_WAVES"): out.append(f"VALU instructions per wavefront {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f}")
    if v.get("SQ_INSTS_VALU"): out.append(f"VALU issue {v['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.3f} of one per SIMD and 4 cycles")
    if v.get("FETCH_SIZE"): out.append(f"HBM-side traffic {(v['FETCH_SIZE'] * 2 + v.get('WRITE_SIZE', 0)) * 1024 / 1e9:.1f} GB (FETCH_SIZE x2 + WRITE_SIZE)")
    print("; ".join(out))

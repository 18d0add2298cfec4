"""MFMA utilisation per kernel from the PMC passes of tools/pmc_iter.sh (run in the container on the merged csv files).

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x clock x 1024 SIMDs): the counter adds, over the chip's
256 CUs x 4 SIMDs, the cycles in which a SIMD's matrix pipe is busy (MI355X_MICROARCH.md: 32 per v_mfma_f32_32x32x16_bf16).
Two clocks are reported: the 2.4 GHz of the 2.5 PFLOP/s peak figure (a lower bound on the pipe's share of its own
time) and the clock GRBM_GUI_ACTIVE / 8 / duration gives for that kernel (reads high on dispatches under ~0.3 ms).
usage: python tools/mfma_util.py TAG [out.json]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in sorted(glob.glob(f"gpurun_out/pmc_{tag}/pass*.csv")):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", ""))[:70]
        c = r["Counter_Name"]
        agg[name][c] += float(r["Counter_Value"])
        agg[name][c + "::ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        calls[name][c] += 1
rows = []
for name, a in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in a:
        continue
    ns = a["SQ_VALU_MFMA_BUSY_CYCLES::ns"]
    n = calls[name]["SQ_VALU_MFMA_BUSY_CYCLES"]
    row = {"kernel": name, "dispatches": n, "avg_us": ns / n / 1e3,
           "mfma_busy_cycles_per_dispatch": a["SQ_VALU_MFMA_BUSY_CYCLES"] / n,
           "mfma_instructions_per_dispatch": a.get("SQ_INSTS_MFMA", 0.0) / max(1, calls[name].get("SQ_INSTS_MFMA", 1)),
           "mfma_util_at_2.4GHz": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns * 1e-9 * 2.4e9 * 1024)}
    if "GRBM_GUI_ACTIVE" in a:
        cycles = a["GRBM_GUI_ACTIVE"] / 8.0
        row["clock_GHz_from_GRBM"] = cycles / (a["GRBM_GUI_ACTIVE::ns"])
        row["mfma_util_at_measured_clock"] = (a["SQ_VALU_MFMA_BUSY_CYCLES"] / n) / ((cycles / calls[name]["GRBM_GUI_ACTIVE"]) * 1024)
    rows.append(row)
rows.sort(key=lambda r: -r["avg_us"] * r["dispatches"])
out = {"source": f"gpurun_out/pmc_{tag} (rocprofv3 --pmc, separate passes; eager iterations of tools/iter_profile.py)",
       "kernels": rows}
text = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
for r in rows[:25]:
    print(f"{r['kernel'][:60]:60s} {r['avg_us']:8.1f} us x{r['dispatches']:3d}  MFMA util {100 * r['mfma_util_at_2.4GHz']:5.1f} % "
          f"(at measured clock {100 * r.get('mfma_util_at_measured_clock', float('nan')):5.1f} %)")

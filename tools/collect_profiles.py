"""Copy the rocprofv3 summaries of a gpurun session from gpurun_out/ (scratch) into profiles/ (tracked).
usage: python tools/collect_profiles.py TAG [bench_json]"""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
bench_json = Path(sys.argv[2]) if len(sys.argv) > 2 else None
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
src_prof = ROOT / "gpurun_out" / f"prof_{tag}"
src_pmc = ROOT / "gpurun_out" / f"pmc_{tag}"

stats = src_prof / "bench_kernel_stats.csv"
if stats.exists():
    rows = list(csv.DictReader(open(stats)))
    keep = rows[:60] + [r for r in rows[60:] if "seld::" in r["Name"]]
    with open(out / f"{tag}_bench_kernel_stats.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    for name in ("bench_stdout.log",):
        if (src_prof / name).exists():
            text = [l for l in open(src_prof / name) if l.startswith("{")]
            (out / f"{tag}_bench_under_rocprof.json").write_text("".join(text))
timed = src_prof / "bench_timed_region_stats.csv"
if timed.exists():                      # the timed region only (MIOpen find-mode trials of the warm-up excluded)
    shutil.copy(timed, out / f"{tag}_bench_timed_region_stats.csv")
    if (src_prof / "timed_region.log").exists():
        shutil.copy(src_prof / "timed_region.log", out / f"{tag}_bench_timed_region.txt")
timeline = src_prof / "iteration_timeline.txt"
if timeline.exists():                   # every dispatch of one optimiser iteration (tools/iter_timeline.py)
    (out / f"{tag}_iteration_timeline.txt").write_text("".join(l[:170].rstrip() + "\n" for l in open(timeline)))
if src_pmc.exists():
    for p in sorted(src_pmc.glob("pass*.csv")):
        shutil.copy(p, out / f"{tag}_pmc_logmel_{p.name}")
if bench_json and bench_json.exists():
    shutil.copy(bench_json, out / f"{tag}_bench.json")
# the other workloads of tools/round_artifacts.sh part b: bench lines and the timed-region statistics of their profiled runs
# PMC passes of other micro-benchmarks (tools/pmc_kernel.sh TAG_name ...): the rows kept there + a per-kernel summary with
# the HBM traffic (FETCH_SIZE KiB x 2 -- the guide's gfx950 correction for coalesced streams -- and WRITE_SIZE KiB)
for pmc_dir in sorted((ROOT / "gpurun_out").glob(f"pmc_{tag}_*")):
    name = pmc_dir.name[len(f"pmc_{tag}_"):]
    kernels = {}
    for p in sorted(pmc_dir.glob("pass*.csv")):
        shutil.copy(p, out / f"{tag}_pmc_{name}_{p.name}")
        rows = list(csv.DictReader(open(p)))
        for kn in sorted(set(r["Kernel_Name"] for r in rows)):
            kr = [r for r in rows if r["Kernel_Name"] == kn]
            last = kr[-1]["Dispatch_Id"]
            k = kernels.setdefault(kn[:120], {})
            for r in kr:
                if r["Dispatch_Id"] == last:
                    k[r["Counter_Name"]] = float(r["Counter_Value"])
                    k.setdefault("dur_us_" + p.stem, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in kernels.values():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["hbm_read_bytes_corrected"] = k["FETCH_SIZE"] * 1024 * 2
            k["hbm_write_bytes"] = k["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k and k["GRBM_GUI_ACTIVE"]:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs (r02's mfma_util.py: duration x 2.4 GHz x 1024)
            k["mfma_pipe_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if kernels:
        (out / f"{tag}_pmc_{name}_summary.json").write_text(json.dumps(kernels, indent=1))

for extra in ("eager", "conformer", "resnet_conformer", "mic8_gcc", "rehearsal_2ranks"):
    src = ROOT / "gpurun_out" / f"bench_{tag}_{extra}.json"
    if src.exists():
        lines = [l for l in open(src) if l.startswith("{")]
        (out / f"{tag}_bench_{extra}.json").write_text("".join(lines))
    prof = ROOT / "gpurun_out" / f"prof_{tag}_{extra}"
    if (prof / "bench_timed_region_stats.csv").exists():
        shutil.copy(prof / "bench_timed_region_stats.csv", out / f"{tag}_{extra}_timed_region_stats.csv")
        if (prof / "timed_region.log").exists():
            shutil.copy(prof / "timed_region.log", out / f"{tag}_{extra}_timed_region.txt")
        if (prof / "iteration_timeline.txt").exists():
            (out / f"{tag}_{extra}_iteration_timeline.txt").write_text(
                "".join(l[:170].rstrip() + "\n" for l in open(prof / "iteration_timeline.txt")))

# traffic of the log-mel main kernel from the PMC passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads
# half of a coalesced stream on gfx950 -> x2; WRITE_SIZE exact; both in KiB)
summary = {}
for p in sorted(src_pmc.glob("pass*.csv")) if src_pmc.exists() else []:
    for r in csv.DictReader(open(p)):
        if "logmel_main" in r["Kernel_Name"]:
            summary[r["Counter_Name"]] = float(r["Counter_Value"])
            summary.setdefault("dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    clips = 32
    summary["hbm_read_bytes_corrected"] = summary["FETCH_SIZE"] * 1024 * 2
    summary["hbm_write_bytes"] = summary["WRITE_SIZE"] * 1024
    summary["traffic_bytes_per_clip"] = (summary["hbm_read_bytes_corrected"] + summary["hbm_write_bytes"]) / clips
    summary["algorithmic_bytes_per_clip"] = 26112896
    (out / f"{tag}_pmc_logmel_summary.json").write_text(json.dumps(summary, indent=1))
    print(json.dumps(summary, indent=1))
print(sorted(p.name for p in out.iterdir()))

"""Summarise gpurun_out/pmc_<tag>/pass*.csv (last dispatch of every kernel)."""
import csv, sys, glob
tag = sys.argv[1]
for path in sorted(glob.glob(f"gpurun_out/pmc_{tag}/pass*.csv")):
    rows = list(csv.DictReader(open(path)))
    for kn in sorted(set(r["Kernel_Name"] for r in rows)):
        kr = [r for r in rows if r["Kernel_Name"] == kn]
        disp = kr[-1]["Dispatch_Id"]
        meta = kr[-1]
        dur = (int(meta["End_Timestamp"]) - int(meta["Start_Timestamp"])) / 1e3
        print(f"{path.split('/')[-1]} {kn[:60]} grid={meta['Grid_Size']} wg={meta['Workgroup_Size']} vgpr={meta['VGPR_Count']} dur={dur:.1f}us")
        for r in kr:
            if r["Dispatch_Id"] == disp:
                print(f"     {r['Counter_Name']:30s} {float(r['Counter_Value']):16.0f}")

"""Per-kernel statistics of the TIMED region of a bench.py run from a rocprofv3 kernel trace (run on the GPU box).

rocprofv3 --stats aggregates the whole process, warm-up included -- and the warm-up of a PyTorch-ROCm training step
contains MIOpen's find-mode trial kernels, which swamp the summary.  The timed region is recognised from the trace
itself: bench.py launches seld::logmel_main_kernel exactly once per step, so with W warm-up steps the region starts
at the (W+1)-th such dispatch.

usage: python tools/trace_timed_region.py kernel_trace.csv WARMUP out_stats.csv [iterations_per_step]"""
import csv
import sys
from collections import defaultdict

trace, warmup, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
iters_per_step = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rows = []
with open(trace, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [s for s, _, n in rows if "logmel_main_kernel" in n]
if len(marks) <= warmup:
    raise SystemExit(f"only {len(marks)} logmel_main dispatches in the trace, warm-up {warmup}")
t0 = marks[warmup]
steps = len(marks) - warmup
agg = defaultdict(lambda: [0, 0, 1 << 62, 0])
busy = 0
for s, e, n in rows:
    if s < t0:
        continue
    a = agg[n]
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
    busy += d
span = max(e for _, e, _ in rows) - t0
total = sum(a[1] for a in agg.values())
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsPerIteration",
                "UsPerIteration"])
    n_iter = steps * iters_per_step
    for name, (calls, dur, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name[:200], calls, dur, f"{dur / calls:.1f}", f"{100.0 * dur / total:.3f}", mn, mx,
                    f"{calls / n_iter:.2f}", f"{dur / n_iter / 1e3:.2f}"])
print(f"timed region: {steps} steps, {span / 1e6:.2f} ms wall, {total / 1e6:.2f} ms of kernel time "
      f"({100.0 * total / span:.1f}% busy), {len(agg)} distinct kernels, {total / steps / iters_per_step / 1e3:.1f} us "
      f"of kernels per optimiser iteration")

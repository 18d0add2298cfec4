"""Timeline of ONE optimiser iteration from a rocprofv3 kernel trace of bench.py (run on the GPU box).

The iteration is the span between two consecutive seld::softmax_mse_kernel dispatches in the middle of the timed
region.  Prints every dispatch with its offset, duration, the idle gap before it (time during which NO kernel of the
iteration was running) and the number of kernels running beside it, then the totals: wall, union-busy, summed kernel
time, idle.  usage: python tools/iter_timeline.py kernel_trace.csv [which_iteration_from_the_end=40]"""
import csv
import sys

trace = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
with open(trace, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "softmax_mse_kernel" in r[2]]
lo, hi = marks[-back - 1], marks[-back]
it = rows[lo:hi]
t0 = it[0][0]
end_so_far = t0
busy = 0
total = 0
print(f"{'t_us':>9} {'dur_us':>8} {'idle_before':>11} {'conc':>4} queue  kernel")
for k, (s, e, name, q) in enumerate(it):
    gap = max(0, s - end_so_far)
    conc = sum(1 for (s2, e2, _, _) in it if s2 < e and e2 > s) - 1
    short = name.replace("void ", "")[:110]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:11.1f} {conc:4d} {q:>5}  {short}")
    if e > end_so_far:
        busy += e - max(s, end_so_far)
        end_so_far = e
    total += e - s
wall = rows[hi][0] - t0
print(f"iteration: {len(it)} dispatches, wall {wall / 1e3:.1f} us, union-busy {busy / 1e3:.1f} us, "
      f"summed kernel time {total / 1e3:.1f} us, idle {(wall - busy) / 1e3:.1f} us")

"""Developer tool: from a rocprofv3 kernel trace of tools/iter_profile.py, print what ran beside each BiGRU backward
recurrence of the last iterations (name, start relative to the recurrence's start, duration, queue), and the
recurrence durations with / without company.   usage: overlap_timeline.py kernel_trace.csv [iterations]"""
import csv
import sys

trace = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
with open(trace, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
rec = [r for r in rows if "gru_backward_kernel" in r[2]]
durs = [(e - s) / 1e3 for s, e, _, _ in rec]
print(f"{len(rec)} backward recurrences; duration us: min {min(durs):.0f} median {sorted(durs)[len(durs) // 2]:.0f} "
      f"max {max(durs):.0f}")
for s, e, name, q in rec[-2 * last:]:
    print(f"\nrecurrence queue {q}: {(e - s) / 1e3:.0f} us")
    for s2, e2, n2, q2 in rows:
        if s2 < e + 150000 and e2 > s - 30000 and (s2, e2, n2) != (s, e, name):
            print(f"   {(s2 - s) / 1e3:8.1f} +{(e2 - s2) / 1e3:7.1f} us  q{q2}  {n2[:90]}")

"""Timeline of one batch in a rocprofv3 --kernel-trace CSV: start, gap to the
previous kernel, duration and name of every kernel between two consecutive
launches of the first pass (k_mum_first) -- the shortest such period of the
trace, i.e. a step of the timed region and not one with a host phase inside
(the per-step breakdown quoted in DESIGN.md)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_mum_first" in r["Kernel_Name"]]
pairs = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]),
          a, b) for a, b in zip(first, first[1:])
         if rows[a]["Kernel_Name"] == rows[b]["Kernel_Name"]]
if pairs:
    _, a, b = min(pairs)
else:
    a, b = (first[-1] if first else 0), len(rows) - 1
t0 = int(rows[a]["Start_Timestamp"])
prev = None
busy = gaps = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    if r is not rows[b]:
        busy += (e - s) / 1e3
    gaps += max(gap, 0.0)
    print("%9.1f gap %7.1f dur %8.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:80]))
    prev = e
print("one step: %.1f us from first pass to first pass; kernels %.1f us, gaps "
      "%.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy, gaps))

"""Timeline of the last batch in a rocprofv3 --kernel-trace CSV: start, gap
to the previous kernel, duration and name of every kernel between the
last two k_mum_first launches (the per-step breakdown quoted in DESIGN.md)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_mum_first" in r["Kernel_Name"]]
# one whole period: from the second-to-last k_mum_first up to the last one
a = first[-2] if len(first) > 1 else (first[-1] if first else 0)
b = first[-1] if len(first) > 1 else len(rows)
t0 = int(rows[a]["Start_Timestamp"])
prev = None
busy = gaps = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    busy += (e - s) / 1e3
    gaps += max(gap, 0.0)
    print("%9.1f gap %7.1f dur %8.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:80]))
    prev = e
print("kernels %.1f us, gaps %.1f us" % (busy, gaps))

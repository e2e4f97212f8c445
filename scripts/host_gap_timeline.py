"""Host side of one step from a rocprofv3 --hip-trace --kernel-trace CSV pair:
every HIP API call between the last two k_mum_first launches with its start
(relative to the first launch call), duration and name -- shows where the host
spends the time the GPU idles between steps and around the read-backs."""
import csv
import sys

api = list(csv.DictReader(open(sys.argv[1])))
ker = list(csv.DictReader(open(sys.argv[2])))
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [r for r in ker if "k_mum_first" in r["Kernel_Name"]]
t0, t1 = int(first[-2]["Start_Timestamp"]), int(first[-1]["Start_Timestamp"])
api.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 - 300000 or s > t1:
        continue
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%9.1f hostgap %7.1f dur %8.1f  %s" % ((s - t0) / 1e3, gap,
                                                 (e - s) / 1e3, r["Function"]))
    prev = e

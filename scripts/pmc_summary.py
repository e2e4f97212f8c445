#!/usr/bin/env python3
"""Summarises the counter passes of scripts/pmc_passes.sh (one rocprofv3
--pmc run per line of counters, bench.py --steps 1 --warmup 0) into a table
per kernel, and writes profiles/hbm_traffic.json for the dominant kernel.
usage: pmc_summary.py gpurun_out/DIR OUT.txt [--traffic profiles/hbm_traffic.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    return name[:80]


def main():
    d, out = sys.argv[1], sys.argv[2]
    traffic = sys.argv[4] if len(sys.argv) > 4 else None
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    # The traffic of ONE whole step: in the passes with one step, everything
    # between the first k_mum_first launch and the next one (the untimed call
    # that dumps the plans for bench.py's byte count) in dispatch order.
    # (Passes that ran two steps -- pmc_passes.sh, "2 COUNTER" -- were meant for
    # a difference of totals; the builder's kernels move 100 times the bytes of
    # a step and their run-to-run noise drowned it: not used.)
    step = collections.defaultdict(float)
    for f in sorted(glob.glob(d + "/p*/*/*counter_collection.csv")):
        pdir = f[len(d) + 1:].split("/")[0]
        marker = os.path.join(d, pdir + ".steps")
        steps = int(open(marker).read()) if os.path.exists(marker) else 1
        rows = list(csv.DictReader(open(f)))
        if steps == 2:
            continue
        for r in rows:
            k = short(r["Kernel_Name"])
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        firsts = [i for i, r in enumerate(rows)
                  if "k_mum_first" in r["Kernel_Name"]]
        ids = sorted({int(rows[i]["Dispatch_Id"]) for i in firsts})
        if len(ids) >= 2:
            for r in rows:
                if ids[0] <= int(r["Dispatch_Id"]) < ids[1] and \
                        r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    step[(short(r["Kernel_Name"]), r["Counter_Name"])] += \
                        float(r["Counter_Value"])
    step_bytes = None
    if step:
        # FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but a request is an aligned
        # 128-byte line -- for wide coalesced streams (MI355X_MICROARCH.md) and,
        # measured in round 4, for random 16-byte reads as well
        # (profiles/r04/README.md, pmc_line_probe_summary.txt: two sectors of
        # one line = ONE request, and lines arrive at the streaming rate):
        # doubled for every kernel
        step_bytes, per_kernel = 0.0, collections.defaultdict(float)
        for (k, c), v in step.items():
            fac = 2 if c == "FETCH_SIZE" else 1
            per_kernel[k] += v * fac * 1024
            step_bytes += v * fac * 1024
    counters = sorted({c for k in sums for c in sums[k]})
    want = [k for k in sums if k.startswith("k_") or "k_" in k]
    lines = ["rocprofv3 --pmc, one pass per counter group (scripts/"
             "pmc_passes.sh), python3 bench.py --steps 1 --warmup 0 "
             "--cpu-sample 0; mean per launch", ""]
    for k in sorted(want):
        lines.append(k)
        for c in counters:
            if c in sums[k]:
                lines.append("    %-34s %16.4e   (%d launches)" % (
                    c, sums[k][c] / launches[k][c], launches[k][c]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:4]))
    if traffic:
        # the MUM search of the headline workload: the planned form since
        # round 2 (mum_workplan.inc), the work-list form before
        # (template arguments: index width, workgroup, deep tables, reads as
        # 2-bit rows, MUM; the default bench searches packed reads)
        dom = ([k for k in sums if k.startswith(
                    "k_query_search_planned<unsigned int, 256, true, true, true")] or
               [k for k in sums if k.startswith(
                    "k_query_search_planned<unsigned int, 256, true, false, true")])
        if dom:
            k = dom[0]
            mean = {c: sums[k][c] / launches[k][c] for c in sums[k]}
            j = {"kernel": k,
                 "index_bp": 3000000000, "queries": 10000000,
                 "FETCH_SIZE_KiB": mean.get("FETCH_SIZE"),
                 "WRITE_SIZE_KiB": mean.get("WRITE_SIZE"),
                 "TCC_HIT_sum": mean.get("TCC_HIT_sum"),
                 "TCC_MISS_sum": mean.get("TCC_MISS_sum"),
                 "read_requests_per_launch":
                     mean.get("FETCH_SIZE", 0) * 1024 / 64.0,
                 "hbm_bytes_per_launch":
                     (2 * mean.get("FETCH_SIZE", 0) + mean.get("WRITE_SIZE", 0))
                     * 1024,
                 "hbm_bytes_per_launch_at_64B_per_request":
                     (mean.get("FETCH_SIZE", 0) + mean.get("WRITE_SIZE", 0))
                     * 1024,
                 "note": "FETCH_SIZE = read requests x 64 B; a request is an "
                         "aligned 128-byte line, for random 16-byte reads as "
                         "for wide streams (profiles/r04/README.md: two "
                         "sectors of one line are ONE request; "
                         "MI355X_MICROARCH.md prescribes the factor 2 for "
                         "streams): hbm_bytes_per_launch = 2 x FETCH_SIZE + "
                         "WRITE_SIZE.  Rounds 1-3 quoted the 64-byte figure "
                         "(kept next to it)",
                 "kernel_source_sha16": kernel_source_hash(),
                 "round": 4, "source": out}
            if step_bytes:
                j["step_hbm_bytes"] = step_bytes
                j["step_hbm_bytes_by_kernel"] = dict(sorted(
                    per_kernel.items(), key=lambda kv: -kv[1])[:16])
            # the other kernel families bench.py prices: (2 x FETCH_SIZE +
            # WRITE_SIZE) KiB per launch (see the note above)
            fams = {}
            for key, pattern, factor in (
                    ("k_mum_first", "k_mum_first<unsigned int, true, true, true", 2),
                    ("k_mum_plan", "k_mum_plan<unsigned int, true, true, true", 2),
                    ("k_complete_search", "k_complete_search<unsigned int, true, true, true", 2),
                    ("k_query_search_mem", "k_query_search_planned<unsigned int, 256, true, false, false", 2),
                    ("k_mem_plan", "k_mem_plan", 2),
                    ("k_apm_banded", "k_apm_banded", 2),
                    ("k_selfmum_peaks", "k_selfmum_peaks", 2)):
                for k2 in sums:
                    if k2.startswith(pattern) and "FETCH_SIZE" in sums[k2]:
                        m2 = {c: sums[k2][c] / launches[k2][c]
                              for c in sums[k2]}
                        fams[key] = {
                            "kernel": k2,
                            "hbm_bytes_per_launch":
                                (m2.get("FETCH_SIZE", 0) * factor +
                                 m2.get("WRITE_SIZE", 0)) * 1024,
                            "fetch_correction": factor,
                            "launches_averaged": launches[k2]["FETCH_SIZE"]}
            j["families"] = fams
            json.dump(j, open(traffic, "w"), indent=1)
            print(json.dumps(j))


if __name__ == "__main__":
    main()

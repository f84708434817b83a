"""Builds profiles/<tag>_hbm_traffic.json from the two PMC passes of tools/pmc_traffic.sh.

FETCH_SIZE and WRITE_SIZE are reported by rocprofv3 in KB.  On gfx950 FETCH_SIZE tallies a
128-byte read request as 64 bytes, so it is doubled; WRITE_SIZE needs no correction.  Both were
checked against kernels whose traffic is known exactly (k_generate only writes, k_resolve's
reads are the per-sample colours).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    m = re.search(r"(k_[a-z_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def collect(directory, counter):
    files = sorted(glob.glob(f"gpurun_out/{directory}/*/*_counter_collection.csv"), key=os.path.getmtime)
    per = collections.OrderedDict()
    for row in csv.DictReader(open(files[-1])):   # newest pass
        if row["Counter_Name"] != counter:
            continue
        e = per.setdefault(short(row["Kernel_Name"]), {"n": 0, "kb": 0.0, "ns": 0})
        e["n"] += 1
        e["kb"] += float(row["Counter_Value"])
        e["ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return per


fetch, write = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
kernels = collections.OrderedDict()
for name, f in fetch.items():
    w = write.get(name, {"n": f["n"], "kb": 0.0, "ns": 0})
    fb, wb = 2.0 * f["kb"] * 1024 / f["n"], w["kb"] * 1024 / max(w["n"], 1)
    ms = f["ns"] / f["n"] / 1e6
    kernels[name] = {"launches": f["n"], "fetch_bytes": round(fb), "write_bytes": round(wb), "avg_ms": round(ms, 4),
                     "hbm_side_GBps": round((fb + wb) / ms / 1e6, 1)}
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/pmc_traffic.sh) on "
              "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`",
    "units": "bytes per launch, averaged over all launches of the kernel in the run (counted pass + one timed step)",
    "correction": "FETCH_SIZE (KB) is doubled: on gfx950 it tallies a 128-byte read request as 64 bytes. WRITE_SIZE (KB) is used as is.",
    "kernels": kernels,
}
path = f"profiles/{tag}_hbm_traffic.json"
json.dump(out, open(path, "w"), indent=1)
for k, v in kernels.items():
    print(f"{k:40s} {v}")
print("wrote", path)

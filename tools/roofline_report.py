#!/usr/bin/env python3
"""Builds profiles/<tag>_limits.json (and a readable .txt) from the counter passes of
tools/roofline_limits.sh: for the traversal kernel of the timed bench step, how close it runs to each
limit of the machine that could bind it.

    python3 tools/roofline_report.py TAG

Limits and their denominators:
  valu_issue     VALU instructions issued x 4 cycles / (SIMD-cycles the kernel was resident), from
                 SQ_ACTIVE_INST_VALU and SQ_BUSY_CYCLES: the share of issue slots the vector unit used.
  lane_use       SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes enabled per VALU instruction.
  l1_requests    TCP_TOTAL_CACHE_ACCESSES per second (one per 16-byte lane load of a gather: checked
                 on gather64 itself, 4.00 per 64-byte record) / what tools/microbench/gather64
                 sustains from a table that fits the L1 (16 KiB): the vector L1's request rate.
  l2_bandwidth   TCP_TCC_READ_REQ x 64 B per second / 34.5 TB/s (aggregate L2 read bandwidth).
  fabric         bytes crossing L2 -> fabric (FETCH_SIZE + WRITE_SIZE) per second / what gather64
                 sustains from a 64 MiB table (beyond the L2s, inside the Infinity Cache).
  hbm            the same bytes / 8 TB/s.
FETCH_SIZE is calibrated on known byte counts in this kernel's own access shapes
(tools/microbench/fetch_calib): exact for per-lane 64-byte gathers, half for 16-byte-per-lane streams.
The traversal kernel mixes both, so its fabric bytes lie between the raw and the doubled count.
"""
import collections
import csv
import glob
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "cur"
L2_PEAK_GBS, HBM_PEAK_GBS = 34500.0, 8000.0


def rows(kind):
    f = sorted(glob.glob(f"gpurun_out/lim_{tag}_{kind}/*/*_counter_collection.csv"))[-1]
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(by.values())


def timed_trace(entries):
    """Launches of the timed step's traversal kernel: no visit counting (first template argument false)."""
    return [e for e in entries if re.search(r"k_trace<false, \d+, (true|false), (true|false), (true|false), true, false>", e["name"])]


def total(entries, key):
    return sum(e.get(key, 0.0) for e in entries)


sq, tc, fetch, write = (timed_trace(rows(k)) for k in ("sq", "tc", "fetch", "write"))
n = len(sq)
ms = total(sq, "ms")
valu_busy = total(sq, "SQ_ACTIVE_INST_VALU") * 4 / (total(sq, "SQ_BUSY_CYCLES") / 32 * 1024)
lane_use = total(sq, "SQ_THREAD_CYCLES_VALU") / (64 * total(sq, "SQ_ACTIVE_INST_VALU"))
tc_ms = total(tc, "ms")
l1_rate = total(tc, "TCP_TOTAL_CACHE_ACCESSES_sum") / (tc_ms * 1e-3)
l2_gbs = total(tc, "TCP_TCC_READ_REQ_sum") * 64 / (tc_ms * 1e-3) / 1e9
l1_hit = 1 - total(tc, "TCP_TCC_READ_REQ_sum") / total(tc, "TCP_TOTAL_CACHE_ACCESSES_sum")
l2_hit = total(tc, "TCC_HIT_sum") / (total(tc, "TCC_HIT_sum") + total(tc, "TCC_MISS_sum"))
fetch_raw = total(fetch, "FETCH_SIZE") * 1024
write_b = total(write, "WRITE_SIZE") * 1024
f_ms, w_ms = total(fetch, "ms"), total(write, "ms")

# calibration: FETCH_SIZE against known bytes
known = dict(re.findall(r"(k_calib<\d+>) bytes (\d+)", open(f"gpurun_out/lim_{tag}_calib.log").read()))
calib = {}
for e in rows("calib"):
    m = re.search(r"k_calib<\d+>", e["name"])
    if m:
        calib[m.group(0)] = round(e["FETCH_SIZE"] * 1024 / int(known[m.group(0)]), 4)

# microbenchmarks: sustained 64-byte gathers (records/s) from tables inside the L2s and inside the Infinity Cache
gather = {}
for line in open(f"gpurun_out/lim_{tag}_gather64.txt"):
    m = re.match(r"table (\d+) KB, 7 blocks/CU, shape A: .* ([\d.]+) G records/s, ([\d.]+) TB/s", line)
    if m:
        gather[int(m.group(1))] = (float(m.group(2)) * 1e9, float(m.group(3)) * 1e3)
chain = re.findall(r"(\d+) blocks, per-lane chains of 40 dependent 64-B loads: [\d.]+ us per launch \(([\d.]+) us per step\)", open(f"gpurun_out/lim_{tag}_chain.txt").read())
l1_peak = gather[16][0] * 4          # four 16-byte lane loads per record = accesses the L1 takes per second
fabric_peak_gbs = gather[65536][1]

fabric_lo = (fetch_raw / f_ms + write_b / w_ms) / 1e6        # GB/s, FETCH_SIZE as counted (exact for 64-byte gathers)
fabric_hi = (2 * fetch_raw / f_ms + write_b / w_ms) / 1e6    # doubled (what a 16-byte-per-lane stream needs)
out = {
    "source": f"tools/roofline_limits.sh {tag}: rocprofv3 --pmc passes on `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin`, timed traversal launches only; tools/microbench/{{fetch_calib,gather64,chain}}",
    "kernel": re.search(r"k_trace<[^>]*>", sq[0]["name"]).group(0),
    "launches": n,
    "avg_launch_ms": round(ms / n, 4),
    "fetch_size_calibration": {"counted_over_known_bytes": calib,
                               "reading": "64-byte per-lane gathers are counted exactly, 16-byte-per-lane streams at half; 48-byte records read 1.67x their size (two 64-byte pieces for most records)"},
    "limits": {
        "valu_issue": {"frac": round(valu_busy, 4), "what": "share of VALU issue cycles used (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles busy)"},
        "l1_requests": {"frac": round(l1_rate / l1_peak, 4), "achieved_per_s": round(l1_rate), "peak_per_s": round(l1_peak), "what": "vector L1 accesses (16-byte lane loads) per second against gather64 from a 16 KiB table"},
        "l2_bandwidth": {"frac": round(l2_gbs / L2_PEAK_GBS, 4), "achieved_GBps": round(l2_gbs, 1), "peak_GBps": L2_PEAK_GBS, "what": "L1 -> L2 read requests x 64 B"},
        "fabric": {"frac_low": round(fabric_lo / fabric_peak_gbs, 4), "frac_high": round(fabric_hi / fabric_peak_gbs, 4), "achieved_GBps_low": round(fabric_lo, 1), "achieved_GBps_high": round(fabric_hi, 1),
                   "peak_GBps": fabric_peak_gbs, "what": "bytes crossing L2 -> fabric against gather64 from a 64 MiB table (served by the Infinity Cache)"},
        "hbm": {"frac_low": round(fabric_lo / HBM_PEAK_GBS, 4), "frac_high": round(fabric_hi / HBM_PEAK_GBS, 4), "peak_GBps": HBM_PEAK_GBS},
    },
    "lane_use": round(lane_use, 4),
    "l1_hit": round(l1_hit, 4),
    "l2_hit": round(l2_hit, 4),
    "fabric_bytes_per_launch": {"fetch_as_counted": round(fetch_raw / len(fetch)), "fetch_doubled": round(2 * fetch_raw / len(fetch)), "write": round(write_b / len(write))},
    "gather64_TBps": {f"{kb}_KiB_table": round(v[1] / 1e3, 2) for kb, v in sorted(gather.items())},
    "l2_miss_bytes_per_launch": round(total(tc, "TCC_MISS_sum") * 64 / len(tc)),
    "dependent_load_latency_us": {f"{b}_blocks_each_lane_its_own_chain": float(us) for b, us in chain},
}
lim = out["limits"]
binding = max(("valu_issue", "l1_requests", "l2_bandwidth"), key=lambda k: lim[k]["frac"])
if lim["fabric"]["frac_high"] > lim[binding]["frac"]:
    binding = "fabric"
out["binding"] = binding
json.dump(out, open(f"profiles/{tag}_limits.json", "w"), indent=1)
with open(f"profiles/{tag}_limits.txt", "w") as f:
    f.write(f"# {out['source']}\n# {out['kernel']}: {n} launches, {out['avg_launch_ms']} ms on average\n")
    for k, v in lim.items():
        f.write(f"{k:14s} {json.dumps(v)}\n")
    f.write(f"lane use {lane_use:.3f}, L1 hit {l1_hit:.3f}, L2 hit {l2_hit:.3f}; FETCH_SIZE calibration {calib}; binding: {binding}\n")
    f.write("per launch (timed traversal kernel), raw counters:\n")
    for a, b, c, d in zip(sq, tc, fetch, write):
        f.write(f"  ms={a['ms']:.3f} VALU_inst={a['SQ_INSTS_VALU']:.0f} ACTIVE_INST_VALU={a['SQ_ACTIVE_INST_VALU']:.0f} THREAD_CYCLES_VALU={a['SQ_THREAD_CYCLES_VALU']:.0f} BUSY_CYCLES={a['SQ_BUSY_CYCLES']:.0f} WAVE_CYCLES={a.get('SQ_WAVE_CYCLES', 0):.0f} "
                f"TCP_ACCESSES={b['TCP_TOTAL_CACHE_ACCESSES_sum']:.0f} TCP_TCC_READ_REQ={b['TCP_TCC_READ_REQ_sum']:.0f} TCC_HIT={b['TCC_HIT_sum']:.0f} TCC_MISS={b['TCC_MISS_sum']:.0f} FETCH_KB={c['FETCH_SIZE']:.0f} WRITE_KB={d['WRITE_SIZE']:.0f}\n")
print(open(f"profiles/{tag}_limits.txt").read())

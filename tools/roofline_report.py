#!/usr/bin/env python3
"""Builds profiles/<tag>_limits.json (and a readable .txt) from the counter passes of
tools/roofline_limits.sh: for every kernel of the timed bench step, how close it runs to each limit of the
machine that could bind it, stamped with the build of the library it was measured on.

    python3 tools/roofline_report.py TAG [GIT_COMMIT]

Per kernel (k_trace, k_shade_rgb, k_connect<false>, k_generate, k_resolve from the default RGB step;
k_shade_spectral, k_connect<true> from the Spectral step):
  valu_issue     VALU instructions issued x 4 cycles / (SIMD-cycles the kernel was resident), from
                 SQ_ACTIVE_INST_VALU and SQ_BUSY_CYCLES: the share of issue slots the vector unit used.
  lane_use       SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes enabled per VALU instruction.
  waves_per_simd SQ_WAVE_CYCLES / SIMD-cycles the kernel was resident: waves in flight per SIMD on average.
  wait_share     SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of their resident cycles waves spend waiting on a dependency
                 (s_waitcnt on outstanding memory instructions, mostly); issue_wait_share is SQ_WAIT_INST_ANY /
                 SQ_WAVE_CYCLES: cycles a wave had an instruction ready but the issue slot went to another wave.
  l1_requests    TCP_TOTAL_CACHE_ACCESSES per second / what tools/microbench/gather64 sustains from a
                 table that fits the L1 (16 KiB): the vector L1's request rate.
  l2_bandwidth   TCP_TCC_READ_REQ x 64 B per second / 34.5 TB/s (aggregate L2 read bandwidth).
  fabric         bytes crossing L2 -> fabric per second / what gather64 sustains from a 64 MiB table
                 (beyond the L2s, inside the Infinity Cache).
  hbm            the same bytes / 8 TB/s.

Bytes across the L2 -> fabric boundary.  rocprofv3's FETCH_SIZE counts per-lane gathers exactly (in 32-byte
sectors) and wide consecutive streams at half their bytes (tools/microbench/fetch_calib; the guide's gfx950
note); WRITE_SIZE is exact.  A kernel's fetched bytes are therefore
    FETCH_SIZE + 0.5 x (bytes it reads as streams from beyond the L2s),
and the stream bytes are known from the algorithm (queue entries, ray records, path state: read once each).
tools/microbench/mix_calib replays the traversal kernel's mix of gathers and streams with known byte
counts; the report prints how far this model is off on each of its mixes (`mix_calibration`), which is
the uncertainty of the `fabric` and `hbm` figures.
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "cur"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
L2_PEAK_GBS, HBM_PEAK_GBS = 34500.0, 8000.0
SIMDS = 1024


def rows(kind):
    files = sorted(glob.glob(f"gpurun_out/lim_{tag}_{kind}/*/*_counter_collection.csv"))
    if not files:
        return []
    by = collections.OrderedDict()
    for r in csv.DictReader(open(files[-1])):
        e = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                                                  "vgpr": r.get("VGPR_Count", ""), "scratch": r.get("Scratch_Size", ""), "lds": r.get("LDS_Block_Size", "")})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(by.values())


def total(entries, key):
    return sum(e.get(key, 0.0) for e in entries)


# Bytes a kernel reads as wide consecutive streams from beyond the L2s, per work item (FETCH_SIZE counts them at half).
#   k_trace      queue entry 4 + ray record 32 per ray
#   k_shade_*    queue entry 4 + ray 32 + hit 16 + path state 32 per path (stage 0 of the RGB integrator reads no state: 16 fewer on average)
#   k_connect    queue entry 4 + shadow ray 32 + hit 16 + pending NEE term 48 + radiance 16 per shadow ray
#   k_resolve    radiance 16 per sample
#   k_generate   none (writes only)
STREAM_READ_BYTES = {"k_trace": 36.0, "k_shade_rgb": 76.0, "k_shade_spectral": 84.0, "k_connect": 116.0, "k_resolve": 16.0, "k_generate": 0.0}

FAMILIES = [   # (key, prefix of the passes, regular expression on the kernel name, work-item stream key)
    ("k_trace", "", r"k_trace<false, \d+, (true|false), (true|false), (true|false), true, false>", "k_trace"),
    ("k_trace_second_walk", "", r"k_trace<false, \d+, (true|false), (true|false), false, false, true>", None),
    ("k_shade_rgb", "", r"k_shade_rgb", "k_shade_rgb"),
    ("k_connect_rgb", "", r"k_connect<false>", "k_connect"),
    ("k_generate_rgb", "", r"k_generate<false>", "k_generate"),
    ("k_resolve_rgb", "", r"k_resolve<false>", "k_resolve"),
    ("k_trace_spectral_step", "spec_", r"k_trace<false, \d+, (true|false), (true|false), (true|false), true, false>", "k_trace"),
    ("k_shade_spectral", "spec_", r"k_shade_spectral", "k_shade_spectral"),
    ("k_connect_spectral", "spec_", r"k_connect<true>", "k_connect"),
]

# microbenchmarks: sustained 64-byte gathers (records/s) from tables inside the L1, the L2s and the Infinity Cache
gather = {}
for line in open(f"gpurun_out/lim_{tag}_gather64.txt"):
    m = re.match(r"table (\d+) KB, 7 blocks/CU, shape A: .* ([\d.]+) G records/s, ([\d.]+) TB/s", line)
    if m:
        gather[int(m.group(1))] = (float(m.group(2)) * 1e9, float(m.group(3)) * 1e3)
chain = re.findall(r"(\d+) blocks, per-lane chains of 40 dependent 64-B loads: [\d.]+ us per launch \(([\d.]+) us per step\)", open(f"gpurun_out/lim_{tag}_chain.txt").read())
l1_peak = gather[16][0] * 4          # four 16-byte lane loads per record = accesses the L1 takes per second
fabric_peak_gbs = gather[65536][1]

# calibration of the byte counters: pure shapes, then the traversal kernel's mix
known = dict(re.findall(r"(k_calib<\d+>) bytes (\d+)", open(f"gpurun_out/lim_{tag}_calib.log").read()))
calib = {}
for e in rows("calib"):
    m = re.search(r"k_calib<\d+>", e["name"])
    if m:
        calib[m.group(0)] = round(e["FETCH_SIZE"] * 1024 / int(known[m.group(0)]), 4)
mix = {}
mixlog = open(f"gpurun_out/lim_{tag}_mix_fetch.log").read()
fetched = {re.search(r"k_mix<\d+, \d+>", e["name"]).group(0): e["FETCH_SIZE"] * 1024 for e in rows("mix_fetch") if "k_mix" in e["name"]}
written = {re.search(r"k_mix<\d+, \d+>", e["name"]).group(0): e["WRITE_SIZE"] * 1024 for e in rows("mix_write") if "k_mix" in e["name"]}
for name, rays, req, sectors, sread, swritten in re.findall(r"(k_mix<\d+, \d+>) rays (\d+) gather_requested (\d+) gather_sectors (\d+) stream_read (\d+) stream_written (\d+)", mixlog):
    req, sectors, sread, swritten = float(req), float(sectors), float(sread), float(swritten)
    counted = fetched.get(name)
    if counted is None:
        continue
    mix[name] = {
        "stream_share_of_bytes_read": round(sread / (sectors + sread), 4),
        "FETCH_SIZE_over_bytes_read": round(counted / (sectors + sread), 4),
        "model_over_bytes_read": round((counted + 0.5 * sread) / (sectors + sread), 4),      # the correction the report applies
        "doubled_over_bytes_read": round(2 * counted / (sectors + sread), 4),                  # what doubling everything would claim
        "WRITE_SIZE_over_bytes_written": round(written.get(name, 0.0) / swritten, 4),
    }
model_error = max((abs(v["model_over_bytes_read"] - 1.0) for v in mix.values()), default=None)


def family(key, prefix, pattern, stream_key):
    def sel(kind):
        # The bench command runs one untimed step with the visit counters on before the timed step: only the launches
        # that follow the first launch of the timed traversal kernel (and the k_generate just before it) belong to the step.
        every = rows(prefix + kind)
        first_timed = next((i for i, e in enumerate(every) if re.search(FAMILIES[0][2], e["name"])), 0)
        start = max((i for i in range(first_timed) if "k_generate" in every[i]["name"]), default=0)
        return [e for e in every[start:] if re.search(pattern, e["name"])]
    sq, tc, fetch, write = sel("sq"), sel("tc"), sel("fetch"), sel("write")
    if not sq:
        return None
    n = len(sq)
    ms = total(sq, "ms")
    simd_cycles = total(sq, "SQ_BUSY_CYCLES") / 32 * SIMDS
    out = {"kernel": re.search(pattern, sq[0]["name"]).group(0), "mangled_or_full_name": sq[0]["name"][:160], "launches_per_step": n, "avg_launch_ms": round(ms / n, 4), "ms_per_step": round(ms, 3),
           "vgprs": (2 * int(sq[0]["vgpr"]) if str(sq[0]["vgpr"]).isdigit() else sq[0]["vgpr"]),
           "vgprs_note": "twice rocprofv3's VGPR_Count (it reports half of the allocation)", "scratch_bytes": sq[0]["scratch"], "lds_bytes_per_block": sq[0]["lds"]}
    out["valu_issue"] = round(total(sq, "SQ_ACTIVE_INST_VALU") * 4 / simd_cycles, 4)
    out["lane_use"] = round(total(sq, "SQ_THREAD_CYCLES_VALU") / max(64 * total(sq, "SQ_ACTIVE_INST_VALU"), 1), 4)
    out["waves_per_simd"] = round(4 * total(sq, "SQ_WAVE_CYCLES") / simd_cycles, 2)   # (the counter ticks once per four cycles)
    out["wait_share"] = round(total(sq, "SQ_WAIT_ANY") / max(total(sq, "SQ_WAVE_CYCLES"), 1), 4) if total(sq, "SQ_WAIT_ANY") else None
    out["issue_wait_share"] = round(total(sq, "SQ_WAIT_INST_ANY") / max(total(sq, "SQ_WAVE_CYCLES"), 1), 4)
    out["valu_instructions_per_wave"] = round(total(sq, "SQ_INSTS_VALU") / max(total(sq, "SQ_WAVES"), 1), 1)
    if tc:
        tc_ms = total(tc, "ms")
        l1_rate = total(tc, "TCP_TOTAL_CACHE_ACCESSES_sum") / (tc_ms * 1e-3)
        l2_gbs = total(tc, "TCP_TCC_READ_REQ_sum") * 64 / (tc_ms * 1e-3) / 1e9
        out["l1_requests"] = {"frac": round(l1_rate / l1_peak, 4), "achieved_per_s": round(l1_rate), "peak_per_s": round(l1_peak)}
        out["l2_bandwidth"] = {"frac": round(l2_gbs / L2_PEAK_GBS, 4), "achieved_GBps": round(l2_gbs, 1), "peak_GBps": L2_PEAK_GBS}
        out["l1_hit"] = round(1 - total(tc, "TCP_TCC_READ_REQ_sum") / max(total(tc, "TCP_TOTAL_CACHE_ACCESSES_sum"), 1), 4)
        out["l2_hit"] = round(total(tc, "TCC_HIT_sum") / max(total(tc, "TCC_HIT_sum") + total(tc, "TCC_MISS_sum"), 1), 4)
        out["l2_miss_bytes_per_step"] = round(total(tc, "TCC_MISS_sum") * 64)
    if fetch and write:
        fetch_raw, write_b = total(fetch, "FETCH_SIZE") * 1024, total(write, "WRITE_SIZE") * 1024
        f_ms, w_ms = total(fetch, "ms"), total(write, "ms")
        out["fabric_bytes_per_step"] = {"FETCH_SIZE_as_counted": round(fetch_raw), "WRITE_SIZE": round(write_b)}
        out["_fetch_raw"], out["_write"], out["_f_ms"], out["_w_ms"] = fetch_raw, write_b, f_ms, w_ms
    out["_stream_key"] = stream_key
    out["_per_launch"] = [(a, b, c, d) for a, b, c, d in zip(sq, tc or sq, fetch or sq, write or sq)]
    return out


kernels = collections.OrderedDict()
for key, prefix, pattern, stream_key in FAMILIES:
    f = family(key, prefix, pattern, stream_key)
    if f:
        kernels[key] = f

# Work items per step, for the stream correction: rays of the step from the bench lines the passes printed.
def bench_line(kind):
    try:
        for line in open(f"gpurun_out/lim_{tag}_{kind}.log"):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


for prefix in ("", "spec_"):
    line = bench_line(prefix + "fetch")
    if not line:
        continue
    rays = line["value"] * 1e6 * line["ms_per_step"] * 1e-3
    samples = line["camera_samples_per_s"] * line["ms_per_step"] * 1e-3
    per_sample = line["config"]["rays_per_camera_sample"]
    # extension rays (= paths shaded) and shadow rays of a step: every shaded non-terminal path sends one of each at most;
    # the split is read from the kernels' own launch structure: rays = paths shaded + shadow rays connected.
    for key, k in kernels.items():
        if ("spectral" in key) != (prefix == "spec_") or "_fetch_raw" not in k:
            continue
        sk = k["_stream_key"]
        if sk is None:
            continue
        items = {"k_trace": rays, "k_generate": samples, "k_resolve": samples}.get(sk)
        if items is None:
            shadow = samples * line["config"].get("shadow_rays_per_camera_sample", per_sample / 2)
            items = shadow if sk == "k_connect" else rays - shadow     # shadow rays connected / path segments shaded
        stream = items * STREAM_READ_BYTES[sk]
        fetched_model = k["_fetch_raw"] + 0.5 * stream
        gbs = (fetched_model / k["_f_ms"] + k["_write"] / k["_w_ms"]) / 1e6
        k["fabric_bytes_per_step"]["stream_bytes_read"] = round(stream)
        k["fabric_bytes_per_step"]["fetched_corrected"] = round(fetched_model)
        k["fabric"] = {"frac": round(gbs / fabric_peak_gbs, 4), "achieved_GBps": round(gbs, 1), "peak_GBps": fabric_peak_gbs,
                       "uncertainty": (round(model_error, 3) if model_error is not None else None)}
        k["hbm"] = {"frac": round(gbs / HBM_PEAK_GBS, 4), "peak_GBps": HBM_PEAK_GBS}

build_id = open(f"gpurun_out/lim_{tag}_buildid.txt").read().strip() if os.path.exists(f"gpurun_out/lim_{tag}_buildid.txt") else None
trace = kernels["k_trace"]
limits = {
    "valu_issue": {"frac": trace["valu_issue"], "what": "share of VALU issue cycles used (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles busy)"},
    "l1_requests": dict(trace["l1_requests"], what="vector L1 accesses (16-byte lane loads) per second against gather64 from a 16 KiB table"),
    "l2_bandwidth": dict(trace["l2_bandwidth"], what="L1 -> L2 read requests x 64 B"),
    "fabric": dict(trace["fabric"], what="bytes crossing L2 -> fabric (FETCH_SIZE + half of the stream bytes, + WRITE_SIZE) against gather64 from a 64 MiB table (served by the Infinity Cache)"),
    "hbm": trace["hbm"],
}
binding = max(("valu_issue", "l1_requests", "l2_bandwidth", "fabric"), key=lambda k: limits[k]["frac"])
n_trace = trace["launches_per_step"]
out = {
    "source": f"tools/roofline_limits.sh {tag}: rocprofv3 --pmc passes on `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin` (and the same with --scene knot_glass --spectral); tools/microbench/{{fetch_calib,mix_calib,gather64,chain}}",
    "library_build_id": build_id,
    "git_commit": commit,
    "kernel": trace["kernel"],
    "kernel_full_name": trace["mangled_or_full_name"],
    "launches": n_trace,
    "avg_launch_ms": trace["avg_launch_ms"],
    "fetch_size_calibration": {"counted_over_known_bytes": calib,
                               "reading": "64-byte per-lane gathers are counted exactly, 16-byte-per-lane streams at half; 48-byte records read 1.67x their size (32-byte sectors)"},
    "mix_calibration": mix,
    "mix_calibration_reading": "k_mix<V, P>: V node gathers + P pair gathers per ray plus the ray/queue/hit streams, every byte from beyond the caches; model = FETCH_SIZE + half of the stream bytes, "
                               f"off by at most {model_error:.1%} over the mixes" if model_error is not None else "not measured",
    "limits": limits,
    "binding": binding,
    "lane_use": trace["lane_use"],
    "l1_hit": trace["l1_hit"],
    "l2_hit": trace["l2_hit"],
    "fabric_bytes_per_launch": {"fetch_as_counted": round(trace["_fetch_raw"] / n_trace), "fetch_corrected": round(trace["fabric_bytes_per_step"]["fetched_corrected"] / n_trace),
                                "write": round(trace["_write"] / n_trace)},
    "gather64_TBps": {f"{kb}_KiB_table": round(v[1] / 1e3, 2) for kb, v in sorted(gather.items())},
    "dependent_load_latency_us": {f"{b}_blocks_each_lane_its_own_chain": float(us) for b, us in chain},
    "kernels": {},
}
for key, k in kernels.items():
    out["kernels"][key] = {kk: vv for kk, vv in k.items() if not kk.startswith("_")}
json.dump(out, open(f"profiles/{tag}_limits.json", "w"), indent=1)

with open(f"profiles/{tag}_limits.txt", "w") as f:
    f.write(f"# {out['source']}\n# library build {build_id}, git commit {commit}\n")
    f.write(f"# traversal kernel {trace['kernel']}: {n_trace} launches per step, {trace['avg_launch_ms']} ms on average; binding limit: {binding}\n")
    for k, v in limits.items():
        f.write(f"{k:14s} {json.dumps(v)}\n")
    f.write(f"FETCH_SIZE calibration, pure shapes: {calib}\n")
    for name, v in mix.items():
        f.write(f"mix {name}: {json.dumps(v)}\n")
    f.write("\nper kernel of the step (time per step, VALU issue, lanes enabled per VALU instruction, waves per SIMD, share of wave cycles waiting, L1/L2 hit, fabric GB/s):\n")
    for key, k in out["kernels"].items():
        fab = k.get("fabric", {})
        f.write(f"  {key:24s} {k['ms_per_step']:8.3f} ms in {k['launches_per_step']:2d} launches  vgpr {k['vgprs']:>4}  VALU issue {k['valu_issue']:.3f}  lane use {k['lane_use']:.3f}  waves/SIMD {k['waves_per_simd']:5.2f}  "
                f"waiting {(k['wait_share'] if k['wait_share'] is not None else float('nan')):.3f} (for an issue slot {k['issue_wait_share']:.3f})  VALU/wave {k['valu_instructions_per_wave']:8.1f}  L1 hit {k.get('l1_hit', float('nan')):.3f}  L2 hit {k.get('l2_hit', float('nan')):.3f}  "
                f"fabric {fab.get('achieved_GBps', float('nan')):7.1f} GB/s ({fab.get('frac', float('nan')):.3f} of the gather rate, {k.get('hbm', {}).get('frac', float('nan')):.3f} of HBM)\n")
    f.write("\nper launch, raw counters:\n")
    for key, k in kernels.items():
        for a, b, c, d in k["_per_launch"]:
            f.write(f"  {key} ms={a['ms']:.3f} WAVES={a.get('SQ_WAVES', 0):.0f} VALU_inst={a.get('SQ_INSTS_VALU', 0):.0f} ACTIVE_INST_VALU={a.get('SQ_ACTIVE_INST_VALU', 0):.0f} THREAD_CYCLES_VALU={a.get('SQ_THREAD_CYCLES_VALU', 0):.0f} "
                    f"BUSY_CYCLES={a.get('SQ_BUSY_CYCLES', 0):.0f} WAVE_CYCLES={a.get('SQ_WAVE_CYCLES', 0):.0f} WAIT_INST_ANY={a.get('SQ_WAIT_INST_ANY', 0):.0f} WAIT_ANY={a.get('SQ_WAIT_ANY', 0):.0f} "
                    f"TCP_ACCESSES={b.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.0f} TCP_TCC_READ_REQ={b.get('TCP_TCC_READ_REQ_sum', 0):.0f} TCC_HIT={b.get('TCC_HIT_sum', 0):.0f} TCC_MISS={b.get('TCC_MISS_sum', 0):.0f} "
                    f"FETCH_KB={c.get('FETCH_SIZE', 0):.0f} WRITE_KB={d.get('WRITE_SIZE', 0):.0f}\n")
print(open(f"profiles/{tag}_limits.txt").read()[:6000])

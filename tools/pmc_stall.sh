# Where the waves of a kernel spend their cycles: issue, waiting, instruction fetch.  tools/pmc_stall.sh TAG KERNEL_SUBSTRING [bench args]
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; KERNEL=$2; shift 2
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin $@"
rm -rf gpurun_out/stall_${TAG}_a gpurun_out/stall_${TAG}_b
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH --output-format csv -d gpurun_out/stall_${TAG}_a -- $BENCH > gpurun_out/stall_${TAG}_a.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --output-format csv -d gpurun_out/stall_${TAG}_b -- $BENCH > gpurun_out/stall_${TAG}_b.log 2>&1
python3 - "$TAG" "$KERNEL" <<'PY'
import csv, glob, collections, sys
tag, kernel = sys.argv[1], sys.argv[2]
def rows(kind):
    f = sorted(glob.glob(f"gpurun_out/stall_{tag}_{kind}/*/*_counter_collection.csv"))[-1]
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]: continue
        e = by.setdefault(int(r["Dispatch_Id"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(by.values())
for a, b in zip(rows("a"), rows("b")):
    wc = a["SQ_WAVE_CYCLES"]
    print(f"{kernel} ms={a['ms']:.3f} of wave cycles: waiting(any) {a['SQ_WAIT_ANY']/wc:.3f} waiting(inst) {a['SQ_WAIT_INST_ANY']/wc:.3f} issuing(any) {a['SQ_ACTIVE_INST_ANY']/wc:.3f} "
          f"valu {a['SQ_ACTIVE_INST_VALU']/wc:.3f} scalar {a['SQ_ACTIVE_INST_SCA']/wc:.3f} vmem {a['SQ_ACTIVE_INST_VMEM']/wc:.3f}; ifetch {a['SQ_IFETCH']:.3g}; "
          f"icache req {b['SQC_ICACHE_REQ']:.3g} hit {b['SQC_ICACHE_HITS']/max(b['SQC_ICACHE_REQ'],1):.3f} miss {b['SQC_ICACHE_MISSES']:.3g} dup {b['SQC_ICACHE_MISSES_DUPLICATE']:.3g}; salu {b['SQ_INSTS_SALU']:.3g} smem {b['SQ_INSTS_SMEM']:.3g} waves {b['SQ_WAVES']:.0f}")
PY

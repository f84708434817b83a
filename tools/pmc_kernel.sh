# SQ counters of one kernel of the default bench: tools/pmc_kernel.sh TAG KERNEL_SUBSTRING
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$1_sq
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_$1_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin > gpurun_out/pmc_$1_sq.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$1_sq/*/*_counter_collection.csv")[0]
by = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "$2" not in r["Kernel_Name"]: continue
    e = by.setdefault(int(r["Dispatch_Id"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "vgpr": r["VGPR_Count"], "scratch": r["Scratch_Size"]})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
for k, e in list(by.items())[-6:]:
    print("$2 ms=%.2f vgpr=%s scratch=%s waves=%d VALUinsts/wave=%.0f VMEMrd/wave=%.1f laneUtil=%.1f%% VALUbusy=%.0f%% wait=%.0f%% of wave cycles" % (
        e["ms"], e["vgpr"], e["scratch"], e["SQ_WAVES"], e["SQ_INSTS_VALU"] / e["SQ_WAVES"], e["SQ_INSTS_VMEM_RD"] / e["SQ_WAVES"],
        100 * e["SQ_THREAD_CYCLES_VALU"] / (e["SQ_ACTIVE_INST_VALU"] * 64), 100 * e["SQ_ACTIVE_INST_VALU"] * 4 / (e["SQ_BUSY_CYCLES"] / 32 * 1024),
        100 * e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]))
PY

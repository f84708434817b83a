# FETCH_SIZE + cache hit counters of the default bench under an environment setting: tools/pmc_env.sh TAG VAR=value
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export $2
rm -rf gpurun_out/pmc_$1_tc gpurun_out/pmc_$1_fetch
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmc_$1_tc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin > gpurun_out/pmc_$1_tc.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_$1_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin > gpurun_out/pmc_$1_fetch.log 2>&1
python3 - <<PY
import csv, glob, collections
for kind in ("tc", "fetch"):
    f = glob.glob("gpurun_out/pmc_$1_%s/*/*_counter_collection.csv" % kind)[0]
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_trace<false" not in r["Kernel_Name"] and "k_traceILb0" not in r["Kernel_Name"]: continue
        e = by.setdefault(int(r["Dispatch_Id"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, e in by.items():
        if "TCC_HIT_sum" in e:
            print("$1 stage ms=%.2f L2hit=%.1f%% L1hit=%.1f%% L2miss=%.0fM L1acc=%.0fM" % (e["ms"], 100 * e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 100 * (1 - e["TCP_TCC_READ_REQ_sum"] / e["TCP_TOTAL_CACHE_ACCESSES_sum"]), e["TCC_MISS_sum"] / 1e6, e["TCP_TOTAL_CACHE_ACCESSES_sum"] / 1e6))
        else:
            print("$1 stage ms=%.2f fetchGB=%.2f" % (e["ms"], 2 * e["FETCH_SIZE"] * 1024 / 1e9))
PY

# Counter passes behind bench.py's roofline block, on the bench command itself (one step of the
# default workload), plus the microbenchmarks that give the limits their denominators.
#   bash tools/roofline_limits.sh TAG      -> gpurun_out/lim_TAG_*; then: python3 tools/roofline_report.py TAG
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-cur}
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin"
pass() { timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/lim_${TAG}_$1 -- $BENCH > gpurun_out/lim_${TAG}_$1.log 2>&1; }
pass sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
pass tc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
pass fetch "FETCH_SIZE"
pass write "WRITE_SIZE"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/lim_${TAG}_calib -- tools/microbench/fetch_calib > gpurun_out/lim_${TAG}_calib.log 2>&1
timeout -k 5 120 tools/microbench/gather64 16 > gpurun_out/lim_${TAG}_gather64.txt 2>&1
timeout -k 5 120 tools/microbench/gather64 16384 >> gpurun_out/lim_${TAG}_gather64.txt 2>&1
timeout -k 5 120 tools/microbench/gather64 65536 >> gpurun_out/lim_${TAG}_gather64.txt 2>&1
timeout -k 5 120 tools/microbench/chain > gpurun_out/lim_${TAG}_chain.txt 2>&1
echo done

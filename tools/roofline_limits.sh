# Counter passes behind bench.py's roofline block, on the bench command itself (one step of the default
# workload, then one step of the Spectral configuration for its kernels), plus the microbenchmarks that give
# the limits their denominators and calibrate the byte counters.  Run on the GPU box:
#   bash tools/roofline_limits.sh TAG      -> gpurun_out/lim_TAG_*; then, here: python3 tools/roofline_report.py TAG
# One counter block per pass (rocprofv3 does not split them), every pass under its own time limit; the
# script stops at the first pass that fails.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-cur}
BENCH="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin"
SPEC="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin --scene knot_glass --spectral"
pass() { name=$1; counters=$2; shift 2; timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d gpurun_out/lim_${TAG}_$name -- "$@" > gpurun_out/lim_${TAG}_$name.log 2>&1; echo "pass $name done"; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES"
TC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
python3 -c "import sys; sys.path.insert(0, '.'); from raytracer_amd import capi; print(capi.build_id())" > gpurun_out/lim_${TAG}_buildid.txt
pass sq "$SQ" $BENCH
pass tc "$TC" $BENCH
pass fetch FETCH_SIZE $BENCH
pass write WRITE_SIZE $BENCH
pass spec_sq "$SQ" $SPEC
pass spec_tc "$TC" $SPEC
pass spec_fetch FETCH_SIZE $SPEC
pass spec_write WRITE_SIZE $SPEC
pass calib FETCH_SIZE tools/microbench/fetch_calib
pass mix_fetch FETCH_SIZE tools/microbench/mix_calib
pass mix_write WRITE_SIZE tools/microbench/mix_calib
timeout -k 5 120 tools/microbench/gather64 16 > gpurun_out/lim_${TAG}_gather64.txt 2>&1
timeout -k 5 120 tools/microbench/gather64 16384 >> gpurun_out/lim_${TAG}_gather64.txt 2>&1
timeout -k 5 120 tools/microbench/gather64 65536 >> gpurun_out/lim_${TAG}_gather64.txt 2>&1
timeout -k 5 120 tools/microbench/chain > gpurun_out/lim_${TAG}_chain.txt 2>&1
echo done

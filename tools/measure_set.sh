# The measurement set on one box: bash tools/measure_set.sh TAG  (outputs under gpurun_out/)
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --no-optin --no-cpu-baseline > gpurun_out/prof_${TAG}.log 2>&1
cp $(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
echo "stats done"
bash tools/bench_configs.sh ${TAG}
timeout -k 10 300 python3 tools/interactive_rate.py > gpurun_out/${TAG}_interactive_rate.txt 2>&1
echo "interactive done"
bash tools/roofline_limits.sh ${TAG}

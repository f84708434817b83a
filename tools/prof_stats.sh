# rocprofv3 kernel statistics of the default bench command.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py --no-optin > gpurun_out/prof_$1.log 2>&1
tail -2 gpurun_out/prof_$1.log
find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs cat

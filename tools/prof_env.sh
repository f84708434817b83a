# Kernel-trace of the default bench under an environment setting: tools/prof_env.sh TAG VAR=value
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export $2
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-optin > gpurun_out/prof_$1.log 2>&1
python3 tools/stage_times.py gpurun_out/prof_$1 | grep -v "fillBuffer\|elementwise"

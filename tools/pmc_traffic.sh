# HBM-side traffic of the bench's kernels: FETCH_SIZE and WRITE_SIZE need a pass each (TCC has 4 slots).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-optin > gpurun_out/pmc_write.log 2>&1
echo done

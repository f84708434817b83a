# Collects SQ and cache counters for the bench's kernels (one short run per pass).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-cur}
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_${TAG}_sq -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-optin > gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_${TAG}_sq2 -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-optin > gpurun_out/pmc_${TAG}_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmc_${TAG}_tc -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-optin > gpurun_out/pmc_${TAG}_tc.log 2>&1
echo done

set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cd raytracer_amd/lib && cp librt_hip.so v_new.so && cd ../..
for v in v_orig v_new; do
  cp raytracer_amd/lib/$v.so raytracer_amd/lib/librt_hip.so
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_${v}_sq -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-optin > gpurun_out/pmc_${v}_sq.log 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmc_${v}_tc -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline --no-optin > gpurun_out/pmc_${v}_tc.log 2>&1
done
cp raytracer_amd/lib/v_new.so raytracer_amd/lib/librt_hip.so
ls gpurun_out/pmc_v_orig_sq/*/

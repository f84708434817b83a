# Confirms the cause of the wrong images from the four-wide alpha-cut counting instances (one run each):
#   x_unfixed_count0.so  counting instances at the plain instances' register budget, clamp of 1/d hand-written
#                        (v_rcp_f32 read by the next instruction in k_trace<true,16,true,false,true,true>: the
#                        `interior` scene's instance; textured_room's instance has two instructions in between)
#   x_fixed_count0.so    the same budget, clamp as v_med3_f32
# then the counting tests on the fixed library at the old budget.
# The two libraries are not kept in the tree.  x_fixed_count0.so = `make EXTRA_HIPFLAGS=-DRT_WCOUNT=0` of this source;
# x_unfixed_count0.so = the same with the clamp of 1/d in k_trace's refill written as
# fminI_(fmaxI_(1/d, -1e18f), 1e18f) (rt_kernels_trace.h of commit 9a89d11).  Result: profiles/r03_a_hazard_experiment.txt.
set -o pipefail
L=raytracer_amd/lib
mkdir -p gpurun_out
cp $L/librt_hip.so $L/x_prod.so
cp $L/x_unfixed_count0.so $L/librt_hip.so
echo "== unfixed, old budget (prediction: interior differs, the others are equal)" | tee gpurun_out/r03_hazard.txt
timeout -k 10 300 python3 tools/counting_instances_check.py 2>&1 | tee -a gpurun_out/r03_hazard.txt
rc=$?
[ $rc -gt 1 ] && { cp $L/x_prod.so $L/librt_hip.so; exit $rc; }
cp $L/x_fixed_count0.so $L/librt_hip.so
echo "== fixed, old budget (prediction: all equal)" | tee -a gpurun_out/r03_hazard.txt
timeout -k 10 300 python3 tools/counting_instances_check.py 2>&1 | tee -a gpurun_out/r03_hazard.txt || { cp $L/x_prod.so $L/librt_hip.so; exit 1; }
timeout -k 10 900 python3 -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_wide_tree.py tests/test_gpu_deep_trees.py tests/test_gpu_eight_wide.py -k "count" 2>&1 | tail -5 | tee -a gpurun_out/r03_hazard.txt
rc=$?
cp $L/x_prod.so $L/librt_hip.so
exit $rc

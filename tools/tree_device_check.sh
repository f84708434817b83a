# Device tree build against the host's: tests, then commit time and tracing rate on the default and the interior scene.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_tree_device.py -q --timeout 120 --timeout-method thread 2>&1 | tail -4
B="timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-optin"
for where in host device; do
  RT_TREE_BUILD=$where RT_COMMIT_TIMING=1 $B --steps 3 --warmup 1 > gpurun_out/tree_${where}_default.json 2> gpurun_out/tree_${where}_default.err
  RT_TREE_BUILD=$where RT_COMMIT_TIMING=1 $B --scene interior --width 3840 --height 2160 --spp 16 --of-spp 1024 --steps 3 --warmup 1 > gpurun_out/tree_${where}_interior.json 2> gpurun_out/tree_${where}_interior.err
done
python3 - <<'PY'
import json
for w in ("host","device"):
    for s in ("default","interior"):
        d=json.load(open(f"gpurun_out/tree_{w}_{s}.json"))
        print(w, s, d["value"], d["roofline"]["nodes_per_ray"], d["roofline"]["tris_per_ray"], d.get("scene_upload_commit_s"))
PY
grep "rt_commit_scene" gpurun_out/tree_device_interior.err | tail -44

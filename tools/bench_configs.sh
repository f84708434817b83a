# The other bench configurations on the library's default trees: one JSON line each under gpurun_out/.
# Usage: bash tools/bench_configs.sh <tag>
set -e
B="timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-optin"
$B --scene prism --width 512 --height 512 --spp 16 --spectral > gpurun_out/$1_bench_cfg1_prism_spectral.json
$B --scene knot_glass --spectral --of-spp 256 > gpurun_out/$1_bench_cfg3_knot_glass_spectral.json
$B --scene interior --width 3840 --height 2160 --spp 16 --of-spp 1024 > gpurun_out/$1_bench_cfg4_interior_4k.json
$B --nu 4096 --nv 512 --of-spp 4096 > gpurun_out/$1_bench_cfg5_4M_triangles.json
$B --scene knot_glass --spectral --hero --of-spp 256 > gpurun_out/$1_bench_cfg3_knot_glass_hero.json
python3 - "$1" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_bench_cfg*.json")):
    d = json.load(open(f))
    print(f, d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["nodes_per_ray"], d["roofline"]["tris_per_ray"])
PY

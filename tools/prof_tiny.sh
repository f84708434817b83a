set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tiny
cat > gpurun_out/tiny.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from raytracer_amd import capi
scene = capi.Scene("cornell_knot", nu=64, nv=12)
cam = capi.make_camera(width=16, height=16, bounces=2, ray_per_pixel=1)
be = capi.Backend(0)
be.upload_scene(scene); be.resize(16, 16); be.set_frame(cam, scene.sky, capi.RT_TYPE_RGB)
for f in range(20):
    be.render(first_frame=1 + f, n_frames=1, stride=1)
be.sync(); be.close()
PY
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tiny -- python3 gpurun_out/tiny.py > gpurun_out/prof_tiny.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_tiny/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
for r in rows[-14:]:
    print(r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us')
PY

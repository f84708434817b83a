# A/B of backend builds in the one-frame-per-call regime: tools/ab_interactive.sh v_a v_b ... (files raytracer_amd/lib/<name>.so)
set -e
cd raytracer_amd/lib && cp librt_hip.so base.so && cd ../..
for v in "$@"; do
  cp raytracer_amd/lib/$v.so raytracer_amd/lib/librt_hip.so
  echo "== $v"
  timeout -k 10 200 python3 tools/interactive_rate.py 2>/dev/null
done
cp raytracer_amd/lib/base.so raytracer_amd/lib/librt_hip.so

# A/B of backend builds: tools/ab_libs.sh v_a v_b ...  (files raytracer_amd/lib/<name>.so; "base" = the built library)
set -e
cd raytracer_amd/lib && cp librt_hip.so base.so && cd ../..
for v in "$@"; do
  cp raytracer_amd/lib/$v.so raytracer_amd/lib/librt_hip.so
  echo -n "$v: "
  timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optin $AB_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], 'ms/step', d['ms_per_step'], 'trace-only', r['trace_only_mrays_per_s'], 'trace share', r['trace_fraction_of_step'])"
done
cp raytracer_amd/lib/base.so raytracer_amd/lib/librt_hip.so

# A/B of environment knobs: tools/ab_env.sh "VAR=a" "VAR=b" ...   (extra bench args in AB_ARGS)
set -e
for v in "$@"; do
  echo -n "$v: "
  env $v timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optin $AB_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], 'ms/step', d['ms_per_step'], 'trace-only', r['trace_only_mrays_per_s'], 'trace share', r['trace_fraction_of_step'])"
done

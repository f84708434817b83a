# Batch-capacity sweep of the default bench (paths in flight per batch).
set -e
for p in 4194304 8388608 16777216 33554432 67108864; do
  echo "RT_MAX_PATHS=$p"
  RT_MAX_PATHS=$p timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optin | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['trace_only_mrays_per_s'], d['roofline']['trace_fraction_of_step'], d['roofline']['launches'])"
done

# Refill-threshold / phase-share sweep of the default bench.
set -e
for r in 16 24 32 40 48; do for s in 40 48 56; do
  echo -n "refill=$r share=$s: "
  RT_REFILL_LANES=$r RT_INNER_SHARE=$s timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optin 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['trace_only_mrays_per_s'])"
done; done

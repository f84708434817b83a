for sh in 36 42 48 54; do for rf in 32 40 48; do
echo -n "inner=$sh refill=$rf: "
RT_INNER_SHARE=$sh RT_REFILL_LANES=$rf timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-optin 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], 'ms/step', d['ms_per_step'], 'trace-only', r['trace_only_mrays_per_s'])"
done; done

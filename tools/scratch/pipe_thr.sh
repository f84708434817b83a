B="timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-optin"
for p in 1 0; do
echo -n "prism RT_PIPELINE=$p: "; RT_PIPELINE=$p $B --scene prism --width 512 --height 512 --spp 16 --spectral 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
echo -n "knot 1080p 1spp RT_PIPELINE=$p: "; RT_PIPELINE=$p $B --spp 1 --steps 60 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
echo -n "knot 1080p 2spp RT_PIPELINE=$p: "; RT_PIPELINE=$p $B --spp 2 --steps 60 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
echo -n "knot 720p 1spp RT_PIPELINE=$p: "; RT_PIPELINE=$p $B --width 1280 --height 720 --spp 1 --steps 60 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done

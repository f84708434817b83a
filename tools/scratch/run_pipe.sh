echo "== pipeline on"; timeout -k 10 200 python3 tools/interactive_rate.py 2>/dev/null
echo "== pipeline off"; RT_PIPELINE=0 timeout -k 10 200 python3 tools/interactive_rate.py 2>/dev/null

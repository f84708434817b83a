for b in 7 4 2 1; do echo "blocks/CU=$b"; RT_TRACE_BLOCKS_PER_CU=$b timeout -k 10 120 python3 tools/interactive_rate.py 2>/dev/null | head -2; done

"""Summarises rocprofv3 --pmc CSVs collected by tools/pmc_trace.sh: one line per dispatch of the chosen kernel."""
import collections, csv, glob, sys
tag, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'k_trace')
for kind in ['sq', 'sq2', 'tc']:
    fs = glob.glob(f'gpurun_out/pmc_{tag}_{kind}/*/*_counter_collection.csv')
    if not fs: continue
    by = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if kernel not in r['Kernel_Name']: continue
        e = by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'][:50], 'ms': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'vgpr': r['VGPR_Count']})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    print(tag, kind)
    ks = [k for k in by if '<true' not in by[k]['name']][:7]
    for k in ks:
        e = by[k]; extra = ''
        if 'SQ_THREAD_CYCLES_VALU' in e: extra = ' VALUutil=%.1f%% VALUbusy=%.0f%%' % (100 * e['SQ_THREAD_CYCLES_VALU'] / (e['SQ_ACTIVE_INST_VALU'] * 64), 100 * e['SQ_ACTIVE_INST_VALU'] * 4 / (e['SQ_BUSY_CYCLES'] / 32 * 1024))
        if 'TCC_HIT_sum' in e: extra = ' L2hit=%.1f%% L1hit=%.1f%% L2missGB/s=%.0f' % (100 * e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum']), 100 * (1 - e['TCP_TCC_READ_REQ_sum'] / e['TCP_TOTAL_CACHE_ACCESSES_sum']), e['TCC_MISS_sum'] * 128 / e['ms'] / 1e6)
        print('   ms=%.3f vgpr=%s' % (e['ms'], e['vgpr']), {a: int(b) for a, b in e.items() if a not in ('name', 'ms', 'vgpr')}, extra)

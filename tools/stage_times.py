"""Prints per-launch kernel durations of the first full batch from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows]
gen = [i for i, (n, _) in enumerate(names) if 'k_generate' in n]
i0 = gen[1] if len(gen) > 1 else gen[0]
i1 = gen[2] if len(gen) > 2 else len(names)
tot = {}
for n, t in names[i0:i1]:
    short = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
    print("   %-18s %8.3f ms" % (short, t))
    tot[short] = tot.get(short, 0) + t
print({k: round(v, 3) for k, v in tot.items()}, 'batch total %.3f' % sum(tot.values()))

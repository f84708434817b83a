set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_inter
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_inter -- python3 tools/interactive_rate.py > gpurun_out/prof_inter.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_inter/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# take frames 100..101 of the first scene: find k_generate launches
gen = [i for i, r in enumerate(rows) if 'k_generate' in r['Kernel_Name']]
i0, i1 = gen[100], gen[101]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
    print("%-14s start %8.1f us  dur %7.1f us  gap %6.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
print("frame total %.1f us" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
PY

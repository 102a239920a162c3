"""Per-kernel time of ONE estimate (the last of the ten in scripts/prof_estimate.py's trace) and the idle time between kernels."""
import csv, glob, collections, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_estimate') + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 10
last = rows[-n:]
busy = collections.defaultdict(lambda: [0, 0.0])
t0, t1 = int(last[0]['Start_Timestamp']), int(last[-1]['End_Timestamp'])
gap = 0.0
for a, b in zip(last, last[1:]):
    gap += max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
for r in last:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    busy[k][0] += 1; busy[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('one estimate: %d kernels, %.1f us first start -> last end, %.1f us idle between kernels' % (n, (t1 - t0) / 1e3, gap))
for k, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print('%-62s %4d launches %8.1f us' % (k, c, t))

"""Per-kernel averages from the kernel trace of scripts/prof_defocus.py (20 calls x {1080p, 4K} x {random depth, smooth depth})."""
import csv, glob, collections, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_defocus') + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'].split('(')[0][:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = [0, 0, 0, 0]
for k, v in d.items():
    if len(v) >= 80:
        a = [sum(v[i * 20 + 3:i * 20 + 20]) / 17 for i in range(4)]
        tot = [x + y for x, y in zip(tot, a)]
        print('%-22s 1080p random %.1f smooth %.1f us | 4k random %.1f smooth %.1f us' % (k, a[0], a[1], a[2], a[3]))
print('%-22s 1080p random %.1f smooth %.1f us | 4k random %.1f smooth %.1f us' % ('total', *tot))

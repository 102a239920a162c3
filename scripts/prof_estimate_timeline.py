"""Timeline of ONE estimate (the last of scripts/prof_estimate.py's ten): consecutive launches of one kernel merged into a line with
count, total busy time, and the idle time inside / before the group."""
import csv, glob, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_estimate') + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 10
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
groups = []
prev_end = None
for r in last:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:58]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = r.get('Grid_Size_X', r.get('Grid_Size', '?')); wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
    key = (k, g, wg)
    gap = 0 if prev_end is None else max(0, s - prev_end)
    if groups and groups[-1]['key'] == key:
        groups[-1]['n'] += 1; groups[-1]['busy'] += e - s; groups[-1]['gap_in'] += gap; groups[-1]['end'] = e
    else:
        groups.append(dict(key=key, n=1, busy=e - s, gap_before=gap, gap_in=0, start=s, end=e))
    prev_end = e
print('one estimate: %d kernels, %.1f us' % (n, (int(last[-1]['End_Timestamp']) - t0) / 1e3))
for g in groups:
    k, grid, wg = g['key']
    print('%8.1f us  +%5.1f gap | %-58s grid %-7s wg %-5s x%-3d busy %7.1f us  idle inside %5.1f  (%.2f us/launch)' %
          ((g['start'] - t0) / 1e3, g.get('gap_before', 0) / 1e3, k, grid, wg, g['n'], g['busy'] / 1e3, g['gap_in'] / 1e3, (g['busy'] + g['gap_in']) / 1e3 / g['n']))

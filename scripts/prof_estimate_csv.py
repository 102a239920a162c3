"""CSV summary of ONE estimate out of scripts/prof_estimate.py's kernel trace (the last whole one): per kernel name -- launches, total
and mean duration -- plus the first-start-to-last-end span and the idle time between kernels."""
import csv, glob, collections, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_estimate_r3') + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# an estimate starts at its first k_pyrdown_annotation and ends with the u8 map (k_finish4 of the finest level); take the last whole one
starts = [i for i, r in enumerate(rows) if 'k_pyrdown_annotation' in r['Kernel_Name'] and (i == 0 or 'k_pyrdown_annotation' not in rows[i - 1]['Kernel_Name'])]
a, b = starts[-2], starts[-1]
est = [r for r in rows[a:b] if 'copyBuffer' not in r['Kernel_Name']]
busy = collections.OrderedDict()
gap = 0
for x, y in zip(est, est[1:]):
    gap += max(0, int(y['Start_Timestamp']) - int(x['End_Timestamp']))
for r in est:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    e = busy.setdefault(k, [0, 0])
    e[0] += 1; e[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('kernel,launches,total_us,mean_us')
for k, (n, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.1f,%.2f' % (k, n, t / 1e3, t / 1e3 / n))
print('"(span: first start -> last end of one estimate, under the profiler)",%d,%.1f,' % (len(est), (int(est[-1]['End_Timestamp']) - int(est[0]['Start_Timestamp'])) / 1e3))
print('"(idle between kernels)",,%.1f,' % (gap / 1e3))

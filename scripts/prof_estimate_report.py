"""ONE estimate out of scripts/prof_estimate.py's kernel trace -- the last whole one -- as a timeline AND as a per-kernel summary, both
cut from the same launches so that they agree: a warm estimate (annotation unchanged: no annotation kernels) runs from the coarsest
level's k_prepare launch to the k_finish launch that writes the finest level and the u8 map (the only k_finish of an estimate: the
coarser levels' copy-back is k_pyrup_inject's).
usage: prof_estimate_report.py DIR OUT_PREFIX   -> OUT_PREFIX_timeline.txt, OUT_PREFIX_kernel_trace_summary.csv"""
import collections, csv, glob, sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'rtdd::' in r['Kernel_Name']]
ends = [i for i, r in enumerate(rows) if 'rtdd::k_finish' in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
est = rows[a:b]
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
t0 = int(est[0]['Start_Timestamp'])
span = (int(est[-1]['End_Timestamp']) - t0) / 1e3
gap = sum(max(0, int(y['Start_Timestamp']) - int(x['End_Timestamp'])) for x, y in zip(est, est[1:])) / 1e3

groups, prev_end = [], None
for r in est:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    key = (name(r)[:58], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
    g = 0 if prev_end is None else max(0, s - prev_end)
    if groups and groups[-1]['key'] == key:
        groups[-1]['n'] += 1; groups[-1]['busy'] += e - s; groups[-1]['gap_in'] += g
    else:
        groups.append(dict(key=key, n=1, busy=e - s, gap_before=g, gap_in=0, start=s))
    prev_end = e
with open(out + '_timeline.txt', 'w') as o:
    o.write('one warm 1080p estimate (the last of ten, under rocprofv3 --kernel-trace): %d kernels, first start -> last end %.1f us, idle between kernels %.1f us\n' % (len(est), span, gap))
    for g in groups:
        k, grid, wg = g['key']
        o.write('%8.1f us  +%5.1f gap | %-58s grid %-7s wg %-5s x%-3d busy %7.1f us  idle inside %5.1f  (%.2f us/launch)\n' %
                ((g['start'] - t0) / 1e3, g['gap_before'] / 1e3, k, grid, wg, g['n'], g['busy'] / 1e3, g['gap_in'] / 1e3, (g['busy'] + g['gap_in']) / 1e3 / g['n']))
busy = collections.OrderedDict()
for r in est:
    e = busy.setdefault(name(r), [0, 0])
    e[0] += 1; e[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open(out + '_kernel_trace_summary.csv', 'w') as o:
    o.write('kernel,launches,total_us,mean_us\n')
    for k, (n, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        o.write('"%s",%d,%.1f,%.2f\n' % (k, n, t / 1e3, t / 1e3 / n))
    o.write('"(span: first start -> last end of the same estimate as the timeline, under the profiler)",%d,%.1f,\n' % (len(est), span))
    o.write('"(idle between kernels)",,%.1f,\n' % gap)
print(open(out + '_timeline.txt').read())

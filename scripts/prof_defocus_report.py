"""Report of the defocus effect's profile (scripts/r3_defocus_prof.sh: one kernel-trace pass and one --pmc pass each for FETCH_SIZE and
WRITE_SIZE of scripts/prof_defocus.py -- 20 calls per case, automatic path): per case and kernel the mean duration and counter
values per launch (first 3 calls of a case dropped), per call the HBM-side bytes against the 10 B/px algorithmic figure (FETCH_SIZE
doubled as MI355X_MICROARCH.md prescribes for wide reads: an upper bound for the 8-byte gathers).  A call is told from the trace by
its last kernel (k_defocus or k_defocus_tile); calls map to cases in order.
usage: prof_defocus_report.py DIR [--json]"""
import csv, glob, collections, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_defocus_r3'
cases = ['1080p_random_depth', '1080p_smooth_depth', '4k_random_depth', '4k_smooth_depth']
px = {'1080p': 1080 * 1920, '4k': 2160 * 3840}

def calls(path, field):
    rows = [r for r in csv.DictReader(open(path)) if 'rtdd::k_' in r['Kernel_Name']]
    key = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else 'Dispatch_Id'
    rows.sort(key=lambda r: int(r[key]))
    out, cur = [], []
    for r in rows:
        name = r['Kernel_Name'].replace('void ', '').split('(')[0].split('<')[0].replace('rtdd::', '')
        val = float(r[field]) if field else (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        cur.append((name, val))
        if name in ('k_defocus', 'k_defocus_tile'):
            out.append(cur); cur = []
    return out

def per_case(path, field):
    cs = calls(path, field)
    assert len(cs) == 80, f'{path}: {len(cs)} calls, expected 80'
    res = []
    for i in range(4):
        acc = collections.OrderedDict()
        for c in cs[i * 20 + 3:i * 20 + 20]:
            for name, val in c: acc.setdefault(name, []).append(val)
        res.append({k: sum(v) / len(v) for k, v in acc.items()})
    return res

dur = per_case(glob.glob(f'{d}/**/df_kernel_trace.csv', recursive=True)[0], None)
fetch = per_case(glob.glob(f'{d}/**/fetch_counter_collection.csv', recursive=True)[0], 'Counter_Value')
write = per_case(glob.glob(f'{d}/**/write_counter_collection.csv', recursive=True)[0], 'Counter_Value')
out = {'command': 'rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python3 scripts/prof_defocus.py (20 calls per case, first 3 dropped)',
       'units': 'us; FETCH_SIZE / WRITE_SIZE in KB per launch as reported', 'cases': {}}
for i, c in enumerate(cases):
    ks = {k: {'us': round(dur[i][k], 2), 'FETCH_SIZE_KB': round(fetch[i].get(k, 0), 1), 'WRITE_SIZE_KB': round(write[i].get(k, 0), 1)} for k in dur[i]}
    us = sum(dur[i].values()); f = sum(fetch[i].values()); w = sum(write[i].values())
    algo = 10.0 * px[c.split('_')[0]]
    out['cases'][c] = {'kernels': ks, 'kernel_us_sum': round(us, 1), 'algorithmic_bytes': algo,
                       'hbm_bytes_fetch_x1_plus_write': (f + w) * 1024, 'hbm_bytes_fetch_x2_plus_write': (2 * f + w) * 1024,
                       'ratio_to_algorithmic_x1': round((f + w) * 1024 / algo, 2), 'ratio_to_algorithmic_x2': round((2 * f + w) * 1024 / algo, 2),
                       'frac_of_hbm_peak_on_algorithmic_bytes': round(algo / (us * 1e-6) / 8e12, 4)}
if '--json' in sys.argv:
    print(json.dumps(out, indent=1))
else:
    for c, v in out['cases'].items():
        print(f"{c}: {v['kernel_us_sum']} us in kernels, HBM-side bytes {v['ratio_to_algorithmic_x1']}x (FETCH doubled: {v['ratio_to_algorithmic_x2']}x) the algorithmic 10 B/px")
        for k, kv in v['kernels'].items():
            print(f"    {k:16s} {kv['us']:8.2f} us   FETCH {kv['FETCH_SIZE_KB']:9.1f} KB   WRITE {kv['WRITE_SIZE_KB']:9.1f} KB")

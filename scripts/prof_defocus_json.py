"""JSON summary of the defocus pipeline's profile (scripts/r3_defocus_prof.sh): per kernel and case -- mean duration from the kernel
trace, FETCH_SIZE and WRITE_SIZE per launch from their --pmc passes -- and the HBM-side bytes per call against the 10 B/px algorithmic
figure (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide reads: an upper bound for the 8-byte gathers)."""
import csv, glob, collections, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_defocus_r3'
cases = ['1080p_random_depth', '1080p_smooth_depth', '4k_random_depth', '4k_smooth_depth']
px = {'1080p': 1080 * 1920, '4k': 2160 * 3840}
def per(path, field):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('void ', '').split('(')[0]
        out[k].append(float(r[field]) if field else (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return {k: [sum(v[i * 20 + 3:i * 20 + 20]) / 17 for i in range(4)] for k, v in out.items() if len(v) >= 80}
dur = per(glob.glob(f'{d}/**/df_kernel_trace.csv', recursive=True)[0], None)
fetch = per(glob.glob(f'{d}/**/fetch_counter_collection.csv', recursive=True)[0], 'Counter_Value')
write = per(glob.glob(f'{d}/**/write_counter_collection.csv', recursive=True)[0], 'Counter_Value')
out = {'command': 'rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python3 scripts/prof_defocus.py (20 calls per case, first 3 dropped)',
       'units': 'us; FETCH_SIZE / WRITE_SIZE in KB per launch as reported', 'kernels': {}, 'per_call': {}}
for k in dur:
    out['kernels'][k] = {c: {'us': round(dur[k][i], 2), 'FETCH_SIZE_KB': round(fetch.get(k, [0] * 4)[i], 1), 'WRITE_SIZE_KB': round(write.get(k, [0] * 4)[i], 1)} for i, c in enumerate(cases)}
for i, c in enumerate(cases):
    us = sum(dur[k][i] for k in dur); f = sum(fetch.get(k, [0] * 4)[i] for k in dur); w = sum(write.get(k, [0] * 4)[i] for k in dur)
    algo = 10.0 * px[c.split('_')[0]]
    out['per_call'][c] = {'kernel_us_sum': round(us, 1), 'algorithmic_bytes': algo, 'hbm_bytes_fetch_x1_plus_write': (f + w) * 1024, 'hbm_bytes_fetch_x2_plus_write': (2 * f + w) * 1024,
                          'ratio_to_algorithmic_x1': round((f + w) * 1024 / algo, 2), 'ratio_to_algorithmic_x2': round((2 * f + w) * 1024 / algo, 2),
                          'frac_of_hbm_peak_on_algorithmic_bytes': round(algo / (us * 1e-6) / 8e12, 4)}
print(json.dumps(out, indent=1))

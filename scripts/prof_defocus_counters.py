"""FETCH_SIZE / WRITE_SIZE per kernel of the defocus pipeline from the --pmc passes of scripts/r3_defocus_prof.sh.
Units and the gfx950 correction as MI355X_MICROARCH.md prescribes: both counters are in KiB-like units of 1 KB... see below."""
import csv, glob, collections, sys, json
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_defocus_r3'
out = {}
for name in ('fetch', 'write'):
    fs = glob.glob(f'{d}/**/{name}_counter_collection.csv', recursive=True)
    if not fs:
        print('no', name, 'pass'); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        per[r['Kernel_Name'].split('(')[0][:40]].append(float(r['Counter_Value']))
    for k, v in per.items():
        if len(v) >= 80:
            a = [sum(v[i * 20 + 3:i * 20 + 20]) / 17 for i in range(4)]
            out.setdefault(k, {})[name] = a
for k, v in out.items():
    print('%-22s' % k, {n: ['%.0f' % x for x in a] for n, a in v.items()})
json.dump(out, open(f'{d}/counters_raw.json', 'w'), indent=1)

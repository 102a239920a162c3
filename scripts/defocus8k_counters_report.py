"""Summary of scripts/defocus8k_counters.sh: per configuration and kernel the mean duration, launches per call and FETCH_SIZE / WRITE_SIZE
per call; HBM-side bytes per call = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 B) against the
algorithmic 10 B/px."""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
px = 4320 * 7680
calls = 10
out = {"command": "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python3 scripts/prof_defocus8k.py <strips> <slice MB> (7680 x 4320, smooth depth map, 10 calls)",
       "corrections": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x1, KB -> bytes x1024", "algorithmic_bytes_per_call": 10.0 * px, "configurations": {}}

def name_of(r):
    return r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0].replace("rtdd::", "")

for cfg in ("rowbands", "strips", "slices64"):
    tr = glob.glob(os.path.join(d, cfg, "**", "t_kernel_trace.csv"), recursive=True)[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(tr)):
        if "rtdd::k_" in r["Kernel_Name"]:
            dur[name_of(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = {}
    for tag, field in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
        path = glob.glob(os.path.join(d, cfg, "**", f"{tag}_counter_collection.csv"), recursive=True)[0]
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if "rtdd::k_" in r["Kernel_Name"] and r.get("Counter_Name", field) == field:
                acc[name_of(r)] += float(r["Counter_Value"])
        cnt[field] = acc
    ks = {}
    for k, v in dur.items():
        ks[k] = {"launches_per_call": len(v) / calls, "us_per_call": round(sum(v) / calls, 1), "FETCH_SIZE_KB_per_call": round(cnt["FETCH_SIZE"][k] / calls, 1),
                 "WRITE_SIZE_KB_per_call": round(cnt["WRITE_SIZE"][k] / calls, 1)}
    f = sum(v["FETCH_SIZE_KB_per_call"] for v in ks.values()); w = sum(v["WRITE_SIZE_KB_per_call"] for v in ks.values())
    us = sum(v["us_per_call"] for v in ks.values())
    out["configurations"][cfg] = {"kernels": ks, "kernel_us_per_call": round(us, 1), "hbm_bytes_per_call_corrected": (2 * f + w) * 1024,
                                  "ratio_to_algorithmic": round((2 * f + w) * 1024 / (10.0 * px), 2), "hbm_GBs_over_kernel_time": round((2 * f + w) * 1024 / (us * 1e-6) / 1e9, 1)}
print(json.dumps(out, indent=1))

"""Turn the rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/prof_<workload>/) into the committed summaries under profiles/."""
import collections, csv, json, sys
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_1080p_jacobi1000"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01_1080p"
workload = sys.argv[3] if len(sys.argv) > 3 else "1080p_jacobi1000"
res = {}
for name, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f"{out}/{name}_counter_collection.csv")):
        if r["Counter_Name"] == cn and "rtdd::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]; acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        res.setdefault(k, {})[cn + "_KB_per_launch"] = acc[k] / n[k]; res[k]["launches_" + cn] = n[k]
for k, v in res.items():
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly half of a wide coalesced stream on gfx950 -> x2; WRITE_SIZE exact
    v["hbm_bytes_per_launch_corrected"] = (2 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
d = {"workload": workload, "source": f"profiles/{tag}_pmc_traffic.json",
     "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
     "correction": "FETCH_SIZE x2 (gfx950 under-reports wide coalesced reads by 2x), WRITE_SIZE x1, unit KB -> bytes x1024",
     "note": "1080p: the sweep kernel is k_sweep_blocked<32, 1024, 3, true, true> (persistent: ONE launch = 1000 sweeps); 4K/8K: k_sweep_blocked<16, 512, 3, true, false>, one launch = 8 sweeps; <16,1024,1,..> rows are the small pyramid levels of the estimate leg",
     "kernels": res}
json.dump(d, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
if workload == "1080p_jacobi1000":           # bench.py looks the default workload up here
    json.dump(d, open("profiles/traffic_latest.json", "w"), indent=1)
stats = open(f"{out}/trace_kernel_stats.csv").read().splitlines()
open(f"profiles/{tag}_kernel_stats.csv", "w").write("\n".join(l[:420] for l in stats[:14]) + "\n")
for k, v in res.items():
    if "sweep" in k: print(k, {a: round(b, 1) for a, b in v.items()})
print("\n".join(l[:200] for l in stats[:6]))

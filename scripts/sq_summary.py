"""Summarise the rocprofv3 --pmc passes of scripts/profile_sq.sh: per kernel, counters per launch, mean duration, and the
VALU-issue fraction   SQ_INSTS_VALU x 2 cycles / (duration x 1024 SIMD-32 x 2.4 GHz)   (a wave64 VALU instruction occupies its
SIMD-32 for 2 cycles; MI355X_MICROARCH.md 'Wave scheduling').  usage: sq_summary.py DIR  -> JSON on stdout"""
import collections, csv, glob, json, os, sys

d = sys.argv[1]
SIMDS, CLOCK = 256 * 4, 2.4e9
out = {"source": d, "valu_frac_definition": "SQ_INSTS_VALU * 2 / (mean_duration_s * 1024 SIMDs * 2.4e9 Hz)", "runs": {}}
for p1 in sorted(glob.glob(os.path.join(d, "*_p1_counter_collection.csv"))):
    name = os.path.basename(p1)[: -len("_p1_counter_collection.csv")]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in (p1, p1.replace("_p1_", "_p2_")):
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if "rtdd::" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    dur = collections.defaultdict(list)
    kt = p1.replace("_counter_collection", "_kernel_trace")
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    run = {}
    for k, v in acc.items():
        per = {c: val / max(len(disp[(k, c)]), 1) for c, val in v.items()}
        ds = dur.get(k, [])
        mean = sum(ds) / len(ds) if ds else None
        e = {"launches": len(ds), "mean_duration_us_under_pmc": mean * 1e6 if mean else None, "per_launch": {c: round(x, 1) for c, x in sorted(per.items())}}
        if mean and "SQ_INSTS_VALU" in per:
            e["valu_issue_frac"] = per["SQ_INSTS_VALU"] * 2 / (mean * SIMDS * CLOCK)
        if "SQ_WAVE_CYCLES" in per and per["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
                if c in per:
                    e[c + "/SQ_WAVE_CYCLES"] = per[c] / per["SQ_WAVE_CYCLES"]
        run[k] = e
    # keep the heavy kernels only
    tot = {k: (e["mean_duration_us_under_pmc"] or 0) * e["launches"] for k, e in run.items()}
    keep = sorted(tot, key=tot.get, reverse=True)[:6]
    out["runs"][name] = {k: run[k] for k in keep}
print(json.dumps(out, indent=1))

"""Turn the rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/prof_<round>_<workload>/) into the committed summaries under
profiles/:

  <round>_<workload>_kernel_stats.csv        per-kernel statistics over the TIMED steps of the driver's command only (the last --steps
                                             solves of the clean --kernel-trace pass; a solve starts at its k_prepare launch): bench.py
                                             runs ~0.15 s of untimed clock-ramp solves and the warm-up steps in front of them, which
                                             rocprofv3's own --stats summary cannot tell apart
  <round>_<workload>_kernel_stats_all_launches.csv   that --stats summary as rocprofv3 wrote it (every launch of the process)
  <round>_<workload>_counters.json           per kernel: launches, mean duration, HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE,
                                             KB -> bytes (MI355X_MICROARCH.md section HBM); SQ counters per launch; VALU issue fraction;
                                             per-SOLVE totals divided by the number of solves COUNTED in each pass's own trace
  profiles/counters_latest.json[workload]    the entry bench.py attaches to its roofline record (dominant kernel)

usage: make_counters_json.py ROUND WORKLOAD [STEPS=20]"""
import collections, csv, json, math, os, sys

rnd, wl = sys.argv[1], sys.argv[2]
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d = f"gpurun_out/prof_{rnd}_{wl}"
SIMDS, CLOCK = 256 * 4, 2.4e9
# VALU operations per useful pixel-sweep: COUNTED in the built kernel (scripts/isa_count.py) and required to equal the constant bench.py
# falls back to -- a record made from a build whose loop no longer is what the roofline prices is refused
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isa_count
_count = isa_count.sweep_pair()
VALU_OPS = _count["per_pixel_sweep"]
import importlib.util
_spec = importlib.util.spec_from_file_location("bench_consts", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
_bench = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_bench)
if abs(VALU_OPS - _bench.VALU_OPS["jacobi"]) > 0.05:
    sys.exit(f"make_counters_json: the built kernel issues {VALU_OPS:.2f} VALU operations per pixel-sweep ({_count['valu']} per sweep pair), bench.py prices {_bench.VALU_OPS['jacobi']}: update bench.py VALU_OPS first")
VALU_PEAK = 256 * 4 * 32 * 2.4e9


def kname(full):
    return full.replace("(anonymous namespace)::", "").split("(")[0]


def is_prepare(k):
    return "rtdd::k_prepare" in k


def trace_rows(prefix):
    f = f"{d}/{prefix}_kernel_trace.csv"
    if not os.path.exists(f):
        return []
    return sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def solves_in(prefix):
    return sum(1 for r in trace_rows(prefix) if is_prepare(kname(r["Kernel_Name"])))


TOTALS = {}          # prefix -> counter -> sum over every rtdd kernel launch of the pass


def per_kernel(prefix):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    TOTALS[prefix] = collections.defaultdict(float)
    f = f"{d}/{prefix}_counter_collection.csv"
    if not os.path.exists(f):
        return {}
    for r in csv.DictReader(open(f)):
        k = kname(r["Kernel_Name"])
        if "rtdd::" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
        TOTALS[prefix][r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: {c: v / len(disp[(k, c)]) for c, v in cs.items()} for k, cs in acc.items()}


def durations(prefix, timed_only=False):
    rows = trace_rows(prefix)
    if timed_only:
        starts = [i for i, r in enumerate(rows) if is_prepare(kname(r["Kernel_Name"]))]
        if len(starts) >= STEPS:
            rows = rows[starts[-STEPS]:]
    out = collections.defaultdict(list)
    for r in rows:
        out[kname(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return out, rows


clean_all, _ = durations("trace")
clean, timed_rows = durations("trace", timed_only=True)
n_solves_all = solves_in("trace")
kernels = {}
fetch, write, sq1, sq2 = per_kernel("fetch"), per_kernel("write"), per_kernel("sq1"), per_kernel("sq2")
dur_sq1, _ = durations("sq1")
for k, ds in clean.items():
    if "rtdd::" not in k:
        continue
    e = {"launches_in_timed_steps": len(ds), "mean_duration_us": sum(ds) / len(ds) * 1e6, "min_us": min(ds) * 1e6, "max_us": max(ds) * 1e6, "total_ms": sum(ds) * 1e3,
         "launches_in_whole_trace": len(clean_all.get(k, [])), "mean_duration_us_whole_trace": sum(clean_all[k]) / len(clean_all[k]) * 1e6}
    fs, ws = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
    if fs is not None and ws is not None:
        e["FETCH_SIZE_KB_per_launch"] = fs; e["WRITE_SIZE_KB_per_launch"] = ws
        e["hbm_bytes_per_launch_corrected"] = (2 * fs + ws) * 1024
    sq = dict(sq1.get(k, {})); sq.update(sq2.get(k, {}))
    if sq:
        e["sq_per_launch"] = {c: round(v, 1) for c, v in sorted(sq.items())}
        dd = dur_sq1.get(k)
        if dd and "SQ_INSTS_VALU" in sq:
            e["mean_duration_us_under_sq_pass"] = sum(dd) / len(dd) * 1e6
            e["valu_issue_frac_counted"] = sq["SQ_INSTS_VALU"] * 2 / (sum(dd) / len(dd) * SIMDS * CLOCK)
        if sq.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in sq:
                    e[c + "/SQ_WAVE_CYCLES"] = sq[c] / sq["SQ_WAVE_CYCLES"]
    kernels[k] = e
bench = {}
for name in ("bench_unprofiled.json", "trace.json"):
    try:
        bench[name] = json.loads([l for l in open(f"{d}/{name}").read().splitlines() if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        pass
span_ms = (int(timed_rows[-1]["End_Timestamp"]) - int(timed_rows[0]["Start_Timestamp"])) * 1e-6 if timed_rows else None
summary = {"workload": wl, "round": rnd,
           "commands": {"clean trace": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {STEPS} --warmup 5 --no-cpu-baseline --no-estimate --workload {wl}",
                        "counters": "rocprofv3 --kernel-trace --pmc <one group per pass: FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2> -- python3 bench.py --steps 3 --warmup 1 ... (scripts/profile_round.sh)"},
           "window": f"kernel statistics are over the last {STEPS} solves of the clean trace (= the timed steps; a solve starts at its k_prepare launch); the process ran {n_solves_all} solves in all "
                     "(clock ramp + warm-up + timed)",
           "valu_ops_per_pixel_sweep": VALU_OPS, "valu_ops_source": f"{_count['valu']} VALU instructions on the fall-through path of the sweep-pair loop / 24 (scripts/isa_count.py, counted in the build that was profiled)",
           "corrections": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x1, KB -> bytes x1024; VALU issue fraction = SQ_INSTS_VALU x 2 cycles / (duration x 1024 SIMD-32 x 2.4 GHz)",
           "bench_line_unprofiled": {k: bench.get("bench_unprofiled.json", {}).get(k) for k in ("value", "ms_per_step", "roofline")},
           "ms_per_step_under_kernel_trace": bench.get("trace.json", {}).get("ms_per_step"),
           "timed_steps_span_ms_per_step_in_trace": span_ms / STEPS if span_ms else None,
           "kernels": kernels}
os.makedirs("profiles", exist_ok=True)


def stats_csv(durs, path):
    tot = sum(sum(v) for v in durs.values())
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"']
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        m = sum(v) / len(v)
        sd = math.sqrt(sum((x - m) ** 2 for x in v) / len(v))
        lines.append('"%s",%d,%d,%.6f,%.4f,%d,%d,%.6f' % (k[:400], len(v), round(sum(v) * 1e9), m * 1e9, 100.0 * sum(v) / tot, round(min(v) * 1e9), round(max(v) * 1e9), sd * 1e9))
    open(path, "w").write("\n".join(lines[:17]) + "\n")


stats_csv(clean, f"profiles/{rnd}_{wl}_kernel_stats.csv")
stats = open(f"{d}/trace_kernel_stats.csv").read().splitlines()
open(f"profiles/{rnd}_{wl}_kernel_stats_all_launches.csv", "w").write("\n".join(l[:420] for l in stats[:16]) + "\n")
# the dominant kernel of this workload -> the entry bench.py attaches to its roofline record
dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
latest_path = "profiles/counters_latest.json"
latest = json.load(open(latest_path)) if os.path.exists(latest_path) else {}
n_fetch, n_write = solves_in("fetch"), solves_in("write")
all_bytes = None
if n_fetch and n_write:
    all_bytes = (2 * TOTALS.get("fetch", {}).get("FETCH_SIZE", 0.0) / n_fetch + TOTALS.get("write", {}).get("WRITE_SIZE", 0.0) / n_write) * 1024
summary["solves_counted"] = {"clean trace": n_solves_all, "fetch pass": n_fetch, "write pass": n_write}
summary["hbm_bytes_per_solve_all_kernels_corrected"] = all_bytes
summary["kernel_ms_per_solve_clean_trace"] = sum(k["total_ms"] for k in kernels.values()) / STEPS
cfg = bench.get("trace.json", {}).get("config", {})      # what the profiled command launched: bench.py attaches these counters only to a run of the same kernel
# the roofline fraction recomputed from THIS profile (Jacobi workloads): useful pixel-sweeps per launch x the counted VALU operations per pixel-sweep / mean launch duration / peak
line = bench.get("bench_unprofiled.json", {})
try:
    rows_, cols_ = {"1080p": (1080, 1920), "4k": (2160, 3840), "8k": (4320, 7680)}[wl.split("_")[0]]
    spl = cfg.get("sweeps_per_launch")
    if "jacobi" in wl and spl:
        frac = rows_ * cols_ * spl * VALU_OPS / (kernels[dom]["mean_duration_us"] * 1e-6) / VALU_PEAK
        summary["roofline_frac_recomputed_from_this_profile"] = frac
        summary["roofline_frac_of_the_unprofiled_line"] = (line.get("roofline") or {}).get("frac")
except (KeyError, TypeError):
    pass
json.dump(summary, open(f"profiles/{rnd}_{wl}_counters.json", "w"), indent=1)
latest[wl] = {"tile": cfg.get("tile"), "persistent": cfg.get("persistent"), "temporal_depth": cfg.get("temporal_depth"),
              "hbm_bytes_per_solve_all_kernels_corrected": all_bytes, "kernel_ms_per_solve_clean_trace": summary["kernel_ms_per_solve_clean_trace"], "kernel": dom, "source": f"profiles/{rnd}_{wl}_counters.json", "mean_duration_us": kernels[dom]["mean_duration_us"],
              "hbm_bytes_per_launch_corrected": kernels[dom].get("hbm_bytes_per_launch_corrected"), "valu_issue_frac_counted": kernels[dom].get("valu_issue_frac_counted")}
json.dump(latest, open(latest_path, "w"), indent=1)
print(dom, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in kernels[dom].items() if k != "sq_per_launch"})
print("frac from profile", summary.get("roofline_frac_recomputed_from_this_profile"), "line", summary.get("roofline_frac_of_the_unprofiled_line"),
      "| per solve: %.3f ms, %s bytes" % (summary["kernel_ms_per_solve_clean_trace"], all_bytes))

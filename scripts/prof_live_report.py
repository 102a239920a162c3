"""Timeline of the last pipelined live frames from a rocprofv3 --kernel-trace --memory-copy-trace run of scripts/prof_live.py:
per frame (from its first D2D staging copy / first kernel to its k_finish4), the kernels' busy time, the gaps, and what runs between frames.
usage: prof_live_report.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv>"""
import csv, glob, os, sys
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
ev = []
for r in rows("*kernel_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void rtdd::", "")[:60], r.get("Stream_Id", r.get("Queue_Id", ""))))
for r in rows("*memory_copy_trace.csv"):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy")), r.get("Stream_Id", "")))
ev.sort()
# frames: a k_finish4 ends a frame's compute
ends = [i for i, e in enumerate(ev) if "k_finish4" in e[2]]
print(f"{len(ev)} events, {len(ends)} estimates")
if len(ends) < 4:
    sys.exit(0)
lo, hi = ends[-3], ends[-2]               # one steady-state frame: from the event after the previous k_finish4 to this frame's k_finish4
t0 = ev[lo][1]
print(f"frame period (k_finish4 end to k_finish4 end): {(ev[hi][1] - ev[lo][1]) / 1e3:.1f} us; previous: {(ev[lo][1] - ev[ends[-4]][1]) / 1e3:.1f} us")
busy = 0; last_end = t0; gaps = []
for s, e, name, st in ev[lo + 1:hi + 1]:
    if name.startswith("K "):
        if s > last_end: gaps.append((s - last_end, name))
        busy += e - s; last_end = max(last_end, e)
print(f"kernel busy {busy / 1e3:.1f} us, idle between kernels {sum(g for g, _ in gaps) / 1e3:.1f} us in {len(gaps)} gaps")
for g, n in sorted(gaps, reverse=True)[:12]:
    print(f"   gap {g / 1e3:7.2f} us before {n}")
print("events of the frame that are not sweep kernels:")
for s, e, name, st in ev[lo + 1:hi + 1]:
    if "k_sweep" not in name:
        print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.2f} us  {name}  [{st}]")

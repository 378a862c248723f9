"""from a rocprofv3 kernel trace of bench.py: how the timed region's wall time splits into scan running / only selection running / only small kernels / idle"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# timed region: steps are delimited by set_online_kernel (one per step); take region [a, b) by dispatch index from argv
ons = [i for i, r in enumerate(rows) if "set_online" in r["Kernel_Name"]]
nsteps_total = len(ons)
first, last = int(sys.argv[2]), int(sys.argv[3])  # e.g. steps -15..-5 : the in-flight timed region of a --steps 5 run is [-10,-5)
lo = int(rows[ons[first]]["Start_Timestamp"]); hi = int(rows[ons[last]]["Start_Timestamp"])
ev = []
def cls(n):
    if "scan_tiles" in n: return 0
    if "replay" in n: return 1
    return 2
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= lo or s >= hi: continue
    c = cls(r["Kernel_Name"])
    ev.append((max(s, lo), 1, c)); ev.append((min(e, hi), -1, c))
ev.sort()
cnt = [0, 0, 0]; t = lo; acc = {"scan": 0, "replay_only": 0, "small_only": 0, "idle": 0, "scan+replay": 0}
for ts, d, c in ev:
    dt = ts - t
    if cnt[0] and cnt[1]: acc["scan+replay"] += dt
    elif cnt[0]: acc["scan"] += dt
    elif cnt[1]: acc["replay_only"] += dt
    elif cnt[2]: acc["small_only"] += dt
    else: acc["idle"] += dt
    cnt[c] += d; t = ts
acc["idle"] += hi - t
tot = hi - lo
print(f"{nsteps_total} steps in trace; region {tot / 1e6:.2f} ms over {last - first} steps = {tot / 1e6 / (last - first):.2f} ms/step")
for k, v in acc.items(): print(f"  {k:12s} {v / 1e6:8.3f} ms  {100 * v / tot:5.1f} %")

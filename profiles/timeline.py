"""print the dispatch timeline of the last bench step from a rocprofv3 --kernel-trace CSV
usage: python profiles/timeline.py <dir with *_kernel_trace.csv> [n_last_dispatches]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sel = rows[-n:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} ms  wg {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):8d}  vgpr {r['VGPR_Count']:>4s} lds {r['LDS_Block_Size']:>6s}  {r['Kernel_Name'][:90]}")

"""Summarise a rocprofv3 --kernel-trace --stats csv dir: per-kernel totals + one decoder position."""
import csv
import glob
import sys

d = sys.argv[1]
stats = glob.glob(d + "/*/*_kernel_stats.csv")[0]
trace = glob.glob(d + "/*/*_kernel_trace.csv")[0]
for r in csv.DictReader(open(stats)):
    n = r["Name"].replace("wt::(anonymous namespace)::", "").replace("void ", "")
    print(f"{n[:64]:64s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
if len(sys.argv) > 2:
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    sel = [i for i, r in enumerate(rows) if "select_token" in r["Kernel_Name"]]
    i0 = sel[len(sel) // 2] + 1
    print("--- one decoder position ---")
    for r in rows[i0:i0 + int(sys.argv[2])]:
        n = r["Kernel_Name"].replace("wt::(anonymous namespace)::", "").replace("void ", "")[:60]
        print(f"{n:60s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us grid {int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))}x{r['Grid_Size_Y']}")

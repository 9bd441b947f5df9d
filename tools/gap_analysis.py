"""Per-queue busy time and launch-to-launch gaps from a rocprofv3 --kernel-trace csv (the pipelined bench run):
python tools/gap_analysis.py gpurun_out/prof_r04_stats  -> for every hardware queue: kernels, busy ms, span ms, and the
distribution of gaps (previous kernel's end -> next kernel's start on the same queue)."""
import csv
import glob
import sys
from collections import defaultdict

import numpy as np

trace = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(trace)) if "spin_ticks" not in r["Kernel_Name"]]
byq = defaultdict(list)
for r in rows:
    byq[(r["Queue_Id"], r["Stream_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in sorted(byq.items()):
    ks.sort()
    # steady-state window: middle 60 % of the queue's kernels
    a, b = len(ks) // 5, len(ks) - len(ks) // 5
    ks = ks[a:b]
    if len(ks) < 50:
        continue
    s = np.array([k[0] for k in ks], np.float64)
    e = np.array([k[1] for k in ks], np.float64)
    gaps = (s[1:] - e[:-1]) / 1e3
    dur = (e - s) / 1e3
    span = (e[-1] - s[0]) / 1e6
    names = defaultdict(int)
    for k in ks:
        names[k[2].split("(")[0].replace("wt::(anonymous namespace)::", "").replace("void ", "")[:28]] += 1
    top = ", ".join(f"{n} x{c}" for n, c in sorted(names.items(), key=lambda t: -t[1])[:3])
    print(f"queue {q[0]} stream {q[1]}: {len(ks)} kernels over {span:.1f} ms, busy {dur.sum() / 1e3:.1f} ms ({100 * dur.sum() / 1e3 / span:.0f} %), "
          f"kernel avg {dur.mean():.1f} us | gap median {np.median(gaps):.1f} mean {gaps.mean():.1f} p90 {np.percentile(gaps, 90):.1f} us, "
          f"gaps > 20 us: {100 * (gaps > 20).mean():.0f} % holding {100 * gaps[gaps > 20].sum() / gaps.sum():.0f} % of gap time | {top}")

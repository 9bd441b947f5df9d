"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: python tools/pmc_traffic.py <kernel-prefix> <fetch_dir> <write_dir> > profiles/r01_pmc_traffic.json"""
import collections
import csv
import glob
import json
import sys

prefix, fetch_dir, write_dir = sys.argv[1:4]


def short(n):
    return n.replace("wt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def collect(d, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


ft, fc = collect(fetch_dir, "FETCH_SIZE")
wt, wc = collect(write_dir, "WRITE_SIZE")
per = {}
for k in sorted(set(ft) | set(wt)):
    per[k] = {"launches": fc.get(k, wc.get(k, 0)),
              "read_MB": round(2 * 1024 * ft.get(k, 0) / max(1, fc.get(k, 1)) / 1e6, 1),
              "write_MB": round(1024 * wt.get(k, 0) / max(1, wc.get(k, 1)) / 1e6, 1)}
sel = [k for k in per if k.startswith(prefix)]
n = sum(fc[k] for k in sel)
rd = sum(2 * 1024 * ft[k] for k in sel) / max(1, n)
wr = sum(1024 * wt[k] for k in sel) / max(1, sum(wc[k] for k in sel))
print(json.dumps({
    "kernel_class": f"{prefix} (all {prefix} launches of the encoder phase)",
    "launches": n,
    "fabric_read_bytes_per_launch": rd,
    "fabric_write_bytes_per_launch": wr,
    "traffic_bytes_per_launch": rd + wr,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 "
              "--no-pipeline --no-cpu-baseline`; bytes = 2*1024*FETCH_SIZE + 1024*WRITE_SIZE (gfx950 FETCH_SIZE counts half "
              "of a wide coalesced read, MI355X_MICROARCH.md HBM section; calibrated on layernorm_rows). The counters sit on "
              "the L2's fabric side, so Infinity-Cache hits are included: L2-miss traffic, an upper bound on HBM traffic.",
    "per_kernel": per}, indent=1))

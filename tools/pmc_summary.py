"""Aggregate rocprofv3 --pmc csv output per kernel: counter sums and per-launch averages.
usage: python tools/pmc_summary.py <dir> [<dir> ...]  -> JSON on stdout
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced
read stream (MI355X_MICROARCH.md §HBM), so hbm_read_bytes = 2 * FETCH_SIZE * 1024."""
import collections
import csv
import glob
import json
import sys


def short(n):
    return n.replace("wt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


out = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
res = {}
for k, cs in out.items():
    res[k] = {}
    for c, v in cs.items():
        n = launches[k][c]
        res[k][c] = {"sum": v, "launches": n, "per_launch": v / n}
    if "FETCH_SIZE" in cs:
        res[k]["hbm_read_bytes_per_launch"] = 2 * 1024 * cs["FETCH_SIZE"] / launches[k]["FETCH_SIZE"]
    if "WRITE_SIZE" in cs:
        res[k]["hbm_write_bytes_per_launch"] = 1024 * cs["WRITE_SIZE"] / launches[k]["WRITE_SIZE"]
print(json.dumps(res, indent=1))

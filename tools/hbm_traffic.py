"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) into
per-kernel HBM traffic per launch.  gfx950 correction: FETCH_SIZE counts 64 B per 128 B request for wide streaming
reads, so fetched bytes = 2 * FETCH_SIZE (units: KiB); WRITE_SIZE is exact.
usage: python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_hbm_traffic.json"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = (glob.glob(f"{d}/*counter_collection.csv") + glob.glob(f"{d}/*/*counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -sum(fetch.get(k, [0]))):
    f, w = fetch.get(k, []), write.get(k, [])
    n = max(len(f), len(w))
    fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)          # bytes per launch, gfx950 x2 correction
    wb = 1024.0 * sum(w) / max(len(w), 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
json.dump({"note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, averaged over all launches of the kernel "
                   "in `python bench.py --steps 2 --warmup 1 --serial` (B=512 bf16)", "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:14]:
    print(f"{v['hbm_bytes_per_launch']/1e6:10.1f} MB/launch  (fetch {v['fetch_bytes_per_launch']/1e6:8.1f} write {v['write_bytes_per_launch']/1e6:8.1f})  n={v['launches']:4d}  {k[:70]}")

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_pmc_traffic.json.

    python tools/pmc_traffic.py <fetch_counter_csv> <write_counter_csv> <out.json> [dominant-kernel-substring]

FETCH_SIZE is reported in KB and, on gfx950, counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM section): read bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE (KB) is used as is."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha256


def per_kernel(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("s3d::", "").split("(")[0][:70]
        a = agg.setdefault(name, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return agg


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
dom = sys.argv[4] if len(sys.argv) > 4 else "k_conv_wino"
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 6 --warmup 2 "
               "--no-cpu-baseline --profile-every 0; per-launch averages. read bytes = 2 * FETCH_SIZE KB * 1024 (gfx950 "
               "correction, MI355X_MICROARCH.md), write bytes = WRITE_SIZE KB * 1024.  Aggregated by tools/pmc_traffic.py.",
       "kernels": {}}
for k in fetch:
    f, n = fetch[k]
    w = write.get(k, [0.0, 1])
    out["kernels"][k] = {"launches": n, "fetch_size_kb": round(f / n, 1), "write_size_kb": round(w[0] / max(w[1], 1), 1),
                         "hbm_read_bytes": int(2 * f / n * 1024), "hbm_write_bytes": int(w[0] / max(w[1], 1) * 1024)}
ds = [v for k, v in out["kernels"].items() if dom in k]      # (both blockings of the mixed Winograd kernel: launch-weighted mean)
n = sum(v["launches"] for v in ds)
d = {"launches": n, "hbm_read_bytes": int(sum(v["hbm_read_bytes"] * v["launches"] for v in ds) / n),
     "hbm_write_bytes": int(sum(v["hbm_write_bytes"] * v["launches"] for v in ds) / n)}
out["dominant_kernel"] = dom + " (every kernel whose name contains it, launch-weighted mean)"
out["csrc_sha256"] = csrc_sha256()      # bench.py drops the traffic field when the kernel sources have changed since
out["dominant_traffic_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(dom, d, out["dominant_traffic_bytes_per_launch"])

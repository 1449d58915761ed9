#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (CSV output) into per-kernel HBM bytes per launch.

usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B read requests at
64 B, so wide coalesced reads are DOUBLED here; WRITE_SIZE is reported as read (uncalibrated: the guide gives no factor),
both counters are in KiB.  Infinity-Cache hits are counted as traffic by these memory-side counters."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    disp = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        disp[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for d, v in disp.items():
        k = re.sub(r"\(.*", "", name[d])
        agg[k].append(v)
    return agg


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -sum(f.get(k, [0]))):
        nf, nw = len(f.get(k, [])), len(w.get(k, []))
        rd = 2.0 * 1024.0 * sum(f.get(k, [0])) / max(nf, 1)
        wr = 1024.0 * sum(w.get(k, [0])) / max(nw, 1)
        out[k] = {"launches": max(nf, nw), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    for k, v in list(out.items())[:12]:
        print("%-70s n=%4d  read %9.1f MB  write %9.1f MB" % (k[:70], v["launches"], v["hbm_read_bytes_per_launch"] / 1e6,
                                                              v["hbm_write_bytes_per_launch"] / 1e6))
    if len(sys.argv) > 3:
        import os
        json.dump({"note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported; KiB units; per-launch averages",
                   "micro_batch": int(os.environ.get("KBNER_PROFILE_MICRO_BATCH", "256")),   # what bench.py ran at (its default)
                   "kernels": out}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

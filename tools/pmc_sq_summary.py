#!/usr/bin/env python
"""Per-kernel averages of the counters collected by tools/pmc_sq.sh (rocprofv3 csv: *counter_collection.csv)."""
import collections, csv, glob, os, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
    for (d, c), v in per.items():
        agg[names[d]][c].append(v)
lines = []
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    lines.append("%s  (n=%d)" % (k, max(len(v) for v in cs.values())))
    wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(len(cs.get("SQ_WAVE_CYCLES", [1])), 1)
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        lines.append("    %-28s %16.0f   %6.1f %% of SQ_WAVE_CYCLES" % (c, m, 100.0 * m / wc if wc else 0.0))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:120]))

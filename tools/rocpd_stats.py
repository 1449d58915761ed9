#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(name, calls, total ms, avg us, % of GPU kernel time).  Usage: python tools/rocpd_stats.py results.db [out.md] [--by-grid]
--by-grid keys the rows by (name, workgroups) as well: the small-batch regimes launch one kernel on several shapes."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    by_grid = "--by-grid" in sys.argv
    if by_grid:
        sys.argv.remove("--by-grid")
    gcol = [c for c in cols if c in ("grid_size_x", "grid_x")]
    wcol = [c for c in cols if c in ("workgroup_size_x", "workgroup_x")]
    if by_grid and gcol and wcol:
        rows = cur.execute("select %s || ' wg=' || (%s / %s), start, end from kernels" % (namecol, gcol[0], wcol[0])).fetchall()
    else:
        rows = cur.execute("select %s, start, end from kernels" % namecol).fetchall()
    agg = {}
    for n, s, e in rows:
        if by_grid and " wg=" in n:
            n = short(n.rsplit(" wg=", 1)[0])[:60] + " wg=" + n.rsplit(" wg=", 1)[1]
        d = agg.setdefault(short(n) if not by_grid else n, [0, 0.0])
        d[0] += 1
        d[1] += (e - s)
    tot = sum(v[1] for v in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.1f | %.1f |" % (n, c, t / 1e6, t / c / 1e3, 100.0 * t / tot))
    lines.append("| TOTAL | %d | %.3f | | 100 |" % (len(rows), tot / 1e6))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()

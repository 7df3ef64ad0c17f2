#!/usr/bin/env python3
"""Per-kernel PMC counter sums from one or more rocprofv3 --pmc runs (rocpd sqlite).
usage: python tools/pmc_summary.py out.md db1 [db2 ...]"""
import re
import sqlite3
import sys

out_path, dbs = sys.argv[1], sys.argv[2:]
acc = {}
for path in dbs:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    q = "select %s, %s, sum(%s), count(distinct %s) from counters_collection group by 1, 2" % (kcol, ccol, vcol, dcol or kcol)
    for k, c, v, n in cur.execute(q):
        k = re.sub(r"ude::NetCfg<ude::IntList<([\d, ]+)>, ude::IntList<([\d, ]+)>\s*>", r"Net[\1|\2]", k).replace("ude::", "")
        acc.setdefault(k, {})[c] = (v, n)
lines = []
for k in sorted(acc, key=lambda k: -max(v for v, _ in acc[k].values())):
    if "kernel" not in k or "at::" in k:
        continue
    lines.append("### `%s`" % k[:120])
    lines.append("| counter | sum over dispatches | dispatches | per dispatch |")
    lines.append("|---|---|---|---|")
    for c in sorted(acc[k]):
        v, n = acc[k][c]
        lines.append("| %s | %.6g | %d | %.6g |" % (c, v, n, v / max(n, 1)))
    lines.append("")
txt = "\n".join(lines)
print(txt)
open(out_path, "w").write(txt + "\n")

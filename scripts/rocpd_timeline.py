#!/usr/bin/env python3
"""Prints a window of the kernel timeline of a rocprofv3 rocpd database: start (us, relative), duration, stream/queue, name.
  python scripts/rocpd_timeline.py <db> <name-substring-to-anchor-on> [occurrence] [count] [before]      (before: kernels shown in front of the anchor, default 4)"""
import re
import sqlite3
import sys

db, anchor = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 10
count = int(sys.argv[4]) if len(sys.argv) > 4 else 40
before = int(sys.argv[5]) if len(sys.argv) > 5 else 4
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = cur.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
if not idx:
    sys.exit("anchor not found; columns: " + ", ".join(cols))
i0 = idx[min(occ, len(idx) - 1)]
t0 = rows[i0][1]
for r in rows[max(0, i0 - before): i0 + count]:
    name = re.sub(r"\(.*", "", r[0])[-48:]
    print(f"{(r[1] - t0) / 1e3:10.1f} us  +{(r[2] - r[1]) / 1e3:9.1f} us  q{r[3]}  {name}")

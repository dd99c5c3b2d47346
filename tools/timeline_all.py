#!/usr/bin/env python3
"""All kernels of a rocprofv3 --kernel-trace database between two times (us, relative to the first kernel of the LAST
occurrence window): python tools/timeline_all.py db.db <first_kernel_substring> [n_back]
prints every kernel from the n_back-th last launch of the marker kernel on."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = next((c for c in cols if "queue" in c), None)
rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
i0 = idx[-back]; t0 = rows[i0][1]
for r in rows[i0:]:
    print("%10.1f %10.1f %9.1f  q%-3s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:48].replace("c25519::", "").replace("void ", "")))

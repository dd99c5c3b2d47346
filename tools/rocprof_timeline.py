#!/usr/bin/env python3
"""Timeline of the LAST call of a workload in a rocprofv3 --kernel-trace database (rocpd SQLite): every
kernel from the last launch of <first_kernel_substring> on, with start offset, duration and queue, so that
the overlap between the two streams can be read off.
    python tools/rocprof_timeline.py gpurun_out/.../verify_results.db k_hram"""
import sqlite3
import sys


def main(path, first):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in cols if "queue" in c), None) or next((c for c in cols if "stream" in c), None)
    sel = "name, start, end" + (", " + qcol if qcol else "")
    rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    if not idx:
        raise SystemExit("no kernel matching %r" % first)
    i0 = idx[-1]
    # the call starts a little before its first marker kernel (the other stream may have launched already)
    while i0 > 0 and rows[i0][1] - rows[i0 - 1][2] < 20000 and idx[-1] - i0 < 4:
        i0 -= 1
    t0 = rows[i0][1]
    print("# %s  (columns: %s)" % (path, ", ".join(cols)))
    print("%10s %10s %10s  %-8s %s" % ("start_us", "end_us", "dur_us", qcol or "-", "kernel"))
    for r in rows[i0:]:
        print("%10.1f %10.1f %10.1f  %-8s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", r[0][:70]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

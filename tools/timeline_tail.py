#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 --kernel-trace database with start offsets, durations and queues (stream sets):
    python tools/timeline_tail.py <results.db> [N] [min_dur_us]"""
import sqlite3
import sys


def main(path, count=150, mind=0.0):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((c for c in cols if "queue" in c), None) or next((c for c in cols if "stream" in c), None)
    rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()[-count:]
    t0 = rows[0][1]
    for name, a, b, q in rows:
        if (b - a) / 1e3 >= mind:
            print("%9.1f %9.1f %8.1f  q%-3s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, name.replace("c25519::", "")[:60]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 150, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)

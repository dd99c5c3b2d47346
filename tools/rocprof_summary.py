#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result database (rocpd SQLite) as a per-kernel table:
    python tools/rocprof_summary.py gpurun_out/prof/msm/msm_results.db [> profiles/r01_msm_kernels.txt]
Columns: calls, total us, avg us, min us, max us, share of GPU time, VGPRs, LDS bytes."""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    # the clock warm-up / roofline probes of bench.py (k_probe_*) are listed but kept out of the share column
    tot = sum(r[2] for r in rows if "k_probe_" not in r[0]) or 1.0
    print("# %s" % path)
    print("%-72s %6s %12s %10s %10s %10s %6s %5s %7s %9s %5s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share", "vgpr", "lds_B", "grid_x", "wg_x"))
    for r in rows:
        share = "  probe" if "k_probe_" in r[0] else "%5.1f%%" % (100 * r[2] / tot)
        print("%-72s %6d %12.1f %10.1f %10.1f %10.1f %s %5d %7d %9d %5d" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], share, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))


if __name__ == "__main__":
    main(sys.argv[1])

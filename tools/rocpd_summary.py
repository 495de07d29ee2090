#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace --stats, default output format on
ROCm 7.2) into the per-kernel table rocprofv3's CSV stats would give: calls, total, average, min, max, %.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--top 40] > profiles/xxx.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                          "max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print("# source: %s   kernels: %d distinct, %d dispatches, %.3f ms total GPU time" % (db, len(rows), sum(r[1] for r in rows), total / 1e6))
    print("%-110s %8s %12s %11s %11s %11s %6s %5s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds"))
    for r in rows[:top]:
        print("%-110s %8d %12.3f %11.2f %11.2f %11.2f %6.2f %5s %7s" % (r[0][:110], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7]))


if __name__ == "__main__":
    main()

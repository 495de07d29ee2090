#!/usr/bin/env python3
"""HBM traffic per launch of EVERY kernel of a profiled train step (not only the dw-conv path): two rocprofv3 PMC passes of
`bench.py --markers --steps K` (FETCH_SIZE, WRITE_SIZE; --pmc with --kernel-trace only) + the kernel trace of a third, counter-free run
for the durations (counter collection serialises and slows the launches).

    python tools/pmc_step.py <fetch.db> <write.db> <trace.db> --steps K [--top 45] [--match slak::]

HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE on gfx950 reports half of a wide streaming read:
MI355X_MICROARCH.md, HBM section; WRITE_SIZE as read).  Per kernel name: calls per step, mean duration inside the un-instrumented step,
mean read / written MB per launch, the bandwidth that is and its fraction of the 8 TB/s peak."""
import collections
import sqlite3
import sys


def arg(name, default=None, cast=str):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def counters(db, counter):
    c = sqlite3.connect(db)
    out = collections.defaultdict(list)
    for n, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        out[n].append(float(v))
    return out


def window(db, steps):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))

    def marker_id(r):
        g, w = r[3], r[4] or 64
        n = g // w if (g % w == 0 and g >= w and g // w in (2, 3)) else g
        return n - 1
    m1 = [r for r in rows if "marker_kernel" in r[0] and marker_id(r) == 1]
    m2 = [r for r in rows if "marker_kernel" in r[0] and marker_id(r) == 2]
    if len(m1) != 1 or len(m2) != 1:
        sys.exit("expected one marker 1 and one marker 2 in the trace (bench.py --markers)")
    t0, t1 = m1[0][2], m2[0][1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e, _, _ in rows:
        if t0 <= s < t1 and "marker_kernel" not in n:
            agg[n][0] += 1; agg[n][1] += e - s
    return {n: (c / steps, d / c / 1e3) for n, (c, d) in agg.items()}


def main():
    fdb, wdb, tdb = sys.argv[1:4]
    steps = arg("--steps", None, int)
    top = arg("--top", 45, int)
    match = arg("--match", "")
    f, w = counters(fdb, "FETCH_SIZE"), counters(wdb, "WRITE_SIZE")
    t = window(tdb, steps)
    rows = []
    for n, (calls, us) in t.items():
        if match and match not in n:
            continue
        if n not in f or n not in w:
            continue
        rd = 2.0 * 1024 * sum(f[n]) / len(f[n]); wr = 1024.0 * sum(w[n]) / len(w[n])
        rows.append((calls * us, n, calls, us, rd, wr))
    print("# HBM traffic per launch (mean over every launch of the name in the PMC runs) and the bandwidth it is at the launch's mean duration INSIDE the un-instrumented step")
    print("%-96s %6s %9s %9s %9s %9s %7s" % ("kernel", "calls", "us", "read MB", "write MB", "GB/s", "of 8T"))
    for _, n, calls, us, rd, wr in sorted(rows, reverse=True)[:top]:
        gbs = (rd + wr) / us / 1e3
        print("%-96s %6.1f %9.1f %9.1f %9.1f %9.0f %7.3f" % (n.replace("void ", "")[:96], calls, us, rd / 1e6, wr / 1e6, gbs, gbs / 8000.0))


if __name__ == "__main__":
    main()

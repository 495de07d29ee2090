#!/usr/bin/env python3
"""Where the GPU idles inside the timed steps of a profiled bench.py run (rocprofv3 --kernel-trace database, bench.py --markers):

    python tools/step_gaps.py <results.db> --steps K [--top 40] [--min-gap-us 2]

For every dispatch between the two markers the idle time in front of it = its start - the latest end of everything dispatched before it
(streams overlap: only time with NOTHING running counts).  Printed: idle ms per step in total, and per kernel name the idle time that
sits in front of its launches (calls per step, mean gap, ms per step) -- the kernels behind stream hand-offs and host stalls lead the list."""
import collections
import sqlite3
import sys


def arg(name, default=None, cast=str):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    db = sys.argv[1]
    steps = arg("--steps", None, int)
    top = arg("--top", 40, int)
    min_gap = arg("--min-gap-us", 0.0, float) * 1e3
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))

    def marker_id(r):
        g, w = r[3], r[4] or 64
        n = g // w if (g % w == 0 and g >= w and g // w in (2, 3)) else g
        return n - 1
    m1 = [r for r in rows if "marker_kernel" in r[0] and marker_id(r) == 1]
    m2 = [r for r in rows if "marker_kernel" in r[0] and marker_id(r) == 2]
    if len(m1) != 1 or len(m2) != 1:
        sys.exit("expected one marker 1 and one marker 2 (bench.py --markers)")
    t0, t1 = m1[0][2], m2[0][1]
    agg = collections.defaultdict(lambda: [0, 0.0, 0])
    busy_end = t0
    idle = 0.0
    n_disp = 0
    for n, s, e, _, _ in rows:
        if not (t0 <= s < t1) or "marker_kernel" in n:
            continue
        n_disp += 1
        gap = max(0, s - busy_end)
        a = agg[n]
        a[0] += 1
        if gap >= min_gap:
            a[1] += gap; a[2] += 1
            idle += gap
        busy_end = max(busy_end, e)
    print("# %s" % db)
    print("# %d dispatches per step; idle %.3f ms per step (nothing running), window %.3f ms per step" % (n_disp // steps, idle / steps / 1e6, (t1 - t0) / steps / 1e6))
    print("%-120s %10s %10s %10s" % ("kernel the gap is in front of", "calls/step", "mean gap us", "idle ms/step"))
    for n, (cnt, g, ng) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-120s %10.2f %10.2f %10.3f" % (n[:120], cnt / steps, g / max(cnt, 1) / 1e3, g / steps / 1e6))


if __name__ == "__main__":
    main()

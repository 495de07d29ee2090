#!/usr/bin/env python3
"""Hardware-timestamp durations of every dw-conv launch of the hot path AS THE STEP LAUNCHES IT, from a rocprofv3 --kernel-trace of
tools/time_all.py (tools/kernel_times.sh):

    python tools/kernel_times.py <results.db> <entries.json>  > profiles/rNN_kernel_times.txt

tools/time_all.py launches the entries of bench.hot_path_kernels in order, each (5 warm-up + reps) times back to back, and writes the entry
list (stage, kernel, op, algorithmic bytes per SURVEY 8d, calls per step, the kernel slak_debug_last_kernel() named) to entries.json.
Consecutive dispatches of one (kernel name, grid) are one entry; the first 5 of each group (warm-up) are dropped, the rest averaged."""
import json
import sqlite3
import sys

HBM_PEAK = 8000.0


def main():
    db, entries = sys.argv[1], json.load(open(sys.argv[2]))
    c = sqlite3.connect(db)
    rows = [r for r in c.execute("select name, start, end, grid_x from kernels order by start") if "slak::dwconv" in r[0]]
    groups = []
    for n, s, e, g in rows:
        if groups and groups[-1][0] == (n, g):
            groups[-1][1].append(e - s)
        else:
            groups.append([(n, g), [e - s]])
    want = list(entries)
    # consecutive entries that dispatch the same (kernel, grid) -- the per-branch launches of the wide-map kernels -- arrive as one run: every entry
    # has the same number of dispatches (warm-up + reps), so runs that are a multiple of it are cut
    total = sum(len(g[1]) for g in groups)
    if len(groups) != len(want) and total % len(want) == 0:
        per = total // len(want)
        cut = []
        for key, d in groups:
            if len(d) % per:
                cut = None
                break
            cut += [[key, d[i:i + per]] for i in range(0, len(d), per)]
        if cut is not None:
            groups = cut
    if len(groups) != len(want):
        print("run-length groups %d != expected %d" % (len(groups), len(want)))
        for g in groups:
            print("  ", g[0][0][:100], g[0][1], len(g[1]))
        sys.exit(1)
    print("# %s: rocprofv3 --kernel-trace of tools/time_all.py; per launch: mean over the timed dispatches (warm-up dropped), hardware timestamps" % db)
    print("%-5s %-22s %-5s %-10s %9s %9s %6s %8s %6s  %s" % ("stage", "kernel", "kind", "op", "us", "min us", "calls", "GB/s", "frac", "dispatched kernel"))
    tot = byt = 0.0
    for e, g in zip(want, groups):
        d = g[1][5:] if len(g[1]) > 5 else g[1]
        us = sum(d) / len(d) / 1e3
        gbs = e["alg_bytes"] / us / 1e3
        tot += us * e["calls_per_step"] / 1e3; byt += e["alg_bytes"] * e["calls_per_step"]
        print("%-5d %-22s %-5s %-10s %9.2f %9.2f %6d %8.0f %6.3f  %s" % (e["stage"], e["kernel"], e["branch"], e["op"], us, min(d) / 1e3, e["calls_per_step"], gbs,
                                                                    gbs / HBM_PEAK, g[0][0].replace("void ", "").replace("slak::", "")[:90]))
    print("dw-conv hot path per step (hardware timestamps): %.3f ms for %.3f GB (SURVEY 8d) = %.0f GB/s = %.3f of the %.0f GB/s HBM peak"
          % (tot, byt / 1e9, byt / tot / 1e6, byt / tot / 1e6 / HBM_PEAK, HBM_PEAK))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/time_all.py (FETCH_SIZE, WRITE_SIZE; separate runs, --pmc with --kernel-trace only) into
profiles/pmc_traffic.json: HBM bytes per launch of every dw-conv launch of the hot path AS THE MODEL RUNS IT.

    python tools/pmc_traffic.py <fetch.db> <write.db> <entries.json>  > profiles/r03_pmc_traffic.txt

HBM bytes per launch = 2 * FETCH_SIZE*1024 + WRITE_SIZE*1024: FETCH_SIZE on gfx950 reports exactly half of a wide coalesced streaming
read (MI355X_MICROARCH.md, HBM section: TCC_EA0_RDREQ counted at 64 B for 128-B requests), WRITE_SIZE is taken as read.
tools/time_all.py launches the entries of bench.hot_path_kernels in order, each (warm-up + reps) times back to back and writes the entry
list to entries.json; consecutive dispatches of one (kernel, grid) are one entry.  The median over an entry's launches is kept.
"""
import json
import os
import sqlite3
import statistics
import sys


def runs_of(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    rows = None
    if "counters_collection" in tabs:
        try:
            rows = list(c.execute("select kernel_name, value, start, grid_size from counters_collection where counter_name = ? order by start", (counter,)))
        except sqlite3.OperationalError:
            rows = [(n, v, s, 0) for (n, v, s) in c.execute("select kernel_name, value, start from counters_collection where counter_name = ? order by start", (counter,))]
    seq = [(n, v, g) for (n, v, _, g) in rows if "slak::dwconv" in n]
    runs = []
    for n, v, g in seq:
        if runs and runs[-1][0] == (n, g):
            runs[-1][1].append(v)
        else:
            runs.append([(n, g), [v]])
    return runs


def main():
    fetch_db, write_db, entries = sys.argv[1], sys.argv[2], json.load(open(sys.argv[3]))
    f, w = runs_of(fetch_db, "FETCH_SIZE"), runs_of(write_db, "WRITE_SIZE")
    if not (len(f) == len(w) == len(entries)):
        print("run-length groups: fetch %d, write %d, entries %d" % (len(f), len(w), len(entries)))
        for r in f:
            print("  ", r[0][0][:90], r[0][1], len(r[1]))
        sys.exit(1)
    out = {}
    print("%-40s %-34s %14s %14s %14s %14s %8s" % ("launch", "kernel", "alg bytes", "2*FETCH", "WRITE", "HBM bytes", "HBM/alg"))
    for e, rf, rw in zip(entries, f, w):
        rd = 2.0 * 1024 * statistics.median(rf[1]); wr = 1024.0 * statistics.median(rw[1])
        key = "s%d_%s_%s" % (e["stage"], e["kernel"], e["op"])
        out[key] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_2xFETCH_SIZE": rd, "write_bytes_WRITE_SIZE": wr, "alg_bytes": e["alg_bytes"],
                    "alg_bytes_incl_acc_read": e.get("alg_bytes_incl_acc_read", e["alg_bytes"]), "hip_kernel": e.get("hip_kernel"), "launches_measured": len(rf[1])}
        print("%-40s %-34s %14d %14.0f %14.0f %14.0f %8.2f" % (key, str(e.get("hip_kernel"))[:34], e["alg_bytes"], rd, wr, rd + wr, (rd + wr) / e["alg_bytes"]))
    tot_alg = sum(e["alg_bytes"] * e["calls_per_step"] for e in entries)
    tot_hbm = sum(out["s%d_%s_%s" % (e["stage"], e["kernel"], e["op"])]["hbm_bytes_per_launch"] * e["calls_per_step"] for e in entries)
    print("per step: algorithmic (SURVEY 8d) %.3f GB, measured HBM %.3f GB, ratio %.3f" % (tot_alg / 1e9, tot_hbm / 1e9, tot_hbm / tot_alg))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    if os.environ.get("SLAK_PMC_WRITE_JSON", "1") == "1":
        with open(path, "w") as fo:
            json.dump(out, fo, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/pmc_workload.py into profiles/pmc_traffic.json.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/pmc_results.db gpurun_out/pmc_write/pmc_results.db > profiles/r01_pmc_traffic.txt

HBM bytes per op launch = 2 * FETCH_SIZE*1024 + WRITE_SIZE*1024: FETCH_SIZE on gfx950 reports exactly half of a wide
coalesced streaming read (MI355X_MICROARCH.md, HBM section: TCC_EA0_RDREQ counted at 64 B for 128-B requests), WRITE_SIZE
is taken as read (uncalibrated).  An "op" is everything one C-ABI call launches (one kernel on the MFMA paths).  The workload issues, per (stage, filter), 3 x (forward, backward_data, backward_filter); the median is kept.
"""
import json
import os
import sqlite3
import statistics
import sys

STAGES = [(96, 56, 51), (192, 28, 49), (384, 14, 47), (768, 7, 13)]


def ops_of(db, counter):
    c = sqlite3.connect(db)
    rows = list(c.execute("select kernel_name, value, start from counters_collection where counter_name = ? order by start", (counter,)))
    seq = [(n, v) for (n, v, _) in rows if "slak::" in n]
    # every C-ABI call of the MFMA paths launches exactly ONE kernel now (fragments are built in the kernel, the slice reduction of
    # the weight gradient is folded into it); the fp32-exact direct path (not part of this workload) would add a reduce launch
    ops = []
    for n, v in seq:
        if "reduce" in n or "toeplitz_pack" in n:
            if ops: ops[-1] = (ops[-1][0], ops[-1][1] + v)
        elif "wgrad" in n:
            ops.append(("wgrad", v))
        elif "dwconv" in n:
            ops.append(("conv", v))
    return ops


def main():
    fetch_db, write_db = sys.argv[1], sys.argv[2]
    f, w = ops_of(fetch_db, "FETCH_SIZE"), ops_of(write_db, "WRITE_SIZE")
    assert len(f) == len(w) == 4 * 3 * 3 * 3 + 2 * 2 * 3 + 2 * 3 * 3 + 2 * 3, (len(f), len(w))
    out = {}
    i = 0
    print("%-28s %14s %14s %14s %14s %8s" % ("op", "alg bytes", "2*FETCH", "WRITE", "HBM bytes", "HBM/alg"))
    for si, (C, H, K) in enumerate(STAGES):
        S = 128 * C * H * H
        for (kh, kw) in ((K, 5), (5, K), (5, 5)):
            vals = {"fwd": [], "bwd_data": [], "bwd_filter": []}
            for rep in range(3):
                for name in ("fwd", "bwd_data", "bwd_filter"):
                    kind = "wgrad" if name == "bwd_filter" else "conv"
                    assert f[i][0] == kind and w[i][0] == kind, (i, f[i], w[i])
                    vals[name].append((2.0 * f[i][1] * 1024, w[i][1] * 1024))
                    i += 1
            for name, v in vals.items():
                rd = statistics.median(a for a, b in v); wr = statistics.median(b for a, b in v)
                alg = 2 * S * 2 + C * kh * kw * 4
                key = "s%d_%dx%d_%s" % (si + 1, kh, kw, name)
                out[key] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_2xFETCH_SIZE": rd, "write_bytes_WRITE_SIZE": wr, "alg_bytes": alg}
                print("%-28s %14d %14.0f %14.0f %14.0f %8.2f" % (key, alg, rd, wr, rd + wr, (rd + wr) / alg))
    # the launches as the model runs them (second part of tools/pmc_workload.py); byte prices as bench.py's hot_path (SURVEY 8d per-op bytes)
    def put(key, alg, v):
        rd = statistics.median(a for a, b in v); wr = statistics.median(b for a, b in v)
        out[key] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_2xFETCH_SIZE": rd, "write_bytes_WRITE_SIZE": wr, "alg_bytes": alg}
        print("%-28s %14d %14.0f %14.0f %14.0f %8.2f" % (key, alg, rd, wr, rd + wr, (rd + wr) / alg))
    for si, (C, H, K) in enumerate(STAGES[:2]):
        S = 128 * C * H * H
        for (kh, kw) in ((5, K), (5, 5)):
            v = []
            for rep in range(3):
                assert f[i][0] == "conv" and w[i][0] == "conv", (i, f[i], w[i])
                v.append((2.0 * f[i][1] * 1024, w[i][1] * 1024)); i += 1
            put("s%d_%dx%d_bwd_data+acc" % (si + 1, kh, kw), 3 * S * 2 + C * kh * kw * 4, v)
    for si, (C, H, K) in enumerate(STAGES):
        if si < 2:
            continue
        S = 128 * C * H * H
        alg = 3 * 2 * S * 2 + C * (2 * K * 5 + 25) * 4
        vals = {"fwd": [], "bwd_data": [], "bwd_filter": []}
        for rep in range(3):
            for name in ("fwd", "bwd_data", "bwd_filter"):
                kind = "wgrad" if name == "bwd_filter" else "conv"
                assert f[i][0] == kind and w[i][0] == kind, (i, f[i], w[i])
                vals[name].append((2.0 * f[i][1] * 1024, w[i][1] * 1024)); i += 1
        for name, v in vals.items():
            put("s%d_%dx5+5x%d+5x5_%s" % (si + 1, K, K, name), alg, v)
    for si, (C, H, K) in enumerate(STAGES[:2]):                  # K x 5 + 5 x 5 weight gradients in one launch: priced at the two ops it replaces
        S = 128 * C * H * H
        v = []
        for rep in range(3):
            assert f[i][0] == "wgrad" and w[i][0] == "wgrad", (i, f[i], w[i])
            v.append((2.0 * f[i][1] * 1024, w[i][1] * 1024)); i += 1
        put("s%d_%dx5+5x5_bwd_filter" % (si + 1, K), 2 * 2 * S * 2 + C * (K * 5 + 25) * 4, v)
    assert i == len(f)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("# wrote", path)


if __name__ == "__main__":
    main()

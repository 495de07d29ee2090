"""Board power and shader clock while one kernel of the hot path runs in a loop (turn "the chip runs these kernels at its power cap" from
an inference -- s_memtime of one kernel, profiles/r04_power_clock_probe.txt -- into a measurement).  A sampler thread reads the driver's own sensors
(hwmon power1_average / power1_input, freq1_input = sclk; falls back to rocm-smi --json) at >= 10 Hz while the main thread keeps the GPU busy for ~2 s per
workload.  -> profiles/rNN_power_trace.txt"""
import ctypes, glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from slak_amd import _lib

dev = torch.device("cuda:0")


def _hwmons():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        pw = [p for p in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(p)]
        fq = os.path.join(hw, "freq1_input")
        if pw:
            out.append((pw[0], fq if os.path.exists(fq) else None, hw))
    return out


def _sensors():
    """The hwmon directory of THE GPU THIS PROCESS RUNS ON (a box has several): by PCI address when torch reports one, else the card whose power rises most
    under a one-second copy loop."""
    hws = _hwmons()
    if not hws:
        return None, None, None
    try:
        pr = torch.cuda.get_device_properties(0)
        bdf = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for pw, fq, hw in hws:
            if bdf in os.path.realpath(os.path.join(hw, "device")):
                return pw, fq, hw
    except Exception:
        pass
    def read_all():
        vals = []
        for pw, _, _ in hws:
            try: vals.append(int(open(pw).read()) / 1e6)
            except Exception: vals.append(float("nan"))
        return vals
    a = torch.empty(64 * 1024 * 1024, device=dev); b = torch.empty_like(a)
    torch.cuda.synchronize(); time.sleep(0.5)
    idle = read_all()
    t0 = time.perf_counter(); peak = list(idle)
    while time.perf_counter() - t0 < 1.5:
        for _ in range(20): b.copy_(a)
        torch.cuda.synchronize()
        peak = [max(x, y) for x, y in zip(peak, read_all())]
    deltas = [p - i for p, i in zip(peak, idle)]
    k = max(range(len(hws)), key=lambda i: deltas[i] if deltas[i] == deltas[i] else -1)
    print("# device picked by power response to a copy loop: %s (+%.0f W; others %s)" % (hws[k][2], deltas[k], ["%+.0f" % d for i, d in enumerate(deltas) if i != k][:8]))
    return hws[k]


PW, FQ, HW = _sensors()


def sample():
    if PW:
        try:
            p = int(open(PW).read()) / 1e6
            f = int(open(FQ).read()) / 1e6 if FQ else float("nan")
            return p, f
        except Exception:
            pass
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out); c = d[sorted(d)[0]]
        p = [float(v) for k, v in c.items() if "ower" in k and "(W)" in k]
        f = [float(v.strip("()Mhz")) for k, v in c.items() if "sclk" in k.lower() and "Mhz" in str(v)]
        return (p[0] if p else float("nan")), (f[0] if f else float("nan"))
    except Exception:
        return float("nan"), float("nan")


def trace(name, fn, seconds=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    rows, stop = [], threading.Event()

    def run():
        while not stop.is_set():
            rows.append((time.perf_counter(),) + sample())
            time.sleep(0.02)
    th = threading.Thread(target=run); th.start()
    t0 = time.perf_counter(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); e1.synchronize()
    stop.set(); th.join()
    ps = [r[1] for r in rows if r[1] == r[1]]; fs = [r[2] for r in rows if r[2] == r[2]]
    rate = len(rows) / max(1e-9, rows[-1][0] - rows[0][0]) if len(rows) > 1 else 0
    us = e0.elapsed_time(e1) * 1e3 / max(1, n)
    print("%-46s %8.1f us/launch | power W min %6.1f mean %6.1f max %6.1f | sclk MHz min %6.0f mean %6.0f max %6.0f | %d samples at %.0f Hz" % (
        name, us, min(ps, default=float("nan")), sum(ps) / max(1, len(ps)), max(ps, default=float("nan")),
        min(fs, default=float("nan")), sum(fs) / max(1, len(fs)), max(fs, default=float("nan")), len(rows), rate))


def main():
    print("# sensors: %s (%s)" % (HW or "rocm-smi --json", "power1_* in uW, freq1_input in Hz" if PW else "subprocess per sample"))
    try:
        cap = open(os.path.join(HW, "power1_cap")).read().strip() if HW else None
        print("# power cap: %s W" % (int(cap) / 1e6 if cap else "?"))
    except Exception:
        pass
    torch.cuda.synchronize(); time.sleep(1.0)
    p, f = sample(); print("%-46s %8s            | power W %6.1f | sclk MHz %6.0f" % ("idle", "", p, f))
    N, C, H, K = 128, 96, 56, 51
    L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream; dt = _lib.SLAK_BF16
    x = torch.randn(N, C, H, H, device=dev).bfloat16()
    ys = [torch.empty_like(x) for _ in range(3)]; dys = [torch.randn_like(x) for _ in range(3)]
    ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.02 for kh, kw in ((K, 5), (5, K), (5, 5))]
    dws = [torch.empty_like(w) for w in ws]
    big = torch.empty(308 * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16); big2 = torch.empty_like(big)
    nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, H, K)); wsp = torch.empty(nb, dtype=torch.uint8, device=dev)
    rows = int(L.slak_dwconv2d_tri_stats_rows(dt, N, C, H, H, K)); stats = torch.empty(max(rows, 1), C, 6, device=dev)
    trace("torch copy_ 308 MB (read + write)", lambda: big2.copy_(big))
    trace("stage-1 forward + BN sums (stream_tri)", lambda: _lib.check(L.slak_dwconv2d_tri_forward_stats(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), stats.data_ptr(), dt, N, C, H, H, K, st)))
    trace("stage-1 data gradient (team_tri)", lambda: _lib.check(L.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), dt, N, C, H, H, K, st)))
    trace("stage-1 weight gradients (tri_wgrad_rows)", lambda: _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, H, H, K, wsp.data_ptr(), nb, st)))
    # the round-5 pointwise GEMMs at the stage-3 shape, for scale
    M, Cc = 128 * 14 * 14, 384
    t = torch.randn(M, Cc, device=dev).bfloat16(); w1 = (torch.randn(4 * Cc, Cc, device=dev) * 0.05).bfloat16(); b1 = torch.randn(4 * Cc, device=dev).bfloat16()
    y1 = torch.empty(M, 4 * Cc, device=dev, dtype=torch.bfloat16); a = torch.empty_like(y1)
    trace("stage-3 pwconv1 + GELU (linear_gemm)", lambda: _lib.check(L.slak_linear_gemm(t.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1.data_ptr(), a.data_ptr(), None, None, M, 4 * Cc, Cc, 1, None, 0, st)))
    trace("stage-3 pwconv1 (library GEMM)", lambda: torch.nn.functional.linear(t, w1, b1))
    time.sleep(1.0)
    p, f = sample(); print("%-46s %8s            | power W %6.1f | sclk MHz %6.0f" % ("idle again", "", p, f))


if __name__ == "__main__":
    main()

"""Stress: slak_linear_gemm's counted-vmcnt ring under perturbed memory latencies.  For every covered K and both epilogues, 150 launches of the same inputs while a second
stream hammers HBM with copies of changing size; every output must be BIT-IDENTICAL to the first launch's (a chunk read before it landed, or a staging tile reused too early,
would show as a changed bit somewhere)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
side = torch.cuda.Stream()
junk = [torch.empty(n, device=dev, dtype=torch.uint8) for n in (1 << 20, 13 << 20, 97 << 20, 311 << 20)]
junk2 = [torch.empty_like(j) for j in junk]
bad = 0
for (M, N, K) in ((25088, 1536, 384), (12544, 768, 192), (6272, 3072, 768), (12544, 2048, 512), (3000, 1024, 256), (50176, 768, 192)):
    torch.manual_seed(K + M)
    t = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
    y1r = torch.randn(M, N, device=dev).bfloat16()
    for epi in (1, 2):
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); out2 = torch.empty_like(out); db = torch.empty(N, device=dev)
        nb = L.slak_linear_gemm_workspace_bytes(M, N, K, epi); ws = torch.empty(max(nb, 16), device=dev, dtype=torch.uint8)
        st = torch.cuda.current_stream().cuda_stream
        def run():
            _lib.check(L.slak_linear_gemm(t.data_ptr(), w.data_ptr(), b.data_ptr() if epi == 1 else None, out.data_ptr(), out2.data_ptr() if epi == 1 else None,
                                          y1r.data_ptr() if epi == 2 else None, db.data_ptr() if epi == 2 else None, M, N, K, epi, ws.data_ptr() if nb else None, nb, st))
        run(); torch.cuda.synchronize()
        ref = (out.clone(), out2.clone() if epi == 1 else None, db.clone() if epi == 2 else None)
        diffs = 0
        for it in range(150):
            out.fill_(0); 
            with torch.cuda.stream(side):
                j = it % len(junk); junk2[j].copy_(junk[j])
            run()
            if it % 10 == 9:
                torch.cuda.synchronize()
            torch.cuda.current_stream().synchronize()
            same = torch.equal(out.view(torch.int16), ref[0].view(torch.int16)) and (epi != 1 or torch.equal(out2.view(torch.int16), ref[1].view(torch.int16))) \
                and (epi != 2 or torch.equal(db.view(torch.int32), ref[2].view(torch.int32)))
            diffs += 0 if same else 1
        print("M=%d N=%d K=%d epi=%d: %d of 150 launches differ from the first" % (M, N, K, epi, diffs))
        bad += diffs
torch.cuda.synchronize()
print("STRESS", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)

#!/bin/bash
# fp32 on the matrix cores: parity tests, per-op timing, and the reference dtype flow (fp32 dw convs under bf16 autocast) end to end
cd $GRAFT_REPO_ROOT; O=gpurun_out/fp32; mkdir -p $O
timeout 900 python -m pytest tests/test_fp32_mfma_gpu.py tests/test_dwconv_gpu.py tests/test_model_reference_gpu.py tests/test_reference_modules_gpu.py tests/test_pybind.py tests/test_mfma_gpu.py -q --timeout 600 -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python tools/time_fp32.py > $O/time_fp32.txt 2>&1; tail -1 $O/time_fp32.txt
timeout 600 python bench.py --fp32-dwconv --fp32-matrix-cores --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 > $O/bench_fp32_dwconv_split.json
timeout 600 python bench.py --fp32-dwconv --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 > $O/bench_fp32_dwconv_exact.json
python - <<'P'
import json
for n in ("split","exact"):
    d=json.loads(open(f"gpurun_out/fp32/bench_fp32_dwconv_{n}.json").read()); print(n, d["value"], d["ms_per_step"], d["config"]["dwconv_dtype"])
P

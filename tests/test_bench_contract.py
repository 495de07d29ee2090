"""bench.py's accounting, checked as CODE (the round-2 version of this file asserted properties of a committed JSON artefact):
the stage table, SURVEY 8(d)'s per-op pricing and the argument contract the driver relies on.  The line bench.py prints on a GPU is
checked where it is produced: `hot_path` asserts inside bench.py that the launches it timed add up to survey_8d_bytes(), and
tests/test_distributed_gpu.py parses a real line of the N > 1 path."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    env = {k: os.environ.get(k) for k in list(os.environ) if k.startswith("PYTORCH_TUNABLEOP_")}
    os.environ["SLAK_TUNED_GEMMS"] = "0"                        # importing must not touch the TunableOp environment of this process
    try:
        spec.loader.exec_module(mod)
    finally:
        os.environ.pop("SLAK_TUNED_GEMMS", None)
        for k in list(os.environ):
            if k.startswith("PYTORCH_TUNABLEOP_") and k not in env:
                os.environ.pop(k)
    return mod


def test_stage_tables_are_survey_appendix_a(bench):
    assert bench.stages_of("tiny", 51, 224) == [(96, 56, 51, 3), (192, 28, 49, 3), (384, 14, 47, 9), (768, 7, 13, 3)]            # cfg 2 / 3
    assert bench.stages_of("base", 51, 224) == [(128, 56, 51, 3), (256, 28, 49, 3), (512, 14, 47, 27), (1024, 7, 13, 3)]         # cfg 4
    assert bench.stages_of("tiny", 61, 384) == [(96, 96, 61, 3), (192, 48, 59, 3), (384, 24, 57, 9), (768, 12, 13, 3)]          # cfg 5
    assert bench.kernel_sizes(51) == [51, 49, 47, 13, 5]


def test_survey_8d_bytes_match_the_survey_totals(bench):
    # SURVEY.md section 8 glossary / Appendix A: 9.88 GB (SLaK-T, 128 images), 10.75 GB (SLaK-B, 64), 14.52 GB (61 x 61 at 384 px, 64)
    for args, batch, gb in ((("tiny", 51, 224), 128, 9.88), (("base", 51, 224), 64, 10.75), (("tiny", 61, 384), 64, 14.52)):
        got = bench.survey_8d_bytes(bench.stages_of(*args), batch) / 1e9
        assert gb <= got <= 1.01 * gb + 0.01, (args, got, gb)           # (the survey totals leave the filter bytes out: < 1 %)
    # the per-op price is 2*S*b whatever the launch structure: fp32 doubles it
    st = bench.stages_of("tiny", 51, 224)
    assert bench.survey_8d_bytes(st, 128, 4) > 1.99 * bench.survey_8d_bytes(st, 128, 2) - 1e8


def test_defaults_are_the_driver_contract(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 5) and a.backend == "nccl" and a.model == "tiny" and a.kernel == 51 and a.res == 224
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)
    assert bench.HBM_PEAK_GBS == 8000.0


def test_dry_nccl_env_prints_the_collective_environment_without_a_gpu():
    """bench.py --dry-nccl-env: the NCCL_* / RCCL_* / HSA_* / rendezvous variables a real multi-GPU run would start with, as one JSON
    line, on a box without a GPU (the same dictionary the bench line records as config.comm_env)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCCL_DEBUG="WARN", WORLD_SIZE="8", RANK="3", LOCAL_RANK="3", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-nccl-env", "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["world_size"] == 8 and d["rank"] == 3 and d["backend"] == "nccl"
    assert d["comm_env"]["NCCL_DEBUG"] == "WARN" and d["comm_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and d["comm_env"]["MASTER_ADDR"] == "127.0.0.1"

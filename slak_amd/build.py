"""Build libslak_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m slak_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU.  The .so is git-ignored but travels with the repo snapshot
to the GPU box.  No JIT, no torch cpp_extension: the library has a plain C ABI (include/slak_hip.h).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libslak_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-unused-value", "-ffp-contract=off"] + os.environ.get("SLAK_BUILD_DEFS", "").split()   # e.g. SLAK_BUILD_DEFS=-DSLAK_TEAM_DEV: dev instrumentation of the team / stream kernels


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "slak_hip.h")]
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


PYBIND_NAME = "_depthwise_conv2d_implicit_gemm_C"          # the module name the reference's depthwise_conv2d_implicit_gemm.py:8 imports


def pybind_path() -> str:
    import sysconfig
    return os.path.join(LIBDIR, PYBIND_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


RUNNER_NAME = "_slak_block_runner_C"                        # slak_amd/pybind/block_runner.cpp: a block's call sequence issued from C++


def runner_path() -> str:
    import sysconfig
    return os.path.join(LIBDIR, RUNNER_NAME + sysconfig.get_config_var("EXT_SUFFIX"))


def build_runner(force: bool = False, verbose: bool = False) -> str:
    """slak_amd/pybind/block_runner.cpp (block_ops._BlockFn's call sequence in C++): built like the pybind boundary module."""
    return build_pybind(force=force, verbose=verbose, name=RUNNER_NAME, source="block_runner.cpp")


def build_pybind(force: bool = False, verbose: bool = False, name: str = None, source: str = "frontend_hip.cpp") -> str:
    """The reference's pybind module (frontend.cpp:3-16) on top of libslak_hip.so: slak_amd/pybind/frontend_hip.cpp, host-only C++
    compiled with g++ against the torch headers (what torch.utils.cpp_extension.CppExtension would run), in-tree next to
    libslak_hip.so so that it travels with the repository snapshot.  (`name` / `source`: the other host-only module, build_runner.)"""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    lib = build(force=False, verbose=verbose)
    name = name or PYBIND_NAME
    src = os.path.join(HERE, "pybind", source)
    out = os.path.join(LIBDIR, name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not (force or _stale(out, [src, lib, os.path.join(HERE, "..", "include", "slak_hip.h")])):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-w", src, "-o", out,
           "-DTORCH_EXTENSION_NAME=" + name, "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in ce.include_paths("cuda") + [sysconfig.get_paths()["include"]]:
        cmd += ["-isystem", inc]
    cmd += ["-L" + tlib, "-L" + LIBDIR, "-L/opt/rocm/lib", "-lslak_hip", "-lamdhip64", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-ltorch_python",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    # the host modules on top of the library: rebuilt when asked for, and ALSO whenever one exists and is now older than the library / header / its
    # source -- `python -m slak_amd.build` must not leave a module behind that calls the new library through old argument lists
    for path, fn in ((pybind_path(), build_pybind), (runner_path(), build_runner)):
        if "--pybind" in sys.argv or os.path.exists(path):
            try:
                print(fn(force="--force" in sys.argv, verbose=True))
            except Exception as e:
                if os.path.exists(path):
                    os.remove(path)
                print("removed %s: rebuilding it failed (%s: %s)" % (path, type(e).__name__, e))
                if "--pybind" in sys.argv:
                    raise

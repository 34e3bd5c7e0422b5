"""Build libsmhip.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m simplemath_amd.build          # incremental
    python -m simplemath_amd.build --force

hipcc cross-compiles without a GPU; the .so lands in simplemath_amd/lib/ (git-ignored,
but it travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "lib", "obj")
LIB = os.path.join(LIBDIR, "libsmhip.so")
SOURCES = ["runtime.hip", "contiguous.hip", "broadcast.hip", "reduce.hip", "fill.hip", "fused.hip", "jit.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every float op is the single IEEE operation the reference's
# intrinsic performs; fusions are written explicitly (__builtin_fma).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "smhip.h"))
    return hs


def build_lib(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if warn.strip() and verbose:
                    print(warn, file=sys.stderr)
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lhiprtc"])
    return LIB


HOST_PROGRAMS = {
    # output (under simplemath_amd/bin/)  ->  source
    "test_reference_suite": os.path.join(ROOT, "tests", "cpp", "test_reference_suite.cpp"),
    "pow_host_check": os.path.join(ROOT, "tests", "cpp", "pow_host_check.cpp"),
    "pow64_host_check": os.path.join(ROOT, "tests", "cpp", "pow64_host_check.cpp"),
    "benchmark_add": os.path.join(PKG, "benchmark", "add.cpp"),
    "benchmark_pow": os.path.join(PKG, "benchmark", "pow.cpp"),
}
BINDIR = os.path.join(PKG, "bin")
CXX = os.environ.get("CXX", "g++")


def build_host_programs(force: bool = False) -> dict:
    """g++-compile the header-only C++ host side's programs against libsmhip.so."""
    os.makedirs(BINDIR, exist_ok=True)
    inc = os.path.join(ROOT, "include")
    hdrs = []
    for dp, _, fs in os.walk(inc):
        hdrs += [os.path.join(dp, f) for f in fs]
    hdrs += [os.path.join(CSRC, "sm_pow.h"), os.path.join(CSRC, "sm_pow64.h"), os.path.join(PKG, "benchmark", "minibench.h")]
    out = {}
    for name, src in HOST_PROGRAMS.items():
        exe = os.path.join(BINDIR, name)
        out[name] = exe
        if not (force or _newer(exe, [src] + hdrs)):
            continue
        if name in ("pow_host_check", "pow64_host_check"):  # pure host check of the pow algorithm: no GPU library involved
            cmd = [CXX, "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", f"-I{CSRC}", src, "-o", exe]
        else:
            cmd = [CXX, "-std=c++20", "-O2", "-Wall", "-Wextra", f"-I{inc}", f"-I{os.path.join(PKG, 'benchmark')}", src, "-o", exe,
                   f"-L{LIBDIR}", "-lsmhip", "-pthread", "-Wl,-rpath,$ORIGIN/../lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    for k, v in build_host_programs(force="--force" in sys.argv).items():
        print(v)

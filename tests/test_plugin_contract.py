"""The Op plugin contract of the drop-in headers (reference README.md:86-133, include/math/add.h:5-14), CPU side.

tests/cpp/readme_recipe.cpp restates the README's three steps.  Without a device form it must be rejected AT COMPILE
TIME with a message that names the fix; with the explicit opt-in it must build unmodified and give the reference's
values through its host apply() (no GPU involved: the three loop templates only ever see host pointers there).  The
device-form build is GPU-marked (tests/test_gpu_cpp.py)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "readme_recipe.cpp")
INC = os.path.join(ROOT, "include")
CXX = os.environ.get("CXX", "g++")


def test_recipe_without_a_device_form_is_a_compile_error_that_says_what_to_add():
    r = subprocess.run([CXX, "-std=c++20", "-mavx2", "-fsyntax-only", f"-I{INC}", SRC], capture_output=True, text=True)
    assert r.returncode != 0
    assert "static assertion failed" in r.stderr and "SM_DEVICE_OP" in r.stderr and "SM_DEFINE_OP" in r.stderr
    assert "SM_ALLOW_HOST_USER_OPS" in r.stderr
    assert "__m256" not in r.stderr.split("static assertion failed")[0].split("error")[-1]  # step 2 parsed: <immintrin.h> is there


def test_recipe_step_2_parses_on_its_own():
    """Only the static_assert stands between the verbatim recipe and a build: no `'__m256' does not name a type`."""
    r = subprocess.run([CXX, "-std=c++20", "-mavx2", "-fsyntax-only", f"-I{INC}", SRC], capture_output=True, text=True)
    assert "does not name a type" not in r.stderr and "was not declared" not in r.stderr


def test_recipe_unmodified_with_the_host_opt_in(smhip, tmp_path):
    from simplemath_amd import build
    exe = str(tmp_path / "recipe_host")
    cmd = [CXX, "-std=c++20", "-O2", "-mavx2", "-DSM_ALLOW_HOST_USER_OPS", f"-I{INC}", SRC, "-o", exe, f"-L{build.LIBDIR}", "-lsmhip",
           f"-Wl,-rpath,{build.LIBDIR}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "host apply" in r.stdout and " 0 mismatches" in r.stdout, r.stdout + r.stderr


def test_builtin_ops_never_take_the_host_path():
    """The opt-in is for USER Ops: a built-in Op on an element type without kernels stays a compile-time error."""
    prog = '#include <sm.h>\nint main() { unsigned short a[2] = {1, 2}, r[2]; handle_contiguous_arrays<unsigned short, AddOp<unsigned short>>(a, a, r, 2); }\n'
    r = subprocess.run([CXX, "-std=c++20", "-fsyntax-only", "-DSM_ALLOW_HOST_USER_OPS", f"-I{INC}", "-x", "c++", "-"], input=prog,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "no gfx950 kernels" in r.stderr
    prog = ('#include <sm.h>\n#include <complex>\nint main() { sm::SMArray<std::complex<double>> a = {{1, 2}}, b = {{3, 4}}; auto c = a + b; }\n')
    r = subprocess.run([CXX, "-std=c++20", "-fsyntax-only", f"-I{INC}", "-x", "c++", "-"], input=prog, capture_output=True, text=True)
    assert r.returncode != 0 and "no gfx950 kernels" in r.stderr  # std::complex arithmetic: diagnosed when compiled, not when run


def test_operands_on_different_gpus_are_refused_on_the_host(smhip):
    """an operator whose operands live on different GPUs (what sm::Sharded<T>::part(g) can hand out) throws
    std::runtime_error before any device call -- it used to launch on the calling thread's device.  No GPU needed."""
    from simplemath_amd import build
    exe = build.build_host_programs()["device_mismatch"]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "device_mismatch ok" in r.stdout, r.stdout + r.stderr

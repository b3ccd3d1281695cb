"""CPU: libdfgpu.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every symbol that
include/dfgpu.h declares; the ctypes binding declares exactly the same set.  No compute call is made here."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dfgpu.h")).read() + open(os.path.join(ROOT, "include", "dfgpu_exec.h")).read()
    return sorted(set(re.findall(r"DFGPU_API[^;(]*?\b(dfgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import dfgpu
    if not os.path.exists(dfgpu.capi.LIB_PATH):
        subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "datafusion-upstream_amd", "csrc")])
    lib = dfgpu.load_library()
    names = declared_symbols()
    assert len(names) >= 88
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dfgpu.h but not exported by libdfgpu.so"
    assert sorted(dfgpu.capi.PROTOTYPES) == names, "capi.PROTOTYPES must bind exactly the symbols of include/dfgpu.h + include/dfgpu_exec.h"


def test_exported_symbols_are_only_the_abi():
    import dfgpu
    out = subprocess.check_output(["nm", "-D", "--defined-only", dfgpu.capi.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "dfgpu_" in l.split()[-1] and not l.split()[-1].startswith("_Z"))
    assert exported == declared_symbols()


def test_context_creation_fails_loudly_without_a_gpu():
    """No CPU fallback: on a host without a HIP device the product raises instead of computing on the CPU."""
    import torch
    import dfgpu
    if torch.cuda.is_available():
        return
    try:
        dfgpu.Context(0)
    except dfgpu.DfgpuError as e:
        assert "no usable HIP device" in str(e)
    else:
        raise AssertionError("Context(0) must fail without a GPU")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "datafusion-upstream_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in src and "libdfo" not in src and "dfo_" not in src, f"{f} references the oracle"


def test_headers_are_plain_c_and_bind_from_a_c_program(tmp_path):
    """include/*.h must be consumable by a C compiler without HIP or C++ (what cgo / bindgen / ctypes see): tests/c/abi_probe.c includes
    both headers as C11, type-checks a handful of entry points against their declarations, links against libdfgpu.so and runs the
    calls that need no device (context creation reports a status; the run-time kernel text compiles)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "datafusion-upstream_amd")
    exe = str(tmp_path / "abi_probe")
    subprocess.check_call([gcc, "-std=c11", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_probe.c"),
                           "-o", exe, "-L", pkg, "-ldfgpu", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", ""))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "jit selftest 0" in out.stdout


def _null_probe(extra=()):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "null_probe.py"), *extra], capture_output=True, text=True, timeout=600)
    lines = [l.split() for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0, "crashed inside %s (exit %d)\n%s" % (lines[-1][0] if lines else "?", r.returncode, r.stderr[-1500:])
    return lines


def test_every_entry_point_refuses_null_handles():
    """A C-ABI entry point handed NULL for its handles answers DFGPU_INVALID_ARGUMENT; it never dereferences them (round 3 found dfgpu_array_slice crashing the
    host process that way).  Every function include/*.h declares is called with NULL pointers and zero scalars, in a child process, without a device."""
    lines = _null_probe()
    assert len(lines) >= 150
    for name, kind, value in lines:
        if kind == "status" and name != "dfgpu_jit_selftest":           # takes no handle: NULL arch / log are its defaults
            assert value == "5", f"{name}(NULL, ...) returned status {value}, expected DFGPU_INVALID_ARGUMENT"


@pytest.mark.gpu
def test_every_entry_point_refuses_null_handles_beside_a_live_context():
    """The same with a real context wherever a dfgpu_ctx* is taken: every other handle NULL.  No crash; an error status except where NULL is a documented
    argument (clearing the row selection, synchronising, reading options ...)."""
    lines = _null_probe(["--ctx"])
    ok_with_null = {"dfgpu_jit_selftest", "dfgpu_ctx_synchronize", "dfgpu_ctx_set_row_selection", "dfgpu_profile_enable", "dfgpu_profile_select", "dfgpu_ctx_trim"}
    for name, kind, value in lines:
        if kind == "status" and name not in ok_with_null:
            assert value != "0", f"{name}(ctx, NULL, ...) returned DFGPU_OK"

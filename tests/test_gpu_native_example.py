"""-m gpu: examples/q3_native.c -- TPC-H Q3 driven through the C ABI from plain C (no Python, no PyTorch in that process): built with
gcc against include/*.h + libdfgpu.so, run on the device, self-checked row for row against a host evaluation of the query."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_q3_through_the_c_abi_from_a_c_program(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    pkg = os.path.join(ROOT, "datafusion-upstream_amd")
    exe = str(tmp_path / "q3_native")
    subprocess.check_call([gcc, "-std=gnu11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "q3_native.c"),
                           "-o", exe, "-L", pkg, "-ldfgpu", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout

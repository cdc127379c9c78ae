"""The C ABI driven by a compiled, torch-free caller (tests/abi/abi_smoke.cpp, built by __graft_entry__.build()):
device memory from the HIP runtime, weights packed on the host as include/irm_hip.h describes, its own stream."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_c_abi_from_compiled_caller(dev):
    exe = os.path.join(HERE, "abi", "abi_smoke")
    if not os.path.exists(exe):
        pytest.fail("tests/abi/abi_smoke is not built: run __graft_entry__.build()")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout

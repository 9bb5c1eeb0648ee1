"""Randomised small systems through fdapde_solve (tools/fuzz_small.py, 40 cases x 3 solves here): the sizes of the reference's own meshes -- one-workgroup
launches, k_small_front, the single-launch BiCGStab -- against scipy's LU of the system the product hands out (fem_solver_base.h:142-155 form), 1e-8."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 19])
def test_random_small_systems_against_lu(seed):
    from fdapde_loader import load_package

    assert load_package().capi.load().fdapde_device_count() >= 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_small.py"), "40", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failures 0" in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_random_handles_and_parabolic_problems_against_lu():
    """tools/fuzz_handle.py: the factor-once handle with 1 / 3 / 70 right-hand sides (zero-copy launch, columns side by side, more columns than one launch
    takes) on mass / stiff / non-symmetric matrices, and implicit Euler over 2-6 steps with and without Dirichlet data, against scipy (1e-8)"""
    from fdapde_loader import load_package

    assert load_package().capi.load().fdapde_device_count() >= 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_handle.py"), "30", "29"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failures 0" in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_random_call_sequences_leave_no_stale_state():
    """tools/fuzz_sequence.py: 25 random sequences of boundary calls on one context; every solve against the same problem on a fresh context (1e-9)"""
    from fdapde_loader import load_package

    assert load_package().capi.load().fdapde_device_count() >= 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sequence.py"), "25", "41"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failures 0" in r.stdout.splitlines()[-1], r.stdout[-500:]

"""The header-only C++20 facade (include/fdapde_amd/pde.h): compiles with g++ -std=c++20 here (CPU), and on the GPU runs the
reference's elliptic fem_pde_test cases re-expressed against it (tests/cpp/fem_pde_test.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "fem_pde_test")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")])


def test_facade_compiles_and_refuses_to_run_without_a_device():
    _build()
    assert os.path.exists(EXE)
    import ctypes

    lib = ctypes.CDLL(os.path.join(ROOT, "fdapde-core_amd", "lib", "libfdapde_hip.so"))
    if lib.fdapde_device_count() == 0:
        r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden", "mesh")], capture_output=True, text=True)
        assert r.returncode == 3 and "no CPU fallback" in r.stdout


def test_host_side_io_of_the_facade():
    """the CSV dialects (dense + the 3-column sparse form, csv_reader.h:119-166) need no device: tests/cpp/fem_pde_test --io-only"""
    _build()
    r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden", "mesh"), "--io-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_fem_pde_cases_through_the_cpp_facade():
    _build()   # make: rebuilt whenever a header of the facade or the C ABI changed
    r = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden", "mesh")], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout

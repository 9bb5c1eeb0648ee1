"""The multi-threaded host-side set-up (csrc/host_setup.cpp: relaxed-atomic counting sorts, uninitialised-resize vectors, open
addressing sets, lane permutations; csrc/host_persist.cpp: the persistent CG's resident layout, whose operator application is replayed on the CPU and
compared with the CSR product) under AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU build -- GPU sanitizers are
not available on this pool.  tests/cpp/host_setup_sanitize.cpp drives P1 / P2 meshes in 2-D / 3-D at sizes where every
parallel_for really runs on several threads."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "cpp", "host_setup_sanitize.cpp"), os.path.join(ROOT, "fdapde-core_amd", "csrc", "host_setup.cpp"),
       os.path.join(ROOT, "fdapde-core_amd", "csrc", "host_persist.cpp"), os.path.join(ROOT, "fdapde-core_amd", "csrc", "tables.cpp")]


@pytest.mark.parametrize("name,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]),
                                        ("tsan", ["-fsanitize=thread"])])
def test_host_setup_is_sanitizer_clean(tmp_path, name, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / f"host_setup_{name}")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-fno-omit-frame-pointer", *flags, *SRC, "-pthread", "-o", exe])
    env = dict(os.environ, FDAPDE_THREADS="8", ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ERROR" not in out.stderr and "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-4000:]
    assert out.stdout.count("dofs") == 6

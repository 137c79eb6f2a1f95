"""The host-side entry points of the C ABI (track chaining, the random stream, the adaptive-loop replay, pose
composition) under AddressSanitizer + UBSan: tests/host_asan_driver.cpp calls them with heap buffers of exactly the
documented sizes and compares the results with plain restatements.  Sanitizers run on host code only (hipcc
-Xarch_host); the driver makes no GPU call, so it runs here."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")
def test_host_entry_points_under_address_and_ub_sanitizers(tmp_path):
    exe = str(tmp_path / "host_asan_driver")
    src = [os.path.join(ROOT, "tests", "host_asan_driver.cpp")] + [os.path.join(ROOT, "slam_amd", "csrc", f)
                                                                  for f in ("tracks_host.cpp", "ransac_host.cpp", "api.cpp")]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-x", "hip",
                           "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
                           "-Xarch_host", "-fno-sanitize-recover=undefined"] + src + ["-o", exe, "-lpthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", SLAM_HOST_THREADS="4")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr[-4000:]
    assert "bad=0" in p.stdout and "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]

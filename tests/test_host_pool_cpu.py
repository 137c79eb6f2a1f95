"""The host worker pool of slam_track_pairs (slam_amd/csrc/host_pool.h) under ThreadSanitizer: jobs of alternating
sizes back to back, every chunk exactly once, nothing running after run() returns, no data race reported."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_pool_stress_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "host_pool_stress")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", os.path.join(ROOT, "tests", "host_pool_stress.cpp"),
                           "-o", exe, "-lpthread"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "bad=0" in p.stdout
    assert "ThreadSanitizer" not in p.stderr, p.stderr[:2000]

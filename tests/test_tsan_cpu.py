"""ThreadSanitizer over the HOST code of the library: tools/tsan_host.sh builds every translation unit with
-Xarch_host -fsanitize=thread (the device code cannot be instrumented, and sanitizers do not run on the GPU pool) and runs
tests/tsan/tsan_host.cpp - the entry points that work without a device, from two threads.  Clean = exit code 0, no report."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("bash") is None, reason="needs the ROCm toolchain")
def test_host_paths_are_clean_under_thread_sanitizer(tmp_path_factory):
    out = os.environ.get("BH_TSAN_DIR", "/tmp/bh_tsan")          # object files are kept between runs (the build takes about a minute)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan_host.sh"), out], capture_output=True, text=True, timeout=900)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    assert "ThreadSanitizer" not in log, log[-3000:]
    assert "tsan host test: bad=0" in log

"""Host code of the C-ABI library under a sanitizer build (the reference has no sanitizer run; GPU sanitizers are not available on the
target pool, so this is the HOST side only): tests/host_asan/build.sh compiles every translation unit of csrc/ with the host pass instrumented
(address + undefined-behaviour checks), tests/host_asan/driver.cpp walks every size / plan / capability query over the bench model's shapes
and a set of hostile arguments, and every compute entry point's argument validation (which must fail before the first HIP call).  No GPU
needed; the first run compiles for ~1.5 min, later runs are incremental.  The recipe lives in tests/host_asan/ (not shipped to the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "host_asan", "build.sh")


@pytest.mark.timeout(1500)
def test_host_paths_clean_under_sanitizers():
    if shutil.which("hipcc") is None or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or not os.path.exists(SCRIPT):
        pytest.skip("needs the ROCm toolchain and tests/host_asan/ (CPU-side only)")
    r = subprocess.run(["bash", SCRIPT], capture_output=True, text=True, timeout=1400)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("ok ") and int(last.split()[1]) >= 200, last

"""CPU sanitizer runs (AddressSanitizer + UndefinedBehaviorSanitizer on the host library and the oracle,
ThreadSanitizer on the hierarchy builder), driven by tools/run_sanitizers.sh: the script rebuilds the
instrumented libraries and runs the host / builder / oracle tests inside them.  No GPU code is involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True)
    path = out.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(os.environ.get("RTMI_HOST_ONLY") == "1", reason="already inside a sanitizer run")
def test_host_and_oracle_are_clean_under_sanitizers():
    if shutil.which("gcc") is None or not all(_runtime(n) for n in ("libasan.so", "libubsan.so", "libtsan.so")):
        pytest.skip("gcc sanitizer runtimes not installed")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_sanitizers.sh"), "quick"], capture_output=True, text=True,
                         timeout=600)
    logs = ""
    for name in ("r03_sanitizers_asan.log", "r03_sanitizers_tsan.log"):
        with open(os.path.join(os.environ.get("TMPDIR") or "/tmp", name)) as f:   # where the quick mode writes
            logs += f.read()[-1500:]
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:] + logs
    assert "passed" in logs and "failed" not in logs

"""The RCCL leg of the multi-GPU path, as far as one GPU allows: tools/rccl_single_gpu_check.py in a
process of its own (it initialises a process group)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_all_reduce_on_the_bound_image():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_single_gpu_check.py")], capture_output=True,
                         text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "OK RCCL" in out.stdout

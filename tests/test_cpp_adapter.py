"""The cbird-side C++ binding (cbird_amd/cpp/gpu_dcthashindex.h): compiles against a mock of the
reference's index.h/Qt types on CPU; built and executed against the real library on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_adapter_compiles_against_index_interface():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_adapter"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(CPP, "test_adapter"))
    # every pure-virtual of the reference's Index (src/index.h:172-258) is overridden
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_dcthashindex.h")).read()
    for name in ("isLoaded", "memoryUsage", "count", "load", "save", "mediaIds", "add", "remove", "find",
                 "slice"):
        assert f" {name}(" in src and "override" in src


@pytest.mark.gpu
def test_adapter_runs_on_gpu(gpu):
    subprocess.check_call(["make", "-C", CPP, "test_adapter"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_adapter")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter ok" in out.stdout

"""The cbird-side C++ bindings (cbird_amd/cpp/gpu_dcthashindex.h and gpu_indexes.h: all five Index
subclasses): compiled against a mock of the reference's headers/Qt types on CPU; built and executed
against the real library on the GPU box."""
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


def test_other_four_adapters_compile():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_adapters4"], stdout=subprocess.DEVNULL)
    src = open(os.path.join(ROOT, "cbird_amd", "cpp", "gpu_indexes.h")).read()
    for cls in ("GpuDctFeaturesIndex", "GpuCvFeaturesIndex", "GpuColorDescIndex", "GpuDctVideoIndex"):
        assert f"class {cls} : public" in src


@pytest.mark.gpu
def test_other_four_adapters_run_on_gpu(gpu, tmp_path):
    subprocess.check_call(["make", "-C", CPP, "test_adapters4"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_adapters4"), str(tmp_path)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapters ok" in out.stdout

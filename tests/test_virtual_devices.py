"""The code that only runs with MORE THAN ONE device ordinal, executed on a one-GPU box.

No box of this pool has two GPUs, so through round 3 the D > 1 branches of cbird_amd/csrc/sharded.hip (DeviceGuard
switching, one stream / arena / workspace set per device, needle replication and the exchange by hipMemcpyPeerAsync,
cross-device event waits, the device-mask plumbing, GpuDeviceSet::all()) had never executed.  tests/shim/vdev.c is an
LD_PRELOAD test double that shows the one physical GPU as CBH_VDEV ordinals; under it tests/test_sharded_capi.py -- all
the one-device suites re-run through sharded handles and held against the oracle and the reference's golden vectors --
runs again with shapes that span several ordinals (all four, a sparse mask 0b1101, 2 devices x 2 shards, and the
collective shape falling back to copies), and the C++ adapter asks for GpuDeviceSet::all().  The shim counts what it
saw, so a pass cannot come from everything quietly landing on ordinal 0.

What this cannot show is anything physical -- real peer mappings, xGMI, RCCL between devices: tools/first_contact.sh
remains the thing to run on the first real multi-GPU box.

(The shim is preloaded into CHILD processes only, started before anything there touches the GPU.)"""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIM_SRC = os.path.join(HERE, "shim", "vdev.c")
SHIM = os.path.join(HERE, "shim", "libvdev.so")
CPP = os.path.join(HERE, "cpp")


def build_shim():
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-o", SHIM, SHIM_SRC, "-ldl"])
    return SHIM


def shim_env(n):
    env = dict(os.environ)
    env["LD_PRELOAD"] = SHIM + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    env["CBH_VDEV"] = str(n)
    return env


def test_shim_builds_and_interposes_only_ordinal_taking_entry_points():
    """CPU: the shim compiles, and every symbol it defines is a HIP runtime entry point that takes or returns a device
    ordinal (plus its own counter) -- it must not shadow anything that computes"""
    build_shim()
    out = subprocess.run(["nm", "-D", "--defined-only", SHIM], capture_output=True, text=True, check=True).stdout
    syms = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert syms == {"hipGetDeviceCount", "hipSetDevice", "hipGetDevice", "hipGetDevicePropertiesR0600",
                    "hipDeviceGetAttribute", "hipDeviceGetPCIBusId", "hipDeviceGetName", "hipDeviceTotalMem",
                    "hipDevicePrimaryCtxGetState", "hipDeviceGetP2PAttribute", "hipDeviceCanAccessPeer",
                    "hipDeviceEnablePeerAccess", "hipMemcpyPeer", "hipMemcpyPeerAsync", "hipDeviceGetDefaultMemPool",
                    "hipDeviceGetMemPool", "hipMemPoolCreate", "vdev_stat",
                    # ... and the ones that create / use a stream or an event, only to check which ordinal they belong to
                    "hipStreamCreate", "hipStreamCreateWithFlags", "hipStreamCreateWithPriority", "hipStreamDestroy",
                    "hipGetStreamDeviceId", "hipEventCreate", "hipEventCreateWithFlags", "hipEventDestroy",
                    "hipEventRecord", "hipLaunchKernel"}


@pytest.mark.gpu
def test_library_sees_the_virtual_ordinals(gpu):
    build_shim()
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from cbird_amd import _lib\n"
            "L = _lib.lib()\n"
            "print('count', L.cbh_device_count(), 'mask', hex(L.cbh_usable_device_mask()))\n"
            "assert L.cbh_idx64_create_sharded(0x1f, 1) is None  # ordinal 4 does not exist\n"
            "h = L.cbh_idx64_create_sharded(0xd, 1)\n"
            "assert h and L.cbh_idx64_device_mask(h) == 0xd and L.cbh_idx64_shard_count(h) == 3\n"
            "L.cbh_idx64_destroy(h)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=shim_env(4), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "count 4 mask 0xf" in out.stdout


@pytest.mark.gpu
def test_sharded_suites_over_four_virtual_devices(gpu):
    """tests/test_sharded_capi.py again, in a child process under the shim, with the shapes `_shapes()` gives when
    CBH_VDEV is set; its own last test reads the shim's counters"""
    build_shim()
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_sharded_capi.py"), "-x", "-q", "-m", "gpu",
                          "-p", "no:cacheprovider"], env=shim_env(4), capture_output=True, text=True, timeout=2400,
                         cwd=ROOT)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    m = re.search(r"(\d+) passed", out.stdout)
    assert m and int(m.group(1)) >= 4 * 25 and "failed" not in out.stdout, tail


@pytest.mark.gpu
def test_adapter_over_all_virtual_devices(gpu):
    """GpuDctHashIndex(GpuDeviceSet::all()) -- what INTEGRATION.md tells Engine::Engine to construct on a multi-GPU node --
    with cbh_usable_device_mask() = 0xff: the whole adapter test, every find also held against the one-device index"""
    build_shim()
    subprocess.check_call(["make", "-C", CPP, "test_adapter"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(CPP, "test_adapter"), "all"], env=shim_env(8), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter ok" in out.stdout and "shards 8 devices 8" in out.stdout
    m = re.search(r"peer copies (\d+)", out.stdout)
    assert m and int(m.group(1)) > 0, out.stdout

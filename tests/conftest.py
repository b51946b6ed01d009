import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def gpu():
    """The product library on a box with a usable device; fails loudly (no skip, no fallback)."""
    import cbird_amd

    cbird_amd.require_device()
    return cbird_amd


@pytest.fixture(params=["mfma", "mfma_pre", "mfma_full", "valu", "join"])
def scan_path(request, gpu):
    """Run a GPU test once per 64-bit scan kernel family, all of which must be bit-exact against the oracle:
    "mfma"      the matrix-core scan forced for any size, kernels chosen as shipped ("scan_mfma_pre_max" -1: thresholds
                <= 8 by the launch's candidate rate -- <= 6 for launches too small to probe --, 9..64 the three-field
                kernel, 65 the two-field one);
    "mfma_pre"  the prefilter kernel for every threshold it can represent (<= 32), however dense its candidates;
    "mfma_full" never the prefilter: thresholds 1..64 on the three-field kernel;
    "valu"      the popcount kernel k_hamm64_scan;
    "join"      the bucketed join (hamm64_join.hip) for every call it can take (thresholds <= 8, no needle masks), whatever its
                candidate count; the rest as "mfma"."""
    from cbird_amd import _lib

    L = _lib.lib()
    L.cbh_set_tuning(b"scan_mfma", {"valu": 0, "join": 4}.get(request.param, 2))
    L.cbh_set_tuning(b"scan_mfma_pre_max", {"mfma_pre": 32, "mfma_full": 0}.get(request.param, -1))
    yield request.param
    L.cbh_set_tuning(b"scan_mfma", 1)
    L.cbh_set_tuning(b"scan_mfma_pre_max", -1)


@pytest.fixture(params=["mfma", "mfma_rows", "valu"])
def scan256_path(request, gpu):
    """As scan_path, for the 256-bit scan: the matrix-core kernels forced for any size -- "mfma": as shipped (searches
    with <= 512 needle descriptors and thresholds <= 40 on the stationary-needle kernel k_hamm256_small, the rest on
    k_hamm256_mfma3 / k_hamm256_mfma), "mfma_rows": without k_hamm256_small (the prefilter with three needle tiles per
    accumulator, k_hamm256_mfma3, from 65 needle descriptors up; fewer, and thresholds > 40: k_hamm256_mfma) -- then
    "valu": k_hamm256_scan."""
    from cbird_amd import _lib

    L = _lib.lib()
    L.cbh_set_tuning(b"scan256_mfma", 0 if request.param == "valu" else 2)
    L.cbh_set_tuning(b"scan256_small", 1 if request.param == "mfma" else 0)
    yield request.param
    L.cbh_set_tuning(b"scan256_mfma", 1)
    L.cbh_set_tuning(b"scan256_small", 1)


@pytest.fixture(params=["band", "valu"])
def hash256_kernel(request, gpu):
    """Run a GPU hash test once per kernel for 256 x 256 tiles, both bit-exact against the oracle: "band" =
    k_dcthash_256_band (the default: horizontal box sums on the matrix cores), "valu" = k_dcthash_256."""
    from cbird_amd import _lib

    _lib.lib().cbh_set_tuning(b"hash_mfma", 2 if request.param == "band" else 0)
    yield request.param
    _lib.lib().cbh_set_tuning(b"hash_mfma", 2)


@pytest.fixture(params=["device", "host"])
def reduce_path(request, gpu):
    """DctFeaturesIndex / DctVideoIndex finds with their per-needle reductions on the device (reduce.hip: K5 votes,
    K8 closest frame + adjacency; the shipped path) and on the host (the round-1 std::map loops, knobs
    "fdct_host_vote" / "video_host_reduce"): two implementations, one oracle."""
    from cbird_amd import _lib

    v = 1 if request.param == "host" else 2  # (0 = auto: device for batches, host for a single needle)
    _lib.lib().cbh_set_tuning(b"fdct_host_vote", v)
    _lib.lib().cbh_set_tuning(b"video_host_reduce", v)
    yield request.param
    _lib.lib().cbh_set_tuning(b"fdct_host_vote", 0)
    _lib.lib().cbh_set_tuning(b"video_host_reduce", 0)


@pytest.fixture(params=["one", "shards5", "rccl3"])
def index_shape(request, gpu):
    """Run a GPU test once per shape of the 64-bit index handle, all of which must give the same answers:
    "one"     the plain one-device index (cbh_idx64_create);
    "shards5" cbh_idx64_create_sharded(1 << 0, 5): five logical shards on the one GPU of this pool, five streams,
              device-to-device copies as the exchange -- ragged shares, the merge, overflow-redo, removal;
    "rccl3"   three logical shards whose concatenated device block additionally travels through ncclAllGather on a
              one-rank communicator ("shard_force_rccl"): librccl's transport as the multi-GPU exchange uses it.
    Every DctHashIndex / DctFeaturesIndex / DctVideoIndex a test creates takes the shape (_lib.set_default_sharding)."""
    from cbird_amd import _lib

    shape = {"one": None, "shards5": (1, 5), "rccl3": (1, 3)}[request.param]
    _lib.set_default_sharding(shape)
    _lib.lib().cbh_set_tuning(b"shard_force_rccl", 1 if request.param == "rccl3" else 0)
    _lib.lib().cbh_set_tuning(b"shard_exchange", 0 if request.param == "rccl3" else 1)  # (default: copies)
    yield request.param
    _lib.set_default_sharding(None)
    _lib.lib().cbh_set_tuning(b"shard_force_rccl", 0)
    _lib.lib().cbh_set_tuning(b"shard_exchange", 1)


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def golden_cases():
    return ["vptree_n4096.npz", "vptree_n32768.npz"]

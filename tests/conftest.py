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


@pytest.fixture(params=["mfma", "mfma_r4", "mfma2", "valu"])
def scan_path(request, gpu):
    """Run a GPU test once per 64-bit scan kernel family, all of which must be bit-exact against the oracle:
    "mfma"    the matrix-core scan forced for any size, as shipped (thresholds <= 6: the prefilter variant on lo ^ hi
              with the deferred re-check, 7..64: three needle tiles per accumulator, 65: two);
    "mfma_r4" the prefilter as rounds 1-4 had it (low word, every candidate group through the per-tile queue path),
              for the thresholds it serves now;
    "mfma2"   as "mfma" with the three-tile variant off (thresholds >= 7 on the two-tile kernel);
    "valu"    the popcount kernel k_hamm64_scan."""
    from cbird_amd import _lib

    L = _lib.lib()
    L.cbh_set_tuning(b"scan_mfma", 0 if request.param == "valu" else 2)
    L.cbh_set_tuning(b"scan_mfma_full3", 0 if request.param == "mfma2" else 1)
    L.cbh_set_tuning(b"scan_pre_fold", 0 if request.param == "mfma_r4" else 1)
    L.cbh_set_tuning(b"scan_pre_lean", 0 if request.param == "mfma_r4" else 1)
    yield request.param
    L.cbh_set_tuning(b"scan_mfma", 1)
    L.cbh_set_tuning(b"scan_mfma_full3", 1)
    L.cbh_set_tuning(b"scan_pre_fold", 1)
    L.cbh_set_tuning(b"scan_pre_lean", 1)


@pytest.fixture(params=["mfma", "mfma_rows", "mfma_rows1", "valu"])
def scan256_path(request, gpu):
    """As scan_path, for the 256-bit scan: the matrix-core kernels forced for any size -- "mfma": as shipped (searches
    with <= 512 needle descriptors and thresholds <= 40 on the stationary-needle kernel k_hamm256_small, the rest on
    k_hamm256_mfma3 / k_hamm256_mfma), "mfma_rows": without k_hamm256_small (the prefilter with three needle tiles per
    accumulator, k_hamm256_mfma3, from 65 needle descriptors up), "mfma_rows1": k_hamm256_mfma alone (one tile per
    accumulator) -- then "valu": k_hamm256_scan."""
    from cbird_amd import _lib

    L = _lib.lib()
    L.cbh_set_tuning(b"scan256_mfma", 0 if request.param == "valu" else 2)
    L.cbh_set_tuning(b"scan256_small", 1 if request.param == "mfma" else 0)
    L.cbh_set_tuning(b"scan256_f3", 0 if request.param == "mfma_rows1" else 1)
    yield request.param
    L.cbh_set_tuning(b"scan256_mfma", 1)
    L.cbh_set_tuning(b"scan256_small", 1)
    L.cbh_set_tuning(b"scan256_f3", 1)


@pytest.fixture(params=["band", "valu"])
def hash256_kernel(request, gpu):
    """Run a GPU hash test once per kernel for 256 x 256 tiles, both bit-exact against the oracle: "band" =
    k_dcthash_256_band (the default: horizontal box sums on the matrix cores), "valu" = k_dcthash_256."""
    from cbird_amd import _lib

    _lib.lib().cbh_set_tuning(b"hash_mfma", 2 if request.param == "band" else 0)
    yield request.param
    _lib.lib().cbh_set_tuning(b"hash_mfma", 2)


@pytest.fixture(params=["cvdct", "canon"])
def hash_dct(request, gpu, orc):
    """Run a GPU hash test under both evaluations of dctHash64's stages 3/5, each bit-exact against the oracle set to
    the same one: "cvdct" = cv::dct / cv::sum as OpenCV 2.4 evaluates them (cv_dct32_dev.h / oracle/cv_dct32.c, the
    default), "canon" = the canonical 9x32 matrix form."""
    from cbird_amd import _lib

    v = 1 if request.param == "cvdct" else 0
    _lib.lib().cbh_set_tuning(b"hash_dct", v)
    orc.set_hash_variant(v)
    yield request.param
    _lib.lib().cbh_set_tuning(b"hash_dct", 1)
    orc.set_hash_variant(1)


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

"""Device-memory soak: handles created and destroyed, batch entry points called in a loop; free device memory before and
after each phase (a leak shows as a steady decrease).    python tools/leak_soak.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    from cbird_amd import DctHashIndex, orb
    from cbird_amd.hashing import process_images as hash_images, template_scores
    from cbird_amd.scanner import IndexParams, process_images
    from cbird_amd.video import VideoIndexer

    orb.set_pattern(orb.synthetic_pattern())
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (64, 180, 240), dtype=np.uint8)
    bgr = rng.integers(0, 256, (64, 120, 160, 3), dtype=np.uint8)
    tmpl = rng.integers(0, 256, (120, 160, 4), dtype=np.uint8)
    hashes = rng.integers(1, 2 ** 63, 20000, dtype=np.uint64)
    ids = np.arange(1, 20001, dtype=np.uint32)

    def free():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    def video():
        ix = VideoIndexer(threshold=8)
        ix.push(frames[:40])
        ix.push(frames[40:])
        ix.finish()

    def index():
        idx = DctHashIndex()
        idx.load(hashes, ids)
        idx.find_batch(hashes[:512], 5, 8)

    def external_stream():
        """a caller that cycles its own streams (cbird's worker threads come and go): every call arrives on a NEW
        torch stream that is dropped afterwards -- the library's per-stream scratch pools must not pile up"""
        from cbird_amd import _lib
        import ctypes as C
        L = _lib.lib()
        s = torch.cuda.Stream()
        dq = torch.from_numpy(hashes[:4096].view(np.int64)).cuda()
        with torch.cuda.stream(s):
            out = torch.empty((4096, 8, 2), dtype=torch.int32, device="cuda")
            cnt = torch.empty(4096, dtype=torch.int32, device="cuda")
            tot = C.c_uint64(0)
            _lib.check(L.cbh_idx64_find_batch_dev(ext_idx.handle, dq.data_ptr(), 4096, 6, 8, out.data_ptr(), cnt.data_ptr(),
                                                  C.byref(tot), s.cuda_stream), "find_batch_dev")
        s.synchronize()
        del s

    ext_idx = DctHashIndex()
    ext_idx.load(hashes, ids)
    phases = {"external streams: one new caller stream per call (300 streams)": external_stream,
              "video indexer create/push/finish/destroy": video,
              "cbh_index_images (all algorithms, 64 images)": lambda: process_images(bgr, IndexParams(algos=15, numFeatures=60)),
              "cbh_process_images (64 frames)": lambda: hash_images(frames, 20),
              "cbh_template_scores (64 candidates)": lambda: template_scores(bgr, tmpl),
              "DctHashIndex create/load/find_batch/destroy": index}
    out = {}
    for name, fn in phases.items():
        for _ in range(5):
            fn()
        f0 = free()
        for _ in range(150):
            fn()
        mid = free()
        for _ in range(150):
            fn()
        f1 = free()
        out[name] = {"free_MB_after_warmup": f0 >> 20, "after_150": mid >> 20, "after_300": f1 >> 20}
    # cbh_trim: what the pools still hold goes back to the driver
    from cbird_amd import _lib
    import ctypes as C
    rel = C.c_ulonglong(0)
    before = free()
    _lib.check(_lib.lib().cbh_trim(0, C.byref(rel)), "trim")
    out["cbh_trim"] = {"released_MB": rel.value >> 20, "free_MB_before": before >> 20, "free_MB_after": free() >> 20}
    print(json.dumps(out))
    bad = [k for k, v in out.items() if "after_150" in v and v["after_150"] - v["after_300"] > 64]
    assert not bad, bad


if __name__ == "__main__":
    main()

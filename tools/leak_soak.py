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

    phases = {"video indexer create/push/finish/destroy": video,
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
    print(json.dumps(out))
    bad = [k for k, v in out.items() if v["after_150"] - v["after_300"] > 64]
    assert not bad, bad


if __name__ == "__main__":
    main()

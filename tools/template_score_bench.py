"""TemplateMatcher scoring block (cbh_template_scores: mask, two dctHash64, hamm64) -- candidates/s host to host, and the
oracle's single-core time beside it.    python tools/template_score_bench.py [--n 512] [--w 640 --h 480]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--w", type=int, default=640)
    ap.add_argument("--h", type=int, default=480)
    a = ap.parse_args()
    from cbird_amd.hashing import template_scores
    from oracle import PrestageOracle

    rng = np.random.default_rng(1)
    tmpl = rng.integers(0, 256, (a.h, a.w, 4), dtype=np.uint8)
    cands = np.zeros((a.n, a.h, a.w, 3), np.uint8)
    for i in range(a.n):
        m = int(rng.integers(0, a.h // 6))
        cands[i, m:a.h - m, m:a.w - m] = np.roll(tmpl[..., :3], int(rng.integers(-8, 9)), axis=1)[m:a.h - m, m:a.w - m]
    template_scores(cands[:8], tmpl)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        s, ch, th = template_scores(cands, tmpl)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    po = PrestageOracle()
    t0 = time.perf_counter()
    k = min(8, a.n)
    want = [po.template_score(cands[i], tmpl)[0] for i in range(k)]
    cpu = (time.perf_counter() - t0) / k
    assert s[:k].tolist() == want
    print(json.dumps({"workload": f"{a.n} warped BGR candidates {a.w}x{a.h} against one BGRA template, host in / host out",
                      "s": round(best, 4), "candidates_per_s": a.n / best, "upload_GBps": cands.nbytes / best / 1e9,
                      "cpu_oracle_ms_per_candidate_1core": round(cpu * 1e3, 2)}))


if __name__ == "__main__":
    main()

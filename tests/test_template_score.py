"""TemplateMatcher::match's scoring block (src/templatematcher.cpp:331-374): cbh_template_scores against the oracle's
literal restatement of the masking loop + the two dctHash64 + hamm64."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def po():
    from oracle import PrestageOracle

    return PrestageOracle()


def scene(rng, h, w, ch):
    img = np.zeros((h, w, ch), np.int32) + 120
    for _ in range(12):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        img[y0:y0 + int(rng.integers(6, h // 2)), x0:x0 + int(rng.integers(6, w // 2))] = rng.integers(0, 256, ch)
    img += rng.integers(-5, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def warped(rng, tmpl, margin, shift):
    """a candidate patch as warpAffine leaves it: the template's content, displaced, inside a black undefined margin"""
    c = np.zeros_like(tmpl)
    h, w = tmpl.shape[:2]
    c[margin:h - margin, margin:w - margin] = np.roll(tmpl, shift, axis=1)[margin:h - margin, margin:w - margin]
    c[margin + 3, margin + 5] = 0  # a genuinely black pixel inside the patch masks the template there as well
    return c


def test_oracle_template_score_rules(po):
    rng = np.random.default_rng(3)
    t = scene(rng, 90, 120, 3)
    # identical, fully defined patch: nothing is masked, score 0
    full = np.maximum(t, 1)
    d, ch, th, cg, tg = po.template_score(full, full)
    assert d == 0 and ch == th and (cg == tg).all()
    # an undefined margin zeroes the same pixels of the template: still comparable
    c = warped(rng, np.maximum(t, 1), 10, 0)
    d, ch, th, cg, tg = po.template_score(c, np.maximum(t, 1))
    assert d == 0 and (tg[:10] == 0).all() and (tg[:, :10] == 0).all() and (cg == tg).all()
    # BGRA template: colour premultiplied by alpha (>> 8), candidate's grey scaled by the same alpha
    a = np.full((90, 120, 1), 128, np.uint8)
    d, ch, th, cg, tg = po.template_score(full, np.dstack([full, a[..., 0]]))
    g = po.bgr2gray(full)
    assert (cg == ((g.astype(np.int32) * 128) >> 8)).all()
    assert (tg == po.bgr2gray((full.astype(np.int32) * 128 >> 8).astype(np.uint8))).all()
    # grey on grey
    d, _, _, cg, tg = po.template_score(g, g)
    assert d == 0 and (cg == g).all() and (tg == np.where(g != 0, g, 0)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(90, 120), (256, 256), (300, 417), (33, 47)])
@pytest.mark.parametrize("cc,tc", [(1, 1), (3, 3), (3, 4), (4, 4), (1, 3), (4, 1)])
def test_template_scores_equal_oracle(gpu, po, geom, cc, tc):
    from cbird_amd.hashing import template_scores

    h, w = geom
    rng = np.random.default_rng(h * 1000 + w + 10 * cc + tc)
    base = scene(rng, h, w, 4)
    base[..., 3] = rng.integers(0, 256, (h, w))
    tmpl = base[..., 0] if tc == 1 else base[..., :tc]
    src = base[..., 0] if cc == 1 else base[..., :cc]
    cands = np.stack([warped(rng, np.ascontiguousarray(src), int(rng.integers(0, min(h, w) // 4)),
                             int(rng.integers(-6, 7))) for _ in range(5)])
    scores, ch, th = template_scores(cands, np.ascontiguousarray(tmpl))
    for i in range(len(cands)):
        d, wc, wt, _, _ = po.template_score(cands[i], tmpl)
        assert (int(scores[i]), int(ch[i]), int(th[i])) == (d, wc, wt), i


@pytest.mark.gpu
def test_template_scores_errors_and_empty(gpu):
    from cbird_amd._lib import CbhError
    from cbird_amd.hashing import template_scores

    s, ch, th = template_scores(np.zeros((0, 40, 40), np.uint8), np.zeros((40, 40), np.uint8))
    assert len(s) == 0
    with pytest.raises(ValueError):
        template_scores(np.zeros((1, 40, 40), np.uint8), np.zeros((41, 40), np.uint8))
    with pytest.raises(CbhError):
        template_scores(np.zeros((1, 40, 40, 2), np.uint8), np.zeros((40, 40), np.uint8))

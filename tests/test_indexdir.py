"""cbird `_index/` directory reader (cbird_amd/indexdir.py): table layouts and blob
encodings follow the reference's createTables/addRecords/load (cited in the module); the writer is the
fixture generator.  GPU part: an index loaded from the directory answers like one filled directly."""
import os
import sqlite3
import struct
import zlib

import numpy as np
import pytest


def _fixture(tmp_path, n_img=120, per=24, n_vid=6):
    from cbird_amd import synth, synth_video
    from cbird_amd.colordesc import COLOR_DTYPE
    from cbird_amd.indexdir import MediaRow, TYPE_IMAGE, TYPE_VIDEO, write_index_dir
    from cbird_amd.video import VideoIndex

    rng = np.random.default_rng(5)
    h, ids = synth.make_hashes(n_img, seed=5, planted_frac=0.3)
    h[7] = np.uint64(0xFFFFFFFFFFFFFFFE)  # needs the signed <-> unsigned mapping of sqlite integers
    media = [MediaRow(int(i), TYPE_IMAGE, f"img/{int(i)}.jpg", 256, 256, f"{int(i):032x}", int(x))
             for i, x in zip(ids, h)]
    kp = [(int(i), rng.integers(1, 1 << 63, int(rng.integers(0, 9)), dtype=np.uint64)) for i in ids]
    mats = [(int(i), rng.integers(0, 256, (per if k % 17 else 0, 32), dtype=np.uint8)) for k, i in enumerate(ids)]
    cols = np.zeros(n_img, COLOR_DTYPE)
    cols["colors"] = rng.integers(0, 65536, (n_img, 32, 4), dtype=np.uint16)
    cols["numColors"] = rng.integers(20, 32, n_img, dtype=np.uint8)
    clips = synth_video.make_clips(n_vid, 60, seed=3, subclip_frac=0.3, max_gap=4)
    vids = []
    for k, (f, hh) in enumerate(clips):
        vid = 100000 + k
        media.append(MediaRow(vid, TYPE_VIDEO, f"vid/{vid}.mp4", 640, 480, f"{vid:032x}", 0))
        vids.append((vid, VideoIndex(f.tolist(), [int(x) for x in hh])))
    d = write_index_dir(str(tmp_path), media, kp, mats, [(int(i), cols[k]) for k, i in enumerate(ids)], vids)
    return d, dict(h=h, ids=ids, kp=kp, mats=mats, cols=cols, vids=vids)


def test_qcompress_is_be_length_plus_zlib():
    from cbird_amd.indexdir import q_compress, q_uncompress

    raw = bytes(range(256)) * 9
    blob = q_compress(raw)
    assert struct.unpack(">I", blob[:4])[0] == len(raw) and zlib.decompress(blob[4:]) == raw
    assert q_uncompress(blob) == raw and q_uncompress(b"") == b""
    with pytest.raises(ValueError):
        q_uncompress(struct.pack(">I", 5) + zlib.compress(b"abc"))


def test_layout_and_roundtrip(tmp_path):
    d, w = _fixture(tmp_path)
    # files and schema are the reference's (database.h:44-55, createTables of each index)
    for k, table in ((0, "media"), (1, "kphash"), (2, "matrix"), (3, "color")):
        p = os.path.join(str(tmp_path), "_index", f"media{k}.db")
        assert os.path.exists(p)
        con = sqlite3.connect(p)
        assert con.execute("select count(0) from " + table).fetchone()[0] > 0
        con.close()
    assert sorted(os.listdir(d.video_path())) == sorted(f"{i}.vdx" for i, _ in w["vids"])
    h, ids = d.dct_columns()
    assert (h == w["h"]).all() and (ids == w["ids"]).all()  # incl. the hash with bit 63 set
    kp = d.kphash_rows()
    assert [i for i, _ in kp] == [i for i, _ in w["kp"]]
    assert all((a == b).all() for (_, a), (_, b) in zip(kp, w["kp"]))
    ms = d.matrix_rows()
    want = [(i, m) for i, m in w["mats"] if len(m)]  # empty descriptor sets are skipped (:207-209)
    assert [i for i, _ in ms] == [i for i, _ in want] and all((a == b).all() for (_, a), (_, b) in zip(ms, want))
    cid, cd = d.color_rows()
    assert (cid == w["ids"]).all() and (cd == w["cols"]).all()
    assert d.video_ids() == [i for i, _ in w["vids"]]
    assert len(d.media()) == len(w["ids"]) + len(w["vids"])


def test_invalid_rows_are_ignored_like_the_reference(tmp_path):
    from cbird_amd.indexdir import IndexDir, q_compress

    d, w = _fixture(tmp_path, n_img=20, per=5, n_vid=1)
    con = sqlite3.connect(d.db_path(1))
    con.execute("insert into kphash (media_id, hashes) values (999, ?)", (b"\x01\x02\x03",))  # not a multiple of 8
    con.commit(); con.close()
    assert 999 not in [i for i, _ in IndexDir(str(tmp_path)).kphash_rows()]
    con = sqlite3.connect(d.db_path(2))
    con.execute("insert into matrix (media_id,rows,cols,type,stride,data) values (5000,2,32,0,31,?)",
                (q_compress(bytes(64)),))  # stride != cols * elemSize
    con.commit(); con.close()
    assert 5000 not in [i for i, _ in IndexDir(str(tmp_path)).matrix_rows()]
    con = sqlite3.connect(d.db_path(3))
    con.execute("insert into color (media_id, color_desc) values (7777, ?)", (b"short",))
    con.commit(); con.close()
    cid, cd = IndexDir(str(tmp_path)).color_rows()
    assert cid[-1] == 7777 and cd[-1]["numColors"] == 0  # cleared descriptor (colordescindex.cpp:145)


@pytest.mark.gpu
def test_gpu_indexes_loaded_from_directory_answer_like_direct_ones(gpu, tmp_path):
    from cbird_amd.colordesc import ColorDescIndex
    from cbird_amd.cvfeatures import CvFeaturesIndex
    from cbird_amd.indexdir import _M
    from cbird_amd.video import DctVideoIndex, VideoSearchParams

    d, w = _fixture(tmp_path)
    p = gpu.SearchParams()
    # DctHashIndex
    a, b = gpu.DctHashIndex(), gpu.DctHashIndex()
    d.load_dct(a)
    b.load(w["h"], w["ids"])
    q = w["h"][::7]
    for x, y in zip(a.find_batch(q, 6, 8), b.find_batch(q, 6, 8)):
        assert (x == y).all()
    # DctFeaturesIndex
    fa, fb = gpu.DctFeaturesIndex(), gpu.DctFeaturesIndex()
    d.load_dct_features(fa)
    fb.load(w["kp"])
    assert fa.count() == fb.count() == sum(len(h) for _, h in w["kp"])
    for i, hs in w["kp"][:40]:
        if len(hs):
            m = _M(id=i, keyPointHashes=[int(x) for x in hs])
            assert [(r.mediaId, r.score) for r in fa.find(m, p)] == [(r.mediaId, r.score) for r in fb.find(m, p)]
    # CvFeaturesIndex
    ca, cb = CvFeaturesIndex(), CvFeaturesIndex()
    d.load_cv_features(ca)
    cb.add([_M(id=i, keyPointDescriptors=m) for i, m in w["mats"] if len(m)])
    assert ca.count() == cb.count()
    for i, m in [x for x in w["mats"] if len(x[1])][:10]:
        n = _M(id=i, keyPointDescriptors=m)
        assert [(r.mediaId, r.score) for r in ca.find(n, p)] == [(r.mediaId, r.score) for r in cb.find(n, p)]
    # ColorDescIndex
    xa, xb = ColorDescIndex(), ColorDescIndex()
    d.load_color(xa)
    xb.add([_M(id=int(i), colorDescriptor=w["cols"][k]) for k, i in enumerate(w["ids"])])
    ga, gb = xa.find_batch(w["cols"][:5], 6), xb.find_batch(w["cols"][:5], 6)
    assert all((x == y).all() for x, y in zip(ga, gb))
    # DctVideoIndex
    va, vb = DctVideoIndex(), DctVideoIndex()
    d.load_video(va)
    vb.add([_M(id=i, videoIndex=v) for i, v in w["vids"]])
    vp = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=5, minFramesNear=30)
    for i, v in w["vids"]:
        n = _M(id=i, videoIndex=v)
        assert [(r.mediaId, r.score) for r in va.findVideo(n, vp)] == [(r.mediaId, r.score) for r in vb.findVideo(n, vp)]


# ---- the rebuildable caches (_index/cache/) -------------------------------------------------------------------

def _gen_htree():
    import importlib.util

    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("gen_htree", os.path.join(here, "golden", "gen_golden_htree_cache.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_hamming_tree_cache_written_by_the_real_tree(tmp_path):
    """tests/golden/htree_cache.npz holds a dctfeatures.cache file written by the reference's own HammingTree::write
    (two leaves under a split root, two media removed): the reader returns exactly the values that went in"""
    from conftest import load_golden

    from cbird_amd.indexdir import read_hamming_tree

    gen = _gen_htree()
    ids, h = gen.make_inputs()
    want_ids = ids.copy()
    want_ids[np.isin(ids, gen.REMOVED)] = 0
    path = tmp_path / "dctfeatures.cache"
    load_golden("htree_cache.npz")["cache_bytes"].tofile(path)
    got_ids, got_h = read_hamming_tree(str(path))
    assert len(got_ids) == len(ids) == 8300
    order_g = np.lexsort((got_ids, got_h))
    order_w = np.lexsort((want_ids, h))
    assert (got_h[order_g] == h[order_w]).all() and (got_ids[order_g] == want_ids[order_w]).all()
    # leaf order: the root splits on bit 0, the left leaf holds the hashes with bit 0 clear
    half = int((h & np.uint64(1) == 0).sum())
    assert ((got_h[:half] & np.uint64(1)) == 0).all() and ((got_h[half:] & np.uint64(1)) == 1).all()
    # damaged files are refused
    bad = tmp_path / "bad.cache"
    bad.write_bytes(b"cbird hamming tree:1:4:8:65536\n")
    with pytest.raises(ValueError):
        read_hamming_tree(str(bad))
    raw = path.read_bytes()
    (tmp_path / "trunc.cache").write_bytes(raw + b"\x00")
    with pytest.raises(ValueError):
        read_hamming_tree(str(tmp_path / "trunc.cache"))
    empty = tmp_path / "empty.cache"
    empty.write_bytes(b"cbird hamming tree:2:4:8:65536\n")
    assert len(read_hamming_tree(str(empty))[0]) == 0


def test_reader_agrees_with_the_real_tree_reading_its_own_file(tmp_path):
    """when the compiled-in-place reference is available: a fresh real tree reads the golden file back and answers
    searches like the tree that wrote it (the file is what cbird would load)"""
    from oracle import RefHammingTree, ref_qt_available

    if not ref_qt_available():
        pytest.skip("needs oracle/_ref/libcbird_ref_qt.so (built where /root/reference exists)")
    from conftest import load_golden

    gen = _gen_htree()
    ids, h = gen.make_inputs()
    path = tmp_path / "dctfeatures.cache"
    load_golden("htree_cache.npz")["cache_bytes"].tofile(path)
    t = RefHammingTree()
    assert t.read(str(path))  # (HammingTree::read does not restore size(): _count stays 0 in the reference)
    t2 = tmp_path / "again.cache"
    t.write(str(t2))
    assert t2.read_bytes() == path.read_bytes()  # read -> write round trip through the real code is the identity


def test_cvfeatures_cache_round_trip(tmp_path):
    """cvfeatures.mat / _idmap.map / _indexmap.map / .touch (src/cvfeaturesindex.cpp:387-419, src/cvutil.cpp:129-163,
    src/ioutil.h:203-231): header layout, map layout, removed media (indexmap value 0) keep their rows"""
    import struct

    from cbird_amd.indexdir import (cache_is_stale, read_cv_matrix, read_cvfeatures_cache, read_u32_map,
                                    write_cvfeatures_cache, write_u32_map)

    rng = np.random.default_rng(3)
    media = [(5, rng.integers(0, 256, (7, 32), dtype=np.uint8)), (9, rng.integers(0, 256, (3, 32), dtype=np.uint8)),
             (12, np.zeros((0, 32), np.uint8)), (20, rng.integers(0, 256, (11, 32), dtype=np.uint8))]
    cache = tmp_path / "cache"
    write_cvfeatures_cache(str(cache), media)
    raw = (cache / "cvfeatures.mat").read_bytes()
    assert struct.unpack("<Iiiii", raw[:20]) == (0, 21, 32, 0, 32) and len(raw) == 20 + 21 * 32
    assert read_cv_matrix(str(cache / "cvfeatures.mat")).shape == (21, 32)
    assert read_u32_map(str(cache / "cvfeatures_idmap.map")) == {5: 0, 9: 7, 20: 10}
    assert (cache / "cvfeatures_idmap.map").read_bytes() == struct.pack("<6I", 5, 0, 9, 7, 20, 10)
    got = read_cvfeatures_cache(str(cache))
    assert [g[0] for g in got] == [5, 9, 20] and all((g[1] == m[1]).all() for g, m in zip(got, [media[0], media[1], media[3]]))
    # the reference's remove() zeroes the indexmap value; its save() also stores the trailing sentinels
    im = read_u32_map(str(cache / "cvfeatures_indexmap.map"))
    im[7] = 0
    im[21] = 0  # _indexMap[rows] = 0 sentinel
    write_u32_map(str(cache / "cvfeatures_indexmap.map"), im)
    got = read_cvfeatures_cache(str(cache))
    assert [g[0] for g in got] == [5, 0, 20] and len(got[1][1]) == 3
    # staleness: missing cache, database newer than cache
    db = tmp_path / "media2.db"
    db.write_bytes(b"x")
    assert cache_is_stale(str(db), str(tmp_path / "nope"))
    os.utime(db, (1, 1))
    assert not cache_is_stale(str(db), str(cache / "cvfeatures.touch"))
    os.utime(db, None)
    os.utime(cache / "cvfeatures.touch", (1, 1))
    assert cache_is_stale(str(db), str(cache / "cvfeatures.touch"))


@pytest.mark.gpu
def test_gpu_indexes_loaded_from_the_cache_files(gpu, tmp_path):
    """_index/cache/: the dctfeatures.cache the REAL HammingTree wrote and a cvfeatures cache (incl. a removed media)
    load into the GPU indexes and answer like indexes fed the same values directly"""
    from conftest import load_golden

    from cbird_amd.cvfeatures import CvFeaturesIndex
    from cbird_amd.indexdir import IndexDir, _M, write_cvfeatures_cache

    gen = _gen_htree()
    ids, h = gen.make_inputs()
    d = IndexDir(str(tmp_path))
    os.makedirs(d.cache_path(), exist_ok=True)
    for k in (1, 2):  # empty databases older than the caches: the caches are current
        sqlite3.connect(d.db_path(k)).close()
        os.utime(d.db_path(k), (1, 1))
    load_golden("htree_cache.npz")["cache_bytes"].tofile(os.path.join(d.cache_path(), "dctfeatures.cache"))
    fa, fb = gpu.DctFeaturesIndex(), gpu.DctFeaturesIndex()
    d.load_dct_features(fa)
    rows = [(int(i), h[ids == i]) for i in np.unique(ids)]
    fb.load(rows)
    fb.remove(gen.REMOVED.tolist())
    assert fa.count() == fb.count() == len(h)
    p = gpu.SearchParams()
    for i in (1, 7, 8, 30, 60, 83):
        m = _M(id=i, keyPointHashes=[int(x) for x in h[ids == i]])
        assert [(r.mediaId, r.score) for r in fa.find(m, p)] == [(r.mediaId, r.score) for r in fb.find(m, p)], i
    # cvfeatures
    rng = np.random.default_rng(8)
    media = [(i, rng.integers(0, 256, (int(rng.integers(1, 40)), 32), dtype=np.uint8)) for i in range(1, 60)]
    write_cvfeatures_cache(d.cache_path(), media)
    from cbird_amd.indexdir import read_u32_map, write_u32_map

    im_path = os.path.join(d.cache_path(), "cvfeatures_indexmap.map")
    im = read_u32_map(im_path)
    start_of_17 = [k for k, v in im.items() if v == 17][0]
    im[start_of_17] = 0  # media 17 removed, the way CvFeaturesIndex::remove leaves the map
    write_u32_map(im_path, im)
    ca, cb = CvFeaturesIndex(), CvFeaturesIndex()
    d.load_cv_features(ca)
    cb.add([_M(id=i, keyPointDescriptors=m) for i, m in media])
    cb.remove([17])
    assert ca.count() == cb.count()
    for i, m in media[:12] + [media[16]]:
        n = _M(id=i, keyPointDescriptors=m)
        assert [(r.mediaId, r.score) for r in ca.find(n, p)] == [(r.mediaId, r.score) for r in cb.find(n, p)], i


@pytest.mark.gpu
def test_slice_of_every_index(gpu, tmp_path):
    """Index::slice of the four non-DctHash indexes (the Python mirrors of the adapters' overrides): the slice answers
    like an index that was only ever given the chosen media"""
    from cbird_amd.colordesc import ColorDescIndex
    from cbird_amd.cvfeatures import CvFeaturesIndex
    from cbird_amd.indexdir import _M
    from cbird_amd.video import DctVideoIndex, VideoSearchParams

    d, w = _fixture(tmp_path)
    p = gpu.SearchParams()
    keep = [i for i, _ in w["kp"]][::3]
    # DctFeaturesIndex
    full, ref = gpu.DctFeaturesIndex(), gpu.DctFeaturesIndex()
    full.load(w["kp"])
    ref.load([(i, h) for i, h in w["kp"] if i in set(keep)])
    sub = full.slice(keep)
    assert sub.count() == ref.count()
    for i, hs in w["kp"][:30]:
        if len(hs):
            m = _M(id=i, keyPointHashes=[int(x) for x in hs])
            assert [(r.mediaId, r.score) for r in sub.find(m, p)] == [(r.mediaId, r.score) for r in ref.find(m, p)]
    # CvFeaturesIndex
    mats = [x for x in w["mats"] if len(x[1])]
    keep_c = [i for i, _ in mats][::2]
    cf, cr = CvFeaturesIndex(), CvFeaturesIndex()
    cf.add([_M(id=i, keyPointDescriptors=m) for i, m in mats])
    cr.add([_M(id=i, keyPointDescriptors=m) for i, m in mats if i in set(keep_c)])
    cs = cf.slice(keep_c)
    assert cs.count() == cr.count()
    for i, m in mats[:8]:
        n = _M(id=i, keyPointDescriptors=m)
        assert [(r.mediaId, r.score) for r in cs.find(n, p)] == [(r.mediaId, r.score) for r in cr.find(n, p)]
    # ColorDescIndex
    xf = ColorDescIndex()
    xf.add([_M(id=int(i), colorDescriptor=w["cols"][k]) for k, i in enumerate(w["ids"])])
    keep_x = [int(i) for i in w["ids"][::4]]
    xs = xf.slice(keep_x)
    xr = ColorDescIndex()
    xr.add([_M(id=int(i), colorDescriptor=w["cols"][k]) for k, i in enumerate(w["ids"]) if int(i) in set(keep_x)])
    assert xs.count() == xr.count() == len(keep_x)
    assert all((a == b).all() for a, b in zip(xs.find_batch(w["cols"][:5], 6), xr.find_batch(w["cols"][:5], 6)))
    # DctVideoIndex (re-reads the .vdx files of the subset)
    vf = DctVideoIndex()
    d.load_video(vf)
    keep_v = [i for i, _ in w["vids"]][::2]
    vs = vf.slice(keep_v)
    vr = DctVideoIndex()
    vr.add([_M(id=i, videoIndex=v) for i, v in w["vids"] if i in set(keep_v)])
    vp = VideoSearchParams(dctThresh=5, skipFrames=0, minFramesMatched=5, minFramesNear=30)
    assert vs.count() == vr.count() == len(keep_v)
    for i, v in w["vids"]:
        n = _M(id=i, videoIndex=v)
        assert [(r.mediaId, r.score) for r in vs.find(n, vp)] == [(r.mediaId, r.score) for r in vr.find(n, vp)]
    with pytest.raises(ValueError):
        vr.slice(keep_v)  # built from memory: no data path to re-read

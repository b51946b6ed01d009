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

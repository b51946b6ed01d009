"""A cbird `_index/` directory, read (and, for tests, written) without cbird.

SURVEY.md section 8(f) rank 3: "lets the engine load a real cbird _index/ directory directly".  The
layout restated here is the reference's own, file by file:

    <root>/_index/media0.db     table `media` (id, type, path, width, height, md5, phash_dct)
                                -- Database::createTables, src/database.cpp:235-249; DctHashIndex reads
                                `select id,phash_dct from media where type=1` (src/dcthashindex.cpp:89) and
                                DctVideoIndex `select id from media where type=2 order by id`
                                (src/dctvideoindex.cpp:185); both use database 0
    <root>/_index/media1.db     table `kphash` (media_id, hashes = raw little-endian u64[]) --
                                DctFeaturesIndex, src/dctfeaturesindex.cpp:41-76,139-156
    <root>/_index/media2.db     table `matrix` (media_id, rows, cols, type, stride, data = qCompress(row bytes))
                                -- CvFeaturesIndex, src/cvfeaturesindex.cpp:50-100,189-232; rows must arrive
                                in ascending unique media_id, empty ones are skipped
    <root>/_index/media3.db     table `color` (media_id, color_desc = the 258-byte ColorDescriptor) --
                                ColorDescIndex, src/colordescindex.cpp:39-75,125-150
    <root>/_index/video/<id>.vdx  one .vdx v2 file per video (src/database.cpp:457, src/videoindex.cpp:271-429)

(database file N belongs to the index whose Index::databaseId() is N = its SearchParams algo id,
src/index.h:192, src/database.h:47-49.)  `_index/cache/` holds rebuildable caches (HammingTree dump, FLANN
matrix) that "can be deleted without affecting the index" (src/database.h:51-52): they are ignored.

qCompress (Qt) = 4-byte big-endian uncompressed length followed by a zlib stream.

Host-side plumbing only: nothing here computes; the loaded columns go into the GPU indexes through
their normal load()/add() entry points.
"""
from __future__ import annotations

import os
import sqlite3
import struct
import zlib
from dataclasses import dataclass, field

import numpy as np

from .colordesc import COLOR_DTYPE

INDEX_DIRNAME = "_index"  # src/global.h:35
TYPE_IMAGE, TYPE_VIDEO = 1, 2  # Media::TypeImage / TypeVideo
CV_8U = 0  # cv::Mat type of ORB descriptors (CV_8UC1)


def q_compress(data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + zlib.compress(data)


def q_uncompress(blob: bytes) -> bytes:
    if len(blob) < 4:
        return b""
    n = struct.unpack(">I", blob[:4])[0]
    out = zlib.decompress(blob[4:])
    if len(out) != n:
        raise ValueError("qUncompress: length header does not match the stream")
    return out


def _u64(v: int) -> int:
    """SQLite integers are signed 64-bit; phash_dct is the uint64 bit pattern."""
    return v & 0xFFFFFFFFFFFFFFFF


def _i64(v: int) -> int:
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


@dataclass
class MediaRow:
    id: int
    type: int
    path: str
    width: int = 0
    height: int = 0
    md5: str = ""
    phash_dct: int = 0


@dataclass
class _M:  # the attributes the index classes read from a Media
    id: int
    path: str = ""
    dctHash: int = 0
    keyPointHashes: list = field(default_factory=list)
    keyPointDescriptors: object = None
    colorDescriptor: object = None
    videoIndex: object = None


class IndexDir:
    def __init__(self, root: str) -> None:
        self.root = root
        self.index_path = os.path.join(root, INDEX_DIRNAME)

    # ---- paths (src/database.h:44-55) ---------------------------------------------------------
    def db_path(self, db_id: int = 0) -> str:
        return os.path.join(self.index_path, f"media{db_id}.db")

    def video_path(self) -> str:
        return os.path.join(self.index_path, "video")

    def _connect(self, db_id: int, must_exist: bool = True):
        p = self.db_path(db_id)
        if must_exist and not os.path.exists(p):
            return None
        return sqlite3.connect(p)

    # ---- readers -----------------------------------------------------------------------------
    def media(self):
        """all rows of the media table"""
        con = self._connect(0)
        if con is None:
            return []
        with con:
            rows = con.execute("select id,type,path,width,height,md5,phash_dct from media").fetchall()
        con.close()
        return [MediaRow(r[0], r[1], r[2], r[3], r[4], r[5], _u64(r[6])) for r in rows]

    def dct_columns(self):
        """(hashes u64[], ids u32[]) exactly as DctHashIndex::load fills its arrays (:89-105)"""
        con = self._connect(0)
        if con is None:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
        rows = con.execute("select id,phash_dct from media where type=1").fetchall()
        con.close()
        ids = np.fromiter((r[0] for r in rows), np.uint32, len(rows))
        h = np.fromiter((_u64(r[1]) for r in rows), np.uint64, len(rows))
        return h, ids

    def kphash_rows(self):
        """[(media_id, u64[])] in table order; blobs whose size is not a multiple of 8 are ignored
        (dctfeaturesindex.cpp:145-148)"""
        con = self._connect(1)
        if con is None:
            return []
        out = []
        for media_id, blob in con.execute("select media_id,hashes from kphash"):
            blob = bytes(blob)
            if len(blob) % 8:
                continue
            out.append((int(media_id), np.frombuffer(blob, "<u8").copy()))
        con.close()
        return out

    def matrix_rows(self):
        """[(media_id, u8[rows, cols])] ascending media_id; empty or inconsistent rows skipped
        (cvfeaturesindex.cpp:189-219)"""
        con = self._connect(2)
        if con is None:
            return []
        out, last = [], 0
        q = "select media_id,rows,cols,type,stride,data from matrix order by media_id"
        for media_id, rows, cols, typ, stride, data in con.execute(q):
            if rows <= 0:
                continue
            raw = q_uncompress(bytes(data))
            if last >= media_id or typ != CV_8U or stride != cols or len(raw) != rows * stride:
                continue  # "sql: ignoring invalid data"
            out.append((int(media_id), np.frombuffer(raw, np.uint8).reshape(rows, cols).copy()))
            last = media_id
        con.close()
        return out

    def color_rows(self):
        """(ids u32[], descriptors COLOR_DTYPE[]); a blob of the wrong size becomes an empty descriptor
        (colordescindex.cpp:141-147)"""
        con = self._connect(3)
        if con is None:
            return np.zeros(0, np.uint32), np.zeros(0, COLOR_DTYPE)
        rows = con.execute("select media_id,color_desc from color").fetchall()
        con.close()
        ids = np.fromiter((r[0] for r in rows), np.uint32, len(rows))
        d = np.zeros(len(rows), COLOR_DTYPE)
        for i, r in enumerate(rows):
            b = bytes(r[1])
            if len(b) == COLOR_DTYPE.itemsize:
                d[i] = np.frombuffer(b, COLOR_DTYPE)[0]
        return ids, d

    def video_ids(self):
        con = self._connect(0)
        if con is None:
            return []
        ids = [r[0] for r in con.execute("select id from media where type=2 order by id")]
        con.close()
        return ids

    # ---- load into the GPU indexes --------------------------------------------------------------
    def load_dct(self, index) -> None:
        h, ids = self.dct_columns()
        index.load(h, ids)

    def load_dct_features(self, index) -> None:
        index.load(self.kphash_rows())

    def load_cv_features(self, index) -> None:
        index.add([_M(id=i, keyPointDescriptors=d) for i, d in self.matrix_rows()])

    def load_color(self, index) -> None:
        ids, d = self.color_rows()
        index.add([_M(id=int(i), colorDescriptor=d[k]) for k, i in enumerate(ids)])

    def load_video(self, index) -> None:
        """DctVideoIndex::load (dctvideoindex.cpp:172-211): every type-2 media id, frames from <id>.vdx"""
        index.load(self.video_ids(), self.video_path())


# ---- writer (test fixture generator; follows createTables/addRecords of each index) ---------------------
def write_index_dir(root: str, media=(), kphash=(), matrices=(), colors=(), videos=()) -> IndexDir:
    """media: iterable of MediaRow; kphash: (media_id, u64 array); matrices: (media_id, u8[rows, 32]);
    colors: (media_id, COLOR_DTYPE scalar); videos: (media_id, VideoIndex)"""
    d = IndexDir(root)
    os.makedirs(d.video_path(), exist_ok=True)
    con = sqlite3.connect(d.db_path(0))
    con.execute("create table media (id integer primary key not null, type integer not null, path text not null,"
                " width integer not null, height integer not null, md5 text not null, phash_dct integer not null);")
    con.execute("create unique index media_id_index on media(id);")
    con.execute("create unique index media_path_index on media(path);")
    con.execute("create index media_md5_index on media(md5);")
    con.executemany("insert into media (id,type,path,width,height,md5,phash_dct) values (?,?,?,?,?,?,?)",
                    [(m.id, m.type, m.path, m.width, m.height, m.md5, _i64(m.phash_dct)) for m in media])
    con.commit()
    con.close()

    con = sqlite3.connect(d.db_path(1))
    con.execute("create table kphash (media_id integer not null, hashes blob not null);")
    con.execute("create index kphash_media_id_index on kphash(media_id);")
    con.executemany("insert into kphash (media_id, hashes) values (?,?)",
                    [(int(i), np.asarray(h, "<u8").tobytes()) for i, h in kphash])
    con.commit()
    con.close()

    con = sqlite3.connect(d.db_path(2))
    con.execute("create table matrix (id integer primary key not null, media_id integer not null, rows integer not"
                " null, cols integer not null, type integer not null, stride integer not null, data blob not null);")
    con.execute("create index matrix_media_id_index on matrix(media_id);")
    rows = []
    for i, m in matrices:
        m = np.ascontiguousarray(m, np.uint8)
        r, c = (m.shape if m.ndim == 2 else (0, 0))
        rows.append((int(i), r, c, CV_8U, c, q_compress(m.tobytes()) if r > 0 else b""))
    con.executemany("insert into matrix (media_id,rows,cols,type,stride,data) values (?,?,?,?,?,?)", rows)
    con.commit()
    con.close()

    con = sqlite3.connect(d.db_path(3))
    con.execute("create table color (media_id integer not null, color_desc blob not null);")
    con.execute("create unique index color_media_id_index on color(media_id);")
    con.executemany("insert into color (media_id, color_desc) values (?,?)",
                    [(int(i), np.asarray(c, COLOR_DTYPE).tobytes()) for i, c in colors])
    con.commit()
    con.close()

    for i, v in videos:
        v.save(os.path.join(d.video_path(), f"{int(i)}.vdx"))
    return d

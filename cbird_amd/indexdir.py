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
matrix) that "can be deleted without affecting the index" (src/database.h:51-52): when they are current they are
read instead of the SQL tables (see the cache section below), never required.

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
            try:
                raw = q_uncompress(bytes(data))
            except (zlib.error, ValueError):
                continue  # the reference treats a blob that does not uncompress as invalid data and skips the row
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

    def cache_path(self) -> str:
        return os.path.join(self.index_path, "cache")  # src/database.h:51-52

    def load_dct_features(self, index, use_cache: bool = True) -> None:
        """the kphash table, or cbird's dctfeatures.cache when it is present and not older than media1.db"""
        cache = os.path.join(self.cache_path(), "dctfeatures.cache")
        if use_cache and not cache_is_stale(self.db_path(1), cache):
            ids, hashes = read_hamming_tree(cache)
            index.load_flat(hashes, ids)
            return
        index.load(self.kphash_rows())

    def load_cv_features(self, index, use_cache: bool = True) -> None:
        cache = os.path.join(self.cache_path(), "cvfeatures.touch")
        if use_cache and not cache_is_stale(self.db_path(2), cache):
            # removed media (id 0) keep their rows in the reference; they are added and removed again here
            media = read_cvfeatures_cache(self.cache_path())
            tmp_ids, removed, nxt = [], [], 0xFFFF0000
            for mid, d in media:
                if mid == 0:
                    mid = nxt
                    nxt += 1
                    removed.append(mid)
                tmp_ids.append((mid, d))
            index.add([_M(id=i, keyPointDescriptors=d) for i, d in tmp_ids])
            if removed:
                index.remove(removed)
            return
        index.add([_M(id=i, keyPointDescriptors=d) for i, d in self.matrix_rows()])

    def load_color(self, index) -> None:
        ids, d = self.color_rows()
        index.add([_M(id=int(i), colorDescriptor=d[k]) for k, i in enumerate(ids)])

    def load_video(self, index) -> None:
        """DctVideoIndex::load (dctvideoindex.cpp:172-211): every type-2 media id, frames from <id>.vdx"""
        index.load(self.video_ids(), self.video_path())


# ---- the rebuildable caches under _index/cache/ ---------------------------------------------------------------
# cbird rebuilds them from the SQL tables when they are missing or older than the database file
# (DBHelper::isCacheFileStale, src/qtutil.cpp:934-937).  They are read here so that a large index loads from the
# flat cache files instead of millions of SQL rows; nothing depends on them.

def cache_is_stale(db_file: str, cache_file: str) -> bool:
    """DBHelper::isCacheFileStale: missing cache, or database modified after it"""
    if not os.path.exists(cache_file):
        return True
    if not os.path.exists(db_file):
        return True
    return os.path.getmtime(db_file) > os.path.getmtime(cache_file)


HAMMING_TREE_HEADER = b"cbird hamming tree:2:4:8:65536"  # FILE_VERSION, sizeof(index_t), sizeof(hash_t), CLUSTER_SIZE


def read_hamming_tree(path: str):
    """`dctfeatures.cache` = HammingTree_t<uint32_t>::write (src/tree/hammingtree.h:156-200, 472-521): a text header
    line, then the nodes in pre-order -- bool isLeaf; internal: int32 bit, left, right; leaf: uint32 count,
    uint32 indices[count], uint64 hashes[count].  Returns (ids u32[], hashes u64[]) in leaf order; removed entries
    keep their hash with id 0 (hammingtree.h:349-361).  The tree shape itself is not needed: the GPU index scans."""
    with open(path, "rb") as f:
        data = f.read()
    nl = data.find(b"\n", 0, 128)
    if nl < 0 or data[:nl] != HAMMING_TREE_HEADER:
        raise ValueError(f"not a cbird hamming tree v2 file: {data[:40]!r}")
    pos = nl + 1
    ids, hashes = [], []
    stack = 1  # nodes still to read
    while stack:
        if pos >= len(data):  # an empty tree writes nothing after the header
            break
        is_leaf = data[pos] != 0
        pos += 1
        stack -= 1
        if not is_leaf:
            pos += 4  # the split bit
            stack += 2
            continue
        if pos >= len(data):  # readNode: "if (f.atEnd()) ;" -- a leaf flag at the very end carries no count
            break
        (count,) = struct.unpack_from("<I", data, pos)
        pos += 4
        if count:
            ids.append(np.frombuffer(data, "<u4", count, pos))
            pos += 4 * count
            hashes.append(np.frombuffer(data, "<u8", count, pos))
            pos += 8 * count
    if pos != len(data):
        raise ValueError("trailing bytes after the last tree node")
    if not ids:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint64)
    return np.concatenate(ids).astype(np.uint32), np.concatenate(hashes).astype(np.uint64)


def read_cv_matrix(path: str) -> np.ndarray:
    """`cvfeatures.mat` = saveMatrix (src/cvutil.cpp:129-163): MatrixHeader {u32 id; i32 rows, cols, type, stride}
    followed by rows x stride bytes"""
    with open(path, "rb") as f:
        hdr = f.read(20)
        if len(hdr) != 20:
            raise ValueError("short matrix header")
        _id, rows, cols, typ, stride = struct.unpack("<Iiiii", hdr)
        if typ != CV_8U or stride != cols or rows < 0:
            raise ValueError(f"unexpected matrix header rows={rows} cols={cols} type={typ} stride={stride}")
        raw = f.read(rows * stride)
    if len(raw) != rows * stride:
        raise ValueError("short matrix data")
    return np.frombuffer(raw, np.uint8).reshape(rows, cols).copy()


def write_cv_matrix(path: str, m: np.ndarray) -> None:
    m = np.ascontiguousarray(m, np.uint8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Iiiii", 0, m.shape[0], m.shape[1], CV_8U, m.shape[1]))
        f.write(m.tobytes())


def read_u32_map(path: str) -> dict:
    """saveMap<uint32_t, uint32_t> (src/ioutil.h:203-231): key/value pairs back to back"""
    a = np.fromfile(path, "<u4")
    if a.size % 2:
        raise ValueError("odd number of words in a uint32 map file")
    return dict(zip(a[0::2].tolist(), a[1::2].tolist()))


def write_u32_map(path: str, m: dict) -> None:
    a = np.array([x for kv in sorted(m.items()) for x in kv], "<u4")  # std::map iterates in key order
    a.tofile(path)


def read_cvfeatures_cache(cache_path: str):
    """CvFeaturesIndex::loadIndex (src/cvfeaturesindex.cpp:387-391): cvfeatures.mat + _idmap (mediaId -> first row) +
    _indexmap (first row -> mediaId, 0 after a removal).  Returns [(media_id, u8[rows, 32])] in row order, removed
    media with id 0 (they keep their rows, like in the reference)."""
    rows = read_cv_matrix(os.path.join(cache_path, "cvfeatures.mat"))
    index_map = read_u32_map(os.path.join(cache_path, "cvfeatures_indexmap.map"))
    starts = sorted(k for k in index_map if k < len(rows))
    out = []
    for i, s0 in enumerate(starts):
        s1 = starts[i + 1] if i + 1 < len(starts) else len(rows)
        out.append((int(index_map[s0]), rows[s0:s1]))
    return out


def write_cvfeatures_cache(cache_path: str, media) -> None:
    """media: [(media_id, u8[rows, 32])] ascending id.  Writes what CvFeaturesIndex::saveIndex writes (:406-419),
    maps without the trailing sentinels (as after load(); loadIndex re-adds them)."""
    os.makedirs(cache_path, exist_ok=True)
    id_map, index_map, parts, pos = {}, {}, [], 0
    for mid, d in media:
        d = np.ascontiguousarray(d, np.uint8).reshape(-1, 32)
        if len(d) == 0:
            continue
        id_map[int(mid)] = pos
        index_map[pos] = int(mid)
        parts.append(d)
        pos += len(d)
    write_cv_matrix(os.path.join(cache_path, "cvfeatures.mat"),
                    np.concatenate(parts) if parts else np.zeros((0, 32), np.uint8))
    write_u32_map(os.path.join(cache_path, "cvfeatures_idmap.map"), id_map)
    write_u32_map(os.path.join(cache_path, "cvfeatures_indexmap.map"), index_map)
    with open(os.path.join(cache_path, "cvfeatures.touch"), "wb") as f:
        f.write(b"this file indicates index was saved successfully")


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

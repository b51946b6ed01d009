"""Host-side mirror of DctVideoIndex / VideoIndex (src/dctvideoindex.{h,cpp}, src/videoindex.{h,cpp}).

  VideoIndex       frames + hashes of one video and the .vdx v2 file format (save/load/isValid)
  make_video_index the frame de-dup of Media::makeVideoIndex over a sequence of frame hashes
  VideoIndexer     Media::makeVideoIndex for frames pushed in chunks: autocrop + dctHash64 on the device, the
                   near-frame filter's state kept between pushes (host arrays or device tensors)
  DctVideoIndex    add/remove/count/find (findFrame for image needles, findVideo for video needles)

Compute and format code live in libcbird_hip.so (include/cbird_hip.h); nothing here falls back to CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import CbhError, cbh_vmatch, check
from .index import Match, MatchRange, SearchParams

CBIRD_VERSION = "0.8.1"


@dataclass
class VideoIndex:
    """src/videoindex.h:40-67"""
    frames: list = field(default_factory=list)
    hashes: list = field(default_factory=list)

    def isEmpty(self) -> bool:
        return len(self.frames) == 0 or len(self.hashes) == 0

    def to_bytes(self) -> bytes:
        L = _lib.lib()
        f = np.ascontiguousarray(self.frames, np.int32)
        h = np.ascontiguousarray(self.hashes, np.uint64)
        n = L.cbh_vdx_encode(f.ctypes.data, h.ctypes.data, len(f), CBIRD_VERSION.encode(), None, 0)
        if n == 0:
            raise ValueError("invalid video index (first frame must be 0, frames strictly increasing)")
        buf = np.zeros(n, np.uint8)
        L.cbh_vdx_encode(f.ctypes.data, h.ctypes.data, len(f), CBIRD_VERSION.encode(), buf.ctypes.data, n)
        return buf.tobytes()

    def save(self, path: str) -> None:
        with open(path, "wb") as fp:
            fp.write(self.to_bytes())

    @staticmethod
    def from_bytes(data: bytes) -> "VideoIndex":
        L = _lib.lib()
        buf = np.frombuffer(data, np.uint8)
        cap = max(1, len(buf))
        f = np.zeros(cap, np.int32)
        h = np.zeros(cap, np.uint64)
        n = L.cbh_vdx_decode(buf.ctypes.data, len(buf), f.ctypes.data, h.ctypes.data, cap)
        if n < 0:
            raise ValueError(f"invalid .vdx data ({n})")
        return VideoIndex(f[:n].tolist(), [int(x) for x in h[:n]])

    @staticmethod
    def load(path: str) -> "VideoIndex":
        with open(path, "rb") as fp:
            return VideoIndex.from_bytes(fp.read())

    @staticmethod
    def isValid(path: str) -> bool:
        """VideoIndex::isValid (src/videoindex.cpp:90-104) -> verify_v2: header fields + the "cbir" trailer"""
        try:
            with open(path, "rb") as fp:
                data = fp.read()
        except OSError:
            return False
        buf = np.frombuffer(data, np.uint8)
        return bool(_lib.lib().cbh_vdx_verify(buf.ctypes.data, len(buf))) if len(buf) else False


def make_video_index(frame_hashes, threshold: int = 8) -> VideoIndex:
    """Media::makeVideoIndex (src/media.cpp:925-1037) given the per-frame dct hashes: keeps the frames the
    reference would store (videoThreshold default 8, src/scanner.h)."""
    L = _lib.lib()
    h = np.ascontiguousarray(frame_hashes, np.uint64)
    keep = np.zeros(max(1, len(h)), np.uint8)
    L.cbh_video_dedup(h.ctypes.data, len(h), int(threshold), keep.ctypes.data)
    ix = np.nonzero(keep[: len(h)])[0]
    return VideoIndex(ix.tolist(), [int(x) for x in h[ix]])


class VideoIndexer:
    """Media::makeVideoIndex (src/media.cpp:925-1037) fed by a decoder that produces frames in chunks.

        ix = VideoIndexer(threshold=8)              # IndexParams::videoThreshold; autocrop(img, 20) as at :961
        for chunk in decoder:                        # (n, h, w) uint8 numpy array or cuda tensor
            ix.push(chunk)
        video_index = ix.finish()

    resume = a VideoIndex written earlier (:929-936): frame numbering continues behind its last frame."""

    def __init__(self, threshold: int = 8, autocrop_range: int = 20, device: int = 0,
                 resume: VideoIndex | None = None) -> None:
        self._L = _lib.lib()
        self._h = self._L.cbh_vindexer_create(int(device), int(threshold), int(autocrop_range))
        if not self._h:
            raise CbhError(_lib.CBH_E_NODEVICE, "cbh_vindexer_create")
        self._device = int(device)
        if resume is not None and not resume.isEmpty():
            f = np.ascontiguousarray(resume.frames, np.int32)
            h = np.ascontiguousarray(resume.hashes, np.uint64)
            check(self._L.cbh_vindexer_resume(self._h, f.ctypes.data, h.ctypes.data, len(f)), "vindexer_resume")

    def __del__(self) -> None:
        if getattr(self, "_h", None):
            self._L.cbh_vindexer_destroy(self._h)
            self._h = None

    def push(self, frames) -> None:
        """n grey frames in decode order: (n, h, w) or (h, w); rows and frames may be strided, pixels contiguous"""
        if hasattr(frames, "data_ptr"):  # torch tensor
            t = frames if frames.dim() == 3 else frames[None]
            if t.dtype.itemsize != 1 or t.dim() != 3 or (t.shape[2] > 1 and t.stride(2) != 1):
                raise ValueError("frames: expected (n, h, w) uint8 with contiguous rows")
            n, h, w = t.shape
            if n == 0:
                return
            if t.is_cuda:
                if t.device.index != self._device:
                    raise ValueError(f"frames live on {t.device}, the indexer on device {self._device}")
                import torch
                torch.cuda.current_stream(t.device).synchronize()  # the indexer runs on its own stream
                check(self._L.cbh_vindexer_push_dev(self._h, t.data_ptr(), n, w, h, t.stride(1), t.stride(0)),
                      "vindexer_push_dev")
                return
            frames = t.numpy()
        a = np.asarray(frames)
        if a.ndim == 2:
            a = a[None]
        if a.dtype != np.uint8 or a.ndim != 3:
            raise ValueError("frames: expected (n, h, w) uint8")
        if a.shape[0] == 0:
            return
        if a.shape[2] > 1 and a.strides[2] != 1 or a.strides[1] < a.shape[2] or a.strides[0] < 0:
            a = np.ascontiguousarray(a)
        n, h, w = a.shape
        check(self._L.cbh_vindexer_push(self._h, a.ctypes.data, n, w, h, a.strides[1], a.strides[0]), "vindexer_push")

    @property
    def frames_seen(self) -> int:
        """makeVideoIndex's frameNumber: the number the next frame gets"""
        return int(self._L.cbh_vindexer_frames_seen(self._h))

    def finish(self) -> VideoIndex:
        n = int(self._L.cbh_vindexer_finish(self._h, None, None, 0))
        f = np.zeros(max(1, n), np.int32)
        h = np.zeros(max(1, n), np.uint64)
        got = int(self._L.cbh_vindexer_finish(self._h, f.ctypes.data, h.ctypes.data, n))
        if got != n:
            raise CbhError(_lib.CBH_E_INVAL, f"cbh_vindexer_finish: {got} != {n}")
        return VideoIndex(f[:n].tolist(), [int(x) for x in h[:n]])


@dataclass
class VideoSearchParams(SearchParams):
    """video fields of SearchParams (src/index.h:103-107)"""
    skipFrames: int = 300
    minFramesMatched: int = 30
    minFramesNear: int = 60
    videoRadix: int = 10  # `-p.vradix`; only honoured by DctVideoIndex(radix_compat=True), else the search is exact


class DctVideoIndex:
    """Detect similar videos with full-frame dct hashes (src/dctvideoindex.h:66-70)."""

    def __init__(self, device: int = 0, data_path: str | None = None, radix_compat: bool = False, shards=None) -> None:
        # False: exact search (the reference's vradix = 0).  True: a needle frame only sees the entries of its
        # RadixMap bucket, (hash >> 1) & (2^videoRadix - 1) (src/tree/radix.h:135-141): the reference's
        # approximate candidate sets for `-p.vradix N`.
        self.radix_compat = bool(radix_compat)
        self._device = device
        self._L = _lib.lib()
        self._id = SearchParams.AlgoVideo
        self._data_path = data_path
        shards = shards if shards is not None else _lib.default_sharding()  # (device_mask, shards_per_device)
        self._h = self._L.cbh_vidx_create_sharded(shards[0], shards[1]) if shards else self._L.cbh_vidx_create(device)
        if not self._h:
            raise CbhError(_lib.CBH_E_NODEVICE, "cbh_vidx_create")
        self._loaded = False

    def __del__(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.cbh_vidx_destroy(h)

    def id(self) -> int:
        return self._id

    def databaseId(self) -> int:
        return 0  # dctvideoindex.h:94

    def isLoaded(self) -> bool:
        return self._loaded

    def count(self) -> int:
        return int(self._L.cbh_vidx_count(self._h))

    def memoryUsage(self) -> int:
        """`_tree ? _tree->stats().memory : 0` (dctvideoindex.cpp:57-59): 0 until the first query builds the structure,
        then 8 + 6 bytes per entry"""
        return int(self._L.cbh_vidx_memory_usage(self._h))

    def _add_one(self, media_id: int, vi: VideoIndex) -> None:
        f = np.ascontiguousarray(vi.frames, np.int32)
        h = np.ascontiguousarray(vi.hashes, np.uint64)
        check(self._L.cbh_vidx_add_video(self._h, media_id, f.ctypes.data, h.ctypes.data, len(f)), "add_video")

    def load(self, media_ids, data_path: str | None = None) -> None:
        """load(): `select id from media where type=video order by id` (:172-211); each id's frames come
        from <dataPath>/<id>.vdx (insertHashes :64-72); a missing file is warned about and skipped."""
        if data_path is not None:
            self._data_path = data_path
        for mid in media_ids:
            path = os.path.join(self._data_path, f"{mid}.vdx")
            if not os.path.exists(path):
                warnings.warn(f"index file missing: {path}")
                self._add_one(mid, VideoIndex())
            else:
                self._add_one(mid, VideoIndex.load(path))
        self._loaded = True

    def add(self, media) -> None:
        """add(): media carry (id, videoIndex) -- dctvideoindex.cpp:250-254"""
        for m in media:
            self._add_one(m.id, m.videoIndex)
        self._loaded = True

    def remove(self, ids) -> None:
        i = np.ascontiguousarray(list(ids), np.uint32)
        check(self._L.cbh_vidx_remove(self._h, i.ctypes.data, len(i)), "remove")

    def slice(self, mediaIds) -> "DctVideoIndex":
        """DctVideoIndex::slice (dctvideoindex.cpp:389-397): "replicate what load() does, but use the subset" -- the
        frames come from <dataPath>/<id>.vdx again, so the index must have been loaded from a data path"""
        if self._data_path is None:
            raise ValueError("slice() re-reads the .vdx files: the index needs a data path")
        copy = DctVideoIndex(self._device, self._data_path, self.radix_compat)
        copy.load(list(mediaIds))
        return copy

    @staticmethod
    def _matches(buf, n):
        return [Match(buf[i].id, buf[i].score, MatchRange(buf[i].src_in, buf[i].dst_in, buf[i].len))
                for i in range(n)]

    def _apply_radix(self, p) -> None:
        r = int(getattr(p, "videoRadix", 0)) if self.radix_compat else 0
        check(self._L.cbh_vidx_set_radix(self._h, r), "set_radix")

    def findFrame(self, needle, p: VideoSearchParams):
        self._apply_radix(p)
        hash_ = int(needle.dctHash)
        if hash_ == 0:
            warnings.warn(f"needle has no dct hash {needle.id} {needle.path}")
            return []
        cap = max(1, self.count())
        buf = (cbh_vmatch * cap)()
        n = C.c_size_t(0)
        src_in = getattr(needle, "matchRange", MatchRange()).dstIn
        check(self._L.cbh_vidx_find_frame(self._h, hash_, int(p.dctThresh), int(p.skipFrames), int(src_in), buf,
                                          cap, C.byref(n)), "find_frame")
        return self._matches(buf, n.value)

    def findVideo(self, needle, p: VideoSearchParams):
        self._apply_radix(p)
        vi = needle.videoIndex
        if vi is None or vi.isEmpty():
            warnings.warn(f"needle video index is empty: {needle.path}")
            return []
        f = np.ascontiguousarray(vi.frames, np.int32)
        h = np.ascontiguousarray(vi.hashes, np.uint64)
        cap = max(1, self.count())
        buf = (cbh_vmatch * cap)()
        n = C.c_size_t(0)
        check(self._L.cbh_vidx_find_video(self._h, f.ctypes.data, h.ctypes.data, len(f), needle.id,
                                          int(p.dctThresh), int(p.skipFrames), int(p.minFramesMatched),
                                          int(p.minFramesNear), int(bool(p.filterSelf)), buf, cap, C.byref(n)),
              "find_video")
        return self._matches(buf, n.value)

    def find(self, needle, p: VideoSearchParams):
        """dctvideoindex.cpp:277-289: image needle -> findFrame, video needle -> findVideo"""
        if getattr(needle, "videoIndex", None) is not None:
            return self.findVideo(needle, p)
        return self.findFrame(needle, p)

    def find_videos_batch(self, needles, p: VideoSearchParams):
        self._apply_radix(p)
        needles = list(needles)
        f = np.concatenate([np.asarray(m.videoIndex.frames, np.int32) for m in needles] or [np.zeros(0, np.int32)])
        h = np.concatenate([np.asarray(m.videoIndex.hashes, np.uint64) for m in needles] or [np.zeros(0, np.uint64)])
        o = np.zeros(len(needles) + 1, np.uint64)
        np.cumsum([len(m.videoIndex.frames) for m in needles], out=o[1:])
        i = np.ascontiguousarray([m.id for m in needles], np.uint32)
        out_offs = np.zeros(len(needles) + 1, np.uint64)
        cap = max(64, 8 * len(needles))
        while True:  # the library reports the full size in out_offs[-1] when cap was too small
            buf = (cbh_vmatch * cap)()
            rc = self._L.cbh_vidx_find_videos_batch(self._h, f.ctypes.data, h.ctypes.data, o.ctypes.data,
                                                    i.ctypes.data, len(needles), int(p.dctThresh),
                                                    int(p.skipFrames), int(p.minFramesMatched),
                                                    int(p.minFramesNear), int(bool(p.filterSelf)), buf, cap,
                                                    out_offs.ctypes.data)
            if rc == _lib.CBH_E_OVERFLOW and int(out_offs[-1]) > cap:
                cap = int(out_offs[-1])
                continue
            check(rc, "find_videos_batch")
            break
        return [self._matches(buf[int(out_offs[k]):int(out_offs[k + 1])], int(out_offs[k + 1] - out_offs[k]))
                for k in range(len(needles))]

    def entries(self, skip_frames: int) -> int:
        return int(self._L.cbh_vidx_entries(self._h, int(skip_frames)))

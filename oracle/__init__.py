"""oracle -- TEST INFRASTRUCTURE ONLY (parity checker for the HIP path).

ctypes bindings for
  * ``libcbird_oracle.so``  -- the plain-C restatement in ``oracle/cbird_oracle.c``
  * ``_ref/libcbird_ref.so`` -- the real reference VP-tree/hamm64 compiled in place from
    ``/root/reference`` by ``oracle/Makefile`` (present when it was built in the build
    container; it travels to the GPU box as a prebuilt file).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``cbird_amd/`` does (tests/test_boundary.py enforces it).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libcbird_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libcbird_ref.so")

_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def _load_ref(path: str):
    """dlopen one of oracle/_ref's libraries.  Both link this image's conda Qt 5.9.7 with RUNPATH /opt/conda/lib, a
    directory that also holds an OLDER libstdc++.so.6 (6.0.28): a process that has not loaded the system's libstdc++ yet
    would bind that one (libcbird_ref_qt.so then fails on `std::__throw_bad_array_new_length`, and whatever is imported
    later -- torch -- would meet the old library).  So the system's is loaded first, by soname: the loader reuses it for
    every later NEEDED libstdc++.so.6."""
    C.CDLL("libstdc++.so.6", mode=C.RTLD_GLOBAL)
    return C.CDLL(path)


def build(force: bool = False) -> None:
    """Compile the C restatement (and, when /root/reference exists, oracle/_ref)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE)
            if (f.endswith(".c") or f == "retain_stl.cpp" or f == "Makefile")]
    need = force or not os.path.exists(_ORACLE_SO) or (
        os.path.getmtime(_ORACLE_SO) < max(os.path.getmtime(f) for f in srcs))
    need_ref = os.path.isdir("/root/reference/src/tree") and (
        force or not os.path.exists(_REF_SO)
        or os.path.getmtime(_REF_SO) < os.path.getmtime(os.path.join(_HERE, "ref_wrap.cpp")))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "libcbird_oracle.so", "-B"],
                              stdout=subprocess.DEVNULL)
    if need_ref:
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class Oracle:
    """Plain-C restatement (kind "port")."""

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.orc_hamm64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_hamm64.restype = C.c_int
        for name in ("orc_scan64", "orc_find64"):
            f = getattr(L, name)
            f.argtypes = [_u64p, _u32p, C.c_size_t, C.c_uint64, C.c_int, _u32p, _i32p, C.c_size_t]
            f.restype = C.c_longlong
        L.orc_find64_batch.argtypes = [_u64p, _u32p, C.c_size_t, _u64p, C.c_size_t, C.c_int,
                                       C.c_int, _u32p, _i32p, _u32p]
        L.orc_find64_batch.restype = None
        L.orc_count64_pairs.argtypes = [_u64p, _u32p, C.c_size_t, _u64p, C.c_size_t, C.c_int]
        L.orc_count64_pairs.restype = C.c_longlong
        L.orc_zigzag81.argtypes = [_i32p]
        L.orc_dct9_table.argtypes = [_f32p]
        L.orc_blur_ksize.argtypes = [C.c_int, C.c_int]
        L.orc_blur_ksize.restype = C.c_int
        L.orc_dcthash64.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, _u64p]
        L.orc_dcthash64.restype = C.c_int
        L.orc_dcthash_tile32.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, _u8p]
        L.orc_dcthash_tile32.restype = C.c_int
        L.orc_dcthash64_batch.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_size_t,
                                          C.c_size_t, _u64p]
        L.orc_dcthash64_batch.restype = C.c_int
        L.orc_hash_from_tile32.argtypes = [_u8p, C.c_void_p, C.c_void_p]
        L.orc_hash_from_tile32.restype = C.c_uint64
        L.orc_hash_from_tile32_v.argtypes = [_u8p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_hash_from_tile32_v.restype = C.c_uint64
        L.orc_hash_from_tile32_f64.argtypes = [_u8p, C.c_void_p, C.c_void_p]
        L.orc_hash_from_tile32_f64.restype = C.c_uint64
        L.orc_set_hash_variant.argtypes = [C.c_int]
        L.orc_get_hash_variant.restype = C.c_int
        L.orc_hash_tiles_stats.argtypes = [_u8p, C.c_size_t, C.c_int, _u64p, C.c_void_p]
        L.orc_hash_tiles_stats.restype = None
        L.orc_hash_tiles_risk.argtypes = [_u8p, C.c_size_t, _u64p, _u64p, _u64p] + [C.c_void_p] * 5
        L.orc_hash_tiles_risk.restype = None
        L.orc_cv_dct32x32.argtypes = [_f32p]
        L.orc_cv_dct32_1d.argtypes = [_f32p, _f32p]
        L.orc_cv_sum_f32.argtypes = [_f32p, C.c_int]
        L.orc_cv_sum_f32.restype = C.c_double
        L.orc_fdct_find.argtypes = [_u64p, _u32p, C.c_size_t, _u64p, C.c_size_t, C.c_uint32, C.c_int,
                                    _u32p, _i32p, C.c_size_t]
        L.orc_fdct_find.restype = C.c_longlong
        for name in ("orc_box_blur", "orc_box_blur_direct"):
            f = getattr(L, name)
            f.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, _u8p]
            f.restype = None

    # -- search -------------------------------------------------------------------------
    def hamm64(self, a: int, b: int) -> int:
        return self.L.orc_hamm64(a, b)

    def _run(self, fn, hashes, ids, target, thresh):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        cap = max(1, len(hashes))
        oi = np.zeros(cap, np.uint32)
        od = np.zeros(cap, np.int32)
        m = fn(hashes, ids, len(hashes), int(target), int(thresh), oi, od, cap)
        return oi[:m].copy(), od[:m].copy()

    def scan64(self, hashes, ids, target, thresh):
        """matches in haystack order: (ids, dists)"""
        return self._run(self.L.orc_scan64, hashes, ids, target, thresh)

    def find64(self, hashes, ids, target, thresh):
        """matches in (score, mediaId) order: (ids, dists)"""
        return self._run(self.L.orc_find64, hashes, ids, target, thresh)

    def find64_batch(self, hashes, ids, queries, thresh, k):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        queries = np.ascontiguousarray(queries, np.uint64)
        nq = len(queries)
        oi = np.zeros((nq, k), np.uint32)
        od = np.zeros((nq, k), np.int32)
        cnt = np.zeros(nq, np.uint32)
        self.L.orc_find64_batch(hashes, ids, len(hashes), queries, nq, int(thresh), int(k),
                                oi.reshape(-1), od.reshape(-1), cnt)
        return oi, od, cnt

    def count64_pairs(self, hashes, ids, queries, thresh) -> int:
        hashes = np.ascontiguousarray(hashes, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        queries = np.ascontiguousarray(queries, np.uint64)
        return int(self.L.orc_count64_pairs(hashes, ids, len(hashes), queries, len(queries),
                                            int(thresh)))

    def fdct_find(self, hashes, ids, needle_hashes, needle_id, thresh):
        """DctFeaturesIndex::find with exact candidates: (ids, scores) ascending mediaId"""
        hashes = np.ascontiguousarray(hashes, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        nh = np.ascontiguousarray(needle_hashes, np.uint64)
        cap = len(nh) * 10 + 1
        oi = np.zeros(cap, np.uint32)
        osc = np.zeros(cap, np.int32)
        m = self.L.orc_fdct_find(hashes, ids, len(hashes), nh, len(nh), int(needle_id), int(thresh),
                                 oi, osc, cap)
        return oi[:m].copy(), osc[:m].copy()

    def htree_leaf_masks(self, hashes, needle_hashes):
        """equal-bits masks that restrict a needle hash to its HammingTree leaf (orc_htree_leaf_masks)"""
        hashes = np.ascontiguousarray(hashes, np.uint64)
        nh = np.ascontiguousarray(needle_hashes, np.uint64)
        out = np.zeros(len(nh), np.uint64)
        self.L.orc_htree_leaf_masks.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t, _u64p]
        self.L.orc_htree_leaf_masks.restype = None
        self.L.orc_htree_leaf_masks(hashes, len(hashes), nh, len(nh), out)
        return out

    def fdct_find_tree(self, hashes, ids, needle_hashes, needle_id, thresh):
        """DctFeaturesIndex::find with the reference tree's (approximate) candidate sets"""
        hashes = np.ascontiguousarray(hashes, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        nh = np.ascontiguousarray(needle_hashes, np.uint64)
        mk = self.htree_leaf_masks(hashes, nh)
        cap = len(nh) * 10 + 1
        oi = np.zeros(cap, np.uint32)
        osc = np.zeros(cap, np.int32)
        f = self.L.orc_fdct_find_masked
        f.argtypes = [_u64p, _u32p, C.c_size_t, _u64p, _u64p, C.c_size_t, C.c_uint32, C.c_int, _u32p, _i32p,
                      C.c_size_t]
        f.restype = C.c_longlong
        m = f(hashes, ids, len(hashes), nh, mk, len(nh), int(needle_id), int(thresh), oi, osc, cap)
        return oi[:m].copy(), osc[:m].copy()

    # -- hashing ------------------------------------------------------------------------
    def zigzag81(self):
        z = np.zeros(81, np.int32)
        self.L.orc_zigzag81(z)
        return z

    def dct9_table(self):
        t = np.zeros(9 * 32, np.float32)
        self.L.orc_dct9_table(t)
        return t.reshape(9, 32)

    def blur_ksize(self, w, h):
        return self.L.orc_blur_ksize(w, h)

    def box_blur(self, img, k, direct=False):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros_like(img)
        (self.L.orc_box_blur_direct if direct else self.L.orc_box_blur)(img, w, h, w, k, out)
        return out

    def tile32(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        t = np.zeros((32, 32), np.uint8)
        rc = self.L.orc_dcthash_tile32(img, w, h, w, t.reshape(-1))
        if rc:
            raise ValueError(f"orc_dcthash_tile32 rc={rc}")
        return t

    def hash_from_tile32(self, tile, with_coefs=False):
        tile = np.ascontiguousarray(tile, np.uint8).reshape(-1)
        if not with_coefs:
            return int(self.L.orc_hash_from_tile32(tile, None, None))
        co = np.zeros(64, np.float32)
        th = np.zeros(1, np.float32)
        hv = self.L.orc_hash_from_tile32(tile, co.ctypes.data, th.ctypes.data)
        return int(hv), co, float(th[0])

    # ---- stage 3/5 arithmetic: 1 = cv::dct / cv::sum as OpenCV 2.4 evaluates them (default), 0 = canonical ----
    def set_hash_variant(self, v: int):
        self.L.orc_set_hash_variant(int(v))

    def hash_variant(self) -> int:
        return int(self.L.orc_get_hash_variant())

    def hash_from_tile32_v(self, tile, variant, with_coefs=False):
        tile = np.ascontiguousarray(tile, np.uint8).reshape(-1)
        if variant == 2:  # float64 yardstick
            co, th = np.zeros(64, np.float64), np.zeros(1, np.float64)
            hv = self.L.orc_hash_from_tile32_f64(tile, co.ctypes.data, th.ctypes.data)
        else:
            co, th = np.zeros(64, np.float32), np.zeros(1, np.float32)
            hv = self.L.orc_hash_from_tile32_v(tile, int(variant), co.ctypes.data, th.ctypes.data)
        return (int(hv), co, float(th[0])) if with_coefs else int(hv)

    def hash_tiles_stats(self, tiles, variant):
        """hashes u64[n] and the per-tile smallest |coef - threshold| for variant 0 / 1 / 2 (float64)"""
        tiles = np.ascontiguousarray(tiles, np.uint8).reshape(-1, 1024)
        n = len(tiles)
        hs, mm = np.zeros(n, np.uint64), np.zeros(n, np.float64)
        self.L.orc_hash_tiles_stats(tiles.reshape(-1), n, int(variant), hs, mm.ctypes.data)
        return hs, mm

    def hash_tiles_risk(self, tiles):
        """dict of per-tile arrays: h0 (canonical), h1 (cv::dct), h2 (float64), min_margin, min_bit, err0, err1, thr64"""
        tiles = np.ascontiguousarray(tiles, np.uint8).reshape(-1, 1024)
        n = len(tiles)
        r = {k: np.zeros(n, np.uint64) for k in ("h0", "h1", "h2")}
        r.update({k: np.zeros(n, np.float64) for k in ("min_margin", "err0", "err1", "thr64")})
        r["min_bit"] = np.zeros(n, np.int32)
        self.L.orc_hash_tiles_risk(tiles.reshape(-1), n, r["h0"], r["h1"], r["h2"], r["min_margin"].ctypes.data,
                                   r["min_bit"].ctypes.data, r["err0"].ctypes.data, r["err1"].ctypes.data,
                                   r["thr64"].ctypes.data)
        return r

    def cv_dct32x32(self, m):
        m = np.ascontiguousarray(m, np.float32).copy()
        self.L.orc_cv_dct32x32(m.reshape(-1))
        return m

    def cv_dct32_1d(self, x):
        x = np.ascontiguousarray(x, np.float32)
        o = np.zeros(32, np.float32)
        self.L.orc_cv_dct32_1d(x, o)
        return o

    def cv_sum_f32(self, x) -> float:
        x = np.ascontiguousarray(x, np.float32)
        return float(self.L.orc_cv_sum_f32(x, len(x)))

    def similar_dct(self, hay_hash, hay_id, hay_rank, idx_hash, idx_id, thresh, max_thresh, min_matches, max_matches,
                    filter_self=True, filter_groups=True):
        """oracle/search_index.c: Database::similar over DctHashIndex (searchIndex + filterMatch + filterMatches).
        Returns a list of (needle position, [(id, score), ...]) in result order."""
        hh, hi = np.ascontiguousarray(hay_hash, np.uint64), np.ascontiguousarray(hay_id, np.uint32)
        hr = np.ascontiguousarray(hay_rank, np.int32)
        ih, ii = np.ascontiguousarray(idx_hash, np.uint64), np.ascontiguousarray(idx_id, np.uint32)
        n = len(hh)
        cap_g, cap_i = max(1, n), max(1, n * max(1, max_matches))
        on, of = np.zeros(cap_g, np.uint32), np.zeros(cap_g + 1, np.uint64)
        oi, os_ = np.zeros(cap_i, np.uint32), np.zeros(cap_i, np.int32)
        f = self.L.orc_similar_dct
        f.argtypes = [_u64p, _u32p, _i32p, C.c_size_t, _u64p, _u32p, C.c_size_t] + [C.c_int] * 6 + \
                     [_u32p, _u64p, _u32p, _i32p, C.c_size_t, C.c_size_t]
        f.restype = C.c_longlong
        g = f(hh, hi, hr, n, ih, ii, len(ih), int(thresh), int(max_thresh), int(min_matches), int(max_matches),
              int(bool(filter_self)), int(bool(filter_groups)), on, of, oi, os_, cap_g, cap_i)
        if g < 0:
            raise ValueError("orc_similar_dct: capacity")
        return [(int(on[k]), list(zip(oi[int(of[k]):int(of[k + 1])].tolist(), os_[int(of[k]):int(of[k + 1])].tolist())))
                for k in range(g)]

    def filter_groups_paths(self, paths, groups, db_path, param_path, in_path, filter_parent, min_matches,
                            filter_groups, merge_groups, expand_groups):
        """oracle/search_index.c orc_filter_groups_paths: filterMatch + filterMatches on path strings.  groups = lists
        of (index into paths, score), the needle first.  Returns the surviving groups in the same form."""
        bp = [p.encode() for p in paths]
        arr = (C.c_char_p * len(bp))(*bp)
        first = np.zeros(len(groups) + 1, np.uint64)
        np.cumsum([len(g) for g in groups], out=first[1:])
        mem = np.array([m for g in groups for m, _ in g], np.int32)
        sc = np.array([s for g in groups for _, s in g], np.int32)
        cap_m = max(1, 2 * len(mem) * (4 if expand_groups else 1))
        cap_g = max(1, len(mem))
        of, om, os_ = np.zeros(cap_g + 1, np.uint64), np.zeros(cap_m, np.int32), np.zeros(cap_m, np.int32)
        f = self.L.orc_filter_groups_paths
        f.argtypes = [C.POINTER(C.c_char_p), _u64p, _i32p, _i32p, C.c_size_t, C.c_char_p, C.c_char_p] + [C.c_int] * 6 + \
                     [_u64p, _i32p, _i32p, C.c_size_t, C.c_size_t]
        f.restype = C.c_longlong
        n = f(arr, first, mem if len(mem) else np.zeros(1, np.int32), sc if len(sc) else np.zeros(1, np.int32),
              len(groups), db_path.encode(), param_path.encode(), int(in_path), int(filter_parent), int(min_matches),
              int(filter_groups), int(merge_groups), int(expand_groups), of, om, os_, cap_g, cap_m)
        if n < 0:
            raise ValueError("orc_filter_groups_paths: capacity")
        return [[(int(om[t]), int(os_[t])) for t in range(int(of[g]), int(of[g + 1]))] for g in range(n)]

    def dcthash64(self, img) -> int:
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        o = np.zeros(1, np.uint64)
        rc = self.L.orc_dcthash64(img, w, h, w, o)
        if rc:
            raise ValueError(f"orc_dcthash64 rc={rc}")
        return int(o[0])

    def dcthash64_fast256_batch(self, imgs):
        """oracle/fast_hash.c: the CPU-baseline form of dctHash64 for 256 x 256 images (same hashes as dcthash64_batch)"""
        imgs = np.ascontiguousarray(imgs, np.uint8)
        n, h, w = imgs.shape
        if (h, w) != (256, 256):
            raise ValueError("fast256: 256 x 256 images only")
        o = np.zeros(n, np.uint64)
        f = self.L.orc_dcthash64_fast256_batch
        f.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C"), C.c_size_t, C.c_size_t, C.c_size_t, _u64p]
        f.restype = C.c_int
        f(imgs.reshape(-1), n, w, w * h, o)
        return o

    def dcthash64_batch(self, imgs):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        n, h, w = imgs.shape
        o = np.zeros(n, np.uint64)
        rc = self.L.orc_dcthash64_batch(imgs.reshape(-1), n, w, h, w, w * h, o)
        if rc:
            raise ValueError(f"orc_dcthash64_batch rc={rc}")
        return o

    # ---- sizeLongestSide (src/cvutil.cpp:1932-1950) ----
    def longest_side_dims(self, w, h, size):
        ow, oh = C.c_int(0), C.c_int(0)
        f = self.L.orc_longest_side_dims
        f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        f.restype = None
        f(w, h, size, C.byref(ow), C.byref(oh))
        return ow.value, oh.value

    def lanczos4_tab(self, ssize, dsize):
        ofs = np.zeros(dsize, np.int32)
        coef = np.zeros((dsize, 8), np.int16)
        f = self.L.orc_lanczos4_tab
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        f.restype = None
        f(ssize, dsize, ofs.ctypes.data, coef.ctypes.data)
        return ofs, coef

    def resize_lanczos4(self, img, dw, dh):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((dh, dw), np.uint8)
        f = self.L.orc_resize_lanczos4_u8
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        f.restype = C.c_int
        rc = f(img.ctypes.data, w, h, w, dw, dh, out.ctypes.data)
        if rc:
            raise ValueError(f"orc_resize_lanczos4_u8 rc={rc}")
        return out

    def size_longest_side(self, img, size=400):
        h, w = img.shape
        dw, dh = self.longest_side_dims(w, h, size)
        if dw == 0 or dh == 0:
            raise ValueError("sizeLongestSide: computed width or height is 0")
        return self.resize_lanczos4(img, dw, dh)

    # ---- Media::makeKeyPointHashes (src/media.cpp:874-923) ----
    def keypoint_rects(self, cols, rows, kp):
        kp = np.ascontiguousarray(kp, np.float32).reshape(-1, 3)
        r = np.zeros((max(1, len(kp)), 3), np.int32)
        f = self.L.orc_keypoint_rects
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        f.restype = C.c_int
        n = f(cols, rows, kp.ctypes.data, len(kp), r.ctypes.data)
        return r[:n].copy()

    def keypoint_hashes(self, img, kp):
        """returns (hashes u64[], image after the in-place blurs)"""
        img = np.ascontiguousarray(img, np.uint8).copy()
        kp = np.ascontiguousarray(kp, np.float32).reshape(-1, 3)
        h, w = img.shape
        o = np.zeros(max(1, len(kp)), np.uint64)
        f = self.L.orc_keypoint_hashes
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]
        f.restype = C.c_int
        n = f(img.ctypes.data, w, h, w, kp.ctypes.data, len(kp), o.ctypes.data)
        if n < 0:
            raise ValueError(f"orc_keypoint_hashes rc={n}")
        return o[:n].copy(), img

    def dcthash64_rect_inplace(self, img, x, y, rw, rh):
        """img (2-D u8, C-contiguous) is modified like cv::blur on the view would; returns the hash"""
        assert img.dtype == np.uint8 and img.flags.c_contiguous
        h, w = img.shape
        o = np.zeros(1, np.uint64)
        f = self.L.orc_dcthash64_rect_inplace
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t] + [C.c_int] * 4 + [C.c_void_p]
        f.restype = C.c_int
        rc = f(img.ctypes.data, w, h, w, x, y, rw, rh, o.ctypes.data)
        if rc:
            raise ValueError(f"orc_dcthash64_rect_inplace rc={rc}")
        return int(o[0])

    def resize_linear_tab(self, ssize, is_x):
        ofs = np.zeros(32, np.int32)
        c0 = np.zeros(32, np.int16)
        c1 = np.zeros(32, np.int16)
        f = self.L.orc_resize_linear_tab
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = None
        f(ssize, int(is_x), ofs.ctypes.data, c0.ctypes.data, c1.ctypes.data)
        return ofs, c0, c1


_REF_QT_SO = os.path.join(_HERE, "_ref", "libcbird_ref_qt.so")


def ref_qt_available() -> bool:
    return os.path.exists(_REF_QT_SO)


class RefHammingTree:
    """The real HammingTree_t<uint32_t> behind DctFeaturesIndex (oracle/ref_wrap_qt.cpp)."""

    _L = None

    @classmethod
    def lib(cls):
        if cls._L is None:
            L = _load_ref(_REF_QT_SO)
            L.ref_htree_create.restype = C.c_void_p
            L.ref_htree_destroy.argtypes = [C.c_void_p]
            L.ref_htree_size.argtypes = [C.c_void_p]
            L.ref_htree_size.restype = C.c_size_t
            L.ref_htree_insert.argtypes = [C.c_void_p, _u32p, _u64p, C.c_size_t]
            L.ref_htree_remove.argtypes = [C.c_void_p, _u32p, C.c_size_t]
            L.ref_htree_search.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _u32p, _u64p, _i32p, C.c_int]
            L.ref_htree_search.restype = C.c_int
            L.ref_fdct_find.argtypes = [C.c_void_p, _u64p, C.c_int, C.c_int, C.c_int, _u32p, _i32p, C.c_int]
            L.ref_fdct_find.restype = C.c_int
            cls._L = L
        return cls._L

    def __init__(self):
        self.h = self.lib().ref_htree_create()

    def insert(self, ids, hashes):
        ids = np.ascontiguousarray(ids, np.uint32)
        hashes = np.ascontiguousarray(hashes, np.uint64)
        self.lib().ref_htree_insert(self.h, ids, hashes, len(ids))

    def remove(self, ids):
        ids = np.ascontiguousarray(ids, np.uint32)
        self.lib().ref_htree_remove(self.h, ids, len(ids))

    def size(self):
        return self.lib().ref_htree_size(self.h)

    def write(self, path):
        """HammingTree::write: the dctfeatures.cache file"""
        f = self.lib().ref_htree_write
        f.argtypes = [C.c_void_p, C.c_char_p]
        f.restype = C.c_int
        if not f(self.h, os.fsencode(path)):
            raise OSError(f"ref_htree_write({path}) failed")

    def read(self, path):
        f = self.lib().ref_htree_read
        f.argtypes = [C.c_void_p, C.c_char_p]
        f.restype = C.c_int
        return bool(f(self.h, os.fsencode(path)))

    def search(self, target, thresh, cap=1 << 16):
        oi = np.zeros(cap, np.uint32)
        oh = np.zeros(cap, np.uint64)
        od = np.zeros(cap, np.int32)
        m = self.lib().ref_htree_search(self.h, int(target), int(thresh), oi, oh, od, cap)
        return oi[:m].copy(), oh[:m].copy(), od[:m].copy()

    def fdct_find(self, needle_hashes, needle_id, thresh):
        nh = np.ascontiguousarray(needle_hashes, np.uint64)
        cap = len(nh) * 10 + 1
        oi = np.zeros(cap, np.uint32)
        osc = np.zeros(cap, np.int32)
        m = self.lib().ref_fdct_find(self.h, nh, len(nh), int(needle_id), int(thresh), oi, osc, cap)
        return oi[:m].copy(), osc[:m].copy()

    def __del__(self):
        try:
            self.lib().ref_htree_destroy(self.h)
        except Exception:
            pass


def ref_available() -> bool:
    if not os.path.exists(_REF_SO) and os.path.isdir("/root/reference/src/tree"):
        build()
    return os.path.exists(_REF_SO)


class RefTree:
    """The real reference VP-tree (kind "reference"); see oracle/ref_wrap.cpp."""

    _L = None

    @classmethod
    def lib(cls):
        if cls._L is None:
            if not ref_available():
                raise FileNotFoundError(_REF_SO)
            L = _load_ref(_REF_SO)
            L.ref_hamm64.argtypes = [C.c_uint64, C.c_uint64]
            L.ref_hamm64.restype = C.c_int
            L.ref_dcttree_create.argtypes = [_u64p, _u32p, C.c_int]
            L.ref_dcttree_create.restype = C.c_void_p
            L.ref_dcttree_destroy.argtypes = [C.c_void_p]
            L.ref_dcttree_search.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _u32p, _i32p, C.c_int]
            L.ref_dcttree_search.restype = C.c_int
            L.ref_dcttree_search_many.argtypes = [C.c_void_p, _u64p, C.c_int, C.c_int, C.c_int,
                                                  C.c_void_p]
            L.ref_dcttree_search_many.restype = C.c_longlong
            if hasattr(L, "ref_dcttree_search_lists"):  # (absent from an oracle/_ref built before round 4)
                L.ref_dcttree_search_lists.argtypes = [C.c_void_p, _u64p, C.c_int, C.c_int, C.c_int]
                L.ref_dcttree_search_lists.restype = C.c_void_p
                L.ref_lists_total.argtypes = [C.c_void_p]
                L.ref_lists_total.restype = C.c_ulonglong
                L.ref_lists_copy.argtypes = [C.c_void_p, _u64p, _u32p, _i32p]
                L.ref_lists_copy.restype = None
                L.ref_lists_free.argtypes = [C.c_void_p]
                L.ref_lists_free.restype = None
            cls._L = L
        return cls._L

    def __init__(self, hashes, ids) -> None:
        L = self.lib()
        self.hashes = np.ascontiguousarray(hashes, np.uint64)
        self.ids = np.ascontiguousarray(ids, np.uint32)
        self.n = len(self.hashes)
        self.h = L.ref_dcttree_create(self.hashes, self.ids, self.n)

    def search(self, target, thresh):
        """DctTree::search: (ids, dists) ascending by distance (tie order = heap order)."""
        if not self.h:
            return np.zeros(0, np.uint32), np.zeros(0, np.int32)
        cap = max(1, self.n)
        oi = np.zeros(cap, np.uint32)
        od = np.zeros(cap, np.int32)
        m = self.lib().ref_dcttree_search(self.h, int(target), int(thresh), oi, od, cap)
        return oi[:m].copy(), od[:m].copy()

    def search_many(self, needles, thresh, threads=1, want_counts=False):
        needles = np.ascontiguousarray(needles, np.uint64)
        cnt = np.zeros(len(needles), np.uint32) if want_counts else None
        tot = self.lib().ref_dcttree_search_many(
            self.h, needles, len(needles), int(thresh), int(threads),
            cnt.ctypes.data if cnt is not None else None)
        return (int(tot), cnt) if want_counts else int(tot)

    def search_lists(self, needles, thresh, threads=1):
        """Every needle's full result list from the real tree, in CSR form: (offsets u64[nq+1], ids u32[total],
        dists i32[total]); each list in canonical order (distance, then mediaId, ascending)."""
        needles = np.ascontiguousarray(needles, np.uint64)
        L = self.lib()
        lp = L.ref_dcttree_search_lists(self.h, needles, len(needles), int(thresh), int(threads))
        try:
            tot = int(L.ref_lists_total(lp))
            off = np.zeros(len(needles) + 1, np.uint64)
            oi = np.zeros(max(1, tot), np.uint32)
            od = np.zeros(max(1, tot), np.int32)
            L.ref_lists_copy(lp, off, oi, od)
        finally:
            L.ref_lists_free(lp)
        return off, oi[:tot], od[:tot]

    def close(self):
        if self.h:
            self.lib().ref_dcttree_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- video (DctVideoIndex / VideoIndex) -----------------------------------------------------------
class _OrcVMatch(C.Structure):
    _fields_ = [("id", C.c_uint32), ("score", C.c_int32), ("src_in", C.c_int32), ("dst_in", C.c_int32),
                ("len", C.c_int32)]


class VideoOracle:
    """oracle/cbird_oracle.c: .vdx v2 codec, insertHashes filter, findFrame/findVideo, frame de-dup."""

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        i32p = _i32p
        L.orc_vdx_encode.argtypes = [i32p, _u64p, C.c_size_t, C.c_char_p, _u8p, C.c_size_t]
        L.orc_vdx_encode.restype = C.c_size_t
        L.orc_vdx_decode.argtypes = [_u8p, C.c_size_t, i32p, _u64p, C.c_size_t]
        L.orc_vdx_decode.restype = C.c_longlong
        L.orc_video_insert_filter.argtypes = [i32p, _u64p, C.c_size_t, C.c_int, _u8p]
        L.orc_video_insert_filter.restype = C.c_size_t
        L.orc_video_find_frame.argtypes = [_u32p, i32p, _u64p, C.c_size_t, _u32p, C.c_size_t, C.c_uint,
                                           C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_video_find_frame.restype = C.c_longlong
        L.orc_video_find_video.argtypes = [_u32p, i32p, _u64p, C.c_size_t, _u32p, C.c_size_t, C.c_uint, i32p,
                                           _u64p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_size_t]
        L.orc_video_find_video.restype = C.c_longlong
        L.orc_video_dedup.argtypes = [_u64p, C.c_size_t, C.c_int, _u8p]
        L.orc_video_dedup.restype = C.c_size_t
        L.orc_make_video_index.argtypes = [_u64p, C.c_size_t, C.c_int, C.c_int, i32p, _u64p, C.c_size_t, C.c_size_t]
        L.orc_make_video_index.restype = C.c_longlong

    def vdx_encode(self, frames, hashes, version="0.8.1") -> bytes:
        f = np.ascontiguousarray(frames, np.int32)
        h = np.ascontiguousarray(hashes, np.uint64)
        cap = 256 + 13 * len(f) + 64
        out = np.zeros(cap, np.uint8)
        n = self.L.orc_vdx_encode(f, h, len(f), version.encode(), out, cap)
        if n == 0:
            raise ValueError("invalid frames")
        return out[:n].tobytes()

    def vdx_decode(self, data: bytes):
        buf = np.frombuffer(data, np.uint8).copy()
        cap = max(1, len(buf))
        f = np.zeros(cap, np.int32)
        h = np.zeros(cap, np.uint64)
        n = self.L.orc_vdx_decode(buf, len(buf), f, h, cap)
        if n < 0:
            raise ValueError(f"vdx decode error {n}")
        return f[:n].copy(), h[:n].copy()

    def vdx_any_decode(self, data: bytes):
        """VideoIndex::load: version 1 or 2 by the magic; None where the loader fails"""
        f = self.L.orc_vdx_any_decode
        f.argtypes = [_u8p, C.c_size_t, _i32p, _u64p, C.c_size_t]
        f.restype = C.c_longlong
        buf = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
        cap = len(data) + 2
        fr, hs = np.zeros(cap, np.int32), np.zeros(cap, np.uint64)
        n = f(buf, len(data), fr, hs, cap)
        return None if n < 0 else (fr[:n].copy(), hs[:n].copy())

    def vdx_any_verify(self, data: bytes) -> bool:
        f = self.L.orc_vdx_any_verify
        f.argtypes = [_u8p, C.c_size_t]
        f.restype = C.c_int
        buf = np.frombuffer(data, np.uint8) if len(data) else np.zeros(1, np.uint8)
        return bool(f(buf, len(data)))

    def vdx_encode_v1(self, frames, hashes) -> bytes:
        f = self.L.orc_vdx_encode_v1
        f.argtypes = [_i32p, _u64p, C.c_size_t, _u8p, C.c_size_t]
        f.restype = C.c_size_t
        fr = np.ascontiguousarray(frames, np.int32)
        hs = np.ascontiguousarray(hashes, np.uint64)
        if len(fr) == 0:
            fr, hs = np.zeros(1, np.int32), np.zeros(1, np.uint64)
            n = 0
        else:
            n = len(frames)
        out = np.zeros(2 + 10 * max(1, n), np.uint8)
        size = f(fr, hs, n, out, len(out))
        return out[:size].tobytes()

    def vdx_verify(self, data: bytes) -> bool:
        buf = np.frombuffer(data, np.uint8).copy()
        f = self.L.orc_vdx_verify
        f.argtypes = [_u8p, C.c_size_t]
        f.restype = C.c_int
        return bool(f(buf, len(buf))) if len(buf) else False

    def build_entries(self, videos, skip):
        """videos: list of (media_id, frames, hashes) in _mediaId order -> (evidx, eframe, ehash, mediaIds)
        after the insertHashes filters"""
        ev, ef, eh, mids = [], [], [], []
        for vi, (mid, frames, hashes) in enumerate(videos):
            mids.append(mid)
            f = np.ascontiguousarray(frames, np.int32)
            h = np.ascontiguousarray(hashes, np.uint64)
            keep = np.zeros(max(1, len(f)), np.uint8)
            self.L.orc_video_insert_filter(f, h, len(f), int(skip), keep)
            k = keep[: len(f)].astype(bool)
            ev.append(np.full(int(k.sum()), vi, np.uint32))
            ef.append(f[k])
            eh.append(h[k])
        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0, dt), dt)
        return cat(ev, np.uint32), cat(ef, np.int32), cat(eh, np.uint64), np.asarray(mids, np.uint32)

    @staticmethod
    def _out(buf, n):
        return [(buf[i].id, buf[i].score, buf[i].src_in, buf[i].dst_in, buf[i].len) for i in range(n)]

    def find_frame(self, entries, hash_, thresh, src_in=-1, radix=0):
        ev, ef, eh, mids = entries
        cap = max(1, len(mids))
        buf = (_OrcVMatch * cap)()
        n = self.L.orc_video_find_frame(ev, ef, eh, len(ev), mids, len(mids), radix, int(hash_), int(thresh),
                                        int(src_in), buf, cap)
        return self._out(buf, n)

    def find_video(self, entries, frames, hashes, needle_id, thresh, skip, vfm, vfn, filter_self=True, radix=0):
        ev, ef, eh, mids = entries
        f = np.ascontiguousarray(frames, np.int32)
        h = np.ascontiguousarray(hashes, np.uint64)
        cap = max(1, len(mids))
        buf = (_OrcVMatch * cap)()
        n = self.L.orc_video_find_video(ev, ef, eh, len(ev), mids, len(mids), radix, f, h, len(f), int(needle_id),
                                        int(thresh), int(skip), int(vfm), int(vfn), int(bool(filter_self)), buf,
                                        cap)
        return self._out(buf, n)

    def make_video_index(self, frame_hashes, threshold, resume=None, max_frames=1 << 24):
        """Media::makeVideoIndex given every decoded frame's hash; resume = (frames, hashes) of an earlier index"""
        fh = np.ascontiguousarray(frame_hashes, np.uint64)
        rf, rh = (np.zeros(0, np.int32), np.zeros(0, np.uint64)) if resume is None else resume
        cap = len(rf) + len(fh) + 1
        frames = np.zeros(cap, np.int32)
        hashes = np.zeros(cap, np.uint64)
        frames[: len(rf)] = rf
        hashes[: len(rh)] = rh
        n = self.L.orc_make_video_index(fh if len(fh) else np.zeros(1, np.uint64), len(fh), int(threshold),
                                        int(max_frames), frames, hashes, len(rf), cap)
        assert n >= 0
        return frames[:n].copy(), hashes[:n].copy()

    def dedup(self, hashes, threshold):
        h = np.ascontiguousarray(hashes, np.uint64)
        keep = np.zeros(max(1, len(h)), np.uint8)
        self.L.orc_video_dedup(h, len(h), int(threshold), keep)
        return keep[: len(h)].astype(bool)


class RefRadixMap:
    """The real RadixMap_t<VideoTreeIndex> (src/tree/radix.h) via oracle/ref_wrap_qt.cpp."""

    def __init__(self, radix: int):
        L = _load_ref(_REF_QT_SO)
        self.L = L
        L.ref_radix_create.restype = C.c_void_p
        L.ref_radix_create.argtypes = [C.c_uint]
        L.ref_radix_destroy.argtypes = [C.c_void_p]
        L.ref_radix_insert.argtypes = [C.c_void_p, _u32p, _u32p, _u64p, C.c_size_t]
        L.ref_radix_search.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _u32p, _u32p, _u64p, _i32p, C.c_int]
        L.ref_radix_search.restype = C.c_int
        self.h = L.ref_radix_create(radix)

    def insert(self, vidx, frame, hashes):
        self.L.ref_radix_insert(self.h, np.ascontiguousarray(vidx, np.uint32),
                                np.ascontiguousarray(frame, np.uint32), np.ascontiguousarray(hashes, np.uint64),
                                len(vidx))

    def search(self, hash_, thresh, cap=1 << 16):
        ov = np.zeros(cap, np.uint32)
        of = np.zeros(cap, np.uint32)
        oh = np.zeros(cap, np.uint64)
        od = np.zeros(cap, np.int32)
        m = self.L.ref_radix_search(self.h, int(hash_), int(thresh), ov, of, oh, od, cap)
        return ov[:m].copy(), of[:m].copy(), oh[:m].copy(), od[:m].copy()

    def __del__(self):
        try:
            self.L.ref_radix_destroy(self.h)
        except Exception:
            pass


class CvOracle:
    """oracle/cbird_oracle.c: exact 256-bit knn + CvFeaturesIndex::find scoring."""

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.orc_knn256.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, _u32p, _i32p, _u32p]
        L.orc_knn256.restype = None
        L.orc_cvfeatures_find.argtypes = [_u8p, C.c_size_t, _u32p, _u32p, C.c_size_t, _u8p, C.c_size_t, C.c_int,
                                          C.c_int, _u32p, _i32p, C.c_size_t]
        L.orc_cvfeatures_find.restype = C.c_longlong

    def knn(self, rows, needles, k, thresh):
        rows = np.ascontiguousarray(rows, np.uint8)
        needles = np.ascontiguousarray(needles, np.uint8)
        nq = len(needles)
        r = np.zeros((nq, k), np.uint32)
        d = np.zeros((nq, k), np.int32)
        c = np.zeros(nq, np.uint32)
        self.L.orc_knn256(rows.reshape(-1), len(rows), needles.reshape(-1), nq, k, thresh, r.reshape(-1),
                          d.reshape(-1), c)
        return r, d, c

    def find(self, rows, first_row, media_id, needles, k, thresh):
        rows = np.ascontiguousarray(rows, np.uint8)
        needles = np.ascontiguousarray(needles, np.uint8)
        fr = np.ascontiguousarray(first_row, np.uint32)
        mi = np.ascontiguousarray(media_id, np.uint32)
        cap = len(needles) * k + 1
        oi = np.zeros(cap, np.uint32)
        osc = np.zeros(cap, np.int32)
        m = self.L.orc_cvfeatures_find(rows.reshape(-1), len(rows), fr, mi, len(fr), needles.reshape(-1),
                                       len(needles), k, thresh, oi, osc, cap)
        return oi[:m].copy(), osc[:m].copy()


class ColorOracle:
    """oracle/cbird_oracle.c: ColorDescriptor::distance / ColorDescIndex::find"""

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.orc_color_distance.argtypes = [_u8p, _u8p]
        L.orc_color_distance.restype = C.c_float
        L.orc_color_find.argtypes = [_u8p, _u32p, C.c_size_t, _u8p, _u32p, _i32p, C.c_size_t]
        L.orc_color_find.restype = C.c_longlong

    @staticmethod
    def _bytes(d):
        return np.ascontiguousarray(np.frombuffer(np.ascontiguousarray(d).tobytes(), np.uint8))

    def distance(self, a, b) -> float:
        return float(self.L.orc_color_distance(self._bytes(a), self._bytes(b)))

    def find(self, descs, ids, target):
        ids = np.ascontiguousarray(ids, np.uint32)
        cap = max(1, len(ids))
        oi = np.zeros(cap, np.uint32)
        osc = np.zeros(cap, np.int32)
        m = self.L.orc_color_find(self._bytes(descs), ids, len(ids), self._bytes(target), oi, osc, cap)
        return oi[:m].copy(), osc[:m].copy()


class PrestageOracle:
    """oracle/cbird_oracle.c: grayscale, autocrop and the processImage hash"""

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.orc_bgr2gray.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, _u8p]
        L.orc_autocrop.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, _i32p]
        L.orc_autocrop.restype = C.c_int
        L.orc_process_image.argtypes = [_u8p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, _u64p, _i32p]
        L.orc_process_image.restype = C.c_int

    def bgr2gray(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w, ch = img.shape
        out = np.zeros((h, w), np.uint8)
        self.L.orc_bgr2gray(img.reshape(-1), w, h, w * ch, ch, out.reshape(-1))
        return out

    def autocrop(self, gray, rng=20):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        r = np.zeros(4, np.int32)
        self.L.orc_autocrop(gray.reshape(-1), w, h, w, int(rng), r)
        return r

    def template_score(self, cand, tmpl):
        """TemplateMatcher::match's score (templatematcher.cpp:331-374): (distance, candHash, tmplHash, cand grey as
        hashed, masked template grey as hashed)"""
        cand = np.ascontiguousarray(cand, np.uint8)
        tmpl = np.ascontiguousarray(tmpl, np.uint8)
        h, w = cand.shape[:2]
        cc = 1 if cand.ndim == 2 else cand.shape[2]
        tc = 1 if tmpl.ndim == 2 else tmpl.shape[2]
        assert tmpl.shape[:2] == (h, w)
        f = self.L.orc_template_score
        f.argtypes = [_u8p, C.c_int, C.c_size_t, _u8p, C.c_int, C.c_size_t, C.c_int, C.c_int, _u64p, _u64p, _u8p, _u8p]
        f.restype = C.c_int
        ch, th = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        cg, tg = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
        d = f(cand.reshape(-1), cc, w * cc, tmpl.reshape(-1), tc, w * tc, w, h, ch, th, cg.reshape(-1), tg.reshape(-1))
        if d < 0:
            raise ValueError(f"orc_template_score rc={d}")
        return d, int(ch[0]), int(th[0]), cg, tg

    def process_image(self, img, autocrop=20):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        out = np.zeros(1, np.uint64)
        r = np.zeros(4, np.int32)
        rc = self.L.orc_process_image(img.reshape(-1), w, h, w * ch, ch, -1 if autocrop is None else autocrop,
                                      out, r)
        if rc:
            raise ValueError(f"orc_process_image rc={rc}")
        return int(out[0]), r


KP_DTYPE = np.dtype([("x", np.float32), ("y", np.float32), ("size", np.float32), ("angle", np.float32),
                     ("response", np.float32), ("octave", np.int32)])


class OrbOracle:
    """oracle/orb_oracle.c: OpenCV 2.4 ORB as cbird configures it (media.cpp:859-872), parity unpinned.
    Keypoints are structured arrays of KP_DTYPE (cv::KeyPoint without class_id)."""

    NLEVELS = 12

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.orc_orb_scale.restype = C.c_float
        L.orc_orb_scale.argtypes = [C.c_int]
        L.orc_harris_response.restype = C.c_float
        L.orc_harris_response.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_orb_detect.restype = C.c_long
        L.orc_orb_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_long]
        L.orc_orb_compute.restype = C.c_long
        L.orc_orb_compute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_long, C.c_void_p]
        L.orc_orb_descriptor.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orc_retain_best_stl.restype = C.c_long
        L.orc_retain_best_stl.argtypes = [_f32p, C.c_long, C.c_int, C.c_int, _i32p]
        L.orc_orb_set_retain_order.argtypes = [C.c_int]
        L.orc_orb_set_retain_order.restype = None

    def set_retain_order(self, mode: int) -> None:
        """0: canonical (every tie kept, raster order); 1: what libstdc++'s nth_element + partition leave
        (oracle/retain_stl.cpp).  Process-wide, like the pattern."""
        self.L.orc_orb_set_retain_order(int(mode))

    def retain_best_stl(self, responses, n_points, depth_limit=-1):
        """KeyPointsFilter::retainBest on the real std::nth_element / std::partition: the original positions of the
        survivors, in the order they are left.  depth_limit >= 0 enters introselect with that limit (heap-select
        branch)."""
        r = np.ascontiguousarray(responses, np.float32)
        order = np.zeros(max(1, len(r)), np.int32)
        k = self.L.orc_retain_best_stl(r if len(r) else np.zeros(1, np.float32), len(r), int(n_points),
                                       int(depth_limit), order)
        return order[:k].copy()

    def set_pattern(self, xy) -> None:
        xy = np.ascontiguousarray(xy, np.int8).reshape(1024)
        if self.L.orc_orb_set_pattern(xy.ctypes.data_as(C.c_void_p)):
            raise ValueError("pattern coordinates must lie in [-15, 15]")

    def level_size(self, w, h, level):
        lw, lh = C.c_int(0), C.c_int(0)
        self.L.orc_orb_level_size(w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def scale(self, level) -> float:
        return float(self.L.orc_orb_scale(level))

    def features_per_level(self, nfeatures):
        out = np.zeros(self.NLEVELS, np.int32)
        self.L.orc_orb_features_per_level(int(nfeatures), out.ctypes.data_as(C.c_void_p))
        return out

    def umax(self):
        out = np.zeros(17, np.int32)
        self.L.orc_orb_umax(out.ctypes.data_as(C.c_void_p))
        return out[:16]

    def gauss7_kernel(self):
        out = np.zeros(7, np.int32)
        self.L.orc_gauss7_kernel(out.ctypes.data_as(C.c_void_p))
        return out

    def resize_linear(self, img, dw, dh):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((dh, dw), np.uint8)
        rc = self.L.orc_resize_linear_u8_cv(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), dw, dh,
                                            out.ctypes.data_as(C.c_void_p))
        if rc:
            raise ValueError("resize_linear")
        return out

    def pyramid_level(self, img, level):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        lw, lh = self.level_size(w, h, level)
        out = np.zeros((lh, lw), np.uint8)
        rc = self.L.orc_orb_pyramid_level(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), level,
                                          out.ctypes.data_as(C.c_void_p))
        if rc:
            raise ValueError("pyramid_level")
        return out

    def fast_nms_scores(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((h, w), np.uint8)
        self.L.orc_fast_nms_scores(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), out.ctypes.data_as(C.c_void_p))
        return out

    def harris(self, img, x, y) -> float:
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.L.orc_harris_response(img.ctypes.data_as(C.c_void_p), img.shape[1], int(x), int(y)))

    def ic_angle(self, img, x, y) -> float:
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.L.orc_ic_angle(img.ctypes.data_as(C.c_void_p), img.shape[1], int(x), int(y)))

    def fast_atan2(self, y, x) -> float:
        return float(self.L.orc_fast_atan2(float(y), float(x)))

    def gauss7_blur(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((h, w), np.uint8)
        self.L.orc_gauss7_blur_u8(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), out.ctypes.data_as(C.c_void_p))
        return out

    def descriptor(self, blurred, cx, cy, angle_deg):
        blurred = np.ascontiguousarray(blurred, np.uint8)
        out = np.zeros(32, np.uint8)
        self.L.orc_orb_descriptor(blurred.ctypes.data_as(C.c_void_p), blurred.shape[1], int(cx), int(cy),
                                  float(angle_deg), out.ctypes.data_as(C.c_void_p))
        return out

    def detect(self, img, nfeatures=400):
        """Media::makeKeyPoints"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = 4096
        while True:
            out = np.zeros(cap, KP_DTYPE)
            n = self.L.orc_orb_detect(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), int(nfeatures),
                                      out.ctypes.data_as(C.c_void_p), cap)
            if n < 0:
                raise ValueError(f"orc_orb_detect rc={n}")
            if n <= cap:
                return out[:n].copy()
            cap = int(n)

    def compute(self, img, kps):
        """Media::makeKeyPointDescriptors: returns (keypoints as the call leaves them, descriptors uint8[n, 32])"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        kps = np.ascontiguousarray(kps, KP_DTYPE).copy()
        desc = np.zeros((max(1, len(kps)), 32), np.uint8)
        n = self.L.orc_orb_compute(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w), kps.ctypes.data_as(C.c_void_p),
                                   len(kps), desc.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise ValueError(f"orc_orb_compute rc={n}")
        return kps[:n].copy(), desc[:n].copy()


class ColorCreateOracle:
    """oracle/colordesc_oracle.c: ColorDescriptor::create (cvutil.cpp:790-1099), parity unpinned"""

    def __init__(self) -> None:
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        L.orc_cv_cbrt.restype = C.c_float
        L.orc_cv_cbrt.argtypes = [C.c_float]
        L.orc_cd_bgr2luv.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_color_descriptor_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p,
                                                  C.c_void_p]
        L.orc_cd_kmeans.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]

    def cbrt(self, x) -> float:
        return float(self.L.orc_cv_cbrt(float(x)))

    def resized_dims(self, w, h):
        ow, oh = C.c_int(0), C.c_int(0)
        self.L.orc_cd_resized_dims(int(w), int(h), C.byref(ow), C.byref(oh))
        return ow.value, oh.value

    def ellipse_mask(self, cols, rows):
        m = np.zeros((rows, cols), np.uint8)
        self.L.orc_cd_ellipse_mask(int(cols), int(rows), m.ctypes.data_as(C.c_void_p))
        return m

    def bgr2luv(self, b, g, r):
        out = np.zeros(3, np.float32)
        self.L.orc_cd_bgr2luv(float(b), float(g), float(r), out.ctypes.data_as(C.c_void_p))
        return out

    def tables(self):
        g, c = np.zeros(4096, np.float32), np.zeros(4096, np.float32)
        self.L.orc_cd_tables(g.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        return g, c

    def kmeans(self, samples):
        samples = np.ascontiguousarray(samples, np.float32).reshape(-1, 3)
        labels = np.zeros(len(samples), np.int32)
        centers = np.zeros((32, 3), np.float32)
        it = self.L.orc_cd_kmeans(samples.ctypes.data_as(C.c_void_p), len(samples), labels.ctypes.data_as(C.c_void_p),
                                  centers.ctypes.data_as(C.c_void_p))
        return labels, centers, it

    def create(self, img):
        """img: uint8 [h, w, 3 or 4] (BGR / BGRA).  Returns (258-byte descriptor or None when the reference leaves
        the descriptor untouched, stage = (cols, rows, samples, kmeans iterations))"""
        img = np.ascontiguousarray(img, np.uint8)
        h, w, ch = img.shape
        desc = np.zeros(258, np.uint8)
        stage = np.zeros(4, np.int32)
        rc = self.L.orc_color_descriptor_create(img.ctypes.data_as(C.c_void_p), w, h, C.c_size_t(w * ch), ch,
                                                desc.ctypes.data_as(C.c_void_p), stage.ctypes.data_as(C.c_void_p))
        if rc < 0:
            raise ValueError(f"orc_color_descriptor_create rc={rc}")
        return (None if rc == 1 else desc), tuple(int(v) for v in stage)

"""ctypes binding of libcbird_hip.so (include/cbird_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device
is usable, the compute entry points raise.  ``lib()`` only loads and declares symbols, so it
also works on a machine without a GPU (the "-m 'not gpu'" tests check that every symbol the
header declares is exported).
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CBH_LIB_PATH: a development aid -- another build of the library, e.g. for a same-box A/B of two kernel variants)
LIB_PATH = os.environ.get("CBH_LIB_PATH") or os.path.join(_HERE, "libcbird_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cbird_hip.h")

CBH_OK = 0
CBH_E_INVAL = -1
CBH_E_UNSUPPORTED = -2
CBH_E_NODEVICE = -3
CBH_E_NOMEM = -4
CBH_E_HIP = -5
CBH_E_OVERFLOW = -6
CBH_E_NOTLOADED = -7
CBH_MAX_QUERIES_PER_CALL = 1 << 25


class CbhError(RuntimeError):
    def __init__(self, code: int, what: str = "") -> None:
        self.code = code
        L = _state.get("lib")
        msg = L.cbh_strerror(code).decode() if L is not None else str(code)
        detail = L.cbh_last_error().decode() if (L is not None and code in (CBH_E_HIP, CBH_E_NOMEM)) else ""
        super().__init__(f"{what}: {msg} ({code}) {detail}".strip())


class cbh_stats(C.Structure):
    _fields_ = [("scan_launches", C.c_uint64), ("scan_pairs", C.c_uint64), ("scan_ms", C.c_double)]


class cbh_shard_stats(C.Structure):
    _fields_ = [("shards", C.c_uint32), ("devices", C.c_uint32), ("device_mask", C.c_uint32),
                ("segments", C.c_uint64), ("scans", C.c_uint64), ("rescans", C.c_uint64), ("collectives", C.c_uint64),
                ("peer_copies", C.c_uint64), ("local_copies", C.c_uint64),
                ("collective_fallbacks", C.c_uint64)]


class cbh_filter_params(C.Structure):
    _fields_ = [("min_matches", C.c_int), ("filter_groups", C.c_int), ("filter_parent", C.c_int), ("path_mode", C.c_int),
                ("merge_groups", C.c_int), ("expand_groups", C.c_int)]


class cbh_vmatch(C.Structure):
    _fields_ = [("id", C.c_uint32), ("score", C.c_int32), ("src_in", C.c_int32), ("dst_in", C.c_int32),
                ("len", C.c_int32)]


class cbh_match(C.Structure):
    _fields_ = [("id", C.c_uint32), ("score", C.c_int32)]


_state: dict = {"lib": None}

_vp = C.c_void_p
_sz = C.c_size_t
_SIGS = {
    "cbh_version": (C.c_int, []),
    "cbh_device_count": (C.c_int, []),
    "cbh_strerror": (C.c_char_p, [C.c_int]),
    "cbh_last_error": (C.c_char_p, []),
    "cbh_trim": (C.c_int, [C.c_int, C.POINTER(C.c_ulonglong)]),
    "cbh_dcthash_batch": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, _vp, C.c_int]),
    "cbh_dcthash_batch_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, _vp, C.c_int, _vp]),
    "cbh_dcthash_tiles_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, _vp, _vp, C.c_int, _vp]),
    "cbh_keypoint_rects": (C.c_longlong, [C.c_int, C.c_int, _vp, _sz, _vp]),
    "cbh_keypoint_hashes": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "cbh_dcthash_rects": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int]),
    "cbh_keypoint_hashes_dev": (C.c_int, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "cbh_color_descriptor_dims": (None, [C.c_int, C.c_int, _vp, _vp]),
    "cbh_color_ellipse_mask": (C.c_int, [C.c_int, C.c_int, _vp]),
    "cbh_color_descriptors": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int]),
    "cbh_color_descriptors_dev": (C.c_int, [_vp, _sz, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "cbh_index_images": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, C.c_int]),
    "cbh_orb_set_pattern": (C.c_int, [_vp]),
    "cbh_orb_retain_best_dev": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "cbh_orb": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "cbh_orb_describe": (C.c_int, [_vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "cbh_orb_dev": (C.c_int, [_vp, _sz, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "cbh_longest_side_dims": (None, [C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "cbh_size_longest_side": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, _vp, _vp, _vp, C.c_int]),
    "cbh_resize_lanczos4_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "cbh_bgr2gray_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, _vp, C.c_int, _vp]),
    "cbh_autocrop_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, _vp, C.c_int, _vp]),
    "cbh_process_images": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp, _vp, C.c_int]),
    "cbh_process_images_ex": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp,
                                        _vp, C.c_int]),
    "cbh_idx64_create": (_vp, [C.c_int]),
    "cbh_idx64_create_sharded": (_vp, [C.c_uint32, C.c_int]),
    "cbh_idx64_device_mask": (C.c_uint32, [_vp]),
    "cbh_idx64_shards_per_device": (C.c_int, [_vp]),
    "cbh_idx64_shard_count": (C.c_int, [_vp]),
    "cbh_idx64_shard": (_vp, [_vp, C.c_int]),
    "cbh_idx64_shard_stats": (C.c_int, [_vp, C.POINTER(cbh_shard_stats)]),
    "cbh_vidx_create_sharded": (_vp, [C.c_uint32, C.c_int]),
    "cbh_idx64_destroy": (None, [_vp]),
    "cbh_idx64_load": (C.c_int, [_vp, _vp, _vp, _sz]),
    "cbh_idx64_load_dev": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "cbh_idx64_is_loaded": (C.c_int, [_vp]),
    "cbh_idx64_add": (C.c_int, [_vp, _vp, _vp, _sz]),
    "cbh_idx64_remove": (C.c_int, [_vp, _vp, _sz]),
    "cbh_idx64_count": (_sz, [_vp]),
    "cbh_idx64_memory_usage": (_sz, [_vp]),
    "cbh_idx64_media_ids": (C.c_int, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "cbh_idx64_slice": (_vp, [_vp, _vp, _sz]),
    "cbh_idx64_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "cbh_idx64_find": (C.c_int, [_vp, C.c_uint64, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_idx64_find_batch": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp]),
    "cbh_idx64_find_batch_masked": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp]),
    "cbh_idx64_tree_masks": (C.c_int, [_vp, _vp, _sz, _vp]),
    "cbh_idx64_find_batch_dev": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp,
                                           C.POINTER(C.c_uint64), _vp]),
    "cbh_idx64_scan_dev": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, _vp, _vp]),
    "cbh_sort_records_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp]),
    "cbh_select_records_dev": (C.c_int, [_vp, _sz, _sz, C.c_int, _vp, _vp, C.c_int, _vp]),
    "cbh_idx64_find_coalesced": (C.c_int, [_vp, C.c_uint64, C.c_int, _vp, _sz, _vp]),
    "cbh_idx64_coalesce_stats": (C.c_int, [_vp, _vp]),
    "cbh_idx64_coalesce_set_self_join": (C.c_int, [_vp, C.c_int]),
    "cbh_idx256_knn_media": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "cbh_cvfeatures_score": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, _vp, _sz, _vp]),
    "cbh_color_distances": (C.c_int, [_vp, _vp, _sz, _vp]),
    "cbh_search_index_batch": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _sz,
                                         _vp, _vp]),
    "cbh_filter_groups": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, C.c_int, _vp, _vp, _sz, _vp, _vp]),
    "cbh_fdct_search_index_batch": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              _vp, _sz, _vp, _vp]),
    "cbh_vidx_search_index_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, _vp, _sz, _vp, _vp]),
    "cbh_idx256_search_index_batch": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, _vp, _sz, _vp, _vp]),
    "cbh_color_search_index_batch": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp, _vp]),
    "cbh_fdct_find_coalesced": (C.c_int, [_vp, _vp, _sz, C.c_uint32, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_vidx_find_video_coalesced": (C.c_int, [_vp, _vp, _vp, _sz, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_idx256_find_coalesced": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_color_find_coalesced": (C.c_int, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "cbh_color_find_all_batch": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "cbh_combine_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cbh_filter_groups_ex": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp,
                                       _vp]),
    "cbh_vdx_verify": (C.c_int, [_vp, _sz]),
    "cbh_records_topk_dev": (C.c_int, [_vp, _sz, _sz, _sz, _sz, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "cbh_idx64_set_record_capacity": (C.c_int, [_vp, _sz]),
    "cbh_idx64_remove_ids_only": (C.c_int, [_vp, _vp, _sz]),
    "cbh_idx64_hashes_for_id": (C.c_int, [_vp, C.c_uint32, _vp, _sz, C.POINTER(_sz)]),
    "cbh_fdct_find": (C.c_int, [_vp, _vp, _sz, C.c_uint32, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_fdct_find_batch": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, _vp, _sz, _vp]),
    "cbh_fdct_find_ex": (C.c_int, [_vp, _vp, _sz, C.c_uint32, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_fdct_find_batch_ex": (C.c_int, [_vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "cbh_vidx_create": (_vp, [C.c_int]),
    "cbh_vidx_destroy": (None, [_vp]),
    "cbh_vidx_set_radix": (C.c_int, [_vp, C.c_int]),
    "cbh_vidx_add_video": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _sz]),
    "cbh_vidx_remove": (C.c_int, [_vp, _vp, _sz]),
    "cbh_vidx_count": (_sz, [_vp]),
    "cbh_vidx_memory_usage": (_sz, [_vp]),
    "cbh_vidx_entries": (_sz, [_vp, C.c_int]),
    "cbh_vidx_find_frame": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_vidx_find_video": (C.c_int, [_vp, _vp, _vp, _sz, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_vidx_find_videos_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, _vp, _sz, _vp]),
    "cbh_vdx_encode": (_sz, [_vp, _vp, _sz, C.c_char_p, _vp, _sz]),
    "cbh_vdx_decode": (C.c_longlong, [_vp, _sz, _vp, _vp, _sz]),
    "cbh_vdx_version": (C.c_int, [_vp, _sz]),
    "cbh_vdx_encode_v1": (_sz, [_vp, _vp, _sz, _vp, _sz]),
    "cbh_video_dedup": (_sz, [_vp, _sz, C.c_int, _vp]),
    "cbh_template_scores": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, _vp, _sz, C.c_int, _vp, _vp, _vp,
                                      C.c_int]),
    "cbh_template_hashes_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, C.c_int, _vp, _sz, C.c_int, _vp, _vp,
                                          C.c_int, _vp]),
    "cbh_vindexer_create": (_vp, [C.c_int, C.c_int, C.c_int]),
    "cbh_vindexer_destroy": (None, [_vp]),
    "cbh_vindexer_resume": (C.c_int, [_vp, _vp, _vp, _sz]),
    "cbh_vindexer_push": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _sz, _sz]),
    "cbh_vindexer_push_dev": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _sz, _sz]),
    "cbh_vindexer_frames_seen": (C.c_longlong, [_vp]),
    "cbh_vindexer_finish": (C.c_longlong, [_vp, _vp, _vp, _sz]),
    "cbh_idx256_create": (_vp, [C.c_int]),
    "cbh_idx256_create_sharded": (_vp, [C.c_uint32, C.c_int]),
    "cbh_idx256_shard_count": (C.c_int, [_vp]),
    "cbh_idx256_shard_rows": (_sz, [_vp, C.c_int]),
    "cbh_idx256_shard_stats": (C.c_int, [_vp, C.POINTER(cbh_shard_stats)]),
    "cbh_idx256_destroy": (None, [_vp]),
    "cbh_idx256_add": (C.c_int, [_vp, C.c_uint32, _vp, _sz]),
    "cbh_idx256_remove": (C.c_int, [_vp, _vp, _sz]),
    "cbh_idx256_is_loaded": (C.c_int, [_vp]),
    "cbh_idx256_count": (_sz, [_vp]),
    "cbh_idx256_memory_usage": (_sz, [_vp]),
    "cbh_idx256_rows_of": (C.c_int, [_vp, C.c_uint32, C.POINTER(_sz), C.POINTER(_sz)]),
    "cbh_idx256_download_rows": (C.c_int, [_vp, _sz, _sz, _vp]),
    "cbh_idx256_knn": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _vp, _vp]),
    "cbh_idx256_find": (C.c_int, [_vp, _vp, _sz, C.c_int, C.c_int, _vp, _sz, C.POINTER(_sz)]),
    "cbh_idx256_radius_match": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, _vp]),
    "cbh_idx256_find_batch": (C.c_int, [_vp, _vp, _vp, _sz, C.c_int, C.c_int, _vp, _sz, _vp]),
    "cbh_idx256_get_stats": (C.c_int, [_vp, C.POINTER(cbh_stats)]),
    "cbh_color_create": (_vp, [C.c_int]),
    "cbh_color_destroy": (None, [_vp]),
    "cbh_color_add": (C.c_int, [_vp, _vp, _vp, _sz]),
    "cbh_color_remove": (C.c_int, [_vp, _vp, _sz]),
    "cbh_color_count": (_sz, [_vp]),
    "cbh_color_is_loaded": (C.c_int, [_vp]),
    "cbh_color_memory_usage": (_sz, [_vp]),
    "cbh_color_find_index_data": (C.c_int, [_vp, C.c_uint32, _vp]),
    "cbh_color_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "cbh_color_find": (C.c_int, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "cbh_color_find_batch": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _vp]),
    "cbh_usable_device_mask": (C.c_uint32, []),
    "cbh_last_error_code": (C.c_int, []),
    "cbh_clear_error": (None, []),
    "cbh_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "cbh_get_tuning": (C.c_int, [C.c_char_p, C.POINTER(C.c_longlong)]),
    "cbh_idx64_get_stats": (C.c_int, [_vp, C.POINTER(cbh_stats)]),
    "cbh_idx64_reset_stats": (C.c_int, [_vp]),
    "cbh_idx64_time_scan_dev": (C.c_int, [_vp, _vp, _sz, C.c_int, _vp, _sz, _vp, C.c_int,
                                          C.POINTER(C.c_float)]),
    "cbh_time_dcthash_dev": (C.c_int, [_vp, _sz, C.c_int, C.c_int, _sz, _sz, _vp, C.c_int, C.c_int,
                                       C.POINTER(C.c_float)]),
    "cbh_selftest_buffer_range": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
}


def header_symbols() -> list[str]:
    """every function name include/cbird_hip.h declares"""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cbh_[a-z0-9_]+)\s*\(", txt)))


def lib() -> C.CDLL:
    """Load libcbird_hip.so (raises if it has not been built)."""
    if _state["lib"] is not None:
        return _state["lib"]
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cbird_amd/csrc` (the HIP path has no fallback)")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C-ABI itself
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _state["lib"] = L
    # CBH_TUNING="key=value,key=value": kernel-variant / allocator knobs (cbh_set_tuning) for tools and soaks
    for kv in filter(None, os.environ.get("CBH_TUNING", "").split(",")):
        k, _, v = kv.partition("=")
        if L.cbh_set_tuning(k.strip().encode(), int(v)) != CBH_OK:
            raise ValueError(f"CBH_TUNING: unknown knob {k!r}")
    return L


# ---- index shape: one device, or sharded inside the handle (cbh_idx64_create_sharded) --------------------------------
# (device_mask, shards_per_device) used by every 64-bit index constructor that is not told otherwise; None = the plain
# one-device index.  CBH_INDEX_SHARDS="mask:per_device" sets it from the environment (e.g. "0xff:1" = the 8 GPUs of a
# node, "1:8" = eight logical shards on device 0); the test-suite switches it per test (conftest.index_shape).
_state["sharding"] = None


def set_default_sharding(shape) -> None:
    _state["sharding"] = None if shape is None else (int(shape[0]), int(shape[1]))


def default_sharding():
    if _state["sharding"] is None and os.environ.get("CBH_INDEX_SHARDS"):
        m, _, k = os.environ["CBH_INDEX_SHARDS"].partition(":")
        return int(m, 0), int(k or 1)
    return _state["sharding"]


def create_idx64(device: int, shape=None):
    """cbh_idx64_create, or cbh_idx64_create_sharded when `shape` (or the process default) asks for shards"""
    shape = shape if shape is not None else default_sharding()
    L = lib()
    h = L.cbh_idx64_create_sharded(shape[0], shape[1]) if shape else L.cbh_idx64_create(device)
    if not h:
        raise CbhError(CBH_E_NODEVICE, "cbh_idx64_create" + ("_sharded" if shape else ""))
    return h


def check(code: int, what: str) -> None:
    if code != CBH_OK:
        raise CbhError(code, what)


def require_device() -> int:
    """Number of usable gfx950 devices; raises CbhError(CBH_E_NODEVICE) when there is none."""
    n = lib().cbh_device_count()
    if n <= 0:
        raise CbhError(CBH_E_NODEVICE, "cbird_amd")
    return n

"""ctypes host binding of libr3dm.so -- the C ABI in include/r3dm.h.

This is the Python face of the drop-in boundary used by tests/, bench.py and the multi-GPU
driver.  It contains NO arithmetic: every call goes through the C ABI into the HIP kernels.
If the shared library (or a gfx950 GPU) is missing the calls raise -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libr3dm.so")

F32, U8, BIN = 0, 1, 2
LAYOUT_ROWS, LAYOUT_BF16, LAYOUT_SPLIT, LAYOUT_COUNTS, LAYOUT_BIN8 = 1, 2, 4, 8, 16
NONE = 0xFFFFFFFF

EXPORTS = [
    "r3dm_create", "r3dm_destroy", "r3dm_last_error", "r3dm_device_info", "r3dm_set_image", "r3dm_set_images", "r3dm_images_wait", "r3dm_view_info", "r3dm_memory_info", "r3dm_clear_images", "r3dm_trim",
    "r3dm_match_pairs", "r3dm_filter_F", "r3dm_filter_H", "r3dm_knn2", "r3dm_graph_num_pairs", "r3dm_graph_num_matches",
    "r3dm_graph_pairs", "r3dm_graph_offsets", "r3dm_graph_matches", "r3dm_graph_free", "r3dm_graph_from_csr",
    "r3dm_graph_merge", "r3dm_save_matches", "r3dm_load_matches", "r3dm_get_stats", "r3dm_filter_report",
    "r3dm_compute_matches_dir", "r3dm_compute_matches_stage", "r3dm_stage_create", "r3dm_stage_run", "r3dm_stage_destroy", "r3dm_liop_describe_patches", "r3dm_extract_liop",
    "r3dm_set_intrinsics", "r3dm_filter_E", "r3dm_ann_params_for_algorithm", "r3dm_detect_akaze", "r3dm_detect_akaze_mldb", "r3dm_gray_from_bgr8", "r3dm_extract_features_to_files", "r3dm_multi_extract_features",
    "r3dm_detect_akaze_batch", "r3dm_extract_features_batch", "r3dm_multi_extract_features_ex", "r3dm_get_features_totals", "r3dm_kgraph_preset", "r3dm_match_pairs_kgraph", "r3dm_exhaustive_is_faster", "r3dm_kgraph_knn2", "r3dm_kgraph_index", "r3dm_drop_indices",
    "r3dm_filter_FEH", "r3dm_host_threads", "r3dm_set_features_sink", "r3dm_multi_set_features_sink", "r3dm_set_deferred_feature_files", "r3dm_set_background_nice", "r3dm_multi_set_background_nice", "r3dm_features_files_wait", "r3dm_multi_set_deferred_feature_files", "r3dm_multi_features_files_wait", "r3dm_hnsw_preset", "r3dm_match_pairs_hnsw", "r3dm_hnsw_knn2", "r3dm_hnsw_knn2_on_index", "r3dm_hnsw_index",
    "r3dm_mrpt_preset", "r3dm_match_pairs_mrpt", "r3dm_mrpt_knn2", "r3dm_mrpt_index", "r3dm_multi_match_pairs_mrpt",
    "r3dm_set_integer_mfma", "r3dm_set_split_mfma", "r3dm_set_hamming_mfma", "r3dm_index_create", "r3dm_index_knn2", "r3dm_index_destroy",
    "r3dm_multi_create", "r3dm_multi_destroy", "r3dm_multi_num_devices", "r3dm_multi_ctx", "r3dm_multi_last_error",
    "r3dm_multi_set_image", "r3dm_multi_transfer_counts", "r3dm_multi_set_intrinsics", "r3dm_multi_clear_images", "r3dm_multi_set_integer_mfma",
    "r3dm_multi_match_pairs", "r3dm_multi_match_pairs_kgraph", "r3dm_multi_match_pairs_hnsw", "r3dm_multi_filter_F", "r3dm_multi_filter_H", "r3dm_multi_filter_E", "r3dm_shard_pairs",
    "r3dm_comm_unique_id", "r3dm_comm_create", "r3dm_comm_destroy", "r3dm_comm_rank", "r3dm_comm_world", "r3dm_comm_last_error",
    "r3dm_allgather_graphs", "r3dm_graphs_pack", "r3dm_words_free", "r3dm_graphs_unpack_merge",
    "r3dm_set_device_graphs", "r3dm_graph_on_device", "r3dm_comm_last_device_graphs",
]


class R3dmError(RuntimeError):
    pass


class ViewDesc(C.Structure):
    """r3dm_view_desc (include/r3dm.h): one view of an r3dm_set_images batch"""
    _fields_ = [("view_id", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("n", C.c_uint32), ("dim", C.c_uint32),
                ("dtype", C.c_int32), ("desc", C.c_void_p), ("xy", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("ms_match_kernels", C.c_double), ("n_match_launches", C.c_uint64),
                ("ms_filter_kernels", C.c_double), ("n_pairs", C.c_uint64), ("n_queries", C.c_uint64),
                ("n_exact_fallback", C.c_uint64), ("algorithmic_flops", C.c_double),
                ("algorithmic_bytes", C.c_double), ("ms_wall_match", C.c_double),
                ("ms_wall_match_post", C.c_double), ("ms_wall_filter", C.c_double), ("ms_liop_kernel", C.c_double),
                ("ms_ann_build", C.c_double), ("ms_ann_search", C.c_double), ("n_ann_built", C.c_uint64),
                ("n_ann_dist", C.c_uint64), ("ms_detect", C.c_double), ("n_integer_mfma", C.c_uint64),
                ("n_split_mfma", C.c_uint64), ("n_views_staged", C.c_uint64),
                ("n_hamming_mfma", C.c_uint64), ("n_detect_images", C.c_uint64), ("n_ann_rows16", C.c_uint64), ("n_ann_rows8", C.c_uint64), ("n_ann_dot8", C.c_uint64),
                ("ms_detect_kernels", C.c_double), ("detect_algorithmic_bytes", C.c_double),
                ("ms_liop_wall", C.c_double), ("ms_feature_files", C.c_double),
                ("n_hnsw_launches", C.c_uint64), ("n_hnsw_retries", C.c_uint64), ("n_counts_mfma", C.c_uint64),
                ("detect_compulsory_bytes", C.c_double), ("n_filter_workgroups", C.c_uint64), ("n_filter_coop_pairs", C.c_uint64)]


class FeaturesTotals(C.Structure):
    """r3dm_features_totals: the features work of one context since its creation"""
    _fields_ = [("n_images", C.c_uint64), ("n_passes", C.c_uint64), ("n_keypoints", C.c_uint64), ("n_regrows", C.c_uint64),
                ("ms_detect_kernels", C.c_double), ("detect_algorithmic_bytes", C.c_double), ("ms_liop_kernels", C.c_double),
                ("ms_wall", C.c_double), ("ms_files", C.c_double)]


class ViewImage(C.Structure):
    """r3dm_view_image (include/r3d_compute_matches.hpp)"""
    _fields_ = [("id", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("basename", C.c_char_p),
                ("bgr8", C.c_void_p), ("gray", C.c_void_p), ("focal_px", C.c_double), ("ppx", C.c_double), ("ppy", C.c_double)]


class StageReport(C.Structure):
    """r3dm_stage_report: wall time of the phases of R3DComputeMatches::computeMatches (ms), kernel times, counts"""
    _fields_ = [(k, C.c_double) for k in ("ms_features", "ms_load", "ms_match", "ms_filter_F", "ms_filter_E", "ms_filter_H", "ms_files", "ms_total",
                                          "ms_match_kernels", "ms_F_kernels", "ms_E_kernels", "ms_H_kernels", "ms_filters_wall", "ms_match_post")] + \
               [(k, C.c_uint64) for k in ("images_extracted", "n_keypoints", "n_putative_pairs", "n_putative_matches", "n_F_pairs", "n_F_matches",
                                          "n_E_pairs", "n_E_matches", "n_H_pairs", "n_H_matches", "match_was_exhaustive")] + [("features", FeaturesTotals)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "features"}
        d["features"] = {k: getattr(self.features, k) for k, _ in FeaturesTotals._fields_}
        return d


def _stage_views(views, keep):
    arr = (ViewImage * max(len(views), 1))()
    for k, v in enumerate(views):
        def ptr(a, dt):
            if a is None:
                return None
            if not hasattr(a, "data_ptr"):
                a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data
        arr[k] = ViewImage(int(v["id"]), int(v["width"]), int(v["height"]), v["basename"].encode(), ptr(v.get("bgr"), np.uint8),
                           ptr(v.get("gray"), np.float32), float(v.get("focal_px", -1.0)), float(v.get("ppx", 0.0)), float(v.get("ppy", 0.0)))
    return arr


_STAGE_ARGS = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
               C.c_uint32, C.c_void_p, C.c_char_p, C.c_size_t]


class Stage:
    """r3dm_stage_*: the R3DComputeMatches facade kept alive between calls (contexts, detector work buffers and page-locked memory
    are allocated once), as a long-lived host process would keep it."""

    def __init__(self, device_ids):
        L = load_library()
        L.r3dm_stage_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.r3dm_stage_run.argtypes = [C.c_void_p] + _STAGE_ARGS
        L.r3dm_stage_destroy.argtypes = [C.c_void_p]
        L.r3dm_stage_destroy.restype = None
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = L.r3dm_stage_create(ids, len(device_ids), C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_stage_create({list(device_ids)}) -> {rc} (no gfx950 GPU visible? there is no CPU fallback)")
        self._h, self._L = h.value, L

    def run(self, matches_dir: str, views, threshold: float = 0.001, dist_ratio: float = 0.6, matching_algorithm: int = 9, compute_F: bool = True,
            compute_E: bool = True, compute_H: bool = True, seed: int = 5489, batches_in_flight: int = 3, images_per_batch: int = 8,
            arms_as_requested: bool = False, split_mfma: bool = False, integer_mfma: bool = False, f32_tiles: bool = False, background_nice: bool = False) -> StageReport:
        keep = []
        arr = _stage_views(views, keep)
        rep = StageReport(); err = C.create_string_buffer(1024)
        rc = self._L.r3dm_stage_run(self._h, matches_dir.encode(), arr, len(views), threshold, dist_ratio, matching_algorithm, int(compute_F),
                                    int(compute_E), int(compute_H), seed, batches_in_flight, images_per_batch,
                                    (1 if arms_as_requested else 0) | (2 if split_mfma else 0) | (4 if integer_mfma else 0) | (8 if f32_tiles else 0) | (16 if background_nice else 0), C.byref(rep), err, 1024)
        if rc != 0:
            raise R3dmError(f"r3dm_stage_run -> {rc}: {err.value.decode()}")
        return rep

    def close(self):
        if getattr(self, "_h", None):
            self._L.r3dm_stage_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compute_matches_stage(device_ids, matches_dir: str, views, threshold: float = 0.001, dist_ratio: float = 0.6,
                          matching_algorithm: int = 9, compute_F: bool = True, compute_E: bool = True, compute_H: bool = True,
                          seed: int = 5489, batches_in_flight: int = 3, images_per_batch: int = 8, arms_as_requested: bool = False,
                          split_mfma: bool = False, integer_mfma: bool = False, f32_tiles: bool = False, background_nice: bool = False) -> StageReport:
    """R3DComputeMatches::computeMatches from pixels (r3dm_compute_matches_stage): features stage for the views whose .feat/.desc
    are missing, matching, F / E / H filters, match files.  views: dicts with id, width, height, basename and optionally
    gray ([h, w] float32) or bgr ([h, w, 3] uint8) -- numpy or torch (host or device) -- and focal_px / ppx / ppy."""
    L = load_library()
    keep = []
    arr = _stage_views(views, keep)
    ids = (C.c_int * len(device_ids))(*device_ids)
    rep = StageReport(); err = C.create_string_buffer(1024)
    L.r3dm_compute_matches_stage.argtypes = [C.c_void_p, C.c_int] + _STAGE_ARGS
    rc = L.r3dm_compute_matches_stage(ids, len(device_ids), matches_dir.encode(), arr, len(views), threshold, dist_ratio, matching_algorithm,
                                      int(compute_F), int(compute_E), int(compute_H), seed, batches_in_flight, images_per_batch,
                                      (1 if arms_as_requested else 0) | (2 if split_mfma else 0) | (4 if integer_mfma else 0) | (8 if f32_tiles else 0) | (16 if background_nice else 0), C.byref(rep), err, 1024)
    if rc != 0:
        raise R3dmError(f"r3dm_compute_matches_stage -> {rc}: {err.value.decode()}")
    return rep


class KGraphParams(C.Structure):
    """r3dm_kgraph_params: index_K forward neighbours per row, search_P start rows, search_S neighbours per step."""
    _fields_ = [("index_K", C.c_uint32), ("search_P", C.c_uint32), ("search_S", C.c_uint32), ("reserved", C.c_uint32),
                ("seed", C.c_uint64)]

    @staticmethod
    def preset(which) -> "KGraphParams":
        """which: 0 / "fast", 1 / "medium", 2 / "precise", anything else the reference's default block"""
        code = {"fast": 0, "medium": 1, "precise": 2}.get(which, which if isinstance(which, int) else 3)
        kp = KGraphParams()
        if load_library().r3dm_kgraph_preset(int(code), C.byref(kp)) != 0:
            raise R3dmError("r3dm_kgraph_preset failed")
        return kp


class HnswParams(C.Structure):
    """r3dm_hnsw_params: M links per row (2M on layer 0), ef_construction (unused by the batch build), ef search beam, seed of the level draw"""
    _fields_ = [("M", C.c_uint32), ("ef_construction", C.c_uint32), ("ef", C.c_uint32), ("seed", C.c_uint32)]

    @staticmethod
    def preset(which) -> "HnswParams":
        """which: 0 / "fast", 1 / "medium", 2 / "precise" (matchingAlgorithm 6 / 7 / 8 of the reference)"""
        code = {"fast": 0, "medium": 1, "precise": 2}.get(which, which if isinstance(which, int) else 2)
        hp = HnswParams()
        if load_library().r3dm_hnsw_preset(int(code), C.byref(hp)) != 0:
            raise R3dmError("r3dm_hnsw_preset failed")
        return hp


class MrptParams(C.Structure):
    """r3dm_mrpt_params: n_trees, depth (clamped per view), votes, density (<= 0: 1 / sqrt(dim)), seed of the random vectors"""
    _fields_ = [("n_trees", C.c_uint32), ("depth", C.c_uint32), ("votes", C.c_uint32), ("density", C.c_float), ("seed", C.c_uint64)]

    @staticmethod
    def preset() -> "MrptParams":
        mp = MrptParams()
        if load_library().r3dm_mrpt_preset(C.byref(mp)) != 0:
            raise R3dmError("r3dm_mrpt_preset failed")
        return mp


class HnswArrays(C.Structure):
    """r3dm_hnsw_arrays: an HNSW index in hnswlib's own shape"""
    _fields_ = [("M", C.c_uint32), ("links0", C.c_void_p), ("up_off", C.c_void_p), ("up_links", C.c_void_p), ("up_rows", C.c_uint32),
                ("enterpoint", C.c_int32), ("maxlevel", C.c_int32)]


class PairReport(C.Structure):
    _fields_ = [("threshold_px", C.c_double), ("nfa", C.c_double), ("iterations", C.c_uint32),
                ("models", C.c_uint32), ("inliers", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None
DEV_LIB_PATH = os.path.join(_HERE, "libr3dm_dev.so")


def use_developer_library():
    """tools/ and the fallback-path tests: load the developer build (build.sh dev: -DR3DM_DEVTOOLS, the A/B kernel variants,
    traces and test hooks that read R3DM_* environment variables) instead of the product library.  Must be called before the
    first load_library(); the product library itself never reads the environment."""
    global LIB_PATH
    if _lib is not None:
        raise R3dmError("the library is already loaded")
    LIB_PATH = DEV_LIB_PATH


def use_library(path: str):
    """load a specific build of the library (bisecting tools); before the first load_library()"""
    global LIB_PATH
    if _lib is not None:
        raise R3dmError("the library is already loaded")
    LIB_PATH = path


def load_library():
    """dlopen libr3dm.so and declare prototypes.  Raises if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise R3dmError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.r3dm_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.r3dm_destroy.argtypes = [vp]; L.r3dm_destroy.restype = None
    L.r3dm_last_error.argtypes = [vp]; L.r3dm_last_error.restype = C.c_char_p
    L.r3dm_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(u64)]
    L.r3dm_set_image.argtypes = [vp, u32, u32, u32, vp, u32, u32, C.c_int, vp]
    L.r3dm_set_images.argtypes = [vp, vp, u32]
    L.r3dm_images_wait.argtypes = [vp]
    L.r3dm_view_info.argtypes = [vp, u32, vp, vp, vp, vp]
    L.r3dm_memory_info.argtypes = [vp, vp, vp, vp]
    L.r3dm_clear_images.argtypes = [vp]
    L.r3dm_trim.argtypes = [vp]
    L.r3dm_set_integer_mfma.argtypes = [vp, C.c_int]
    L.r3dm_set_split_mfma.argtypes = [vp, C.c_int]
    L.r3dm_set_hamming_mfma.argtypes = [vp, C.c_int]
    L.r3dm_index_create.argtypes = [vp, vp, u32, u32, C.c_int, C.POINTER(vp)]
    L.r3dm_index_knn2.argtypes = [vp, vp, vp, u32, vp, vp]
    L.r3dm_index_destroy.argtypes = [vp]; L.r3dm_index_destroy.restype = None
    L.r3dm_match_pairs.argtypes = [vp, vp, u64, C.c_float, C.c_int, C.POINTER(vp)]
    L.r3dm_filter_F.argtypes = [vp, vp, C.c_double, u32, u64, C.c_int, C.POINTER(vp), vp]
    L.r3dm_filter_H.argtypes = [vp, vp, C.c_double, u32, u64, C.POINTER(vp), vp]
    L.r3dm_filter_E.argtypes = [vp, vp, C.c_double, u32, u64, u32, C.c_float, C.POINTER(vp), vp]
    L.r3dm_filter_FEH.argtypes = [vp, vp, C.c_double, u32, u64, C.c_int, u32, C.c_float, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp]
    L.r3dm_set_intrinsics.argtypes = [vp, u32, vp]
    L.r3dm_liop_describe_patches.argtypes = [vp, vp, u32, u32, vp, C.POINTER(u32)]
    L.r3dm_extract_liop.argtypes = [vp, vp, u32, u32, vp, u32, C.c_float, vp, vp]
    L.r3dm_knn2.argtypes = [vp, vp, u32, vp, u32, u32, C.c_int, vp, vp]
    L.r3dm_detect_akaze.argtypes = [vp, vp, u32, u32, C.c_float, vp, vp, u32, C.POINTER(u32)]
    L.r3dm_detect_akaze_mldb.argtypes = [vp, vp, u32, u32, C.c_float, vp, vp, u32, C.POINTER(u32)]
    L.r3dm_gray_from_bgr8.argtypes = [vp, vp, u32, u32, vp]
    L.r3dm_extract_features_to_files.argtypes = [vp, vp, u32, u32, C.c_float, C.c_char_p, C.c_char_p, C.POINTER(u32)]
    L.r3dm_multi_extract_features.argtypes = [vp, u32, vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_char_p, C.c_size_t]
    L.r3dm_multi_extract_features_ex.argtypes = [vp, u32, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, u32, C.c_char_p, C.c_size_t]
    L.r3dm_get_features_totals.argtypes = [vp, vp]
    L.r3dm_detect_akaze_batch.argtypes = [vp, u32, vp, u32, u32, C.c_float, vp, vp, u32, vp]
    L.r3dm_extract_features_batch.argtypes = [vp, u32, vp, vp, u32, u32, C.c_float, vp, vp, vp]
    L.r3dm_kgraph_preset.argtypes = [C.c_int, vp]
    L.r3dm_ann_params_for_algorithm.argtypes = [C.c_int, vp]
    L.r3dm_match_pairs_kgraph.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(vp)]
    L.r3dm_kgraph_knn2.argtypes = [vp, vp, u32, vp, u32, u32, vp, u32, u32, vp, vp]
    L.r3dm_kgraph_index.argtypes = [vp, u32, u32, vp, vp]
    L.r3dm_hnsw_preset.argtypes = [C.c_int, vp]
    L.r3dm_mrpt_preset.argtypes = [vp]
    L.r3dm_match_pairs_mrpt.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(vp)]
    L.r3dm_mrpt_knn2.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.r3dm_mrpt_index.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, vp]
    L.r3dm_multi_match_pairs_mrpt.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(vp)]
    L.r3dm_match_pairs_hnsw.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(vp)]
    L.r3dm_hnsw_knn2.argtypes = [vp, vp, u32, vp, u32, u32, vp, vp, vp]
    L.r3dm_hnsw_knn2_on_index.argtypes = [vp, vp, u32, u32, vp, vp, u32, u32, vp, vp]
    L.r3dm_hnsw_index.argtypes = [vp, u32, vp, vp, vp, vp, u32, vp, vp, vp]
    L.r3dm_drop_indices.argtypes = [vp]
    L.r3dm_graph_num_pairs.argtypes = [vp]; L.r3dm_graph_num_pairs.restype = u64
    L.r3dm_graph_num_matches.argtypes = [vp]; L.r3dm_graph_num_matches.restype = u64
    L.r3dm_graph_pairs.argtypes = [vp]; L.r3dm_graph_pairs.restype = vp
    L.r3dm_graph_offsets.argtypes = [vp]; L.r3dm_graph_offsets.restype = vp
    L.r3dm_graph_matches.argtypes = [vp]; L.r3dm_graph_matches.restype = vp
    L.r3dm_graph_free.argtypes = [vp]; L.r3dm_graph_free.restype = None
    L.r3dm_graph_from_csr.argtypes = [vp, u64, vp, vp, C.POINTER(vp)]
    L.r3dm_graph_merge.argtypes = [vp, u32, C.POINTER(vp)]
    L.r3dm_comm_unique_id.argtypes = [vp]
    L.r3dm_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.r3dm_comm_destroy.argtypes = [vp]; L.r3dm_comm_destroy.restype = None
    L.r3dm_comm_rank.argtypes = [vp]; L.r3dm_comm_world.argtypes = [vp]
    L.r3dm_comm_last_error.argtypes = [vp]; L.r3dm_comm_last_error.restype = C.c_char_p
    L.r3dm_allgather_graphs.argtypes = [vp, vp, u32, vp]
    L.r3dm_set_device_graphs.argtypes = [vp, C.c_int]
    L.r3dm_graph_on_device.argtypes = [vp]
    L.r3dm_set_deferred_feature_files.argtypes = [vp, C.c_int]
    L.r3dm_features_files_wait.argtypes = [vp]
    L.r3dm_multi_set_deferred_feature_files.argtypes = [vp, C.c_int]
    L.r3dm_multi_features_files_wait.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.r3dm_comm_last_device_graphs.argtypes = [vp]
    L.r3dm_graphs_pack.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
    L.r3dm_words_free.argtypes = [vp]; L.r3dm_words_free.restype = None
    L.r3dm_graphs_unpack_merge.argtypes = [vp, vp, u32, u32, vp]
    L.r3dm_save_matches.argtypes = [vp, C.c_char_p]
    L.r3dm_load_matches.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.r3dm_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.r3dm_filter_report.argtypes = [vp, vp, u64]
    L.r3dm_multi_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.r3dm_multi_destroy.argtypes = [vp]; L.r3dm_multi_destroy.restype = None
    L.r3dm_multi_num_devices.argtypes = [vp]
    L.r3dm_multi_ctx.argtypes = [vp, C.c_int]; L.r3dm_multi_ctx.restype = vp
    L.r3dm_multi_last_error.argtypes = [vp]; L.r3dm_multi_last_error.restype = C.c_char_p
    L.r3dm_multi_set_image.argtypes = [vp, u32, u32, u32, vp, u32, u32, C.c_int, vp]
    L.r3dm_multi_set_intrinsics.argtypes = [vp, u32, vp]
    L.r3dm_multi_clear_images.argtypes = [vp]
    L.r3dm_multi_set_integer_mfma.argtypes = [vp, C.c_int]
    L.r3dm_multi_match_pairs.argtypes = [vp, vp, u64, C.c_float, C.c_int, C.POINTER(vp)]
    L.r3dm_multi_match_pairs_kgraph.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(vp)]
    L.r3dm_multi_match_pairs_hnsw.argtypes = [vp, vp, u64, C.c_float, vp, C.POINTER(vp)]
    L.r3dm_multi_filter_F.argtypes = [vp, vp, C.c_double, u32, u64, C.POINTER(vp), vp]
    L.r3dm_multi_filter_H.argtypes = [vp, vp, C.c_double, u32, u64, C.POINTER(vp), vp]
    L.r3dm_multi_filter_E.argtypes = [vp, vp, C.c_double, u32, u64, u32, C.c_float, C.POINTER(vp), vp]
    L.r3dm_shard_pairs.argtypes = [vp, u64, u32, vp]
    _lib = L
    return L


def _ptr(a) -> Optional[int]:
    """address of a numpy array or torch tensor (host or device) -- the ABI takes plain pointers"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a.data_ptr())           # torch tensor


class Graph:
    """PairWiseMatches: pairs [P,2] u32 ordered by (I,J); offsets [P+1] u64; matches [M,2] u32 (i_, j_)."""

    def __init__(self, handle: int):
        self._h = handle

    @property
    def on_device(self) -> int:
        """device id of the graph's device mirror, -1 without one"""
        return load_library().r3dm_graph_on_device(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load_library().r3dm_graph_free(self._h)
                self._h = None
        except Exception:          # interpreter shutdown
            pass

    @property
    def num_pairs(self) -> int:
        return int(load_library().r3dm_graph_num_pairs(self._h))

    @property
    def num_matches(self) -> int:
        return int(load_library().r3dm_graph_num_matches(self._h))

    def _view(self, fn, n, dtype):
        if n == 0:
            return np.zeros(0, dtype)
        addr = fn(self._h)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=dtype).copy()

    @property
    def pairs(self) -> np.ndarray:
        return self._view(load_library().r3dm_graph_pairs, 2 * self.num_pairs, np.uint32).reshape(-1, 2)

    @property
    def offsets(self) -> np.ndarray:
        if self.num_pairs == 0:
            return np.zeros(1, np.uint64)
        return self._view(load_library().r3dm_graph_offsets, self.num_pairs + 1, np.uint64)

    @property
    def matches(self) -> np.ndarray:
        return self._view(load_library().r3dm_graph_matches, 2 * self.num_matches, np.uint32).reshape(-1, 2)

    def as_dict(self):
        """{(I, J): ndarray [m, 2]} -- the std::map view used by the parity tests"""
        p, o, m = self.pairs, self.offsets, self.matches
        return {(int(p[k, 0]), int(p[k, 1])): m[int(o[k]):int(o[k + 1])] for k in range(p.shape[0])}

    def save(self, path: str) -> None:
        rc = load_library().r3dm_save_matches(self._h, path.encode())
        if rc != 0:
            raise R3dmError(f"r3dm_save_matches({path}) -> {rc}")

    @staticmethod
    def load(path: str) -> "Graph":
        h = C.c_void_p()
        rc = load_library().r3dm_load_matches(path.encode(), C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_load_matches({path}) -> {rc}")
        return Graph(h.value)

    @staticmethod
    def from_csr(pairs: np.ndarray, offsets: np.ndarray, matches: np.ndarray) -> "Graph":
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        matches = np.ascontiguousarray(matches, np.uint32).reshape(-1, 2)
        h = C.c_void_p()
        rc = load_library().r3dm_graph_from_csr(_ptr(pairs) if pairs.size else None, pairs.shape[0], _ptr(offsets),
                                                 _ptr(matches) if matches.size else None, C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_graph_from_csr -> {rc}")
        return Graph(h.value)

    @staticmethod
    def merge(parts: Sequence["Graph"]) -> "Graph":
        arr = (C.c_void_p * len(parts))(*[p._h for p in parts])
        h = C.c_void_p()
        rc = load_library().r3dm_graph_merge(arr, len(parts), C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_graph_merge -> {rc}")
        return Graph(h.value)


def graphs_pack(graphs: Sequence["Graph"]) -> np.ndarray:
    """r3dm_graphs_pack: the wire format of a rank's graphs (uint32 words), for a transport of the caller's own"""
    L = load_library()
    arr = (C.c_void_p * len(graphs))(*[g._h for g in graphs])
    words = C.c_void_p(); n = C.c_uint64()
    rc = L.r3dm_graphs_pack(arr, len(graphs), C.byref(words), C.byref(n))
    if rc != 0:
        raise R3dmError(f"r3dm_graphs_pack -> {rc}")
    try:
        return np.ctypeslib.as_array(C.cast(words, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()
    finally:
        L.r3dm_words_free(words)


def graphs_unpack_merge(rank_words: Sequence[np.ndarray], n_graphs: int) -> List["Graph"]:
    """r3dm_graphs_unpack_merge: the packed graphs of every rank -> the graphs of the whole collection, ordered by (I, J)"""
    L = load_library()
    bufs = [np.ascontiguousarray(w, np.uint32) for w in rank_words]
    ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    lens = (C.c_uint64 * len(bufs))(*[b.size for b in bufs])
    out = (C.c_void_p * max(n_graphs, 1))()
    rc = L.r3dm_graphs_unpack_merge(ptrs, lens, len(bufs), n_graphs, out)
    if rc != 0:
        raise R3dmError(f"r3dm_graphs_unpack_merge -> {rc}")
    return [Graph(out[k]) for k in range(n_graphs)]


class Comm:
    """r3dm_comm: the RCCL communicator of the one-process-per-GPU route (r3dm_allgather_graphs).  Rank 0 draws the id
    (Comm.unique_id()) and hands its 128 bytes to the other ranks; every rank then creates its end."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        L = load_library()
        h = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        rc = L.r3dm_comm_create(buf, rank, world, device, C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_comm_create -> {rc}")
        self._h = h.value

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * 128)()
        rc = load_library().r3dm_comm_unique_id(buf)
        if rc != 0:
            raise R3dmError(f"r3dm_comm_unique_id -> {rc}" + (" (librccl.so not found)" if rc == -5 else ""))
        return bytes(buf)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load_library().r3dm_comm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def rank(self) -> int:
        return load_library().r3dm_comm_rank(self._h)

    @property
    def world(self) -> int:
        return load_library().r3dm_comm_world(self._h)

    @property
    def last_device_graphs(self) -> int:
        """local graphs of the last exchange that went on the wire from their device mirror (Context.set_device_graphs)"""
        return load_library().r3dm_comm_last_device_graphs(self._h)

    def allgather_graphs(self, local: Sequence["Graph"]) -> List["Graph"]:
        L = load_library()
        arr = (C.c_void_p * len(local))(*[g._h for g in local])
        out = (C.c_void_p * max(len(local), 1))()
        rc = L.r3dm_allgather_graphs(self._h, arr, len(local), out)
        if rc != 0:
            raise R3dmError(f"r3dm_allgather_graphs -> {rc}: {L.r3dm_comm_last_error(self._h).decode()}")
        return [Graph(out[k]) for k in range(len(local))]


class Index:
    """r3dm_index: a dataset staged once for many 2-NN searches"""

    def __init__(self, handle: int):
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            load_library().r3dm_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU, one context (one process per GPU)."""

    def __init__(self, device: int = 0):
        L = load_library()
        h = C.c_void_p()
        rc = L.r3dm_create(device, C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_create(device={device}) -> {rc} (no gfx950 GPU visible? there is no CPU fallback)")
        self._h = h.value
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.r3dm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: ctypes globals may already be gone
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise R3dmError(f"{what} -> {rc}: {self._L.r3dm_last_error(self._h).decode()}")

    def device_info(self) -> Tuple[str, int, int]:
        arch = C.create_string_buffer(64); cu = C.c_int(); hbm = C.c_uint64()
        self._check(self._L.r3dm_device_info(self._h, arch, 64, C.byref(cu), C.byref(hbm)), "r3dm_device_info")
        return arch.value.decode(), cu.value, hbm.value

    def set_image(self, view_id: int, desc, xy=None, width: int = 0, height: int = 0, binary: bool = False):
        """desc: [n, dim] float32 / uint8 numpy array or torch tensor (host or device memory)."""
        if isinstance(desc, np.ndarray):
            desc = np.ascontiguousarray(desc)
            is_f32 = desc.dtype == np.float32
            if not is_f32 and desc.dtype != np.uint8:
                raise TypeError(desc.dtype)
        else:
            import torch
            desc = desc.contiguous()
            is_f32 = desc.dtype == torch.float32
            if not is_f32 and desc.dtype != torch.uint8:
                raise TypeError(desc.dtype)
        n, dim = int(desc.shape[0]), int(desc.shape[1])
        dt = F32 if is_f32 else (BIN if binary else U8)
        if xy is not None:
            xy = np.ascontiguousarray(xy, np.float32) if isinstance(xy, np.ndarray) else xy.contiguous().float()
        self._check(self._L.r3dm_set_image(self._h, view_id, width, height, _ptr(desc), n, dim, dt, _ptr(xy)),
                    "r3dm_set_image")

    def set_images(self, view_ids, descs, xys=None, width: int = 0, height: int = 0, binary: bool = False, wait: bool = False):
        """a whole collection in one call (r3dm_set_images): descs / xys are sequences of [n, dim] / [n, 2] arrays (numpy: host memory,
        copied through the library's page-locked ring by helper threads; torch: device or page-locked tensors, copied by the copy engine straight
        into the ring's device slot).  wait: return when every view is
        resident and laid out (r3dm_images_wait) instead of when the caller's buffers are consumed."""
        keep = []
        arr = (ViewDesc * len(view_ids))()
        for k, vid in enumerate(view_ids):
            d = descs[k]
            if isinstance(d, np.ndarray):
                d = np.ascontiguousarray(d)
                is_f32 = d.dtype == np.float32
                if not is_f32 and d.dtype != np.uint8:
                    raise TypeError(d.dtype)
            else:
                import torch
                d = d.contiguous()
                is_f32 = d.dtype == torch.float32
                if not is_f32 and d.dtype != torch.uint8:
                    raise TypeError(d.dtype)
            x = None if xys is None else xys[k]
            if x is not None:
                x = np.ascontiguousarray(x, np.float32) if isinstance(x, np.ndarray) else x.contiguous().float()
            keep.append((d, x))
            arr[k] = ViewDesc(int(vid), width, height, int(d.shape[0]), int(d.shape[1]), F32 if is_f32 else (BIN if binary else U8), _ptr(d), _ptr(x))
        self._check(self._L.r3dm_set_images(self._h, C.cast(arr, C.c_void_p), len(view_ids)), "r3dm_set_images")
        if wait:
            self.images_wait()

    def images_wait(self):
        self._check(self._L.r3dm_images_wait(self._h), "r3dm_images_wait")

    def memory_info(self):
        """-> (device bytes of the slabs the registered views' layouts are cut from, device bytes of the upload ring, its page-locked host bytes)"""
        a = C.c_uint64(); b = C.c_uint64(); h = C.c_uint64()
        self._check(self._L.r3dm_memory_info(self._h, C.byref(a), C.byref(b), C.byref(h)), "r3dm_memory_info")
        return a.value, b.value, h.value

    def view_info(self, view_id: int):
        """-> (layout bits LAYOUT_*, bytes of HBM the view's layouts and indices occupy, views whose rows went through the page-locked ring, views copied straight from the caller's device / page-locked buffers)"""
        lay = C.c_uint32(); b = C.c_uint64(); ru = C.c_uint64(); du = C.c_uint64()
        self._check(self._L.r3dm_view_info(self._h, view_id, C.byref(lay), C.byref(b), C.byref(ru), C.byref(du)), "r3dm_view_info")
        return lay.value, b.value, ru.value, du.value

    def clear_images(self):
        self._check(self._L.r3dm_clear_images(self._h), "r3dm_clear_images")

    def trim(self):
        """give the staging buffers kept by clear_images() back to the device"""
        self._check(self._L.r3dm_trim(self._h), "r3dm_trim")

    def match_pairs(self, pairs, dist_ratio: float = 0.6, squared_metric: bool = True) -> Graph:
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        h = C.c_void_p()
        self._check(self._L.r3dm_match_pairs(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0],
                                             dist_ratio, int(squared_metric), C.byref(h)), "r3dm_match_pairs")
        return Graph(h.value)

    def match_pairs_kgraph(self, pairs, dist_ratio: float = 0.6, params: "KGraphParams" = None) -> Graph:
        """kgraph_match: approximate 2-NN through a per-view graph index (config C5)"""
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        kp = params if params is not None else KGraphParams.preset(3)
        h = C.c_void_p()
        self._check(self._L.r3dm_match_pairs_kgraph(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0], dist_ratio,
                                                    C.addressof(kp), C.byref(h)), "r3dm_match_pairs_kgraph")
        return Graph(h.value)

    def kgraph_knn2(self, dataset, query, params: "KGraphParams" = None, pair=(0, 1)):
        dataset = np.ascontiguousarray(dataset, np.float32); query = np.ascontiguousarray(query, np.float32)
        kp = params if params is not None else KGraphParams.preset(3)
        nq = query.shape[0]
        idx = np.full((max(nq, 1), 2), -1, np.int32); dist = np.zeros((max(nq, 1), 2), np.float32)
        self._check(self._L.r3dm_kgraph_knn2(self._h, _ptr(dataset), dataset.shape[0], _ptr(query), nq, dataset.shape[1],
                                             C.addressof(kp), pair[0], pair[1], _ptr(idx), _ptr(dist)), "r3dm_kgraph_knn2")
        return idx[:nq], dist[:nq]

    def drop_indices(self):
        """r3dm_drop_indices: the next match_pairs_kgraph rebuilds the graph index of every view it uses"""
        self._check(self._L.r3dm_drop_indices(self._h), "r3dm_drop_indices")

    def kgraph_index(self, view_id: int, n_rows: int, index_K: int = 24):
        """-> (adj [n, 64] uint32 padded with 0xFFFFFFFF, deg [n] uint32) of a registered view"""
        adj = np.zeros((n_rows, 64), np.uint32); deg = np.zeros(n_rows, np.uint32)
        self._check(self._L.r3dm_kgraph_index(self._h, view_id, index_K, _ptr(adj), _ptr(deg)), "r3dm_kgraph_index")
        return adj, deg

    def match_pairs_hnsw(self, pairs, dist_ratio: float = 0.6, params: "HnswParams" = None) -> Graph:
        """hnsw_match: hnswlib's searchKnn on a batch-built HNSW index per first view (matchingAlgorithm 6 / 7 / 8)"""
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        hp = params if params is not None else HnswParams.preset(2)
        h = C.c_void_p()
        self._check(self._L.r3dm_match_pairs_hnsw(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0], dist_ratio,
                                                  C.addressof(hp), C.byref(h)), "r3dm_match_pairs_hnsw")
        return Graph(h.value)

    def match_pairs_mrpt(self, pairs, dist_ratio: float = 0.6, params: "MrptParams" = None) -> Graph:
        """mrpt_match (matchingAlgorithm 5): random projection trees per first view, vote, exact re-rank (r3dm_match_pairs_mrpt)"""
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        mp = params if params is not None else MrptParams.preset()
        h = C.c_void_p()
        self._check(self._L.r3dm_match_pairs_mrpt(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0], dist_ratio,
                                                  C.addressof(mp), C.byref(h)), "r3dm_match_pairs_mrpt")
        return Graph(h.value)

    def mrpt_knn2(self, dataset, query, params: "MrptParams" = None):
        """ArrayMatcher_mrpt-shaped: (idx [nq, 2] int32 (-1: dropped), dist [nq, 2] float32 = SQUARE ROOTS of the squared L2 distances)"""
        dataset = np.ascontiguousarray(dataset, np.float32); query = np.ascontiguousarray(query, np.float32)
        mp = params if params is not None else MrptParams.preset()
        nq = query.shape[0]
        idx = np.full((max(nq, 1), 2), -1, np.int32); dist = np.zeros((max(nq, 1), 2), np.float32)
        self._check(self._L.r3dm_mrpt_knn2(self._h, _ptr(dataset), dataset.shape[0], _ptr(query), nq, dataset.shape[1],
                                           C.addressof(mp), _ptr(idx), _ptr(dist)), "r3dm_mrpt_knn2")
        return idx[:nq], dist[:nq]

    def mrpt_index(self, view_id: int, n_rows: int, dim: int, params: "MrptParams" = None) -> dict:
        """the MRPT index of a registered view as arrays (r3dm_mrpt_index), cut to the view's clamped depth"""
        mp = params if params is not None else MrptParams.preset()
        nl = 1 << mp.depth
        R = np.zeros((mp.n_trees * mp.depth, dim), np.float32); sp = np.zeros((mp.n_trees, nl - 1), np.float32)
        lv = np.zeros((mp.n_trees, n_rows), np.int32); lf = np.zeros(nl + 1, np.int32); d = C.c_uint32(0)
        self._check(self._L.r3dm_mrpt_index(self._h, view_id, C.addressof(mp), _ptr(R), _ptr(sp), _ptr(lv), _ptr(lf), C.byref(d)), "r3dm_mrpt_index")
        d = d.value; nl = 1 << d
        return dict(R=R.reshape(-1)[:mp.n_trees * d * dim].reshape(mp.n_trees * d, dim).copy(), splits=sp.reshape(-1)[:mp.n_trees * (nl - 1)].reshape(mp.n_trees, nl - 1).copy(),
                    leaves=lv, leaf_first=lf[:nl + 1].copy(), depth=d)

    def hnsw_knn2(self, dataset, query, params: "HnswParams" = None):
        dataset = np.ascontiguousarray(dataset, np.float32); query = np.ascontiguousarray(query, np.float32)
        hp = params if params is not None else HnswParams.preset(2)
        nq = query.shape[0]
        idx = np.full((max(nq, 1), 2), -1, np.int32); dist = np.zeros((max(nq, 1), 2), np.float32)
        self._check(self._L.r3dm_hnsw_knn2(self._h, _ptr(dataset), dataset.shape[0], _ptr(query), nq, dataset.shape[1],
                                           C.addressof(hp), _ptr(idx), _ptr(dist)), "r3dm_hnsw_knn2")
        return idx[:nq], dist[:nq]

    def hnsw_knn2_on_index(self, dataset, index: dict, M: int, query, ef: int):
        """searchKnn(row, 2) with setEf(ef) on an index given as arrays (keys links0, up_off, up_links, enterpoint, maxlevel)"""
        dataset = np.ascontiguousarray(dataset, np.float32); query = np.ascontiguousarray(query, np.float32)
        l0 = np.ascontiguousarray(index["links0"], np.int32); uo = np.ascontiguousarray(index["up_off"], np.int32)
        ul = np.ascontiguousarray(index["up_links"], np.int32).reshape(-1, 1 + M)
        a = HnswArrays(M, l0.ctypes.data, uo.ctypes.data, ul.ctypes.data if ul.size else None, ul.shape[0], int(index["enterpoint"]), int(index["maxlevel"]))
        nq = query.shape[0]
        idx = np.full((max(nq, 1), 2), -1, np.int32); dist = np.zeros((max(nq, 1), 2), np.float32)
        self._check(self._L.r3dm_hnsw_knn2_on_index(self._h, _ptr(dataset), dataset.shape[0], dataset.shape[1], C.addressof(a),
                                                    _ptr(query), nq, ef, _ptr(idx), _ptr(dist)), "r3dm_hnsw_knn2_on_index")
        return idx[:nq], dist[:nq]

    def hnsw_index(self, view_id: int, n_rows: int, params: "HnswParams" = None) -> dict:
        """the HNSW index of a registered view as arrays in hnswlib's shape (r3dm_hnsw_arrays)"""
        hp = params if params is not None else HnswParams.preset(2)
        M = hp.M
        l0 = np.zeros((n_rows, 1 + 2 * M), np.int32); uo = np.zeros(n_rows + 1, np.int32)
        cap = n_rows + 64
        ul = np.zeros((cap, 1 + M), np.int32)
        rows = C.c_uint32(0); ep = C.c_int32(0); ml = C.c_int32(0)
        self._check(self._L.r3dm_hnsw_index(self._h, view_id, C.addressof(hp), _ptr(l0), _ptr(uo), _ptr(ul), cap, C.byref(rows), C.byref(ep), C.byref(ml)),
                    "r3dm_hnsw_index")
        return dict(links0=l0, up_off=uo, up_links=ul[:rows.value].copy(), enterpoint=ep.value, maxlevel=ml.value,
                    levels=np.diff(uo).astype(np.int32))

    def filter_F(self, putative: Graph, max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489,
                 want_F: bool = False):
        h = C.c_void_p()
        Fbuf = np.zeros((max(putative.num_pairs, 1), 9), np.float64) if want_F else None
        self._check(self._L.r3dm_filter_F(self._h, putative._h, max_residual_px, max_iter, seed, 0, C.byref(h),
                                          _ptr(Fbuf)), "r3dm_filter_F")
        g = Graph(h.value)
        return (g, Fbuf[:g.num_pairs].copy()) if want_F else g

    def filter_H(self, putative: Graph, max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489,
                 want_H: bool = False):
        h = C.c_void_p()
        Hbuf = np.zeros((max(putative.num_pairs, 1), 9), np.float64) if want_H else None
        self._check(self._L.r3dm_filter_H(self._h, putative._h, max_residual_px, max_iter, seed, C.byref(h), _ptr(Hbuf)),
                    "r3dm_filter_H")
        g = Graph(h.value)
        return (g, Hbuf[:g.num_pairs].copy()) if want_H else g

    def set_device_graphs(self, enable: bool = True):
        """r3dm_set_device_graphs: graphs produced from now on keep a device mirror (sent by Comm.allgather_graphs without a host round trip)"""
        self._check(self._L.r3dm_set_device_graphs(self._h, int(bool(enable))), "r3dm_set_device_graphs")

    def set_integer_mfma(self, enable: bool = True):
        """opt-in bf16-exact MFMA path for integer-valued descriptors (include/r3dm.h: r3dm_set_integer_mfma)"""
        self._check(self._L.r3dm_set_integer_mfma(self._h, int(bool(enable))), "r3dm_set_integer_mfma")

    def set_hamming_mfma(self, enable: bool = True):
        """opt-in exact MFMA formulation of the Hamming matcher (include/r3dm.h: r3dm_set_hamming_mfma)"""
        self._check(self._L.r3dm_set_hamming_mfma(self._h, int(bool(enable))), "r3dm_set_hamming_mfma")

    def set_split_mfma(self, enable: bool = True):
        """opt-in split-f16 nominator for real-valued descriptors (include/r3dm.h: r3dm_set_split_mfma)"""
        self._check(self._L.r3dm_set_split_mfma(self._h, int(bool(enable))), "r3dm_set_split_mfma")

    def set_intrinsics(self, view_id: int, K):
        K = None if K is None else np.ascontiguousarray(K, np.float64).reshape(9)
        self._check(self._L.r3dm_set_intrinsics(self._h, view_id, _ptr(K)), "r3dm_set_intrinsics")

    def filter_E(self, putative: Graph, max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489,
                 min_count: int = 50, min_ratio: float = 0.3, want_E: bool = False):
        h = C.c_void_p()
        Ebuf = np.zeros((max(putative.num_pairs, 1), 9), np.float64) if want_E else None
        self._check(self._L.r3dm_filter_E(self._h, putative._h, max_residual_px, max_iter, seed, min_count, min_ratio,
                                          C.byref(h), _ptr(Ebuf)), "r3dm_filter_E")
        g = Graph(h.value)
        return (g, Ebuf[:g.num_pairs].copy()) if want_E else g

    def filter_FEH(self, putative: Graph, which: str = "FEH", max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489,
                   min_count: int = 50, min_ratio: float = 0.3):
        """r3dm_filter_FEH: the requested filters side by side -> ({"F": Graph, ...}, ms_kernels [F, E, H], ms_wall [F, E, H])"""
        bits = sum({"F": 1, "E": 2, "H": 4}[k] for k in which)
        hs = {k: C.c_void_p() for k in "FEH"}
        msk = np.zeros(3); msw = np.zeros(3)
        self._check(self._L.r3dm_filter_FEH(self._h, putative._h, max_residual_px, max_iter, seed, bits, min_count, min_ratio,
                                            C.byref(hs["F"]), C.byref(hs["E"]), C.byref(hs["H"]), _ptr(msk), _ptr(msw)), "r3dm_filter_FEH")
        return {k: Graph(hs[k].value) for k in which}, msk, msw

    def knn2(self, dataset: np.ndarray, query: np.ndarray, binary: bool = False):
        dataset = np.ascontiguousarray(dataset); query = np.ascontiguousarray(query)
        dt = F32 if dataset.dtype == np.float32 else (BIN if binary else U8)
        nq = query.shape[0]
        idx = np.full((max(nq, 1), 2), -1, np.int32); dist = np.zeros((max(nq, 1), 2), np.float32)
        self._check(self._L.r3dm_knn2(self._h, _ptr(dataset), dataset.shape[0], _ptr(query), nq, dataset.shape[1], dt,
                                      _ptr(idx), _ptr(dist)), "r3dm_knn2")
        return idx[:nq], dist[:nq]

    def index_create(self, dataset: np.ndarray, binary: bool = False) -> "Index":
        """ArrayMatcher::Build: stage the dataset once (r3dm_index_create)"""
        dataset = np.ascontiguousarray(dataset)
        dt = F32 if dataset.dtype == np.float32 else (BIN if binary else U8)
        h = C.c_void_p()
        self._check(self._L.r3dm_index_create(self._h, _ptr(dataset), dataset.shape[0], dataset.shape[1], dt, C.byref(h)), "r3dm_index_create")
        return Index(h.value)

    def index_knn2(self, index: "Index", query: np.ndarray):
        """ArrayMatcher::SearchNeighbours(NN = 2) against a staged dataset; any context of the index's device"""
        query = np.ascontiguousarray(query)
        nq = query.shape[0]
        idx = np.full((max(nq, 1), 2), -1, np.int32); dist = np.zeros((max(nq, 1), 2), np.float32)
        self._check(self._L.r3dm_index_knn2(self._h, index._h, _ptr(query), nq, _ptr(idx), _ptr(dist)), "r3dm_index_knn2")
        return idx[:nq], dist[:nq]

    def liop_describe_patches(self, patches):
        """patches: [n, 41, 41] float32 (numpy or torch, host or device) -> (desc [n, 144] float32, n_resorted)"""
        if isinstance(patches, np.ndarray):
            patches = np.ascontiguousarray(patches, np.float32)
        n, side = int(patches.shape[0]), int(patches.shape[1])
        out = np.zeros((max(n, 1), 144), np.float32)
        nt = C.c_uint32(0)
        self._check(self._L.r3dm_liop_describe_patches(self._h, _ptr(patches), n, side, _ptr(out), C.byref(nt)),
                    "r3dm_liop_describe_patches")
        return out[:n], int(nt.value)

    def extract_liop(self, image, keypoints, kp_size_factor: float = 8.0, want_patches: bool = False):
        """image: [h, w] float32 gray/255; keypoints: [n, 4] float32 (x, y, size, angle_deg) -> desc [n, 144]"""
        image = np.ascontiguousarray(image, np.float32) if isinstance(image, np.ndarray) else image.contiguous()
        keypoints = np.ascontiguousarray(keypoints, np.float32)
        n = keypoints.shape[0]
        desc = np.zeros((max(n, 1), 144), np.float32)
        patches = np.zeros((max(n, 1), 41, 41), np.float32) if want_patches else None
        self._check(self._L.r3dm_extract_liop(self._h, _ptr(image), int(image.shape[1]), int(image.shape[0]), _ptr(keypoints), n,
                                              kp_size_factor, _ptr(desc), _ptr(patches)), "r3dm_extract_liop")
        return (desc[:n], patches[:n]) if want_patches else desc[:n]

    def detect_akaze(self, image, threshold: float = 0.001, cap: int = 200000):
        """image: [h, w] float32 in [0, 1] (numpy or torch, host or device) -> (keypoints [n, 4] (x, y, size, angle_deg), responses [n])"""
        h, w = int(image.shape[0]), int(image.shape[1])
        if isinstance(image, np.ndarray):
            image = np.ascontiguousarray(image, np.float32)
        kps = np.zeros((cap, 4), np.float32); resp = np.zeros(cap, np.float32)
        n = C.c_uint32(0)
        self._check(self._L.r3dm_detect_akaze(self._h, _ptr(image), w, h, threshold, _ptr(kps), _ptr(resp), cap, C.byref(n)),
                    "r3dm_detect_akaze")
        k = min(n.value, cap)
        return kps[:k].copy(), resp[:k].copy()

    def detect_akaze_mldb(self, image, threshold: float = 0.001, cap: int = 200000):
        """AKAZE2::detectAndCompute(DESCRIPTOR_MLDB) -> (keypoints [n, 4], descriptors [n, 61] uint8)"""
        h, w = int(image.shape[0]), int(image.shape[1])
        if isinstance(image, np.ndarray):
            image = np.ascontiguousarray(image, np.float32)
        kps = np.zeros((cap, 4), np.float32); desc = np.zeros((cap, 61), np.uint8)
        n = C.c_uint32(0)
        self._check(self._L.r3dm_detect_akaze_mldb(self._h, _ptr(image), w, h, threshold, _ptr(kps), _ptr(desc), cap, C.byref(n)),
                    "r3dm_detect_akaze_mldb")
        k = min(n.value, cap)
        return kps[:k].copy(), desc[:k].copy()

    def detect_akaze_batch(self, images, threshold: float = 0.001, cap: int = 200000):
        """r3dm_detect_akaze_batch: B same-size images (numpy or torch, host or device) in one pass of the detector
        -> list of (keypoints [n, 4], responses [n])"""
        imgs = [im if hasattr(im, "data_ptr") else np.ascontiguousarray(im, np.float32) for im in images]
        B = len(imgs); h, w = int(imgs[0].shape[0]), int(imgs[0].shape[1])
        assert all(tuple(im.shape) == (h, w) for im in imgs), "a batch holds images of one size"
        ip = (C.c_void_p * B)(*[(im.data_ptr() if hasattr(im, "data_ptr") else im.ctypes.data) for im in imgs])
        kps = [np.zeros((cap, 4), np.float32) for _ in range(B)]; resp = [np.zeros(cap, np.float32) for _ in range(B)]
        kp_p = (C.c_void_p * B)(*[k.ctypes.data for k in kps]); rp_p = (C.c_void_p * B)(*[r.ctypes.data for r in resp])
        n = np.zeros(B, np.uint32)
        self._check(self._L.r3dm_detect_akaze_batch(self._h, B, ip, w, h, threshold, kp_p, rp_p, cap, _ptr(n)), "r3dm_detect_akaze_batch")
        return [(kps[b][:min(int(n[b]), cap)].copy(), resp[b][:min(int(n[b]), cap)].copy()) for b in range(B)]

    def extract_features_batch(self, images, feat_paths, desc_paths, threshold: float = 0.001, bgr: bool = False):
        """r3dm_extract_features_batch: detector + LIOP + .feat / .desc files of B same-size images in one pass.
        images: [h, w] float32 gray / 255, or with bgr=True [h, w, 3] uint8 as cv::imread decodes -> n_features [B]"""
        if bgr:
            imgs = [im if hasattr(im, "data_ptr") else np.ascontiguousarray(im, np.uint8) for im in images]
        else:
            imgs = [im if hasattr(im, "data_ptr") else np.ascontiguousarray(im, np.float32) for im in images]
        B = len(imgs); h, w = int(imgs[0].shape[0]), int(imgs[0].shape[1])
        ip = (C.c_void_p * B)(*[(im.data_ptr() if hasattr(im, "data_ptr") else im.ctypes.data) for im in imgs])
        fp = (C.c_char_p * B)(*[p.encode() for p in feat_paths]); dp = (C.c_char_p * B)(*[p.encode() for p in desc_paths])
        nf = np.zeros(B, np.uint32)
        self._check(self._L.r3dm_extract_features_batch(self._h, B, None if bgr else ip, ip if bgr else None, w, h, threshold, fp, dp, _ptr(nf)),
                    "r3dm_extract_features_batch")
        return nf

    def set_deferred_feature_files(self, on: bool = True):
        """r3dm_set_deferred_feature_files: features calls return when the images are computed; their files are written on the
        context's writer thread (features_files_wait joins it and reports its I/O error)"""
        self._check(self._L.r3dm_set_deferred_feature_files(self._h, 1 if on else 0), "r3dm_set_deferred_feature_files")

    def features_files_wait(self):
        self._check(self._L.r3dm_features_files_wait(self._h), "r3dm_features_files_wait")

    def gray_from_bgr8(self, bgr: np.ndarray) -> np.ndarray:
        bgr = np.ascontiguousarray(bgr, np.uint8)
        out = np.zeros(bgr.shape[:2], np.float32)
        self._check(self._L.r3dm_gray_from_bgr8(self._h, _ptr(bgr), bgr.shape[1], bgr.shape[0], _ptr(out)), "r3dm_gray_from_bgr8")
        return out

    def extract_features_to_files(self, gray, feat_path: str, desc_path: str, threshold: float = 0.001) -> int:
        gray = np.ascontiguousarray(gray, np.float32)
        n = C.c_uint32(0)
        self._check(self._L.r3dm_extract_features_to_files(self._h, _ptr(gray), gray.shape[1], gray.shape[0], threshold,
                                                           feat_path.encode(), desc_path.encode(), C.byref(n)), "r3dm_extract_features_to_files")
        return n.value

    def filter_report(self):
        """per putative pair of the last filter_F call: (threshold_px, nfa, iterations, models, inliers)"""
        n = self._L.r3dm_filter_report(self._h, None, 0)
        arr = (PairReport * max(n, 1))()
        self._L.r3dm_filter_report(self._h, arr, n)
        return [(r.threshold_px, r.nfa, r.iterations, r.models, r.inliers) for r in arr[:n]]

    def features_totals(self) -> FeaturesTotals:
        t = FeaturesTotals()
        self._check(self._L.r3dm_get_features_totals(self._h, C.byref(t)), "r3dm_get_features_totals")
        return t

    def stats(self) -> Stats:
        s = Stats()
        self._check(self._L.r3dm_get_stats(self._h, C.byref(s)), "r3dm_get_stats")
        return s


def shard_owner(pairs: np.ndarray, world: int) -> np.ndarray:
    """r3dm_shard_pairs: the device / rank that the snake deal of rows I gives every pair (host code, no GPU needed)"""
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    owner = np.zeros(pairs.shape[0], np.uint32)
    rc = load_library().r3dm_shard_pairs(_ptr(pairs) if pairs.size else None, pairs.shape[0], int(world), _ptr(owner) if pairs.size else None)
    if rc != 0:
        raise R3dmError(f"r3dm_shard_pairs -> {rc}")
    return owner


class MultiContext:
    """r3dm_multi_*: one process driving several GPUs (one context + one host thread per device); device ids may repeat."""

    def __init__(self, device_ids: Sequence[int]):
        L = load_library()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = L.r3dm_multi_create(ids, len(device_ids), C.byref(h))
        if rc != 0:
            raise R3dmError(f"r3dm_multi_create({list(device_ids)}) -> {rc} (no gfx950 GPU visible? there is no CPU fallback)")
        self._h = h.value
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.r3dm_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise R3dmError(f"{what} -> {rc}: {self._L.r3dm_multi_last_error(self._h).decode()}")

    @property
    def num_devices(self) -> int:
        return int(self._L.r3dm_multi_num_devices(self._h))

    def device_stats(self, k: int) -> Stats:
        s = Stats()
        rc = self._L.r3dm_get_stats(self._L.r3dm_multi_ctx(self._h, k), C.byref(s))
        self._check(rc, "r3dm_get_stats")
        return s

    def set_image(self, view_id: int, desc, xy=None, width: int = 0, height: int = 0, binary: bool = False):
        desc = np.ascontiguousarray(desc)
        dt = F32 if desc.dtype == np.float32 else (BIN if binary else U8)
        if xy is not None:
            xy = np.ascontiguousarray(xy, np.float32)
        self._check(self._L.r3dm_multi_set_image(self._h, view_id, width, height, _ptr(desc), desc.shape[0], desc.shape[1], dt, _ptr(xy)),
                    "r3dm_multi_set_image")

    def transfer_counts(self):
        """(views uploaded from host memory, device-to-device copies) of the set_image calls so far"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._L.r3dm_multi_transfer_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self._check(self._L.r3dm_multi_transfer_counts(self._h, C.byref(a), C.byref(b)), "r3dm_multi_transfer_counts")
        return int(a.value), int(b.value)

    def set_intrinsics(self, view_id: int, K):
        K = None if K is None else np.ascontiguousarray(K, np.float64).reshape(9)
        self._check(self._L.r3dm_multi_set_intrinsics(self._h, view_id, _ptr(K)), "r3dm_multi_set_intrinsics")

    def set_integer_mfma(self, enable: bool = True):
        self._check(self._L.r3dm_multi_set_integer_mfma(self._h, int(bool(enable))), "r3dm_multi_set_integer_mfma")

    def extract_features(self, images, feat_paths, desc_paths, threshold: float = 0.001, bgr: bool = False, batch: int = 0):
        """r3dm_multi_extract_features(_bgr8): the features stage over an image list, one BATCH of same-size images in flight per context.
        images: list of [h, w] float32 arrays (gray / 255) -- or, with bgr=True, [h, w, 3] uint8 as cv::imread decodes --
        host (numpy) or device (anything with data_ptr(): torch tensors).
        -> (n_features [N], skipped [N] bool)"""
        imgs = [im if hasattr(im, "data_ptr") else np.ascontiguousarray(im, np.uint8 if bgr else np.float32) for im in images]
        n = len(imgs)
        gp = (C.c_void_p * n)(*[(im.data_ptr() if hasattr(im, "data_ptr") else im.ctypes.data) for im in imgs])
        ws = np.array([im.shape[1] for im in imgs], np.uint32); hs = np.array([im.shape[0] for im in imgs], np.uint32)
        fp = (C.c_char_p * n)(*[p.encode() for p in feat_paths]); dp = (C.c_char_p * n)(*[p.encode() for p in desc_paths])
        nf = np.zeros(n, np.uint32); sk = np.zeros(n, np.uint32)
        err = C.create_string_buffer(512)
        if bgr or batch:
            rc = self._L.r3dm_multi_extract_features_ex(self._h, n, None if bgr else gp, gp if bgr else None, _ptr(ws), _ptr(hs), threshold, fp, dp,
                                                        _ptr(nf), _ptr(sk), batch, err, 512)
        else:
            rc = self._L.r3dm_multi_extract_features(self._h, n, gp, _ptr(ws), _ptr(hs), threshold, fp, dp, _ptr(nf), _ptr(sk), err, 512)
        if rc != 0:
            raise R3dmError(f"r3dm_multi_extract_features -> {rc}: {err.value.decode()}")
        return nf, sk.astype(bool)

    def match_pairs(self, pairs, dist_ratio: float = 0.6, squared_metric: bool = True) -> Graph:
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        h = C.c_void_p()
        self._check(self._L.r3dm_multi_match_pairs(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0],
                                                   dist_ratio, int(squared_metric), C.byref(h)), "r3dm_multi_match_pairs")
        return Graph(h.value)

    def match_pairs_kgraph(self, pairs, dist_ratio: float = 0.6, params: "KGraphParams" = None) -> Graph:
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        kp = params if params is not None else KGraphParams.preset(3)
        h = C.c_void_p()
        self._check(self._L.r3dm_multi_match_pairs_kgraph(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0], dist_ratio,
                                                          C.addressof(kp), C.byref(h)), "r3dm_multi_match_pairs_kgraph")
        return Graph(h.value)

    def match_pairs_hnsw(self, pairs, dist_ratio: float = 0.6, params: "HnswParams" = None) -> Graph:
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        hp = params if params is not None else HnswParams.preset(2)
        h = C.c_void_p()
        self._check(self._L.r3dm_multi_match_pairs_hnsw(self._h, _ptr(pairs) if pairs.size else None, pairs.shape[0], dist_ratio,
                                                        C.addressof(hp), C.byref(h)), "r3dm_multi_match_pairs_hnsw")
        return Graph(h.value)

    def _filter(self, fn, what, putative, args, want):
        h = C.c_void_p()
        buf = np.zeros((max(putative.num_pairs, 1), 9), np.float64) if want else None
        self._check(fn(self._h, putative._h, *args, C.byref(h), _ptr(buf)), what)
        g = Graph(h.value)
        return (g, buf[:g.num_pairs].copy()) if want else g

    def filter_F(self, putative: Graph, max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489, want_F: bool = False):
        return self._filter(self._L.r3dm_multi_filter_F, "r3dm_multi_filter_F", putative, (max_residual_px, max_iter, seed), want_F)

    def filter_H(self, putative: Graph, max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489, want_H: bool = False):
        return self._filter(self._L.r3dm_multi_filter_H, "r3dm_multi_filter_H", putative, (max_residual_px, max_iter, seed), want_H)

    def filter_E(self, putative: Graph, max_residual_px: float = 4.0, max_iter: int = 2048, seed: int = 5489,
                 min_count: int = 50, min_ratio: float = 0.3, want_E: bool = False):
        return self._filter(self._L.r3dm_multi_filter_E, "r3dm_multi_filter_E", putative,
                            (max_residual_px, max_iter, seed, min_count, min_ratio), want_E)



"""ctypes binding of libchordvis.so (the C ABI declared in include/chordvis.h).

The product path has no CPU fallback: if the HIP library is missing this module
raises at import, and every device entry point fails loudly without a GPU.
"""
import ctypes as C
import os

import numpy as np

from . import records as R

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHORDVIS_LIB") or os.path.join(_HERE, "_build", "libchordvis.so")

OK, E_INVALID, E_HIP, E_NO_DEVICE, E_CAPACITY, E_COMM = 0, -1, -2, -3, -4, -5


class ChordvisError(RuntimeError):
    pass


class CountAndCmd(C.Structure):
    _fields_ = [("count", C.c_void_p), ("cmds", C.c_void_p), ("capacity", C.c_uint32)]


class Limits(C.Structure):
    _fields_ = [("maxTriangleRecords", C.c_uint64), ("binPoolChunks", C.c_uint32), ("binMaxChunksPerTile", C.c_uint32)]


class TileMarker(C.Structure):
    _fields_ = [("marker", C.c_void_p), ("visibilityDim", C.c_uint32 * 2), ("markerDim", C.c_uint32 * 2)]


class ShadingTiles(C.Structure):
    _fields_ = [("tileCmd", C.c_void_p), ("count", C.c_void_p), ("dispatchIndirect", C.c_void_p), ("capacity", C.c_uint32)]


class DepthTarget(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class HZB(C.Structure):
    _fields_ = [("desc", R.HZBDesc), ("minTexels", C.c_void_p), ("maxTexels", C.c_void_p), ("validRange", C.c_void_p)]


class CameraDesc(C.Structure):
    _fields_ = [
        ("position", C.c_double * 3), ("front", C.c_double * 3), ("worldUp", C.c_double * 3),
        ("fovy", C.c_float), ("jitter", C.c_float * 2),
        ("zNear", C.c_double), ("zFar", C.c_double),
        ("width", C.c_uint32), ("height", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("msClear", C.c_float), ("msInstanceCulling", C.c_float), ("msStage0", C.c_float), ("msHzbStage0", C.c_float),
        ("msStage1", C.c_float), ("msHzbFinal", C.c_float), ("msFrame", C.c_float),
        ("msRasterCluster", C.c_float), ("msRasterClip", C.c_float), ("msRasterChunk", C.c_float),
        ("framesTimed", C.c_uint32), ("rasterLaunches", C.c_uint32), ("overflow", C.c_uint32), ("countInstanceCulled", C.c_uint32), ("countStage0Visible", C.c_uint32),
        ("countStage0Rejected", C.c_uint32), ("countStage1Visible", C.c_uint32), ("trianglesSubmitted", C.c_uint64),
        ("triangleRecords", C.c_uint64), ("binEntries", C.c_uint64), ("tilesTouched", C.c_uint32 * 2),
        ("triangleRecordsCompact", C.c_uint64),
        ("pixelBlockBytes", C.c_uint64),
        ("pixelBlocks", C.c_uint64),
        ("msExchangeHzb", C.c_float), ("msExchangeVis", C.c_float), ("msExchangeCull", C.c_float), ("msExchangeFinal", C.c_float),
        ("kernelLaunches", C.c_uint32), ("largeRecords", C.c_uint32 * 2), ("clipTriangles", C.c_uint32 * 2),
        ("stampsPerFrame", C.c_float),
    ]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n in ("tilesTouched", "largeRecords", "clipTriangles") else getattr(self, n)) for n, _ in self._fields_}


def _preload_hip_runtime():
    """Make exactly one HIP runtime live in this process, with global symbol visibility.

    libchordvis.so is linked with -no-hip-rt (no DT_NEEDED on libamdhip64).  If PyTorch-ROCm is
    installed its bundled runtime must be the one (two HSA runtimes in one process cannot both open
    the device), otherwise the system ROCm runtime is used.
    """
    candidates = []
    try:
        import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    candidates += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"), "libamdhip64.so"]
    for path in candidates:
        try:
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise ChordvisError("no HIP runtime (libamdhip64.so) found; tried: %s" % ", ".join(candidates))


def _load():
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ChordvisError(
            "libchordvis.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python chord_amd/build.py`; there is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    P = C.POINTER
    protos = {
        "chordvis_version": (C.c_char_p, []),
        "chordvis_hzb_desc": (i32, [u32, u32, P(R.HZBDesc)]),
        "chordvis_camera_fill_view": (i32, [P(CameraDesc), vp, vp, vp]),
        "chordvis_cascade_setup": (i32, [vp, vp, vp, vp, vp, u32, i32, vp]),
        "chordvis_object_basic_data": (i32, [vp, vp, vp, vp, vp]),
        "chordvis_object_basic_data_batch": (i32, [u32, vp, vp, vp, vp, vp]),
        "chordvis_nanite_build": (i32, [vp, u32, vp, u32, vp, P(vp)]),
        "chordvis_meshlet_bounds": (i32, [vp, u32, vp, u32, vp]),
        "chordvis_built_asset_desc": (i32, [vp, P(R.AssetDesc), vp, P(u32)]),
        "chordvis_free_built_asset": (None, [vp]),
        "chordvis_save_asset": (i32, [vp, C.c_char_p]),
        "chordvis_load_asset": (i32, [C.c_char_p, P(vp)]),
        "chordvis_save_gltf_binary": (i32, [vp, C.c_char_p, i32]),
        "chordvis_load_gltf_binary": (i32, [C.c_char_p, P(vp)]),
        "chordvis_create": (i32, [i32, vp, P(vp)]),
        "chordvis_destroy": (i32, [vp]),
        "chordvis_last_error": (C.c_char_p, [vp]),
        "chordvis_sync": (i32, [vp]),
        "chordvis_upload_scene": (i32, [vp, P(R.SceneDesc)]),
        "chordvis_update_objects": (i32, [vp, vp, u32]),
        "chordvis_bind_objects": (i32, [vp, vp, u32]),
        "chordvis_set_view": (i32, [vp, vp, vp, u32]),
        "chordvis_allocate_gbuffer": (i32, [vp, u32, u32, vp]),
        "chordvis_set_shard": (i32, [vp, u32, u32]),
        "chordvis_tile_count": (u32, [u32, u32]),
        "chordvis_tile_slots_per_rank": (u32, [u32, u32, u32]),
        "chordvis_tile_slot_capacity": (u32, [u32, u32, u32]),
        "chordvis_tile_layout": (i32, [u32, u32, u32, vp, u32, vp]),
        "chordvis_set_tile_owners": (i32, [vp, vp, u32]),
        "chordvis_get_tile_owners": (i32, [vp, vp, u32]),
        "chordvis_read_tile_loads": (i32, [vp, vp, u32]),
        "chordvis_rebalance": (i32, [vp, vp]),
        "chordvis_set_cull_mode": (i32, [vp, i32]),
        "chordvis_set_tile_schedule_keep": (i32, [vp, u32]),
        "chordvis_tile_schedule_keep": (u32, [vp]),
        "chordvis_visibility_words": (u64, [vp]),
        "chordvis_visibility_chunk_words": (u64, [vp]),
        "chordvis_visibility_ptr": (vp, [vp]),
        "chordvis_clear_gbuffer": (i32, [vp]),
        "chordvis_instance_culling": (i32, [vp, P(CountAndCmd)]),
        "chordvis_hzb_culling": (i32, [vp, P(HZB), i32, CountAndCmd, P(CountAndCmd), P(CountAndCmd)]),
        "chordvis_render_mesh": (i32, [vp, CountAndCmd]),
        "chordvis_visibility_stage0": (i32, [vp, P(HZB), CountAndCmd, P(CountAndCmd), P(i32)]),
        "chordvis_visibility_stage1": (i32, [vp, P(HZB), CountAndCmd]),
        "chordvis_build_hzb": (i32, [vp, i32, i32, i32, i32, P(HZB)]),
        "chordvis_render_frame": (i32, [vp]),
        "chordvis_frame_phase_cull": (i32, [vp]),
        "chordvis_cull_exchange_ptr": (vp, [vp]),
        "chordvis_cull_exchange_chunk_bytes": (u64, [vp]),
        "chordvis_debug_fill_cull_exchange": (i32, [vp]),
        "chordvis_frame_phase_a": (i32, [vp]),
        "chordvis_frame_phase_b": (i32, [vp]),
        "chordvis_frame_phase_c": (i32, [vp]),
        "chordvis_frame_phase_c_finish": (i32, [vp]),
        "chordvis_frame_resolve_visibility": (i32, [vp, vp]),
        "chordvis_swap_visibility": (i32, [vp]),
        "chordvis_hzb_final_exchange_ptr": (vp, [vp]),
        "chordvis_hzb_final_exchange_chunk_bytes": (u64, [vp]),
        "chordvis_reset_history": (i32, [vp]),
        "chordvis_hzb_exchange_ptr": (vp, [vp]),
        "chordvis_hzb_exchange_halves": (u64, [vp]),
        "chordvis_hzb_exchange_chunk_halves": (u64, [vp]),
        "chordvis_resolved_visibility_ptr": (vp, [vp]),
        "chordvis_last_frame_cmds": (i32, [vp, P(CountAndCmd)]),
        "chordvis_history_hzb": (i32, [vp, P(HZB)]),
        "chordvis_readback_visibility": (i32, [vp, vp]),
        "chordvis_readback_previous_visibility": (i32, [vp, vp]),
        "chordvis_readback_cmds": (i32, [vp, CountAndCmd, vp, u32, P(u32)]),
        "chordvis_readback_hzb": (i32, [vp, P(HZB), vp, vp, vp]),
        "chordvis_upload_history_hzb": (i32, [vp, vp]),
        "chordvis_set_limits": (i32, [vp, P(Limits)]),
        "chordvis_visibility_mark": (i32, [vp, CountAndCmd, P(TileMarker)]),
        "chordvis_wait_visibility": (i32, [vp, vp]),
        "chordvis_prepare_shading_tile_param": (i32, [vp, u32, P(TileMarker), P(ShadingTiles)]),
        "chordvis_readback_tile_marker": (i32, [vp, P(TileMarker), vp]),
        "chordvis_readback_shading_tiles": (i32, [vp, P(ShadingTiles), vp, u32, P(u32), vp]),
        "chordvis_comm_unique_id": (i32, [vp]),
        "chordvis_comm_init_rank": (i32, [vp, u32, u32, vp]),
        "chordvis_comm_destroy": (i32, [vp]),
        "chordvis_comm_set_pipelined": (i32, [vp, vp]),
        "chordvis_comm_info": (i32, [vp, P(i32), P(u32), vp, u32]),
        "chordvis_create_group": (i32, [u32, P(i32), P(vp)]),
        "chordvis_destroy_group": (i32, [vp]),
        "chordvis_group_size": (u32, [vp]),
        "chordvis_group_ctx": (vp, [vp, u32]),
        "chordvis_group_last_error": (C.c_char_p, [vp]),
        "chordvis_group_set_limits": (i32, [vp, P(Limits)]),
        "chordvis_group_upload_scene": (i32, [vp, P(R.SceneDesc)]),
        "chordvis_group_allocate_gbuffer": (i32, [vp, u32, u32]),
        "chordvis_group_rebalance": (i32, [vp, vp]),
        "chordvis_group_update_objects": (i32, [vp, vp, u32]),
        "chordvis_group_set_view": (i32, [vp, vp, vp, u32]),
        "chordvis_group_render_frame": (i32, [vp]),
        "chordvis_group_sync": (i32, [vp]),
        "chordvis_group_enqueue_ms": (i32, [vp, vp, u32]),
        "chordvis_group_set_pipelined": (i32, [vp, i32]),
        "chordvis_allocate_depth_views": (i32, [vp, u32, u32]),
        "chordvis_set_instance_views": (i32, [vp, vp, u32]),
        "chordvis_instance_culling_view": (i32, [vp, u32, P(CountAndCmd)]),
        "chordvis_hzb_culling_generic": (i32, [vp, P(HZB), C.c_float, u32, i32, CountAndCmd, P(CountAndCmd)]),
        "chordvis_render_mesh_depth": (i32, [vp, u32, i32, C.c_float, C.c_float, CountAndCmd, P(DepthTarget)]),
        "chordvis_build_hzb_from_depth": (i32, [vp, P(DepthTarget), P(HZB)]),
        "chordvis_readback_depth": (i32, [vp, P(DepthTarget), vp]),
        "chordvis_depth_view_stats": (i32, [vp, P(Stats)]),
        "chordvis_render_shadow": (i32, [vp, vp, vp, vp, u32, i32, vp, vp, P(u32)]),
        "chordvis_enable_timers": (i32, [vp, i32]),
        "chordvis_stats": (i32, [vp, P(Stats)]),
        "chordvis_set_debug": (i32, [vp, u32]),
        "chordvis_debug_tile_profile": (i32, [vp, i32, vp, vp, u32]),
        "chordvis_debug_setup_profile": (i32, [vp, i32, vp, P(u32)]),
        "chordvis_debug_read": (i32, [vp, i32, C.c_uint64, C.c_uint64, vp]),
        "chordvis_debug_slab_nonzero": (i32, [vp, P(C.c_uint64)]),
        "chordvis_debug_graph_frames": (i32, [vp, u32, P(C.c_float), P(C.c_float)]),
    }
    missing = []
    for name, (res, args) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    if missing and os.environ.get("CHORDVIS_AB_OLD_LIB") == "1":
        missing = []                                     # measurement only: A/B against a library of an earlier round (tools/)
    if missing:
        raise ChordvisError("libchordvis.so lacks symbols declared in include/chordvis.h: %s" % ", ".join(missing))
    return lib, sorted(protos)


lib, EXPORTED = _load()


# ------------------------------------------------------------------------------- host helpers ---

def hzb_desc(width, height):
    d = R.HZBDesc()
    rc = lib.chordvis_hzb_desc(width, height, C.byref(d))
    if rc != OK:
        raise ChordvisError("chordvis_hzb_desc(%d, %d) -> %d" % (width, height, rc))
    return d


def make_views(camera, last_view=None):
    """(ChordCameraView, ChordInstanceCullingView) numpy records for a scenes.Camera."""
    cd = CameraDesc()
    cd.position[:] = camera.position
    cd.front[:] = camera.front
    cd.worldUp[:] = camera.world_up
    cd.fovy = camera.fovy
    cd.jitter[:] = camera.jitter
    cd.zNear, cd.zFar = camera.z_near, camera.z_far
    cd.width, cd.height = camera.width, camera.height
    view = np.zeros(1, dtype=R.CAMERA_VIEW)
    iv = np.zeros(1, dtype=R.INSTANCE_CULLING_VIEW)
    lv = last_view.ctypes.data if last_view is not None else None
    rc = lib.chordvis_camera_fill_view(C.byref(cd), lv, view.ctypes.data, iv.ctypes.data)
    if rc != OK:
        raise ChordvisError("chordvis_camera_fill_view -> %d" % rc)
    return view, iv


def fill_objects(scene, camera, camera_last=None, local_to_world_last=None):
    """SceneNode::getObjectBasicData for every object of a scenes.Scene (static objects unless local_to_world_last -- the
    objects' transforms of the previous frame, same layout as scene.local_to_world -- says otherwise)."""
    cam = (C.c_double * 3)(*camera.position)
    cam_last = (C.c_double * 3)(*(camera_last or camera).position)
    l2w = scene.local_to_world
    if local_to_world_last is not None:
        local_to_world_last = np.ascontiguousarray(local_to_world_last, dtype=np.float64)
        assert local_to_world_last.shape == l2w.shape
    rc = lib.chordvis_object_basic_data_batch(len(scene.objects), l2w.ctypes.data,
                                              local_to_world_last.ctypes.data if local_to_world_last is not None else None, cam, cam_last,
                                              scene.objects.ctypes.data)
    if rc != OK:
        raise ChordvisError("chordvis_object_basic_data_batch -> %d" % rc)
    return scene.objects


def cascade_setup(config, view, main_iv, light_dir, valid_range=None, tick=0, cache_valid=False, views=None):
    """cascadeComputeCS on the host: InstanceCullingViewInfo records of the shadow cascades (in/out `views`)."""
    n = int(config["cascadeCount"][0])
    if views is None:
        views = np.zeros(n, dtype=R.INSTANCE_CULLING_VIEW)
    ld = np.asarray(light_dir, dtype=np.float32)
    vr = None if valid_range is None else np.asarray(valid_range, dtype=np.uint32)
    rc = lib.chordvis_cascade_setup(config.ctypes.data, view.ctypes.data, main_iv.ctypes.data, ld.ctypes.data,
                                    vr.ctypes.data if vr is not None else None, int(tick), int(cache_valid), views.ctypes.data)
    if rc != OK:
        raise ChordvisError("chordvis_cascade_setup -> %d" % rc)
    return views


class BuiltAsset:
    """chordvis_nanite_build: meshlets / groups / BVH of a triangle mesh as numpy copies (the native object is freed)."""

    def __init__(self, handle):
        ad, prim, lods = R.AssetDesc(), np.zeros(1, dtype=R.PRIMITIVE), C.c_uint32(0)
        rc = lib.chordvis_built_asset_desc(handle, C.byref(ad), prim.ctypes.data, C.byref(lods))
        if rc != OK:
            raise ChordvisError("chordvis_built_asset_desc -> %d" % rc)

        def arr(ptr, n, dt):
            return np.frombuffer((C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy() if n else np.zeros(0, dtype=dt)
        self.meshlets = arr(ad.meshlets, ad.meshletCount, R.MESHLET)
        self.groups = arr(ad.meshletGroups, ad.meshletGroupCount, R.MESHLET_GROUP)
        self.group_indices = arr(ad.meshletGroupIndices, ad.meshletGroupIndexCount, np.uint32)
        self.meshlet_data = arr(ad.meshletData, ad.meshletDataCount, np.uint32)
        self.positions = arr(ad.positions, ad.vertexCount * 3, np.float32).reshape(-1, 3)
        self.texcoord0 = arr(ad.texcoord0, ad.texcoord0Count * 2, np.float32).reshape(-1, 2) if ad.texcoord0 else None
        self.bvh_nodes = arr(ad.bvhNodes, ad.bvhNodeCount, R.BVH_NODE)
        self.primitive = prim
        self.lod_count = lods.value


def nanite_build(positions, indices, texcoord0=None, keep_handle=False):
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    uv = None if texcoord0 is None else np.ascontiguousarray(texcoord0, dtype=np.float32).reshape(-1, 2)
    h = C.c_void_p()
    rc = lib.chordvis_nanite_build(pos.ctypes.data, len(pos), idx.ctypes.data, len(idx), uv.ctypes.data if uv is not None else None, C.byref(h))
    if rc != OK:
        raise ChordvisError("chordvis_nanite_build -> %d" % rc)
    if keep_handle:
        return h
    try:
        return BuiltAsset(h)
    finally:
        lib.chordvis_free_built_asset(h)

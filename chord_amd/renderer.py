"""Host-side mirror of the reference's pass surface over the C ABI (libchordvis.so).

Names follow source/renderer/mesh/gltf_rendering.h:37-108 and postprocessing.h:41-54:
instance_culling, hzb_culling, render_mesh, visibility_stage0/1, build_hzb, and render_frame for
the hot segment of DeferredRenderer::render (renderer.cpp:315-345).  Everything here only forwards
to the HIP library; there is no CPU implementation of any pass in this package.
"""
import ctypes as C

import numpy as np

from . import lib as L
from . import records as R


class VisibilityRenderer:
    """One device context (graphics::Context + DeferredRenderer state for this path)."""

    def __init__(self, device=0, stream=None, _borrowed_ctx=None):
        self._borrowed = _borrowed_ctx is not None
        if self._borrowed:
            self._ctx = C.c_void_p(_borrowed_ctx)             # a rank of a VisibilityGroup: the group owns it
        else:
            self._ctx = C.c_void_p()
            rc = L.lib.chordvis_create(device, stream, C.byref(self._ctx))
            if rc != L.OK:
                raise L.ChordvisError(
                    "chordvis_create(device=%d) failed with %d: no usable HIP device (the product path has no CPU fallback)" % (device, rc))
        self.width = self.height = 0
        self.scene = None

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != L.OK:
            msg = L.lib.chordvis_last_error(self._ctx)
            raise L.ChordvisError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def close(self):
        if self._ctx and not self._borrowed:
            L.lib.chordvis_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(L.lib.chordvis_sync(self._ctx), "sync")

    # -- scene / view ---------------------------------------------------------------------------
    def upload_scene(self, scene):
        self.scene = scene
        self._check(L.lib.chordvis_upload_scene(self._ctx, C.byref(scene.desc)), "upload_scene")

    def update_objects(self, objects):
        objects = np.ascontiguousarray(objects, dtype=R.OBJECT)
        self._check(L.lib.chordvis_update_objects(self._ctx, objects.ctypes.data, len(objects)), "update_objects")

    def bind_objects(self, device_ptr, count):
        self._check(L.lib.chordvis_bind_objects(self._ctx, device_ptr, count), "bind_objects")

    def allocate_gbuffer(self, width, height, device_visibility=None):
        self._check(L.lib.chordvis_allocate_gbuffer(self._ctx, width, height, device_visibility), "allocate_gbuffer")
        self.width, self.height = width, height

    def set_cull_mode(self, hierarchical):
        """0: flat group cull (the reference's dispatch); 1: walk the primitives' BVHs (same command list)."""
        self._check(L.lib.chordvis_set_cull_mode(self._ctx, int(hierarchical)), "set_cull_mode")

    def set_tile_schedule_keep(self, frames):
        """Non-zero (default 1): a raster pass runs under the tile schedule the frame before made for it; 0: a fresh schedule every pass."""
        self._check(L.lib.chordvis_set_tile_schedule_keep(self._ctx, int(frames)), "set_tile_schedule_keep")

    def tile_schedule_keep(self):
        return int(L.lib.chordvis_tile_schedule_keep(self._ctx))

    def set_shard(self, ranks, rank):
        """Screen ownership by 64 x 64 tiles: the default map (compact regions along a generalised Hilbert curve)."""
        self._check(L.lib.chordvis_set_shard(self._ctx, ranks, rank), "set_shard")

    def set_tile_owners(self, owners):
        """An explicit tile map (one owner per tile); None: the default map."""
        if owners is None:
            self._check(L.lib.chordvis_set_tile_owners(self._ctx, None, 0), "set_tile_owners")
            return
        owners = np.ascontiguousarray(owners, dtype=np.uint8)
        self._check(L.lib.chordvis_set_tile_owners(self._ctx, owners.ctypes.data, len(owners)), "set_tile_owners")

    def tile_owners(self):
        out = np.zeros(int(L.lib.chordvis_tile_count(self.width, self.height)), dtype=np.uint8)
        self._check(L.lib.chordvis_get_tile_owners(self._ctx, out.ctypes.data, len(out)), "get_tile_owners")
        return out

    def read_tile_loads(self):
        """Bin entries per tile of the last frame (every rank's tiles, after the end-of-frame exchange)."""
        out = np.zeros(int(L.lib.chordvis_tile_count(self.width, self.height)), dtype=np.uint32)
        self._check(L.lib.chordvis_read_tile_loads(self._ctx, out.ctypes.data, len(out)), "read_tile_loads")
        return out

    def rebalance(self):
        """Tile map from the last frame's loads; returns the old map's heaviest rank / mean."""
        imb = C.c_uint32(0)
        self._check(L.lib.chordvis_rebalance(self._ctx, C.byref(imb)), "rebalance")
        return imb.value / 1000.0

    def set_view(self, view, instance_view, flags):
        self._views = (view, instance_view)       # keep alive
        self._check(L.lib.chordvis_set_view(self._ctx, view.ctypes.data, instance_view.ctypes.data, flags), "set_view")

    # -- buffers for collectives ------------------------------------------------------------------
    def visibility_words(self):
        return int(L.lib.chordvis_visibility_words(self._ctx))

    def visibility_chunk_words(self):
        return int(L.lib.chordvis_visibility_chunk_words(self._ctx))

    def visibility_ptr(self):
        return L.lib.chordvis_visibility_ptr(self._ctx)

    def resolved_visibility_ptr(self):
        return L.lib.chordvis_resolved_visibility_ptr(self._ctx)

    def hzb_exchange(self):
        """(device pointer, halves, halves per rank) of the mid-frame exchange buffer."""
        return (L.lib.chordvis_hzb_exchange_ptr(self._ctx), int(L.lib.chordvis_hzb_exchange_halves(self._ctx)),
                int(L.lib.chordvis_hzb_exchange_chunk_halves(self._ctx)))

    def hzb_final_exchange(self):
        """(device pointer, bytes per rank) of the end-of-frame exchange buffer."""
        return (L.lib.chordvis_hzb_final_exchange_ptr(self._ctx), int(L.lib.chordvis_hzb_final_exchange_chunk_bytes(self._ctx)))

    # -- passes -------------------------------------------------------------------------------------
    def clear_gbuffer(self):
        self._check(L.lib.chordvis_clear_gbuffer(self._ctx), "clear_gbuffer")

    def instance_culling(self):
        out = L.CountAndCmd()
        self._check(L.lib.chordvis_instance_culling(self._ctx, C.byref(out)), "instance_culling")
        return out

    def hzb_culling(self, hzb, first_stage, in_list):
        vis, rej = L.CountAndCmd(), L.CountAndCmd()
        self._check(L.lib.chordvis_hzb_culling(self._ctx, C.byref(hzb), int(first_stage), in_list, C.byref(vis), C.byref(rej)), "hzb_culling")
        return vis, rej

    def render_mesh(self, in_list):
        self._check(L.lib.chordvis_render_mesh(self._ctx, in_list), "render_mesh")

    def visibility_stage0(self, hzb_prev, in_list):
        rej, flag = L.CountAndCmd(), C.c_int(0)
        self._check(L.lib.chordvis_visibility_stage0(self._ctx, C.byref(hzb_prev) if hzb_prev is not None else None,
                                                      in_list, C.byref(rej), C.byref(flag)), "visibility_stage0")
        return bool(flag.value), rej

    def visibility_stage1(self, hzb, in_list):
        self._check(L.lib.chordvis_visibility_stage1(self._ctx, C.byref(hzb), in_list), "visibility_stage1")

    def build_hzb(self, build_min=True, build_max=False, build_valid_range=False, slot=0):
        out = L.HZB()
        self._check(L.lib.chordvis_build_hzb(self._ctx, int(build_min), int(build_max), int(build_valid_range), slot, C.byref(out)), "build_hzb")
        return out

    def render_frame(self):
        self._check(L.lib.chordvis_render_frame(self._ctx), "render_frame")

    def frame_phase_cull(self):
        """Sharded group cull, first half: this rank's share of the group tests -> rank-mask words (all-gather cull_exchange() next)."""
        self._check(L.lib.chordvis_frame_phase_cull(self._ctx), "frame_phase_cull")

    def cull_exchange(self):
        """(device pointer, bytes per rank) of the sharded cull's exchange buffer; (None, 0) when the sharded cull does not apply."""
        return (L.lib.chordvis_cull_exchange_ptr(self._ctx), int(L.lib.chordvis_cull_exchange_chunk_bytes(self._ctx)))

    def debug_fill_cull_exchange(self):
        """Measurement aid: every rank's chunk of the cull exchange buffer computed on this context (one rank timed alone)."""
        self._check(L.lib.chordvis_debug_fill_cull_exchange(self._ctx), "debug_fill_cull_exchange")

    def frame_phase_a(self):
        self._check(L.lib.chordvis_frame_phase_a(self._ctx), "frame_phase_a")

    def frame_phase_b(self):
        self._check(L.lib.chordvis_frame_phase_b(self._ctx), "frame_phase_b")

    def frame_phase_c(self):
        self._check(L.lib.chordvis_frame_phase_c(self._ctx), "frame_phase_c")

    # -- one process per GPU: RCCL communicator owned by the library ---------------------------------
    def comm_init_rank(self, nranks, rank, unique_id):
        """Attach an RCCL communicator (after set_shard); render_frame() then runs the two all-gathers itself."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._check(L.lib.chordvis_comm_init_rank(self._ctx, nranks, rank, buf), "comm_init_rank")

    def comm_set_pipelined(self, unique_id):
        """Second communicator (its own unique id, or None to switch off): the image of frame i travels beside frame i + 1."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id)) if unique_id is not None else None
        self._check(L.lib.chordvis_comm_set_pipelined(self._ctx, buf), "comm_set_pipelined")

    def comm_destroy(self):
        self._check(L.lib.chordvis_comm_destroy(self._ctx), "comm_destroy")

    def comm_info(self):
        return comm_info(self._ctx)

    def reset_history(self):
        self._check(L.lib.chordvis_reset_history(self._ctx), "reset_history")

    def last_frame_cmds(self):
        out = L.CountAndCmd()
        self._check(L.lib.chordvis_last_frame_cmds(self._ctx, C.byref(out)), "last_frame_cmds")
        return out

    def history_hzb(self):
        out = L.HZB()
        self._check(L.lib.chordvis_history_hzb(self._ctx, C.byref(out)), "history_hzb")
        return out

    def upload_history_hzb(self, hzb_min):
        hzb_min = np.ascontiguousarray(hzb_min, dtype=np.uint16)
        self._check(L.lib.chordvis_upload_history_hzb(self._ctx, hzb_min.ctypes.data), "upload_history_hzb")

    def set_limits(self, max_triangle_records=0, bin_pool_chunks=0, bin_max_chunks_per_tile=0):
        """Work-list capacities (before upload_scene / allocate_gbuffer); 0 keeps a default."""
        lim = L.Limits(int(max_triangle_records), int(bin_pool_chunks), int(bin_max_chunks_per_tile))
        self._check(L.lib.chordvis_set_limits(self._ctx, C.byref(lim)), "set_limits")

    # -- depth-only views (renderShadow's passes, mesh_raster.cpp:331-546) ------------------------------------
    def allocate_depth_views(self, dim, view_count):
        self._check(L.lib.chordvis_allocate_depth_views(self._ctx, dim, view_count), "allocate_depth_views")

    def set_instance_views(self, views):
        views = np.ascontiguousarray(views, dtype=R.INSTANCE_CULLING_VIEW)
        self._check(L.lib.chordvis_set_instance_views(self._ctx, views.ctypes.data, len(views)), "set_instance_views")

    def instance_culling_view(self, view_offset):
        out = L.CountAndCmd()
        self._check(L.lib.chordvis_instance_culling_view(self._ctx, view_offset, C.byref(out)), "instance_culling_view")
        return out

    def hzb_culling_generic(self, hzb, extent_scale, view_offset, use_last_frame, in_list):
        out = L.CountAndCmd()
        self._check(L.lib.chordvis_hzb_culling_generic(self._ctx, C.byref(hzb), extent_scale, view_offset, int(use_last_frame), in_list, C.byref(out)),
                    "hzb_culling_generic")
        return out

    def render_mesh_depth(self, view_offset, in_list, depth_clamped=True, bias_const=0.0, bias_slope=0.0):
        out = L.DepthTarget()
        self._check(L.lib.chordvis_render_mesh_depth(self._ctx, view_offset, int(depth_clamped), bias_const, bias_slope, in_list, C.byref(out)),
                    "render_mesh_depth")
        return out

    def build_hzb_from_depth(self, depth):
        out = L.HZB()
        self._check(L.lib.chordvis_build_hzb_from_depth(self._ctx, C.byref(depth), C.byref(out)), "build_hzb_from_depth")
        return out

    def read_depth(self, depth):
        out = np.empty(depth.width * depth.height, dtype=np.float32)
        self._check(L.lib.chordvis_readback_depth(self._ctx, C.byref(depth), out.ctypes.data), "readback_depth")
        return out

    def render_shadow(self, config, light_dir, tick, valid_range=None, hzb_culling=True):
        """renderShadow: returns (depth targets, views, mask of the cascades re-rendered by this call)."""
        n = int(config["cascadeCount"][0])
        depths = (L.DepthTarget * n)()
        views = np.zeros(n, dtype=R.INSTANCE_CULLING_VIEW)
        ld = np.asarray(light_dir, dtype=np.float32)
        vr = None if valid_range is None else np.asarray(valid_range, dtype=np.uint32)
        mask = C.c_uint32(0)
        self._check(L.lib.chordvis_render_shadow(self._ctx, config.ctypes.data, ld.ctypes.data, vr.ctypes.data if vr is not None else None,
                                                 int(tick), int(hzb_culling), depths, views.ctypes.data, C.byref(mask)), "render_shadow")
        return list(depths), views, mask.value

    def depth_view_stats(self):
        st = L.Stats()
        self._check(L.lib.chordvis_depth_view_stats(self._ctx, C.byref(st)), "depth_view_stats")
        return st.as_dict()

    # -- consumers' first step (visibility_tile.cpp) -------------------------------------------------------
    def visibility_mark(self, drawed_meshlet_cmd=None):
        """visibilityMark (visibility_tile.cpp:20-57); the command list defaults to last_frame_cmds()."""
        out = L.TileMarker()
        cmd = drawed_meshlet_cmd if drawed_meshlet_cmd is not None else self.last_frame_cmds()
        self._check(L.lib.chordvis_visibility_mark(self._ctx, cmd, C.byref(out)), "visibility_mark")
        return out

    def wait_visibility(self, stream=None):
        """Orders `stream` (default: the context's) behind the completion of the last frame's resolved image."""
        self._check(L.lib.chordvis_wait_visibility(self._ctx, stream), "wait_visibility")

    def prepare_shading_tile_param(self, shading_type, marker):
        """prepareShadingTileParam (visibility_tile.cpp:59-110)."""
        out = L.ShadingTiles()
        self._check(L.lib.chordvis_prepare_shading_tile_param(self._ctx, int(shading_type), C.byref(marker), C.byref(out)),
                    "prepare_shading_tile_param")
        return out

    def read_tile_marker(self, marker):
        out = np.zeros((marker.markerDim[1], marker.markerDim[0], 4), dtype=np.uint32)
        self._check(L.lib.chordvis_readback_tile_marker(self._ctx, C.byref(marker), out.ctypes.data), "readback_tile_marker")
        return out

    def read_shading_tiles(self, tiles):
        out = np.zeros((max(1, tiles.capacity), 2), dtype=np.uint32)
        n = C.c_uint32(0)
        args = np.zeros(4, dtype=np.uint32)
        self._check(L.lib.chordvis_readback_shading_tiles(self._ctx, C.byref(tiles), out.ctypes.data, tiles.capacity,
                                                           C.byref(n), args.ctypes.data), "readback_shading_tiles")
        return out[:n.value].copy(), args

    # -- readback -------------------------------------------------------------------------------------
    def read_visibility(self):
        out = np.empty(self.width * self.height, dtype=np.uint64)
        self._check(L.lib.chordvis_readback_visibility(self._ctx, out.ctypes.data), "readback_visibility")
        return out

    def read_previous_visibility(self):
        """Pipelined sharded frames: the image of the frame before the last submitted one."""
        out = np.zeros(self.width * self.height, dtype=np.uint64)
        self._check(L.lib.chordvis_readback_previous_visibility(self._ctx, out.ctypes.data), "readback_previous_visibility")
        return out

    def read_cmds(self, handle):
        n = C.c_uint32(0)
        self._check(L.lib.chordvis_readback_cmds(self._ctx, handle, None, 0, C.byref(n)), "readback_cmds")
        out = np.zeros(max(1, n.value), dtype=R.DRAW_CMD)
        self._check(L.lib.chordvis_readback_cmds(self._ctx, handle, out.ctypes.data, len(out), C.byref(n)), "readback_cmds")
        return out[:n.value].copy()

    def read_hzb(self, hzb):
        n = hzb.desc.totalTexels
        mn = np.zeros(n, dtype=np.uint16)
        mx = np.zeros(n, dtype=np.uint16) if hzb.maxTexels else None
        rng = np.zeros(2, dtype=np.uint32) if hzb.validRange else None
        self._check(L.lib.chordvis_readback_hzb(self._ctx, C.byref(hzb), mn.ctypes.data,
                                                mx.ctypes.data if mx is not None else None,
                                                rng.ctypes.data if rng is not None else None), "readback_hzb")
        return mn, mx, rng

    def enable_timers(self, mode=1, period=1):
        """0 off, 1 last frame, 2 accumulate until stats(); only every `period`-th frame is stamped."""
        self._check(L.lib.chordvis_enable_timers(self._ctx, int(mode) | (int(period) << 8)), "enable_timers")

    def last_error(self):
        """Text of the last error this context reported (chordvis_last_error)."""
        msg = L.lib.chordvis_last_error(self._ctx)
        return msg.decode(errors="replace") if msg else ""

    def set_debug(self, flags):
        """Measurement-only ablation switches (0 = production)."""
        self._check(L.lib.chordvis_set_debug(self._ctx, int(flags)), "set_debug")

    def stats(self):
        st = L.Stats()
        self._check(L.lib.chordvis_stats(self._ctx, C.byref(st)), "stats")
        return st.as_dict()


def decode_visibility(vis):
    """(depth float32, slot int64 (-1 = empty), triangle uint8) from packed words — base.hlsli:437-447."""
    vis = np.asarray(vis, dtype=np.uint64)
    depth = (vis >> np.uint64(32)).astype(np.uint32).view(np.float32)
    low = (vis & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    tri = (low & 0xFF).astype(np.uint8)
    slot = ((low >> 8) & 0xFFFFFF).astype(np.int64) - 1
    return depth, slot, tri


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0; the host distributes the 128 bytes)."""
    buf = (C.c_char * 128)()
    rc = L.lib.chordvis_comm_unique_id(buf)
    if rc != L.OK:
        raise L.ChordvisError("chordvis_comm_unique_id failed (%d): librccl not loadable" % rc)
    return bytes(buf.raw)


def comm_info(ctx=None):
    ver, n = C.c_int(0), C.c_uint32(0)
    origin = C.create_string_buffer(256)
    rc = L.lib.chordvis_comm_info(ctx, C.byref(ver), C.byref(n), origin, 256)
    if rc != L.OK:
        raise L.ChordvisError("chordvis_comm_info failed (%d)" % rc)
    return {"nccl_version_code": ver.value, "ranks": n.value, "library": origin.value.decode()}


class VisibilityGroup:
    """ChordGroup: one process, n devices, one call per frame; the library issues the exchanges (direct peer copies)."""

    def __init__(self, devices):
        devices = list(devices)
        arr = (C.c_int * len(devices))(*devices)
        self._g = C.c_void_p()
        rc = L.lib.chordvis_create_group(len(devices), arr, C.byref(self._g))
        if rc != L.OK:
            raise L.ChordvisError("chordvis_create_group(%r) failed with %d" % (devices, rc))
        self.size = len(devices)
        self.ranks = [VisibilityRenderer(_borrowed_ctx=L.lib.chordvis_group_ctx(self._g, r)) for r in range(self.size)]

    def _check(self, rc, what):
        if rc != L.OK:
            msg = L.lib.chordvis_group_last_error(self._g)
            raise L.ChordvisError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def close(self):
        if self._g:
            for r in self.ranks:
                r.close()
            L.lib.chordvis_destroy_group(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_limits(self, max_triangle_records=0, bin_pool_chunks=0, bin_max_chunks_per_tile=0):
        lim = L.Limits(int(max_triangle_records), int(bin_pool_chunks), int(bin_max_chunks_per_tile))
        self._check(L.lib.chordvis_group_set_limits(self._g, C.byref(lim)), "group_set_limits")

    def upload_scene(self, scene):
        self.scene = scene
        self._check(L.lib.chordvis_group_upload_scene(self._g, C.byref(scene.desc)), "group_upload_scene")

    def allocate_gbuffer(self, width, height):
        self._check(L.lib.chordvis_group_allocate_gbuffer(self._g, width, height), "group_allocate_gbuffer")
        for r in self.ranks:
            r.width, r.height = width, height

    def rebalance(self):
        """chordvis_group_rebalance: every rank's tile map from the last frame's loads; returns the old map's heaviest rank / mean."""
        imb = C.c_uint32(0)
        self._check(L.lib.chordvis_group_rebalance(self._g, C.byref(imb)), "group_rebalance")
        return imb.value / 1000.0

    def update_objects(self, objects):
        objects = np.ascontiguousarray(objects, dtype=R.OBJECT)
        self._check(L.lib.chordvis_group_update_objects(self._g, objects.ctypes.data, len(objects)), "group_update_objects")

    def set_view(self, view, instance_view, flags):
        self._views = (view, instance_view)
        self._check(L.lib.chordvis_group_set_view(self._g, view.ctypes.data, instance_view.ctypes.data, flags), "group_set_view")

    def render_frame(self):
        self._check(L.lib.chordvis_group_render_frame(self._g), "group_render_frame")

    def sync(self):
        self._check(L.lib.chordvis_group_sync(self._g), "group_sync")

    def enqueue_ms(self):
        """Mean host time per frame of every rank's worker inside render_frame since the last call."""
        out = (C.c_double * self.size)()
        self._check(L.lib.chordvis_group_enqueue_ms(self._g, out, self.size), "group_enqueue_ms")
        return [float(v) for v in out]

    def set_pipelined(self, enable=True):
        """The visibility all-gather of frame i travels beside frame i + 1 (chordvis_group_set_pipelined)."""
        self._check(L.lib.chordvis_group_set_pipelined(self._g, 1 if enable else 0), "group_set_pipelined")

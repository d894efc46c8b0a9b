#!/usr/bin/env python3
"""bench.py — Gtri/s of the visibility hot path on N MI355X (one process per GPU).

A "step" is one frame of the hot path (clear -> instanceCulling -> stage 0 [HZB phase 0 + raster]
-> buildHZB -> stage 1 [HZB phase 1 + raster] -> buildHZB(min,max,range)) over a synthetic meshlet
scene already resident in HBM.  Consecutive steps alternate between two camera positions 0.5 m
apart so that every frame culls against the HZB of a *different* previous frame (BASELINE config 3:
"two frames, camera advanced 0.5 m").

  N = 1   workload "street_4k_hzb"  BASELINE config 3 (Bistro-class, 3840x2160, two-pass HZB)
  N > 1   workload "subpixel_1g"    BASELINE config 5 (1.07 G sub-pixel triangles per frame), the screen sharded by 64 x 64 tiles
          across the ranks, the tiles' HZB texels (two-pass workloads) + visibility words reassembled with all-gathers
          issued by the library itself over RCCL (chordvis_comm_init_rank: --exchange lib), by torch.distributed on the
          context's stream (--exchange torch), or by ONE process driving all N devices with peer copies (ChordGroup:
          --exchange group).  `auto` tries them in that order, so a node whose RCCL does not come up still yields a curve;
          the line says which one ran (`exchange`) and why the others did not (`exchange_fallbacks`).  The control plane
          (barriers, the communicator id, the max over ranks) is a gloo group: it does not depend on RCCL.  Every N > 1
          line carries the same workload rendered unsharded on rank 0's GPU (single_gpu_same_workload, speedup_vs_single)
          and, per rank, the GPU time of every phase and exchange of the frame (`phases_ms`).  --workload overrides either
          default, so the N = 1 point of any curve can be re-run on the N > 1 workload.  One N > 1 line holds BOTH protocols -- the top
          level is the unpipelined frame, `pipelined` the same frames with the image gather beside the next frame (its own ms_per_step,
          value, speedup_vs_single) --, the figure under the default tile map beside the re-balanced one (`default_map`), the achieved
          GB/s of every exchange (`exchange_gbs`, `rccl_schedule_ok`; when the image gather over RCCL stays under 250 GB/s per rank the
          peer-copy transport is measured in the same run: `group_transport`), what bounds the frame (`bound`: launches x launch floor +
          exchanges), and `also` = BASELINE config 4 with all of the same.
  `python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one process per GPU).

value = triangles of the clusters submitted to the rasterizer per frame (post-cull, the unit of
SURVEY §8d) x steps / wall time of the timed region, max over ranks.  Prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLUSTER_BYTES = 64 + 12   # meshlet header + draw command; + 4*(V+T) + 12*V per cluster


def build_workload(name):
    from chord_amd import scenes
    if name == "street_4k_hzb":
        return scenes.config3_street(3840, 2160)
    if name == "street_4k_masked":       # config 3 with alpha-tested materials on every prop and every other building (same triangles)
        return scenes.config3_street(3840, 2160, masked=True)
    if name == "street_4k_masked_twin":  # ... and its opaque twin: the same two-sided materials without the alpha test (equal triangle count)
        return scenes.config3_street(3840, 2160, masked="twin")
    if name == "street_x64_4k_hzb":
        return scenes.config4_street_x64(3840, 2160, grid=8)
    if name == "street_x16_4k_hzb":
        return scenes.config4_street_x64(3840, 2160, grid=4)
    if name == "subpixel_1g":            # BASELINE config 5: 1 Mi unique ~8x8 px patches x 8 instances = 1.07 G triangles
        return scenes.config5_subpixel(3840, 2160)
    if name == "subpixel_64m":           # the same at 1/16 size (64 Ki patches x 8): fits the default work-list limits
        return scenes.config5_subpixel(3840, 2160, prims=64)
    if name == "subpixel_1g_hotspot":    # config 5, variant "hotspot" (SURVEY 8d): the same 1.07 G triangles, centres Gaussian (sigma 64 px) around the screen centre
        return scenes.config5_subpixel(3840, 2160, hotspot_sigma_px=64.0)
    if name == "subpixel_64m_hotspot":
        return scenes.config5_subpixel(3840, 2160, prims=64, hotspot_sigma_px=64.0)
    if name == "street_720p_hzb":        # config 3 at 1280x720: the small two-pass workload of the N > 1 protocol tests
        return scenes.config3_street(1280, 720)
    if name == "street_x64_720p_hzb":    # config 4 at 1280x720 (the protocol tests' stand-in for the `also` workload)
        return scenes.config4_street_x64(1280, 720, grid=8)
    if name == "atrium_1080p":
        return scenes.config2_atrium(1920, 1080)
    raise SystemExit("unknown workload %r" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", help="street_4k_hzb | street_4k_masked | street_x64_4k_hzb | street_x16_4k_hzb | subpixel_1g | subpixel_1g_hotspot | subpixel_64m | atrium_1080p")
    ap.add_argument("--exchange", default="auto", choices=("auto", "lib", "torch", "group"),
                    help="N > 1: who issues the all-gathers -- the library over RCCL, torch.distributed, or one process with N devices (ChordGroup, peer copies); auto = the first that works")
    ap.add_argument("--pipelined", action="store_true", help="(kept for old command lines: every N > 1 line now carries the pipelined protocol beside the unpipelined one)")
    ap.add_argument("--no-pipelined", action="store_true", help="N > 1: do not also measure the pipelined protocol")
    ap.add_argument("--no-group-fallback", action="store_true", help="N > 1, --exchange lib: do not measure the peer-copy transport when the RCCL image gather is slow")
    ap.add_argument("--no-also", action="store_true", help="N > 1, default workload: do not also measure BASELINE config 4 (street_x64_4k_hzb) for the line's `also` field")
    ap.add_argument("--no-rebalance", action="store_true", help="N > 1: keep the default tile map (compact regions of equal area) instead of re-balancing it from the warm-up frames' tile loads")
    ap.add_argument("--cpu-baseline-frames", type=int, default=48, help="oracle frames timed on the host (rank 0, N=1); 0 disables")
    ap.add_argument("--cull", default="flat", choices=("flat", "hierarchical"),
                    help="instanceCulling: the reference's flat group dispatch, or the BVH walk (same command list)")
    ap.add_argument("--no-hzb", action="store_true", help="disable HZB occlusion culling (frustum+cone only)")
    ap.add_argument("--debug-flags", type=int, default=0, help="raster ablation switches (measurement only; voids parity)")
    ap.add_argument("--no-stamps", action="store_true", help="no hipEvent stamps anywhere in the run (kernel traces: every frame of the process is then a product frame; the line's gpu_ms / roofline times are 0)")
    ap.add_argument("--no-path", action="store_true", help="N = 1: do not also time the moving camera path (`moving_path`)")
    ap.add_argument("--path-views", type=int, default=64, help="distinct views of the moving camera path (even)")
    ap.add_argument("--tile-schedule-keep", type=int, default=-1, help="frames a tile schedule is kept for (chordvis_set_tile_schedule_keep; -1: the library's default -- measurement runs only: the line says what ran)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_spawn(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    # test hooks (a 1-GPU box cannot host two RCCL ranks): CHORDVIS_BENCH_BACKEND=gloo + CHORDVIS_BENCH_ONE_DEVICE=1
    # run all ranks on device 0 with host-staged collectives, to exercise the N>1 control flow
    backend = os.environ.get("CHORDVIS_BENCH_BACKEND", "nccl")
    if os.environ.get("CHORDVIS_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # control plane on gloo (CPU): barriers, the communicator id and the max over ranks work whatever state RCCL is in
        import datetime
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=10))
    # Every kernel of the path AND every collective of the frame is enqueued on ONE explicit stream.  (PyTorch's
    # default stream has handle 0; handed to chordvis_create that reads as "no stream given" and the context would run
    # on a private non-blocking stream that nothing orders against the collectives.)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    from chord_amd import lib as L, records as R
    from chord_amd.renderer import VisibilityRenderer

    # ---- the workload of the line; N > 1 with the default workload (config 5) also measures BASELINE config 4, the configuration
    #      BASELINE.json names for 2/4/8 GPUs (a sub-millisecond frame that the exchanges dominate: DESIGN.md 6), in the same line
    env = dict(world=world, rank=rank, local_rank=local_rank, backend=backend, dev=dev, stream=stream)
    line = measure(args, args.workload, env)
    # (CHORDVIS_BENCH_ALSO: test hook -- a small second workload for the one-GPU protocol tests)
    also_wl = os.environ.get("CHORDVIS_BENCH_ALSO") or ("street_x64_4k_hzb" if args.workload == "auto" else None)
    if world > 1 and also_wl and not args.no_also:
        # (the second measurement must not cost the first its line: a failure here -- on any rank; the others leave their
        # collectives when the control plane times out -- is reported inside `also` instead)
        try:
            also = measure(args, also_wl, env)
            if line is not None and also is not None:
                line["also"] = {k: also[k] for k in ("config", "value", "unit", "ms_per_step", "steps", "timed_region_s", "exchange", "pipelined", "tile_map", "exchange_fallbacks", "default_map",
                                                       "exchange_gbs", "rccl_schedule_ok", "group_transport", "bound", "cull",
                                                       "phases_ms", "single_gpu_same_workload", "speedup_vs_single", "triangles_submitted_per_step", "gpu_ms") if k in also}
        except Exception as e:                      # noqa: BLE001
            if line is not None:
                line["also"] = {"config": {"workload": also_wl}, "error": "%s: %s" % (type(e).__name__, e)}
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:                           # noqa: BLE001  (the line is out; a peer that failed above cannot meet this barrier)
            pass
    return 0


def measure(args, workload, env):
    """One workload through the whole protocol of the bench (scene, renderer, exchange set-up, warm-up, timed region, roofline, references);
    returns the JSON line as a dict on rank 0, None on the other ranks."""
    world, rank, local_rank, backend, dev, stream = (env[k] for k in ("world", "rank", "local_rank", "backend", "dev", "stream"))
    args = argparse.Namespace(**vars(args))          # (per-workload switches below must not leak into the next measurement)
    from chord_amd import lib as L, records as R
    from chord_amd.renderer import VisibilityRenderer
    line = None
    wl = workload
    if wl == "auto":
        # N = 1: BASELINE config 3 (the largest single-GPU configuration).  N > 1: config 5, the multi-GPU stress
        # configuration (1 G sub-pixel triangles per frame): a frame long enough (40 ms on one GPU) for the
        # fixed cost of the all-gathers to amortise.  Config 4 (0.5 ms per frame on one GPU: less than the 66 MB
        # visibility all-gather alone) stays available as --workload street_x64_4k_hzb.  DESIGN.md 5 and 6.
        wl = "street_4k_hzb" if world == 1 else "subpixel_1g"
    scene, cam_a = build_workload(wl)
    f = np.array(cam_a.front, dtype=np.float64)
    f /= np.linalg.norm(f)
    cam_b = cam_a.moved(tuple(0.5 * f))
    W, H = cam_a.width, cam_a.height
    if wl.startswith("subpixel"):
        args.no_hzb = True               # config 5 is a single pass, frustum + cone only
    flags = R.FLAG_FRUSTUM_CULL | R.FLAG_CONE_CULL | (0 if args.no_hzb else R.FLAG_HZB_CULL)

    # per-view inputs (host, once): frame at A follows a frame at B and vice versa
    view_a0, _ = L.make_views(cam_a)
    view_b0, _ = L.make_views(cam_b)
    view_a, iv_a = L.make_views(cam_a, view_b0)
    view_b, iv_b = L.make_views(cam_b, view_a0)
    obj_a = L.fill_objects(scene, cam_a, cam_b).copy()
    obj_b = L.fill_objects(scene, cam_b, cam_a).copy()
    d_obj = [torch.from_numpy(o.view(np.uint8).reshape(-1)).to(dev) for o in (obj_a, obj_b)]
    views = [(view_a, iv_a), (view_b, iv_b)]

    assert stream.cuda_stream != 0
    r = VisibilityRenderer(local_rank, stream.cuda_stream)
    if wl.startswith("subpixel_1g"):     # ~1 G records of 48 B and as many bin entries in one pass (a rank holds 1/N of them)
        share = max(1, world // 2) if world > 1 else 1
        r.set_limits(max_triangle_records=(1152 << 20) // share, bin_pool_chunks=(1200 << 10) // share, bin_max_chunks_per_tile=2048)
    if args.cull == "hierarchical":
        r.set_cull_mode(1)
    if args.tile_schedule_keep >= 0:
        r.set_tile_schedule_keep(args.tile_schedule_keep)
    r.upload_scene(scene)
    if world > 1:
        r.set_shard(world, rank)                         # screen tiles, the default map (re-balanced after the warm-up, below)
    r.allocate_gbuffer(W, H)
    vis_t = vis_mine = None                              # (--exchange torch: a caller-owned visibility buffer, below)

    # ---- who issues the two all-gathers of a sharded frame -------------------------------------------------------
    #   lib    the library (RCCL communicator attached to the context: ONE call per frame, like the reference's host)
    #   torch  torch.distributed (an NCCL group) on the context's stream between chordvis_frame_phase_a/b/c
    #   group  ONE process (rank 0) drives all N devices through ChordGroup: peer copies, no RCCL at all
    exchange = "none"
    comm = None
    fallbacks = []
    nccl_pg = None

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def try_lib():
        nonlocal comm
        from chord_amd.renderer import comm_unique_id
        uid, err = [None], None
        if rank == 0:
            try:
                uid[0] = comm_unique_id()
            except Exception as e:                       # noqa: BLE001  (no librccl: every rank falls back together)
                err = "comm_unique_id: %s" % e
        dist.broadcast_object_list(uid, src=0)
        ok = uid[0] is not None
        if ok:
            try:
                r.comm_init_rank(world, rank, uid[0])
                comm = r.comm_info()
            except Exception as e:                       # noqa: BLE001
                ok, err = False, "rank %d comm_init_rank: %s" % (rank, e)
        if not all_agree(ok):
            if ok:
                r.comm_destroy()
            return err or "another rank could not attach its communicator"
        return None

    def try_torch():
        nonlocal nccl_pg
        if backend != "nccl":
            return None                                  # test hook: gloo, host-staged
        ok, err = True, None
        try:
            nccl_pg = dist.new_group(backend="nccl")
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe, group=nccl_pg)
            torch.cuda.synchronize(dev)
            ok = float(probe.item()) == float(world)
        except Exception as e:                           # noqa: BLE001
            ok, err = False, "rank %d nccl group: %s" % (rank, e)
        if not all_agree(ok):
            return err or "another rank's NCCL group failed"
        return None

    group = None
    if world > 1:
        order = {"auto": ["lib", "torch", "group"], "lib": ["lib"], "torch": ["torch"], "group": ["group"]}[args.exchange]
        if backend != "nccl":
            order = [m for m in order if m != "lib"] or ["torch"]      # (test hook: no RCCL ranks on one device)
        for mode in order:
            why = try_lib() if mode == "lib" else try_torch() if mode == "torch" else None
            if why is None:
                exchange = mode
                break
            fallbacks.append({mode: why})
            if rank == 0:
                print("[bench] exchange %r unavailable (%s)" % (mode, why), file=sys.stderr)
        if exchange == "none":
            raise SystemExit("--exchange %s: no exchange could be set up: %r" % (args.exchange, fallbacks))
    if exchange == "group":
        r.close()                                        # (the group makes its own contexts, one per device)
        return run_group(args, wl, scene, views, (obj_a, obj_b), flags, W, H, world, rank, dev, stream, fallbacks)

    if exchange == "torch":
        words = r.visibility_words()
        vis_t = torch.zeros(words, dtype=torch.int64, device=dev)      # caller-owned visibility (all-gather target)
        r.allocate_gbuffer(W, H, vis_t.data_ptr())

    def vis_views():
        # (a rank's chunk is as many tile slots as the largest rank owns under the current map)
        chunk = r.visibility_chunk_words()
        return vis_t[:world * chunk], vis_t[rank * chunk:(rank + 1) * chunk]

    def all_gather(full, mine):
        if backend == "nccl":
            dist.all_gather_into_tensor(full, mine, group=nccl_pg)      # ordered on the current stream (= the context's)
        else:                                            # test hook (gloo has no device all-gather): staged through the host
            torch.cuda.current_stream(dev).synchronize()
            parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
            dist.all_gather(parts, mine.cpu())
            full.copy_(torch.cat(parts).to(dev))

    def frame(i):
        v, iv = views[i & 1]
        r.bind_objects(d_obj[i & 1].data_ptr(), len(scene.objects))
        r.set_view(v, iv, flags)
        if world == 1 or exchange == "lib":
            r.render_frame()
        else:
            if ex.cull is not None:                      # the sharded group cull: this rank's share of the tests, then everybody's rank masks
                r.frame_phase_cull()
                all_gather(ex.cull, ex.cull_mine)
            r.frame_phase_a()
            ex.all_gather_hzb()
            r.frame_phase_b()
            ex.all_gather_final()
            all_gather(*ex.vis)
            r.frame_phase_c()

    class Exchange:
        """all-gathers of the context-owned HZB exchange buffers (per tile slot the tile's HZB texels, rank-major) and of the image."""
        def __init__(self):
            self.remap()

        def remap(self):
            # raw bytes: RCCL has no 16-bit integer type and the payload is opaque f16 bits anyway
            ptr, halves, chunk_h = r.hzb_exchange()
            self.mid = _tensor_from_ptr(ptr, halves * 2, torch.uint8, dev)
            self.mid_mine = self.mid[rank * chunk_h * 2:(rank + 1) * chunk_h * 2]
            fptr, fbytes = r.hzb_final_exchange()
            self.fin = _tensor_from_ptr(fptr, fbytes * world, torch.uint8, dev)
            self.fin_mine = self.fin[rank * fbytes:(rank + 1) * fbytes]
            self.vis = vis_views() if vis_t is not None else None
            cptr, cbytes = r.cull_exchange()
            self.cull = self.cull_mine = None
            if cptr and cbytes and args.cull == "flat" and not os.environ.get("CHORDVIS_BENCH_REPLICATED_CULL"):
                self.cull = _tensor_from_ptr(cptr, cbytes * world, torch.uint8, dev)
                self.cull_mine = self.cull[rank * cbytes:(rank + 1) * cbytes]

        def all_gather_hzb(self):
            if not args.no_hzb:                          # (a frame without stage 1 has nothing to exchange)
                all_gather(self.mid, self.mid_mine)

        def all_gather_final(self):
            all_gather(self.fin, self.fin_mine)

    ex = Exchange() if (world > 1 and exchange == "torch") else None
    if args.debug_flags:
        r.set_debug(args.debug_flags)

    def timed(n, first=0):
        """n frames between barrier + synchronize on both sides; (seconds of this rank, max over ranks)"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        a = time.perf_counter()
        for k in range(n):
            frame(first + k)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        mine_s = time.perf_counter() - a
        if world > 1:
            t = torch.tensor([mine_s], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return mine_s, float(t.item())
        return mine_s, mine_s

    # ---- warm-up (untimed): also collects the deterministic per-view counts ---------------------
    r.enable_timers(0)
    per_view = [None, None]
    # (a short warm-up leaves the GPU below its clocks: the W frames asked for, then -- still untimed -- frames until 60 ms of work
    # have gone through, in pairs so that the timed region starts on view A)
    warm = max(args.warmup, 4) + (max(args.warmup, 4) & 1)
    w0 = time.perf_counter()
    i = 0
    while i < warm or (world == 1 and time.perf_counter() - w0 < 0.06 and i < 4000):
        frame(i)
        frame(i + 1)
        i += 2
        if i % 64 == 0 or i >= warm:
            torch.cuda.synchronize(dev)
    warm = i
    for i in range(2):
        frame(i)
        per_view[i & 1] = r.stats()
    # ---- N > 1: the tile map re-balanced from the loads of the last warm-up frame (every rank holds every tile's load after
    #      the end-of-frame exchange and computes the same map), then two more untimed frames under the new map
    tile_map = None
    default_map = None
    if world > 1:
        imb_before = 1.0
        if not args.no_rebalance:
            # the same frames under the DEFAULT map first (compact regions of equal area): the re-balanced map below is made from the
            # loads of exactly the two views the timed region renders, which a moving camera would not grant it
            nd = max(2, min(args.steps, 20)) & ~1
            _, dm = timed(nd)
            default_map = {"steps": nd, "ms_per_step": round(dm / nd * 1e3, 4)}
            imb_before = r.rebalance()
            if ex is not None:
                ex.remap()
            for i in range(4):
                frame(i)
                per_view[i & 1] = r.stats()
        owners = r.tile_owners()
        loads = r.read_tile_loads().astype(np.float64)
        per = np.bincount(owners, weights=loads, minlength=world)
        tile_map = {"rebalanced": not args.no_rebalance, "entries_max_over_mean_default_map": round(imb_before, 3),
                    "entries_max_over_mean": round(float(per.max() / max(per.mean(), 1.0)), 3),
                    "tiles_per_rank": np.bincount(owners, minlength=world).tolist(), "chunk_slots": r.visibility_chunk_words() // 4096}
    chunks = None
    if world > 1:
        chunks = {"cull": r.cull_exchange()[1] if args.cull == "flat" else 0, "hzb_mid": r.hzb_exchange()[2] * 2,
                  "final": r.hzb_final_exchange()[1], "image": r.visibility_chunk_words() * 8}
    tris_per_pair = per_view[0]["trianglesSubmitted"] + per_view[1]["trianglesSubmitted"]
    clusters_per_pair = sum(pv["countStage0Visible"] + pv["countStage1Visible"] for pv in per_view)

    # ---- timed region --------------------------------------------------------------------------
    # GPU timestamps (hipEvent on the launch stream): each event record is a barrier packet that costs 3-5 us (the kernel behind it is
    # not dispatched under the kernel in front), 15 per frame make a stamped frame ~28 % longer than a product frame.  N = 1: the
    # stamped frames -- 8, or 16 behind runs of 64 steps or more -- are rendered right behind the timed region (same frames, same
    # state, same stream); N > 1 (millisecond frames) stamps every 8th step inside it: `stamped_inside_timed_region` says which it was.
    # (N = 1, round 6: NEVER inside the timed region -- one stamped frame in eight made a 200-step run 3.5 % slower than a 20-step one:
    # 25 frames x 50 us of records in 35 ms; tools/host_time.py: the same 200 frames without a record run in 166.7 us each)
    stamp_inside = world > 1 and not args.no_stamps
    r.enable_timers(2 if stamp_inside else 0, period=8)
    if world == 1:
        for i in range(16):                          # (the count read-backs above left the device idle: a few more untimed frames ahead of the clock)
            frame(i)
    rank_elapsed, elapsed = timed(args.steps)
    # The per-kernel averages behind `roofline` come from hipEvent stamps on the launch stream.  Inside the timed region only every
    # 8th step is stamped (a stamped frame is ~20 % longer); a short run (the driver's 20 steps: 2-3 stamped frames) is topped
    # up to at least 8 stamped frames right after it -- same frames, same state, outside the clock.
    stamped_in_region = (args.steps + 7) // 8 if stamp_inside else 0
    extra = 0
    st = r.stats()                                  # per-frame GPU timestamps averaged over the stamped steps of the timed region
    if stamped_in_region < 8 and world == 1 and not args.no_stamps:
        r.enable_timers(2, period=1)                # (restarts the accumulation)
        extra = (16 if args.steps >= 64 else 8) - stamped_in_region
        for i in range(extra):
            frame(args.steps + i)
        torch.cuda.synchronize(dev)
        st2 = r.stats()
        for k in list(st):                          # the ms* fields are means over stamped frames: weighted mean of the two sets
            if (k.startswith("ms") or k == "stampsPerFrame") and isinstance(st[k], float):
                st[k] = (st[k] * stamped_in_region + st2[k] * extra) / (stamped_in_region + extra)
            elif not stamped_in_region:
                st[k] = st2[k]
    ms_per_step = elapsed / args.steps * 1e3
    # launches of a frame between two tile schedules and of a frame that makes them (two consecutive frames, outside the clock)
    launches_seen = [st["kernelLaunches"]]
    if world == 1:
        for i in range(2):
            frame(args.steps + extra + i)
            torch.cuda.synchronize(dev)
            launches_seen.append(r.stats()["kernelLaunches"])
        launches_seen = launches_seen[1:]
    st["kernelLaunches"] = min(launches_seen)

    # ---- N = 1: the same metric along a MOVING camera (the two-view loop re-makes the kept tile schedule on the same view every time and
    #      renders it on views the schedule was made for or next to: as favourable as a kept schedule gets)
    moving = None
    if world == 1 and not args.no_path and wl.startswith("street") and not args.debug_flags:
        moving = moving_path(args, r, scene, cam_a, flags, dev)
        if moving:
            moving["ratio_to_two_view"] = round(moving["value"] / (tris_per_pair / 2.0 / (ms_per_step * 1e-3) / 1e9), 4)

    # ---- N > 1, library-run exchange: the PIPELINED protocol on the same frames (second communicator: the image of frame i travels
    #      beside frame i + 1; the history HZB waits only for the small end-of-frame exchange) -- DESIGN.md 6
    pipe = None
    if world > 1 and not args.no_pipelined:
        if exchange == "lib":
            from chord_amd.renderer import comm_unique_id
            try:
                uid2 = [comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid2, src=0)
                r.enable_timers(0)
                r.comm_set_pipelined(uid2[0])
                timed(4)
                _, pe = timed(args.steps)
                r.comm_set_pipelined(None)
                pipe = {"steps": args.steps, "ms_per_step": round(pe / args.steps * 1e3, 4), "timed_region_s": round(pe, 6)}
            except Exception as e:                          # noqa: BLE001
                pipe = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            pipe = {"skipped": "exchange %r drives the phases from the host; the pipelined protocol is the library's (chordvis_comm_set_pipelined / chordvis_group_set_pipelined)" % exchange}

    # ---- roofline of the dominant kernel (HIP events on the launch stream, inside the timed region)
    # algorithmic bytes per frame (DESIGN.md "Roofline"): setup reads 1884 B per cluster (64 header + 12 cmd +
    # 4(V+T) indices + 12V positions, V=81 T=128) and writes a 32 B (vertices <= 64 px apart) or 48 B record + 4 B per bin entry; the tile
    # kernel reads 4 + 48 B per bin entry, writes every pixel once on the first pass of a frame (8 B x W x H, this
    # is also the clear) and reads + writes the 64x64 tiles that have bin entries on the second pass.
    pixels = W * H
    V, T = 81, 128
    cluster_bytes = CLUSTER_BYTES + 4 * (V + T) + 12 * V
    clusters_per_frame = clusters_per_pair / 2.0
    recs = sum(pv["triangleRecords"] for pv in per_view) / 2.0
    recs_c = sum(pv["triangleRecordsCompact"] for pv in per_view) / 2.0
    rec_bytes = 32.0 * recs_c + 48.0 * (recs - recs_c)            # what the setup kernel writes
    rec_size = rec_bytes / recs if recs else 32.0                 # average record a bin entry leads to
    bins = sum(pv["binEntries"] for pv in per_view) / 2.0
    # small clusters leave the setup kernel as pixel blocks (one per cluster and tile: 8 B per window pixel + header) instead
    # of records; a block is one bin entry, read once by the tile kernel
    blocks = sum(pv["pixelBlocks"] for pv in per_view) / 2.0
    block_bytes = sum(pv["pixelBlockBytes"] for pv in per_view) / 2.0
    tiles1 = sum(pv["tilesTouched"][1] for pv in per_view) / 2.0
    launches = max(1, st["rasterLaunches"])
    kernels = {
        "raster_setup_kernel": (st["msRasterCluster"], cluster_bytes * clusters_per_frame + rec_bytes + block_bytes + 4.0 * bins),
        "raster_tile_kernel": (st["msRasterChunk"], 4.0 * bins + rec_size * (bins - blocks) + block_bytes + 8.0 * pixels + (launches - 1) * 16.0 * 4096 * tiles1),
    }
    # What an event record costs.  A record between two kernels is a barrier packet on the queue: the kernel behind it is not dispatched
    # under the kernel in front of it, so every stamped interval is a few microseconds longer than its kernels and a stamped frame longer
    # than a product frame by that x the frame's records (round 5 read that difference as "23 % of the frame idle between kernels"; the
    # unstamped frames of a kernel trace have first start -> last end = the sum of their kernels' durations).  Measured in this run: the
    # stamped frame (msFrame: sum of its intervals) against the frame of the timed region, per interval.  The timed region itself holds
    # stamps in one frame of eight when it is 64 steps or longer: its mean is then U + O / 8 for a stamped frame U + O.
    stamps_pf = float(st.get("stampsPerFrame") or 0.0)
    stamp_cost_ms = 0.0
    if world == 1 and stamps_pf > 1.0 and st["msFrame"] > 0.0:
        over = st["msFrame"] - ms_per_step
        if stamp_inside:                                # (one frame in eight of the timed region itself was a stamped one)
            over *= 8.0 / 7.0
        stamp_cost_ms = max(0.0, over) / (stamps_pf - 1.0)
    raw_ms = {k: v[0] for k, v in kernels.items()}
    # (one interval per launch of the kernel ends in a stamp: the set-up interval of a dense launch holds two kernels and one stamp)
    kernels = {k: (max(v[0] - launches * stamp_cost_ms, 0.0) if v[0] > 0 else 0.0, v[1]) for k, v in kernels.items()}
    dom = max(kernels, key=lambda k: kernels[k][0])
    dom_ms, dom_bytes = kernels[dom]
    achieved = (dom_bytes / launches) / (dom_ms / launches * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM traffic of that kernel from the PMC counters: they need their own rocprofv3 passes, so the figure comes from
    # the committed summary of those passes over this same command (profiles/, tools/profile.sh), per launch
    traffic, traffic_src, traffic_head = None, None, None
    prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    tail = {"street_4k_hzb": "config3_4k_hzb_traffic.json", "street_x64_4k_hzb": "config4_x64_4k_hzb_traffic.json",
            "subpixel_1g": "config5_subpixel_1g_traffic.json", "subpixel_1g_hotspot": "config5_hotspot_traffic.json",
            "street_4k_masked": "masked_4k_traffic.json"}.get(wl)
    cands = sorted(f for f in os.listdir(prof_dir) if tail and f.endswith(tail)) if os.path.isdir(prof_dir) else []
    tj = os.path.join(prof_dir, cands[-1]) if cands else ""              # the newest round's
    if world == 1 and not args.debug_flags and (not args.no_hzb or wl.startswith("subpixel")) and os.path.isfile(tj):
        try:
            tk = json.load(open(tj))
            # (the setup time covers both setup kernels of a launch: the pixel-block kernel of dense launches and the record kernel)
            names = [dom] + (["raster_setup_blocks_kernel"] if dom == "raster_setup_kernel" and blocks > 0 else [])
            traffic = int(sum(tk["kernels"][n]["hbm_bytes_per_launch"] for n in names if n in tk["kernels"]))
            traffic_src = "profiles/" + os.path.basename(tj)
            traffic_head = tk.get("profile_head")
        except (KeyError, ValueError):
            traffic = None
    # The block kernel of the sub-pixel workloads is bound by instruction issue, not by memory: beside the HBM figure the
    # line carries its VALU roofline -- wave-instructions per cluster from the committed SQ counter pass (profiles/*_config5_valu.json,
    # like `traffic`), times the clusters of a launch, over the launch's time, against what the chip's SIMDs can issue: SIMDs x clock /
    # cycles per wave-instruction, the divisor MEASURED by tools/microbench/valu_issue.hip (profiles/r06_microbench_valu_issue.txt):
    # 1.31 cycles for two-operand ops at 8 waves per SIMD, 1.64 at 4 waves, 2.1-2.7 for three-operand / multiply ops, 8 cycles between
    # DEPENDENT instructions of one wave -- not the 4 cycles per instruction rounds 4-5 priced the kernel with.
    valu = None
    if dom == "raster_setup_kernel" and blocks > 0 and dom_ms > 0:
        vj = sorted(f for f in os.listdir(prof_dir) if f.endswith("config5_valu.json")) if os.path.isdir(prof_dir) else []
        if vj:
            try:
                vk = json.load(open(os.path.join(prof_dir, vj[-1])))
                simds = torch.cuda.get_device_properties(dev).multi_processor_count * 4
                clock_hz = 2.4e9
                per = float(vk["valu_per_cluster"])
                ach = per * clusters_per_frame / (dom_ms * 1e-3)
                valu = {"insts_per_cluster": per, "salu_per_cluster": vk.get("salu_per_cluster"), "clusters_per_launch": clusters_per_frame / launches,
                        "achieved_ginst_s": round(ach / 1e9, 1), "simds": simds, "clock_ghz": 2.4,
                        "cycles_per_inst_measured": {"two_operand_8_waves": 1.31, "two_operand_4_waves": 1.64, "three_operand_8_waves": 2.09, "dependent_chain_one_wave": 8.06},
                        "peak_ginst_s": round(simds * clock_hz / 1.31 / 1e9, 1), "issue_frac": round(ach / (simds * clock_hz / 1.31), 4),
                        "issue_frac_three_operand_rate": round(ach / (simds * clock_hz / 2.09), 4),
                        "source": "profiles/" + vj[-1], "profile_head": vk.get("profile_head"), "from_committed_profile": True}
            except (KeyError, ValueError):
                valu = None
    # The first-pass tile kernel of the 4K street workloads is held to the same lens: its VALU pipes are busy 71 % of the launch (DESIGN 4.2;
    # per-pass counter rows of the committed passes, tools/counters_by_pass.py -> profiles/*_config3_tile_valu.json).  A committed figure
    # like `traffic`: counters cannot be read inside this process.
    if valu is None and dom == "raster_tile_kernel" and wl == "street_4k_hzb" and args.cull != "hierarchical" and not args.debug_flags and not args.no_hzb:
        vj = sorted(f for f in os.listdir(prof_dir) if f.endswith("config3_tile_valu.json")) if os.path.isdir(prof_dir) else []
        if vj:
            try:
                vk = json.load(open(os.path.join(prof_dir, vj[-1])))
                valu = {"kernel": vk["kernel"], "insts_per_launch": vk["valu_insts_per_launch"], "busy_cycles_per_launch": vk["busy_cycles_per_launch"],
                        "clock_ghz_from_busy_cycles": vk["clock_ghz_from_busy_cycles"], "profiled_launch_us": vk["mean_launch_us"],
                        "issue_frac": vk["valu_busy_frac"], "waves_resident_per_simd": vk["waves_resident_per_simd"], "lanes_active_frac": vk.get("lanes_active_frac"),
                        "source": "profiles/" + vj[-1], "profile_head": vk.get("profile_head"), "from_committed_profile": True}
            except (KeyError, ValueError):
                valu = None
    roofline = {"bound": "hbm", "valu": valu, "kernel": dom + (" (raster_setup_blocks_kernel + raster_setup_kernel)" if dom == "raster_setup_kernel" and blocks > 0 else ""), "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "traffic_profile_head": traffic_head,          # the commit that profile was taken at: a kernel changed since then leaves `traffic` stale
                "from_committed_profile": traffic is not None,   # (PMC passes cannot run inside this process: not observed in THIS run)
                # avg_launch_us: the stamped interval less what the stamp itself costs (stamp_cost_us, measured in this run: stamped frame
                # against the timed region's frame, per interval) -- the figure the rocprofv3 kernel stats of profiles/ must agree with
                "avg_launch_us": round(dom_ms / launches * 1e3, 2), "avg_launch_us_stamped": round(raw_ms[dom] / launches * 1e3, 2),
                "stamp_cost_us": round(stamp_cost_ms * 1e3, 2), "stamps_per_frame": round(stamps_pf, 1), "launches_per_step": launches,
                "stamped_frames": stamped_in_region + extra, "stamped_inside_timed_region": stamped_in_region,
                "algorithmic_bytes_per_launch": int(dom_bytes / launches),
                "other_kernel": {k: {"avg_launch_us": round(v[0] / launches * 1e3, 2), "algorithmic_bytes_per_launch": int(v[1] / launches)}
                                 for k, v in kernels.items() if k != dom}}

    # ---- N > 1: the same workload unsharded on rank 0's GPU, measured live, so that the line carries its own
    #      strong-scaling reference (the N = 1 default workload is BASELINE config 3, the N > 1 one config 4) ----
    single_ref = None
    if world > 1:
        if rank == 0:
            r1 = VisibilityRenderer(local_rank, stream.cuda_stream)
            if wl.startswith("subpixel_1g"):
                r1.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
            r1.upload_scene(scene)
            r1.allocate_gbuffer(W, H)
            r1.enable_timers(0)

            def frame1(i):
                v, iv = views[i & 1]
                r1.bind_objects(d_obj[i & 1].data_ptr(), len(scene.objects))
                r1.set_view(v, iv, flags)
                r1.render_frame()
            n1 = max(8, min(args.steps, 100)) & ~1
            for i in range(8):
                frame1(i)
            torch.cuda.synchronize(dev)
            s0 = time.perf_counter()
            for i in range(n1):
                frame1(i)
            torch.cuda.synchronize(dev)
            s1 = time.perf_counter()
            # the unit of the metric -- triangles of the clusters submitted to the rasterizer -- is the FRAME's: a rank of a
            # sharded frame culls and counts only the clusters that touch its rows, so the count comes from this run
            ref_view = [None, None]
            for i in range(2):
                frame1(i)
                ref_view[i & 1] = r1.stats()
            tris_per_pair = ref_view[0]["trianglesSubmitted"] + ref_view[1]["trianglesSubmitted"]
            single_ref = {"workload": wl, "n_gpus": 1, "steps": n1, "ms_per_step": round((s1 - s0) / n1 * 1e3, 4),
                          "value": round(tris_per_pair * (n1 // 2) / (s1 - s0) / 1e9, 4), "unit": "Gtri/s",
                          "triangles_submitted_view_a": ref_view[0]["trianglesSubmitted"], "triangles_submitted_view_b": ref_view[1]["trianglesSubmitted"]}
            r1.close()
        dist.barrier()
    tris_view_a = (single_ref["triangles_submitted_view_a"] if single_ref else per_view[0]["trianglesSubmitted"])
    tris_total = tris_per_pair * (args.steps // 2) + (tris_view_a if args.steps & 1 else 0)
    value = tris_total / elapsed / 1e9

    # per rank: GPU time of every phase and exchange of the frame (hipEvent stamps on every 8th timed step), host wall time
    phases = None
    if world > 1:
        mine = {"rank": rank, "wall_ms_per_step": round(rank_elapsed / args.steps * 1e3, 4),
                "phase_a_cull": round(st["msClear"] + st["msInstanceCulling"], 4), "phase_a_stage0": round(st["msStage0"], 4),
                "hzb_mid": round(st["msHzbStage0"], 4), "exchange_hzb": round(st["msExchangeHzb"], 4),
                "phase_b_stage1": round(st["msStage1"], 4), "exchange_vis": round(st["msExchangeVis"], 4),
                "exchange_cull": round(st["msExchangeCull"], 4), "exchange_final": round(st["msExchangeFinal"], 4), "kernel_launches": st["kernelLaunches"],
                "phase_c_final_hzb": round(st["msHzbFinal"], 4), "setup_kernels": round(st["msRasterCluster"], 4), "tile_kernels": round(st["msRasterChunk"], 4),
                "clusters": [per_view[0]["countStage0Visible"] + per_view[0]["countStage1Visible"], per_view[1]["countStage0Visible"] + per_view[1]["countStage1Visible"]]}
        allp = [None] * world
        dist.all_gather_object(allp, mine)
        phases = allp

    # ---- CPU baseline: the oracle replaying the same frame on the host (rank 0, N = 1) ----------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline_frames > 0:
        import orc
        prev = orc.frame(scene_with(scene, obj_b), view_b0, L.make_views(cam_b)[1], flags)   # history for view A
        sc_a = scene_with(scene, obj_a)
        c0 = time.perf_counter()
        tri = 0
        for _ in range(args.cpu_baseline_frames):
            out = orc.frame(sc_a, view_a, iv_a, flags, prev_hzb_min=prev["hzb_min"])
            tri += out["stats"].trianglesSubmitted
        c1 = time.perf_counter()
        cpu = {"value": round(tri / (c1 - c0) / 1e9, 6), "unit": "Gtri/s", "cores": 1, "kind": "port",
               "sample": "%d frames of %s (view A, history of view B), oracle/oracle.c single thread, %.1f s"
                         % (args.cpu_baseline_frames, wl, c1 - c0)}
        # the same frames on every host core (culls over ranges, clusters into per-thread tile-private images merged
        # by max, HZB levels over row ranges: oracle.c orc_frame_mt); the image must equal the single-thread one
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = min(avail, 64)                    # (orc_frame_mt's serial part -- one frame's big triangles -- flattens beyond ~64 threads)
        if threads > 1:
            orc.frame_mt(sc_a, view_a, iv_a, flags, prev["hzb_min"], threads, reuse=True)      # (untimed: first touch of the images)
            m0 = time.perf_counter()
            tri_mt = 0
            for _ in range(args.cpu_baseline_frames):
                out_mt = orc.frame_mt(sc_a, view_a, iv_a, flags, prev["hzb_min"], threads, reuse=True)
                tri_mt += out_mt["triangles_submitted"]
            m1 = time.perf_counter()
            assert np.array_equal(out_mt["vis"], out["vis"]), "multi-threaded CPU replay differs from the scalar one"
            cpu["all_cores"] = {"value": round(tri_mt / (m1 - m0) / 1e9, 6), "unit": "Gtri/s", "cores": threads, "cores_available": avail, "thread_cap": 64, "kind": "port",
                                "sample": "same %d frames (after one to map the threads' private images), %d threads: culls over ranges, tile-private images merged by max, HZB over rows (orc_frame_mt), %.1f s"
                                          % (args.cpu_baseline_frames, threads, m1 - m0)}

    if rank == 0:
        line = {
            "metric": "Gtri/s into 4K 64-bit visbuffer", "value": round(value, 4), "unit": "Gtri/s",
            "n_gpus": world, "steps": args.steps,
            # (the frames that really ran before the clock: the W asked for, at least 4, plus -- N = 1 -- frames until 60 ms of GPU work
            # have gone through, because a GPU that has run 5 frames of 0.2 ms has not reached its clocks)
            "warmup": warm + 2, "warmup_requested": args.warmup, "ms_per_step": round(ms_per_step, 4), "timed_region_s": round(elapsed, 6),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32+u64", "data": "synthetic",
            "config": {"workload": wl, **({"ABLATION_debug_flags": args.debug_flags} if args.debug_flags else {}), "resolution": [W, H], "scene_triangles_lod0": scene.triangle_count_lod0(),
                       "objects": len(scene.objects), "hzb": not args.no_hzb, "cull": args.cull,
                       "parallelism": "tiles%d" % world if world > 1 else "single"},
            "triangles_submitted_per_step": tris_per_pair / 2.0,
            # end to end over ALL scene triangles (LOD 0), i.e. including what culling removed
            "scene_gtri_per_s": round(scene.triangle_count_lod0() / (ms_per_step * 1e-3) / 1e9, 3),
            "clusters_rastered_per_step": clusters_per_frame, "triangle_records_per_step": recs, "bin_entries_per_step": bins, "pixel_blocks_per_step": blocks, "pixel_block_bytes_per_step": block_bytes,
            "gpu_ms": {k: round(st[k], 4) for k in ("msClear", "msInstanceCulling", "msStage0", "msHzbStage0", "msStage1",
                                                     "msHzbFinal", "msFrame", "msRasterCluster", "msRasterClip", "msRasterChunk")},
            "kernel_launches": st["kernelLaunches"], "kernel_launches_schedule_frame": max(launches_seen),
            # frames a pass's tile schedule is reused for (chordvis_set_tile_schedule_keep; the library's default unless --tile-schedule-keep):
            # `kernel_launches` is a frame between two schedules, `kernel_launches_schedule_frame` one that makes them (every (this + 1)-th)
            "tile_schedule_keep_frames": r.tile_schedule_keep(),
            "tiles_touched_view_a": per_view[0]["tilesTouched"], "tiles_total": ((W + 63) // 64) * ((H + 63) // 64),
            "counts_view_a": {k: per_view[0][k] for k in ("countInstanceCulled", "countStage0Visible", "countStage0Rejected", "countStage1Visible", "trianglesSubmitted", "binEntries")},
            "counts_view_b": {k: per_view[1][k] for k in ("countInstanceCulled", "countStage0Visible", "countStage0Rejected", "countStage1Visible", "trianglesSubmitted", "binEntries")},
            "roofline": roofline,
            "cpu_baseline": cpu,
            # the same metric over `views` DISTINCT views along the street (0.5 m steps, two 180-degree cuts per loop), every frame culled
            # against the HZB of the view before it; None where it does not apply (N > 1, the sub-pixel workloads, --no-path)
            "moving_path": moving,
        }
        if world > 1:
            line["exchange"] = exchange
            one = single_ref["ms_per_step"] if single_ref else None
            if pipe and "ms_per_step" in pipe:
                pipe["value"] = round(tris_per_pair / 2.0 / (pipe["ms_per_step"] * 1e-3) / 1e9, 4)
                pipe["speedup_vs_single"] = round(one / pipe["ms_per_step"], 4) if one else None
            line["pipelined"] = pipe
            if default_map:
                default_map["speedup_vs_single"] = round(one / default_map["ms_per_step"], 4) if one else None
            line["default_map"] = default_map
            line["cull"] = "sharded" if phases and any(p["exchange_cull"] > 0 for p in phases) or (ex is not None and ex.cull is not None) else "replicated"
            # achieved bandwidth of every exchange: bytes ARRIVING at a rank ((N - 1) chunks) / the shortest time any rank spent in it
            # (the rank that entered last waited least for its peers: the closest a stream stamp gets to the transfer itself)
            def gbs(key, chunk_bytes):
                ts = [p[key] for p in phases if p[key] > 0]
                if not ts or not chunk_bytes:
                    return None
                return {"bytes_per_rank_in": int((world - 1) * chunk_bytes), "ms_min_over_ranks": round(min(ts), 4), "ms_max_over_ranks": round(max(ts), 4),
                        "gbs": round((world - 1) * chunk_bytes / (min(ts) * 1e-3) / 1e9, 2)}
            both_end = exchange != "lib"                   # (host-driven phases: one stamp covers the small end-of-frame exchange AND the image)
            line["exchange_gbs"] = {"cull": gbs("exchange_cull", chunks["cull"]), "hzb_mid": gbs("exchange_hzb", chunks["hzb_mid"]),
                                    "final": gbs("exchange_final", chunks["final"]),
                                    "image" + ("+final" if both_end else ""): gbs("exchange_vis", chunks["image"] + (chunks["final"] if both_end else 0))}
            img = line["exchange_gbs"]["image" + ("+final" if both_end else "")]
            # a ring all-gather of the 66.8 MB image is per-link bound (~0.8 ms), a direct one ~0.13 ms; 250 GB/s into a rank
            # separates the two.  Below it, the peer-copy transport (ChordGroup) is measured in this same run (main()).
            line["rccl_schedule_ok"] = (img["gbs"] >= 250.0) if (img and backend == "nccl" and exchange == "lib") else None
            launches_pf = max(p["kernel_launches"] for p in phases)
            exch = sum(min([p[k] for p in phases if p[k] > 0] or [0.0]) for k in ("exchange_cull", "exchange_hzb", "exchange_final", "exchange_vis"))
            # what a short frame cannot go below: its launches x the launch floor (profiles/r03_microbench_launch_floor.txt: 2.6-3 us back to
            # back, ~5 us with a dependent load in front) + the exchanges at their best rank
            line["bound"] = {"kernel_launches_per_frame": launches_pf, "launch_floor_us": 5.0, "exchanges_ms": round(exch, 4),
                             "bound_ms": round(launches_pf * 5.0e-3 + exch, 4), "ms_per_step": round(ms_per_step, 4)}
            line["tile_map"] = tile_map
            line["exchange_fallbacks"] = fallbacks
            line["phases_ms"] = phases
            line["rccl_ranks"] = comm["ranks"] if (comm and exchange == "lib") else (dist.get_world_size() if backend == "nccl" else 0)
            line["rccl_version"] = (comm["nccl_version_code"] if comm else (_torch_nccl_version() if backend == "nccl" else None))
            line["collective_backend"] = backend
        if single_ref is not None:
            line["single_gpu_same_workload"] = single_ref
            line["speedup_vs_single"] = round(single_ref["ms_per_step"] / ms_per_step, 4)
    r.close()
    # ---- the RCCL image gather stayed under 250 GB/s into a rank (a ring schedule over xGMI: SURVEY 5): the peer-copy transport --
    #      n - 1 concurrent direct copies per rank -- measured in this same run, its line attached
    if world > 1 and exchange == "lib" and not args.no_group_fallback:
        need = [bool(line is not None and line.get("rccl_schedule_ok") is False)]
        dist.broadcast_object_list(need, src=0)
        if need[0]:
            try:
                gl = run_group(args, wl, scene, views, (obj_a, obj_b), flags, W, H, world, rank, dev, stream, [{"lib": "image gather below 250 GB/s per rank"}])
            except Exception as e:                          # noqa: BLE001
                gl = {"error": "%s: %s" % (type(e).__name__, e)}
            if line is not None and gl is not None:
                line["group_transport"] = {k: gl[k] for k in ("value", "ms_per_step", "speedup_vs_single", "pipelined", "phases_ms", "collective_backend", "error") if k in gl}
    return line


PATH_CUT_AHEAD_M = 90.0


def path_cameras(cam_a, n):
    """n distinct views: n / 2 steps of 0.5 m down the street from the bench camera, a cut to the far end looking back (90 m ahead, the
    heading reversed: other tiles are the heavy ones, other bins the long ones), n / 2 steps of 0.5 m back; the loop closes with the cut
    from the last view to the first."""
    from chord_amd import scenes
    f = np.array(cam_a.front, dtype=np.float64)
    f /= np.linalg.norm(f)
    half = n // 2
    cams = [cam_a.moved(tuple(0.5 * i * f)) for i in range(half)]
    b = np.array([-f[0], f[1], -f[2]])
    p0 = np.array(cam_a.position) + PATH_CUT_AHEAD_M * f * np.array([1.0, 0.0, 1.0])
    back = scenes.Camera(tuple(p0), tuple(b), cam_a.width, cam_a.height, cam_a.fovy, cam_a.z_near, cam_a.z_far, cam_a.world_up, cam_a.jitter)
    cams += [back.moved(tuple(0.5 * i * b)) for i in range(n - half)]
    return cams


def moving_path(args, r, scene, cam_a, flags, dev):
    """The metric of the line over a closed path of distinct views (N = 1): view i culls against the HZB of view i - 1 through the
    objects' last-frame transforms, exactly as the two-view loop does.  One untimed loop settles the history, a second one reads every
    view's submitted triangles, then `loops` loops are timed between two synchronisations.  Stamps off."""
    from chord_amd import lib as L
    n = max(4, args.path_views & ~1)
    cams = path_cameras(cam_a, n)
    base = [L.make_views(c)[0] for c in cams]
    views, d_obj = [], []
    for i, c in enumerate(cams):
        last = cams[i - 1]
        views.append(L.make_views(c, base[i - 1]))
        d_obj.append(torch.from_numpy(L.fill_objects(scene, c, last).copy().view(np.uint8).reshape(-1)).to(dev))

    def frame(i):
        k = i % n
        r.bind_objects(d_obj[k].data_ptr(), len(scene.objects))
        r.set_view(views[k][0], views[k][1], flags)
        r.render_frame()
    r.enable_timers(0)
    for i in range(n):
        frame(i)
    tris, over = 0, 0
    for i in range(n):
        frame(i)
        st = r.stats()
        tris += st["trianglesSubmitted"]
        over |= st["overflow"]
    loops = max(2, (args.steps + n - 1) // n)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(loops * n):
        frame(i)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    # ... and the same loops with a tile schedule made in every frame (chordvis_set_tile_schedule_keep(0)): what keeping the schedule
    # is worth -- or costs -- where every frame is another view and two of them follow a cut
    keep = r.tile_schedule_keep()
    fresh = None
    if keep:
        r.set_tile_schedule_keep(0)
        for i in range(n):
            frame(i)
        torch.cuda.synchronize(dev)
        f0 = time.perf_counter()
        for i in range(loops * n):
            frame(i)
        torch.cuda.synchronize(dev)
        fresh = round((time.perf_counter() - f0) / (loops * n) * 1e3, 4)
        r.set_tile_schedule_keep(keep)
    return {"views": n, "steps": loops * n, "fresh_schedule_ms_per_step": fresh, "ms_per_step": round(el / (loops * n) * 1e3, 4), "value": round(tris * loops / el / 1e9, 4), "unit": "Gtri/s",
            "triangles_submitted_per_step": tris / float(n), "overflow": int(over),
            "path": "%d x 0.5 m down the street, cut to %.0f m ahead looking back, %d x 0.5 m, cut to the start" % (n // 2, PATH_CUT_AHEAD_M, n - n // 2),
            "tile_schedule_keep_frames": r.tile_schedule_keep()}


def run_group(args, wl, scene, views, objs, flags, W, H, world, rank, dev, stream, fallbacks):
    """--exchange group: rank 0 alone drives all N devices through ChordGroup (one worker thread per device, direct peer copies
    between the ranks' buffers; no RCCL).  The other processes of the launch only wait.  Same workload, same frames, same
    timing rule (barrier + synchronize on both sides of exactly --steps frames); one JSON line."""
    from chord_amd.renderer import VisibilityGroup, VisibilityRenderer
    line = None
    if rank == 0:
        one_device = os.environ.get("CHORDVIS_BENCH_ONE_DEVICE") == "1"
        g = VisibilityGroup([0] * world if one_device else list(range(world)))
        if wl.startswith("subpixel_1g"):
            share = max(1, world // 2)
            g.set_limits(max_triangle_records=(1152 << 20) // share, bin_pool_chunks=(1200 << 10) // share, bin_max_chunks_per_tile=2048)
        g.upload_scene(scene)
        g.allocate_gbuffer(W, H)

        def frame(i):
            g.update_objects(objs[i & 1])                # (host arrays: the group API has no bind_objects; 224 B per object and rank)
            g.set_view(views[i & 1][0], views[i & 1][1], flags)
            g.render_frame()
        for r_ in g.ranks:
            r_.enable_timers(0)
        for i in range(max(args.warmup, 4)):
            frame(i)
        g.sync()
        imb_before = 1.0
        if not args.no_rebalance:
            imb_before = g.rebalance()                   # the tile map from the last warm-up frame's loads
            for i in range(4):
                frame(i)
            g.sync()
        owners = g.ranks[0].tile_owners()
        tile_map = {"rebalanced": not args.no_rebalance, "entries_max_over_mean_default_map": round(imb_before, 3),
                    "tiles_per_rank": np.bincount(owners, minlength=world).tolist()}
        for r_ in g.ranks:
            r_.enable_timers(2, period=8)
        g.enqueue_ms()                                   # (reset)
        t0 = time.perf_counter()
        for i in range(args.steps):
            frame(i)
        g.sync()
        elapsed = time.perf_counter() - t0
        enq = g.enqueue_ms()
        sts = [r_.stats() for r_ in g.ranks]
        pipe = None
        if not args.no_pipelined:
            # the pipelined protocol on the same frames: the image of frame i travels (on its own copy streams) beside frame i + 1
            for r_ in g.ranks:
                r_.enable_timers(0)
            g.set_pipelined(True)
            for i in range(4):
                frame(i)
            g.sync()
            p0 = time.perf_counter()
            for i in range(args.steps):
                frame(i)
            g.sync()
            pe = time.perf_counter() - p0
            g.set_pipelined(False)
            pipe = {"steps": args.steps, "ms_per_step": round(pe / args.steps * 1e3, 4), "timed_region_s": round(pe, 6)}
        g.close()
        r1 = VisibilityRenderer(0, stream.cuda_stream)
        if wl.startswith("subpixel_1g"):
            r1.set_limits(max_triangle_records=1152 << 20, bin_pool_chunks=1200 << 10, bin_max_chunks_per_tile=2048)
        r1.upload_scene(scene)
        r1.allocate_gbuffer(W, H)
        r1.enable_timers(0)

        def frame1(i):
            r1.update_objects(objs[i & 1])
            r1.set_view(views[i & 1][0], views[i & 1][1], flags)
            r1.render_frame()
        n1 = max(8, min(args.steps, 100)) & ~1
        for i in range(8):
            frame1(i)
        r1.sync()
        s0 = time.perf_counter()
        for i in range(n1):
            frame1(i)
        r1.sync()
        s1 = time.perf_counter()
        ref_view = [None, None]
        for i in range(2):
            frame1(i)
            ref_view[i & 1] = r1.stats()
        r1.close()
        tris_per_pair = ref_view[0]["trianglesSubmitted"] + ref_view[1]["trianglesSubmitted"]
        tris_total = tris_per_pair * (args.steps // 2) + (ref_view[0]["trianglesSubmitted"] if args.steps & 1 else 0)
        ms = elapsed / args.steps * 1e3
        single = {"workload": wl, "n_gpus": 1, "steps": n1, "ms_per_step": round((s1 - s0) / n1 * 1e3, 4),
                  "value": round(tris_per_pair * (n1 // 2) / (s1 - s0) / 1e9, 4), "unit": "Gtri/s"}
        line = {"metric": "Gtri/s into 4K 64-bit visbuffer", "value": round(tris_total / elapsed / 1e9, 4), "unit": "Gtri/s",
                "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 4) + (0 if args.no_rebalance else 4), "warmup_requested": args.warmup, "ms_per_step": round(ms, 4), "timed_region_s": round(elapsed, 6),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32+u64", "data": "synthetic",
                "config": {"workload": wl, "resolution": [W, H], "scene_triangles_lod0": scene.triangle_count_lod0(), "objects": len(scene.objects),
                           "hzb": not args.no_hzb, "cull": args.cull, "parallelism": "tiles%d" % world},
                "triangles_submitted_per_step": tris_per_pair / 2.0,
                "exchange": "group", "exchange_fallbacks": fallbacks, "collective_backend": "hipMemcpyPeerAsync", "tile_map": tile_map,
                "pipelined": (dict(pipe, value=round(tris_per_pair / 2.0 / (pipe["ms_per_step"] * 1e-3) / 1e9, 4),
                                   speedup_vs_single=round(single["ms_per_step"] / pipe["ms_per_step"], 4)) if pipe else None),
                "phases_ms": [{"rank": k, "phase_a_cull": round(st["msClear"] + st["msInstanceCulling"], 4), "phase_a_stage0": round(st["msStage0"], 4),
                               "hzb_mid": round(st["msHzbStage0"], 4), "exchange_hzb": round(st["msExchangeHzb"], 4), "phase_b_stage1": round(st["msStage1"], 4),
                               "exchange_vis": round(st["msExchangeVis"], 4), "phase_c_final_hzb": round(st["msHzbFinal"], 4),
                               "exchange_cull": round(st["msExchangeCull"], 4), "exchange_final": round(st["msExchangeFinal"], 4), "kernel_launches": st["kernelLaunches"],
                               "setup_kernels": round(st["msRasterCluster"], 4), "tile_kernels": round(st["msRasterChunk"], 4),
                               "host_enqueue_ms": round(enq[k], 4)} for k, st in enumerate(sts)],
                "roofline": None, "cpu_baseline": None,
                "single_gpu_same_workload": single, "speedup_vs_single": round(single["ms_per_step"] / ms, 4)}
    dist.barrier()
    return line


def _torch_nccl_version():
    try:
        v = torch.cuda.nccl.version()
        return int(v[0]) * 10000 + int(v[1]) * 100 + int(v[2]) if isinstance(v, tuple) else int(v)
    except Exception:                                    # noqa: BLE001
        return None


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: one process per GPU under torch.distributed.run, same flags."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def scene_with(scene, objects):
    """A shallow scene whose object records are `objects` (the oracle reads host arrays)."""
    return scene.with_objects(objects)


class _CAI:
    """Minimal __cuda_array_interface__ holder to view raw device memory as a torch tensor."""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 3, "strides": None}


def _tensor_from_ptr(ptr, n, dtype, dev):
    typestr = {torch.int16: "<i2", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_CAI(ptr, n, typestr), device=dev)


if __name__ == "__main__":
    sys.exit(main() or 0)

"""The multi-GPU entry points on ONE device: ranks share device 0, so the protocol (screen tiles, the exchanges, event
ordering, buffer reuse, one call per frame) is what is tested; xGMI bandwidth is not."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
import orc
from chord_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, scene, ranks, re-balance the tile map from frame 1's loads before frame 2)
GROUPS = [
    ("small_2", lambda: scenes.small_test_scene(320, 200, seed=17), 2, False),
    ("small_5", lambda: scenes.small_test_scene(320, 200, seed=23), 5, True),
    ("street_720p_8", lambda: scenes.config3_street(1280, 720), 8, True),
    ("street_x64_360p_4", lambda: scenes.config4_street_x64(640, 360), 4, False),
]


@pytest.mark.parametrize("name,builder,ranks,rebalance", GROUPS, ids=[g[0] for g in GROUPS])
def test_group_frames_equal_the_single_gpu_frames(gpu, name, builder, ranks, rebalance):
    """chordvis_create_group / chordvis_group_render_frame: one call per frame, the library issues both exchanges
    (direct peer copies ordered by events, no host synchronisation inside a frame); four frames with a moving camera so
    that buffers are reused while copies of the previous frame may still be in flight."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityGroup, VisibilityRenderer
    scene, cam0 = builder()
    w, h, flags = cam0.width, cam0.height, H.ALL_FLAGS
    f = np.array(cam0.front, dtype=np.float64)
    f /= np.linalg.norm(f)
    cams = [cam0.moved(tuple(0.25 * k * f)) for k in range(4)]
    ref = VisibilityRenderer(0)
    ref.upload_scene(scene)
    ref.allocate_gbuffer(w, h)
    g = VisibilityGroup([0] * ranks)
    g.upload_scene(scene)
    g.allocate_gbuffer(w, h)
    if name == "street_x64_360p_4":
        # one case keeps the REPLICATED group cull (every rank tests every group: the form before the sharded cull, still what
        # hierarchical mode and more than 8 ranks run); all the others exchange rank masks (chordvis_frame_phase_cull)
        for r in g.ranks:
            r.set_debug(524288)
    wants = []
    last_view = None
    inputs = []
    for k, cam in enumerate(cams):
        objs = L.fill_objects(scene, cam, cams[k - 1] if k else None).copy()
        view, iv = L.make_views(cam, last_view)
        last_view = L.make_views(cam)[0]
        inputs.append((objs, view, iv))
    prev_hzb = None
    for k, (objs, view, iv) in enumerate(inputs):         # reference frames first
        ref.update_objects(objs)
        ref.set_view(view, iv, flags)
        ref.render_frame()
        vis = ref.read_visibility()
        if name.startswith("small"):
            # the small cases are held against the ORACLE, not only against the single-GPU HIP frame (a defect shared by both HIP
            # paths on a group-only code path would pass a comparison between them): the ranks below are compared with its image
            o = orc.frame(scene.with_objects(objs), view, iv, flags, prev_hzb_min=prev_hzb)
            prev_hzb = o["hzb_min"]
            H.assert_vis_equal(vis, o["vis"], w, h, "single-GPU frame %d vs oracle" % k)
            vis = o["vis"]
        wants.append((vis, ref.read_hzb(ref.history_hzb()), ref.stats()))
    for objs, view, iv in inputs[:2]:                     # two frames enqueued back to back, then checked ...
        g.update_objects(objs)
        g.set_view(view, iv, flags)
        g.render_frame()
    g.sync()
    _check_ranks(g, wants[1], w, h, "frame 1")
    if rebalance:
        # chordvis_group_rebalance: every rank re-maps the tiles from the loads of frame 1 (the history HZB carries over)
        old = g.ranks[0].tile_owners()
        imb = g.rebalance()
        maps = [r.tile_owners() for r in g.ranks]
        assert imb >= 1.0 and all(np.array_equal(maps[0], m) for m in maps[1:])
        assert not np.array_equal(maps[0], old) or imb < 1.02
    for r in g.ranks:
        r.enable_timers(1)
    for k, (objs, view, iv) in enumerate(inputs[2:], start=2):   # ... and frame by frame
        g.update_objects(objs)
        g.set_view(view, iv, flags)
        g.render_frame()
        _check_ranks(g, wants[k], w, h, "frame %d" % k)
    # the sharded cull ran where it applies (its exchange shows in the stamps), and a rank's frame is a fixed number of launches
    st = [r.stats() for r in g.ranks]
    assert all((s_["msExchangeCull"] > 0) == (name != "street_x64_360p_4") for s_ in st), [s_["msExchangeCull"] for s_ in st]
    assert all(10 <= s_["kernelLaunches"] <= 24 for s_ in st), [s_["kernelLaunches"] for s_ in st]
    # every rank holds the ORACLE's command array of the last frame (slots included): made on demand from the exchanged masks
    objs, view, iv = inputs[-1]
    want_cmds = orc.instance_culling(scene.with_objects(objs), view, iv, flags)
    for r in (g.ranks[0], g.ranks[-1]):
        assert np.array_equal(r.read_cmds(r.last_frame_cmds()), want_cmds)
    g.close()
    ref.close()


def _check_ranks(g, want, w, h, what):
    wvis, (wmn, wmx, wrng), wst = want
    for rk, r in enumerate(g.ranks):
        H.assert_vis_equal(r.read_visibility(), wvis, w, h, "%s rank %d" % (what, rk))
        mn, mx, rng = r.read_hzb(r.history_hzb())
        assert np.array_equal(mn, wmn) and np.array_equal(mx, wmx) and np.array_equal(rng, wrng), "%s rank %d HZB" % (what, rk)
    H.assert_rank_counts([r.stats() for r in g.ranks], wst)


PIPELINED = [
    ("small_2", lambda: scenes.small_test_scene(320, 200, seed=17), 2, 0),
    ("small_3", lambda: scenes.small_test_scene(320, 200, seed=23), 3, 4),
    ("street_720p_4", lambda: scenes.config3_street(1280, 720), 4, 0),
    ("street_x64_360p_8", lambda: scenes.config4_street_x64(640, 360), 8, 3),
]


@pytest.mark.parametrize("name,builder,ranks,rebalance_at", PIPELINED, ids=[g[0] for g in PIPELINED])
def test_pipelined_group_frames_equal_the_single_gpu_frames(gpu, name, builder, ranks, rebalance_at):
    """chordvis_group_set_pipelined: the visibility all-gather and row-major copy of frame i run beside frame i + 1 (two
    buffer pairs per rank), and the history HZB waits only for the small end-of-frame exchange (the tiles' HZB texels out of
    the tile kernel), not for the gathered image.  rebalance_at: the tile map is re-made from the loads before that frame.  Six frames with a moving camera, never synchronised in between: every rank's
    image of every frame (read as 'the frame before' once the next one is submitted, then as the last one), HZB chain and
    counts equal the single-GPU frames."""
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityGroup, VisibilityRenderer
    scene, cam0 = builder()
    w, h, flags = cam0.width, cam0.height, H.ALL_FLAGS
    f = np.array(cam0.front, dtype=np.float64)
    f /= np.linalg.norm(f)
    cams = [cam0.moved(tuple(0.25 * k * f)) for k in range(6)]
    ref = VisibilityRenderer(0)
    ref.upload_scene(scene)
    ref.allocate_gbuffer(w, h)
    g = VisibilityGroup([0] * ranks)
    g.upload_scene(scene)
    g.allocate_gbuffer(w, h)
    g.set_pipelined(True)
    inputs, wants, last_view, prev_hzb = [], [], None, None
    for k, cam in enumerate(cams):
        objs = L.fill_objects(scene, cam, cams[k - 1] if k else None).copy()
        view, iv = L.make_views(cam, last_view)
        last_view = L.make_views(cam)[0]
        inputs.append((objs, view, iv))
        ref.update_objects(objs)
        ref.set_view(view, iv, flags)
        ref.render_frame()
        vis = ref.read_visibility()
        if name.startswith("small"):                       # (against the ORACLE: see test_group_frames_equal_the_single_gpu_frames)
            o = orc.frame(scene.with_objects(objs), view, iv, flags, prev_hzb_min=prev_hzb)
            prev_hzb = o["hzb_min"]
            H.assert_vis_equal(vis, o["vis"], w, h, "single-GPU frame %d vs oracle" % k)
            vis = o["vis"]
        wants.append((vis, ref.read_hzb(ref.history_hzb()), ref.stats()))
    for k, (objs, view, iv) in enumerate(inputs):
        if rebalance_at and k == rebalance_at:
            g.rebalance()                                  # (drains the frames in flight: their images were laid out with the old map)
        g.update_objects(objs)
        g.set_view(view, iv, flags)
        g.render_frame()                                   # frame k enqueued; frame k - 1's image may still be travelling
        if k in (1, 3, 5) and k != rebalance_at:
            for rk, r in enumerate(g.ranks):
                H.assert_vis_equal(r.read_previous_visibility(), wants[k - 1][0], w, h, "frame %d (as the previous one) rank %d" % (k - 1, rk))
        if k >= 2:
            _check_ranks(g, wants[k], w, h, "pipelined frame %d" % k)
    # back to the unpipelined protocol on the same group: the history carries over
    g.set_pipelined(False)
    objs, view, iv = inputs[-1]
    g.update_objects(objs)
    g.set_view(*L.make_views(cams[-1], L.make_views(cams[-1])[0]), flags)
    g.render_frame()
    ref.update_objects(objs)
    ref.set_view(*L.make_views(cams[-1], L.make_views(cams[-1])[0]), flags)
    ref.render_frame()
    _check_ranks(g, (ref.read_visibility(), ref.read_hzb(ref.history_hzb()), ref.stats()), w, h, "unpipelined frame after pipelined ones")
    g.close()
    ref.close()


def test_library_owned_rccl_exchange_world_size_1(gpu):
    """chordvis_comm_*: librccl resolved at run time (the copy PyTorch already loaded), a communicator attached to the
    context, chordvis_render_frame issuing ncclAllGather on the context's stream.  One rank is all a one-GPU box can
    host; the frame (phases a/b/c around two in-place all-gathers) must equal the fused single-GPU frame."""
    import torch  # noqa: F401  (loads PyTorch's librccl next to its HIP runtime)
    from chord_amd.renderer import VisibilityRenderer, comm_unique_id
    scene, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(320, 200, seed=31))
    w, h, flags = cam.width, cam.height, H.ALL_FLAGS
    ref = VisibilityRenderer(0)
    r = VisibilityRenderer(0)
    for x in (ref, r):
        x.upload_scene(scene)
        x.allocate_gbuffer(w, h)
        x.set_view(view, iv, flags)
    r.comm_init_rank(1, 0, comm_unique_id())
    info = r.comm_info()
    assert info["ranks"] == 1 and info["nccl_version_code"] >= 20000, info
    prev_hzb = None
    for frame in range(3):
        ref.render_frame()
        r.render_frame()
        o = orc.frame(scene, view, iv, flags, prev_hzb_min=prev_hzb)     # the ORACLE's frame, not only the other HIP path's
        prev_hzb = o["hzb_min"]
        H.assert_vis_equal(r.read_visibility(), o["vis"], w, h, "frame %d vs oracle" % frame)
        H.assert_vis_equal(r.read_visibility(), ref.read_visibility(), w, h, "frame %d" % frame)
        a, b = r.read_hzb(r.history_hzb()), ref.read_hzb(ref.history_hzb())
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    r.comm_destroy()
    r.close()
    ref.close()


def test_pipelined_rccl_frames_world_size_1(gpu):
    """chordvis_comm_set_pipelined on the one rank a one-GPU box can host: a second communicator (its own unique id), the
    image of every frame gathered on the resolve stream behind the compute stream's "phase b done" event, an "image complete"
    event per frame parity, consumers ordered by chordvis_wait_visibility (on the context's stream and on a foreign one) --
    RcclTransport::image and the event plumbing of the N-rank protocol; the frames must equal the fused single-GPU frames."""
    import torch
    from chord_amd.renderer import VisibilityRenderer, comm_unique_id
    scene, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(320, 200, seed=32))
    w, h, flags = cam.width, cam.height, H.ALL_FLAGS
    ref = VisibilityRenderer(0)
    r = VisibilityRenderer(0)
    for x in (ref, r):
        x.upload_scene(scene)
        x.allocate_gbuffer(w, h)
        x.set_view(view, iv, flags)
    r.comm_init_rank(1, 0, comm_unique_id())
    r.comm_set_pipelined(comm_unique_id())
    side = torch.cuda.Stream(device=0)
    for frame in range(4):
        ref.render_frame()
        r.render_frame()
        r.wait_visibility()
        r.wait_visibility(C.c_void_p(side.cuda_stream))
        side.synchronize()
        H.assert_vis_equal(r.read_visibility(), ref.read_visibility(), w, h, "pipelined frame %d" % frame)
        a, b = r.read_hzb(r.history_hzb()), ref.read_hzb(ref.history_hzb())
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    r.comm_set_pipelined(None)                       # drains the frames in flight, drops the second communicator
    ref.render_frame()
    r.render_frame()
    H.assert_vis_equal(r.read_visibility(), ref.read_visibility(), w, h, "frame after the pipeline was switched off")
    r.comm_destroy()
    r.close()
    ref.close()


def test_sharded_render_frame_without_a_communicator_is_refused(gpu):
    from chord_amd import lib as L
    from chord_amd.renderer import VisibilityRenderer
    scene, cam, view, iv = H.setup_scene(lambda: scenes.small_test_scene(160, 96))
    r = VisibilityRenderer(0)
    r.upload_scene(scene)
    r.set_shard(2, 1)
    r.allocate_gbuffer(cam.width, cam.height)
    r.set_view(view, iv, H.ALL_FLAGS)
    with pytest.raises(L.ChordvisError):
        r.render_frame()
    r.close()


def _run_bench(extra, env_extra, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, "bench.py failed:\n%s\n%s" % (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_self_spawned_on_one_device(gpu):
    """`python bench.py --gpus 2` with no launcher: re-executes itself under torch.distributed.run, two ranks on device 0
    (gloo, host-staged all-gathers: a one-GPU box cannot host two RCCL ranks), the N > 1 control flow of the bench --
    explicit stream, phases, both exchanges, the same-workload single-GPU reference and the speed-up field."""
    line = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "4", "--workload", "street_720p_hzb"],
                      {"CHORDVIS_BENCH_BACKEND": "gloo", "CHORDVIS_BENCH_ONE_DEVICE": "1", "CHORDVIS_BENCH_ALSO": "street_x64_720p_hzb"})
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["exchange"] == "torch"
    # the line explains both multi-GPU configurations: the second workload (on the driver's run: BASELINE config 4) rides in `also`
    also = line["also"]
    assert also["config"]["workload"] == "street_x64_720p_hzb" and also["single_gpu_same_workload"]["workload"] == "street_x64_720p_hzb"
    assert also["value"] > 0 and also["speedup_vs_single"] > 0 and len(also["phases_ms"]) == 2 and also["exchange"] == "torch"
    assert line["tile_map"]["rebalanced"] and sum(line["tile_map"]["tiles_per_rank"]) == line["tiles_total"]
    assert line["config"]["workload"] == "street_720p_hzb" and line["single_gpu_same_workload"]["workload"] == "street_720p_hzb"
    assert line["speedup_vs_single"] > 0 and line["value"] > 0
    assert line["counts_view_a"]["countInstanceCulled"] > 0
    # per rank: GPU time of every phase and both exchanges, so that a scaling record explains itself
    assert len(line["phases_ms"]) == 2 and {p["rank"] for p in line["phases_ms"]} == {0, 1}
    assert all(k in line["phases_ms"][0] for k in ("phase_a_stage0", "exchange_hzb", "phase_b_stage1", "exchange_vis", "phase_c_final_hzb", "exchange_cull", "kernel_launches"))
    # one line holds both protocols (the host-driven exchange of this test hook has no pipelined form: it says so), the figure under
    # the default tile map beside the re-balanced one, every exchange's achieved bandwidth, what bounds the frame, and the cull's form
    for ln in (line, also):
        assert ln["cull"] == "sharded" and all(p_["exchange_cull"] > 0 for p_ in ln["phases_ms"])
        assert "skipped" in ln["pipelined"]
        assert ln["default_map"]["ms_per_step"] > 0 and ln["default_map"]["speedup_vs_single"] > 0
        gb = ln["exchange_gbs"]
        assert gb["cull"]["gbs"] > 0 and gb["hzb_mid"]["gbs"] > 0 and gb["image+final"]["bytes_per_rank_in"] > 0
        assert ln["rccl_schedule_ok"] is None                       # (no RCCL in this run)
        b = ln["bound"]
        assert 10 <= b["kernel_launches_per_frame"] <= 24 and abs(b["bound_ms"] - (b["kernel_launches_per_frame"] * 5e-3 + b["exchanges_ms"])) < 1e-3
    assert line["warmup"] >= line["warmup_requested"] == 4


def test_bench_two_ranks_group_fallback_on_one_device(gpu):
    """`--exchange group`: the third way to a scaling curve when RCCL does not come up -- rank 0 drives every device through
    ChordGroup (peer copies), the other processes wait.  Two 'devices' = device 0 twice here."""
    line = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "4", "--workload", "street_720p_hzb", "--exchange", "group"],
                      {"CHORDVIS_BENCH_BACKEND": "gloo", "CHORDVIS_BENCH_ONE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["exchange"] == "group" and line["value"] > 0 and line["speedup_vs_single"] > 0
    assert len(line["phases_ms"]) == 2 and all("exchange_vis" in p for p in line["phases_ms"])
    # what the one-process transport costs the host, worker by worker
    assert all(p["host_enqueue_ms"] > 0 for p in line["phases_ms"])
    # the group transport measures its pipelined protocol in the same run, and stamps the small end-of-frame exchange apart from the image
    assert line["pipelined"]["ms_per_step"] > 0 and line["pipelined"]["speedup_vs_single"] > 0
    assert all(p["exchange_final"] > 0 and p["exchange_cull"] > 0 for p in line["phases_ms"])


def test_bench_single_gpu_line_has_the_contract_fields(gpu):
    line = _run_bench(["--steps", "6", "--warmup", "4", "--workload", "street_720p_hzb", "--cpu-baseline-frames", "2", "--path-views", "8"], {})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["roofline"]["frac"] < 1.0 and line["cpu_baseline"]["cores"] == 1
    # the stamp-corrected per-launch time of the dominant kernel: the stamped interval less what a record costs, measured in the run
    rf = line["roofline"]
    assert rf["stamps_per_frame"] >= 10 and 0.0 <= rf["stamp_cost_us"] < 20.0
    assert 0 < rf["avg_launch_us"] <= rf["avg_launch_us_stamped"]
    # the same metric along a moving camera (a closed path of distinct views, two cuts per loop)
    mp = line["moving_path"]
    assert mp["views"] == 8 and mp["steps"] >= 16 and mp["value"] > 0 and mp["ms_per_step"] > 0 and mp["overflow"] == 0
    assert 0.2 < mp["ratio_to_two_view"] < 5.0 and mp["tile_schedule_keep_frames"] == line["tile_schedule_keep_frames"]
    # the library's default schedule keep, and the launches of both kinds of frame (between two schedules / making them)
    assert line["tile_schedule_keep_frames"] == 1 and 4 <= line["kernel_launches"] <= line["kernel_launches_schedule_frame"] <= line["kernel_launches"] + 2


def test_bench_without_stamps_renders_only_product_frames(gpu):
    """`--no-stamps` (kernel traces: tools/trace.sh): no event record in the whole process -- the line says so and carries no stamped times."""
    line = _run_bench(["--steps", "6", "--warmup", "4", "--workload", "street_720p_hzb", "--cpu-baseline-frames", "0", "--no-stamps", "--no-path"], {})
    assert line["roofline"]["stamped_frames"] == 0 and line["gpu_ms"]["msFrame"] == 0.0 and line["value"] > 0 and line["moving_path"] is None

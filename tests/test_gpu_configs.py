"""-m gpu parity tests at the level of BASELINE.json's configs: the whole front-end of one configuration through the
C-ABI vs the CPU oracle (bit-exact), on top of the per-stage files.
  config 2 (index 2): ICL-NUIM living-room style scene, ICL intrinsics (fy < 0): ORB + glue + SearchByProjection on a
                      short sequence, LSD + LBD lines, CAPE and AHC planes of its first frame;
  config 3 again, every frame of a 12-frame sequence through the frame-batch entries (device sequential cores);
  config 5 (index 4): 1280x960 RealSense-style frame: LSD + LBD lines and CAPE planes at 4x the pixels (the ORB part
                      of this configuration is tests/test_gpu_orb.py::test_1280x960_config5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_kps(kps, okps):
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kps[f].view(np.uint32), okps[f].view(np.uint32)), f
    assert np.array_equal(kps["octave"], okps["octave"])


def test_config2_icl_living_room_full_front_end(oracle_mod):
    import torch
    from dr_slam_amd import synth
    from dr_slam_amd.pipeline import FrontEnd
    O = oracle_mod
    cam = synth.ICL
    frames = list(synth.sequence(3, 3, cam=cam, kind="living_room"))
    fe = FrontEnd(cam, max_batch=4)
    try:
        gray = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
        depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
        Twc = np.stack([f[2] for f in frames]).astype(np.float64)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        Twc = Twc.astype(np.float32)
        fe.process(gray, depth, Tcw, Twc, th=15.0, check_ori=True, stream=torch.cuda.current_stream().cuda_stream)
        o = O.OrbOracle()
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        inv = np.float32(1.0) / np.float32(cam.depth_factor)
        of = []
        for s, (g, d, _) in enumerate(frames):
            okps, odesc = o(g)
            kps, desc = fe.keypoints(s)
            _same_kps(kps, okps)
            assert np.array_equal(desc, odesc)
            of.append(O.FrameOracle(okps, odesc, O.depth_to_float(d, inv), K4, cam.bf, cam.w, cam.h, o.scale))
            ur, z = fe.ctx.download_stereo(s)
            assert np.array_equal(ur[:of[s].N].view(np.uint32), of[s].uRight.view(np.uint32))
        total = 0
        for s in range(1, 3):
            world, valid = of[s - 1].unproject(Twc[s - 1])
            mp = np.zeros(of[s - 1].N, O.MAPPOINT_DTYPE)
            mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, of[s - 1].desc
            no, mo = O.search_by_projection_last(of[s], of[s - 1], Tcw[s], Tcw[s - 1], mp, 15.0, False, True)
            m, n = fe.matches(s)
            assert n == no and np.array_equal(m[:of[s].N], mo)
            total += n
        assert total > 200
        # lines and planes of the first frame
        g0, d0, _ = frames[0]
        a, b = fe.ctx.lsd_extract(g0), O.extract_lines(g0)
        assert len(a["lines"]) == len(b["lines"]) >= 10
        assert np.array_equal(a["desc"], b["desc"])
        assert np.array_equal(a["lineF"].view(np.uint64), b["lineF"].view(np.uint64))
        for pa, pb in (("start_point_x", "startPointX"), ("end_point_y", "endPointY"), ("angle", "angle"), ("response", "response")):
            assert np.array_equal(a["lines"][pa].view(np.uint32), b["lines"][pb].view(np.uint32)), pa
        dm = O.depth_to_float(d0, inv)
        gp, op = fe.ctx.planes_cape(dm, K4, 20), O.cape_planes(dm, K4, 20)
        assert len(gp["planes"]) == len(op["planes"]) >= 3 and np.array_equal(gp["seg"], op["seg"])
        assert np.array_equal(gp["planes"]["normal"].view(np.uint64), op["planes"][:, 0:3].view(np.uint64))
        ga, oa = fe.ctx.planes_ahc(d0, K4, float(inv)), O.ahc_planes(d0, K4, float(inv))
        assert len(ga["planes"]) == len(oa["planes"]) >= 2 and np.array_equal(ga["seg"], oa["seg"])
    finally:
        fe.ctx.close()


def test_config3_every_frame_through_the_batch_entries(oracle_mod):
    """BASELINE config 3 as bench.py's full_frontend runs it - EVERY frame's lines, AHC planes + Frame::ComputePlanes' per-plane
    loop and CAPE planes through the frame-batch entries, whose sequential cores run on the device (k_lsd_order / k_lsd_grow,
    k_ahc_cluster / k_ahc_refine, k_voxel_grid) - against the CPU oracle, frame by frame, bit for bit."""
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.ICL
    frames = list(synth.sequence(3, 12, cam=cam, kind="living_room"))
    gray = np.stack([f[0] for f in frames]); depth = np.stack([f[1] for f in frames])
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    depth_m = np.stack([O.depth_to_float(d, inv) for d in depth])
    c = lib.Context(max_batch=1)
    try:
        lines = c.lsd_extract_batch(gray, n_threads=4)
        planes, n, post, na, pn, seg = c.planes_ahc_post_batch(depth, K4, float(inv), 9.0, 0.10, n_threads=4, seg=True)
        cplanes, cn, cseg = c.planes_cape_batch(depth_m, K4, 20, n_threads=2, seg=True)
        nl = 0
        for f in range(len(frames)):
            a, b = lines[f], O.extract_lines(gray[f])
            assert a["detected"] == b["detected"] and len(a["lines"]) == len(b["lines"])
            assert np.array_equal(a["desc"], b["desc"]) and np.array_equal(a["lineF"].view(np.uint64), b["lineF"].view(np.uint64))
            for pa, pb in (("start_point_x", "startPointX"), ("start_point_y", "startPointY"), ("end_point_x", "endPointX"),
                           ("end_point_y", "endPointY"), ("angle", "angle"), ("response", "response"), ("line_length", "lineLength")):
                assert np.array_equal(a["lines"][pa].view(np.uint32), b["lines"][pb].view(np.uint32)), (f, pa)
            nl += len(a["lines"])
            oa = O.ahc_planes(depth[f], K4, float(inv))
            assert n[f] == len(oa["planes"]) and np.array_equal(seg[f], oa["seg"]) and np.array_equal(planes[f, :n[f]]["n_points"], oa["N"])
            assert np.array_equal(planes[f, :n[f]]["normal"].view(np.uint64), oa["planes"][:, 0:3].view(np.uint64))
            assert np.array_equal(planes[f, :n[f]]["center"].view(np.uint64), oa["planes"][:, 3:6].view(np.uint64))
            assert np.array_equal(planes[f, :n[f]]["mse"].view(np.uint64), oa["planes"][:, 6].view(np.uint64))
            o2, opn = O.ahc_post_planes(depth[f], K4, float(inv), oa, 9.0, 0.10)
            assert pn[f] == opn and na[f] == sum(1 for r in o2 if r["accepted"])
            for k, rec in enumerate(o2):
                assert bool(post[f, k]["accepted"]) == rec["accepted"] and post[f, k]["n_voxels"] == len(rec["voxels"])
                assert np.array_equal(post[f, k]["coef"].view(np.uint32), rec["coef"].view(np.uint32))
            oc = O.cape_planes(depth_m[f], K4, 20)
            assert cn[f] == len(oc["planes"]) and np.array_equal(cseg[f], oc["seg"])
            assert np.array_equal(cplanes[f, :cn[f]]["normal"].view(np.uint64), oc["planes"][:, 0:3].view(np.uint64))
        assert nl >= 10 * len(frames) and na.sum() >= 2 * len(frames) and cn.sum() >= 3 * len(frames)
    finally:
        c.close()


def test_config5_1280x960_lines_and_cape_planes(oracle_mod):
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.REALSENSE.scaled(2.0)
    g, d, _ = next(synth.sequence(5, 1, cam=cam, kind="corridor"))
    assert g.shape == (960, 1280)
    c = lib.Context(nfeatures=800, max_width=1280, max_height=960)
    try:
        a, b = c.lsd_extract(g), O.extract_lines(g)
        assert b["detected"] >= 40 and len(a["lines"]) == len(b["lines"]) == 40 and a["detected"] == b["detected"]
        assert np.array_equal(a["desc"], b["desc"])
        assert np.array_equal(a["lineF"].view(np.uint64), b["lineF"].view(np.uint64))
        for pa, pb in (("start_point_x", "startPointX"), ("start_point_y", "startPointY"), ("end_point_x", "endPointX"),
                       ("end_point_y", "endPointY"), ("line_length", "lineLength"), ("response", "response")):
            assert np.array_equal(a["lines"][pa].view(np.uint32), b["lines"][pb].view(np.uint32)), pa
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        dm = O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor))
        gp, op = c.planes_cape(dm, K4, 20), O.cape_planes(dm, K4, 20)
        assert len(gp["planes"]) == len(op["planes"]) >= 2
        assert np.array_equal(gp["seg"], op["seg"])
        assert np.array_equal(gp["planes"]["normal"].view(np.uint64), op["planes"][:, 0:3].view(np.uint64))
        assert np.array_equal(gp["planes"]["d"].view(np.uint64), op["planes"][:, 6].view(np.uint64))
        # the batch entries at this size (round 4: the whole line detector and CAPE::process on the device - a 1024 x 768 level-line
        # field, 3072 CAPE cells = the device path's capacity): the frame twice, identical to the single-frame entries above
        lb = c.lsd_extract_batch(np.stack([g, g]), n_threads=2)
        for x in lb:
            assert x["detected"] == a["detected"] and x["lines"].tobytes() == a["lines"].tobytes()
            assert np.array_equal(x["desc"], a["desc"]) and x["lineF"].tobytes() == a["lineF"].tobytes()
        st = c.lsd_stats()
        assert st["frames"] == 2 and st["grow_to_host"] == 0 and st["keylines_to_host"] == 0
        cp, cn, cs = c.planes_cape_batch(np.stack([dm, dm]), K4, 20, n_threads=2, seg=True)
        for f in range(2):
            assert cn[f] == len(op["planes"]) and np.array_equal(cs[f], op["seg"])
            assert np.array_equal(cp[f, :cn[f]]["normal"].view(np.uint64), op["planes"][:, 0:3].view(np.uint64))
        assert c.planes_cape_stats()["frames"] == 2
    finally:
        c.close()


def test_config5_1280x960_ahc_planes_on_the_device(oracle_mod):
    """BASELINE config 5's plane leg with the LIVE extractor (src/Frame.cc:126: PEAC / AHC): 1280 x 960 = 128 x 96 init blocks, four
    times what the reference's hard-coded 640 x 480 holds (include/PlaneExtractor.h:35-36); the oracle defines the generalised
    semantics.  The single-frame entry and the batch entries - whose extractor (k_ahc_cluster_big /
    k_ahc_refine_big: a 12 800-entry queue in LDS, 21-bit pixel indices) and voxel grids run on the device - against
    ahc_run(w, h), bit for bit, with the counters showing that no frame went back to the host."""
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.REALSENSE.scaled(2.0)
    frames = list(synth.sequence(5, 2, cam=cam, kind="corridor"))
    depth = np.stack([f[1] for f in frames])
    assert depth.shape == (2, 960, 1280)
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    c = lib.Context(nfeatures=800, max_width=1280, max_height=960)
    try:
        oas = [O.ahc_planes(d, K4, inv) for d in depth]
        assert all(len(oa["planes"]) >= 4 for oa in oas)
        # the single-frame entry (block fits on the device, the sequential core on the host)
        ga = c.planes_ahc(depth[0], K4, inv)
        assert len(ga["planes"]) == len(oas[0]["planes"]) and np.array_equal(ga["seg"], oas[0]["seg"])
        assert np.array_equal(ga["planes"]["normal"].view(np.uint64), oas[0]["planes"][:, 0:3].view(np.uint64))
        # Realsense.yaml:76-79: Plane.DistanceThreshold 0.10, Point.MaxDistance 5.0
        planes, n, post, na, pn, seg = c.planes_ahc_post_batch(depth, K4, inv, 5.0, 0.10, n_threads=2, seg=True)
        for f, oa in enumerate(oas):
            assert n[f] == len(oa["planes"]) and np.array_equal(seg[f], oa["seg"]) and np.array_equal(planes[f, :n[f]]["n_points"], oa["N"])
            assert np.array_equal(planes[f, :n[f]]["normal"].view(np.uint64), oa["planes"][:, 0:3].view(np.uint64))
            assert np.array_equal(planes[f, :n[f]]["center"].view(np.uint64), oa["planes"][:, 3:6].view(np.uint64))
            assert np.array_equal(planes[f, :n[f]]["mse"].view(np.uint64), oa["planes"][:, 6].view(np.uint64))
            o2, opn = O.ahc_post_planes(depth[f], K4, inv, oa, 5.0, 0.10)
            assert pn[f] == opn and na[f] == sum(1 for r in o2 if r["accepted"])
            for k, rec in enumerate(o2):
                assert bool(post[f, k]["accepted"]) == rec["accepted"] and post[f, k]["n_voxels"] == len(rec["voxels"])
                assert np.array_equal(post[f, k]["coef"].view(np.uint32), rec["coef"].view(np.uint32))
        st = c.planes_ahc_stats()
        assert st["frames"] == 2 and st["to_host"] == 0, st
        assert st["voxel_grids"] == int(n.sum()) and st["voxel_grids_to_host"] == 0, st
        # the extractor-only batch entry: member lists included
        rb = c.planes_ahc_batch(depth, K4, inv, n_threads=2, members=True)
        for f, oa in enumerate(oas):
            assert np.array_equal(rb[f]["seg"], oa["seg"]) and rb[f]["planes"].tobytes() == planes[f, :n[f]].tobytes()
            assert len(rb[f]["members"]) == len(oa["members"])
            for a, b in zip(rb[f]["members"], oa["members"]):
                assert np.array_equal(a, b)
        st = c.planes_ahc_stats()
        assert st["frames"] == 4 and st["to_host"] == 0, st
    finally:
        c.close()


@pytest.mark.parametrize("rank", [0, 1, 2])
def test_config4_rank_workload_batched(oracle_mod, rank):
    """BASELINE config 4 (index 3), one rank's share on one GPU: the sequence bench.py gives rank `rank`
    (seed 10+rank, intrinsics TUM1 / TUM2 / TUM3 - lens distortion live on ranks 0 and 1), 32 frames through the
    batched FrontEnd.process; mvKeys, descriptors, mvKeysUn, uRight, the grid and the SearchByProjection match arrays
    of all 31 consecutive pairs are compared with the oracle."""
    import torch
    from dr_slam_amd import sharding
    from dr_slam_amd.pipeline import FrontEnd
    O = oracle_mod
    seed, cam, kind, seq_len = sharding.rank_workload(4, rank)
    assert seq_len == 256 and seed == 10 + rank
    assert (len(cam.dist) > 0 and cam.dist[0] != 0.0) == (rank % 3 != 2)
    n = 32
    frames = sharding.render_sequence(seed, n, cam, kind, workers=min(8, sharding.host_cpus()))
    fe = FrontEnd(cam, max_batch=n)
    try:
        gray = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
        depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
        Twc = np.stack([f[2] for f in frames]).astype(np.float64)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        Twc = Twc.astype(np.float32)
        fe.process(gray, depth, Tcw, Twc, th=15.0, check_ori=True, stream=torch.cuda.current_stream().cuda_stream)
        o = O.OrbOracle()
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        inv = np.float32(1.0) / np.float32(cam.depth_factor)
        of = []
        for s, (g, d, _) in enumerate(frames):
            okps, odesc = o(g)
            kps, desc = fe.keypoints(s)
            _same_kps(kps, okps)
            assert np.array_equal(desc, odesc)
            fo = O.FrameOracle(okps, odesc, O.depth_to_float(d, inv), K4, cam.bf, cam.w, cam.h, o.scale, dist=cam.dist)
            of.append(fo)
            un, oun = fe.ctx.download_keys_un(s, fo.N), fo.keys_un()
            for f in ("x", "y"):
                assert np.array_equal(un[f].view(np.uint32), oun[f].view(np.uint32)), (s, f)
            ur, z = fe.ctx.download_stereo(s)
            assert np.array_equal(ur[:fo.N].view(np.uint32), fo.uRight.view(np.uint32))
            assert np.array_equal(z[:fo.N].view(np.uint32), fo.depth.view(np.uint32))
            off, idx = fe.ctx.download_grid(s)
            ooff, oidx = fo.grid_csr()
            assert np.array_equal(off, ooff) and np.array_equal(idx, oidx)
        total = 0
        for s in range(1, n):
            world, valid = of[s - 1].unproject(Twc[s - 1])
            mp = np.zeros(of[s - 1].N, O.MAPPOINT_DTYPE)
            mp["valid"], mp["obsPositive"], mp["world"], mp["desc"] = valid, 1, world, of[s - 1].desc
            no, mo = O.search_by_projection_last(of[s], of[s - 1], Tcw[s], Tcw[s - 1], mp, 15.0, False, True)
            m, nm = fe.matches(s)
            assert nm == no and np.array_equal(m[:of[s].N], mo), s
            total += nm
        assert total > 100 * (n - 1)
    finally:
        fe.ctx.close()


def test_batch_entries_of_three_contexts_side_by_side():
    """What bench.py's full_frontend does - steps in flight on separate contexts, each running its line, plane and CAPE batch on
    its own threads at the same time - must give every context the results it gets alone (no state shared between contexts)."""
    import threading
    from dr_slam_amd import lib, synth
    cams = [synth.ICL, synth.TUM3, synth.ICL]
    kinds = ["living_room", "room_boxes", "corridor"]
    data = []
    for k in range(3):
        frames = list(synth.sequence(40 + k, 6, cam=cams[k], kind=kinds[k]))
        cam = cams[k]
        K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
        inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
        gray = np.stack([f[0] for f in frames]); depth = np.stack([f[1] for f in frames])
        data.append((gray, depth, depth.astype(np.float32) * np.float32(inv), K4, inv))
    ctxs = [lib.Context(max_batch=1) for _ in range(3)]

    def run(k, out):
        gray, depth, depth_m, K4, inv = data[k]
        c = ctxs[k]
        res = {}
        th = [threading.Thread(target=lambda: res.__setitem__("lines", c.lsd_extract_batch(gray, n_threads=3))),
              threading.Thread(target=lambda: res.__setitem__("planes", c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=2, seg=True))),
              threading.Thread(target=lambda: res.__setitem__("cape", c.planes_cape_batch(depth_m, K4, 20, n_threads=2, seg=True)))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        out[k] = res

    def same(a, b):
        for la, lb in zip(a["lines"], b["lines"]):
            assert la["lines"].tobytes() == lb["lines"].tobytes() and np.array_equal(la["desc"], lb["desc"]) and la["detected"] == lb["detected"]
        for x, y in zip(a["planes"], b["planes"]):
            assert x.tobytes() == y.tobytes()
        for x, y in zip(a["cape"], b["cape"]):
            assert x.tobytes() == y.tobytes()

    try:
        alone = {}
        for k in range(3):
            run(k, alone)
        for _ in range(2):
            together = {}
            th = [threading.Thread(target=run, args=(k, together)) for k in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for k in range(3):
                same(alone[k], together[k])
        assert sum(len(l["lines"]) for l in alone[0]["lines"]) > 50 and alone[1]["planes"][1].sum() > 6
    finally:
        for c in ctxs:
            c.close()


def test_bench_size_three_in_flight_against_oracle(oracle_mod):
    """The bench's own configuration under the oracle: batches of 512 frames through drfe_pipeline_submit with three contexts
    in flight (bench.py's timed loop), two rounds so that every context is reused while the others run; then 16 random slots
    of every context's last batch - keypoint records, descriptors, SearchByProjection match arrays - against the CPU oracle
    (oracle/spot_check.py, the same checker bench.py runs after its timed loop)."""
    import torch
    from dr_slam_amd import lib, sharding, synth
    from dr_slam_amd.pipeline import FrontEnd
    from oracle.spot_check import SlotChecker
    cam = synth.TUM3
    B, depth_n, distinct = 512, 3, 24
    batches = []
    for i in range(depth_n):
        base = sharding.render_sequence(50 + i, distinct, cam, ("room_boxes", "living_room", "corridor")[i], workers=min(8, sharding.host_cpus()))
        order = sharding.pingpong_order(B, distinct)
        gray = np.stack([base[k][0] for k in order])
        depth = np.stack([base[k][1] for k in order])
        Twc = np.stack([base[k][2] for k in order]).astype(np.float64)
        Tcw = np.linalg.inv(Twc).astype(np.float32)
        batches.append((gray, depth, Tcw, Twc.astype(np.float32), order,
                        torch.from_numpy(gray).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()))
    torch.cuda.synchronize()
    pipe = lib.Pipeline(depth_n, max_width=cam.w, max_height=cam.h, max_batch=B)
    try:
        views = [FrontEnd(cam, max_batch=B, ctx=c) for c in pipe.contexts]
        for rnd in range(2):
            for i, (gray, depth, Tcw, Twc, order, g_t, d_t) in enumerate(batches):
                k = pipe.submit(g_t.data_ptr(), d_t.data_ptr(), cam.w * cam.h, cam.w, cam.w, cam.h, Tcw, Twc, views[0].cam, 15.0, False, True, B)
                assert k == i
        pipe.sync()
        rng = np.random.default_rng(7)
        checked = 0
        for i, (gray, depth, Tcw, Twc, order, g_t, d_t) in enumerate(batches):
            slots = np.sort(rng.choice(np.arange(1, B), size=16, replace=False))
            slots[0], slots[-1] = 1, B - 1                       # the batch's first pair and its last slot are always among them
            checked += SlotChecker(cam).check(views[i], gray, depth, Tcw, Twc, slots, keys=order, what="context %d " % i)
            assert views[i].ctx.orb_counts(B).min() > 300
        assert checked == 16 * depth_n
    finally:
        pipe.close()


def test_native_shard_rccl_single_rank():
    """drfe_shard_* (RCCL loaded at run time) with one rank on this box's one GPU: communicator from a unique id, the vocabulary
    blob broadcast from rank 0 (identity at one rank, but through ncclBroadcast on device memory), the end-of-run MAX / SUM
    reduction.  Two ranks need two devices (RCCL refuses two ranks on one): the N > 1 path is covered by the world_size-2 gloo
    test of dr_slam_amd.sharding on the CPU and runs on the driver's 8-GPU node."""
    from dr_slam_amd import lib, vocabulary as V
    ident = lib.Shard.unique_id()
    assert ident.shape == (128,) and ident.any()
    sh = lib.Shard(ident, 1, 0, device=0)
    try:
        blob = V.make_synthetic(6, 3, seed=2).pack()
        ref = blob.copy()
        sh.broadcast(blob, root=0)
        assert np.array_equal(blob, ref)
        m, s = sh.reduce_report([1.5, 0.25], [512, 7])
        assert m.tolist() == [1.5, 0.25] and s.tolist() == [512, 7]
        with pytest.raises(lib.DrfeError):
            sh.broadcast(blob, root=1)
    finally:
        sh.close()


def test_long_kernel_clock_reports_every_kernel_of_the_batch_entries():
    """drfe_long_kernel_clock / drfe_long_kernel_ms (what bench.py prices the long kernels' roofline entries with): with the clock on,
    the line and plane batch entries report a positive duration for every kernel that ran on the device, the results are the ones of
    an unclocked call, and with it off the last values stay."""
    from dr_slam_amd import lib, synth
    cam = synth.ICL
    frames = list(synth.sequence(3, 6, cam=cam, kind="living_room"))
    gray = np.stack([f[0] for f in frames]); depth = np.stack([f[1] for f in frames])
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    c = lib.Context(max_batch=1)
    try:
        plain_l = c.lsd_extract_batch(gray, n_threads=2)
        plain_p = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=2)
        assert all(v == 0 for v in c.long_kernel_ms().values())
        c.long_kernel_clock(True)
        timed_l = c.lsd_extract_batch(gray, n_threads=2)
        timed_p = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=2)
        ms = c.long_kernel_ms()
        for k in ("lines_image_passes", "k_lsd_keys", "k_lsd_order", "k_lsd_grow", "k_rect_improve", "k_lsd_keylines+k_lbd",
                  "k_ahc_blocks", "k_ahc_cluster", "k_ahc_refine", "k_ahc_labels", "k_voxel_grid", "k_plane_refit"):
            assert 0.0 < ms[k] < 5000.0, (k, ms)
        for a, b in zip(plain_l, timed_l):
            assert np.array_equal(a["lines"].view(np.uint8), b["lines"].view(np.uint8)) and np.array_equal(a["desc"], b["desc"])
        for a, b in zip(plain_p[:2], timed_p[:2]):
            assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
        c.long_kernel_clock(False)
        c.lsd_extract_batch(gray, n_threads=2)
        assert c.long_kernel_ms() == ms
    finally:
        c.close()

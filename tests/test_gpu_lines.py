"""-m gpu parity tests of the line-feature path (LineSegment::ExtractLineSegment): device image passes
(Gaussian + 0.8 exact downscale, gradient magnitude / level-line angle, Gaussian + Sobel) and the
product's region growing / NFA / LBD vs the CPU oracle.  Bar: identical bytes for the stage images,
identical float bit patterns for key-line fields, identical 256-bit LBD descriptors, bit-identical
line equations.  (The oracle itself is an unpinned restatement of OpenCV 3.4 LSD/LBD — DESIGN.md §5.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PAIRS = [("angle", "angle"), ("class_id", "class_id"), ("octave", "octave"), ("pt_x", "ptX"), ("pt_y", "ptY"),
         ("response", "response"), ("size", "size"), ("start_point_x", "startPointX"), ("start_point_y", "startPointY"),
         ("end_point_x", "endPointX"), ("end_point_y", "endPointY"), ("s_point_in_octave_x", "sPointInOctaveX"),
         ("s_point_in_octave_y", "sPointInOctaveY"), ("e_point_in_octave_x", "ePointInOctaveX"),
         ("e_point_in_octave_y", "ePointInOctaveY"), ("line_length", "lineLength"), ("num_of_pixels", "numOfPixels")]


@pytest.fixture(scope="module")
def ctx():
    from dr_slam_amd import lib
    c = lib.Context(max_batch=1)
    yield c
    c.close()


def _frame(seed, kind):
    from dr_slam_amd import synth
    g, _, _ = next(synth.sequence(seed, 1, kind=kind))
    return g


@pytest.fixture(params=[0, 1, 2], ids=["lsd_source_text", "lsd_paper", "lsd_round4"])
def rect_mode(request, ctx):
    """The reading of OpenCV 3.4's lsd.cpp (drfe_lsd_configure_rect): 0 the source text (rect_nfa's integer corners, nfa()'s
    `double(n) + 1` first term; default), 1 the LSD paper's reading of both (rounds 2-3), 2 integer corners with log_gamma(n + 1)
    (round 4's default).  Every oracle-parity test of the line path runs under all three."""
    ctx.lsd_configure_rect(request.param)
    yield request.param
    ctx.lsd_configure_rect(0)


@pytest.mark.parametrize("seed,kind", [(2, "room_boxes"), (1, "planar_lowtexture"), (5, "corridor"), (3, "living_room")])
def test_lines_bit_exact(ctx, oracle_mod, seed, kind, rect_mode):
    g = _frame(seed, kind)
    a = ctx.lsd_extract(g, stages=True)
    o = oracle_mod.extract_lines(g, stages=True, rect_mode=rect_mode)
    for k in ("scaled", "gx", "gy"):
        assert np.array_equal(a[k], o[k]), k
    assert np.array_equal(a["modgrad"].view(np.uint64), o["modgrad"].view(np.uint64))
    assert np.array_equal(a["angles"].view(np.uint64), o["angles"].view(np.uint64))
    assert a["detected"] == o["detected"] and len(a["lines"]) == len(o["lines"]) == min(40, o["detected"])
    assert len(a["lines"]) >= 20
    for gk, ok in PAIRS:
        assert np.array_equal(a["lines"][gk].view(np.uint32), o["lines"][ok].view(np.uint32)), gk
    assert np.array_equal(a["desc"], o["desc"])
    assert np.array_equal(a["lineF"].view(np.uint64), o["lineF"].view(np.uint64))
    assert (np.unpackbits(a["desc"], axis=1).sum(1) > 20).all()


def test_lines_analytic_polygons_bit_exact(ctx, oracle_mod, rect_mode):
    """The analytic polygon scene whose segments tests/test_oracle_cpu2.py checks in closed form: the device path reports
    the same segments as the oracle, bit for bit - all 11 under the real-valued rect_nfa, the ones the literal OpenCV 3.4
    rect_nfa validates (it rejects most oblique rectangles) under the default."""
    from line_scenarios import analytic_polygons
    g, edges = analytic_polygons()
    a = ctx.lsd_extract(g)
    o = oracle_mod.extract_lines(g, rect_mode=rect_mode)
    assert a["detected"] == o["detected"] and len(edges) == 11
    assert (o["detected"] == 11) if rect_mode == 1 else (2 <= o["detected"] < 11)
    for gk, ok in PAIRS:
        assert np.array_equal(a["lines"][gk].view(np.uint32), o["lines"][ok].view(np.uint32)), gk
    assert np.array_equal(a["desc"], o["desc"])
    assert np.array_equal(a["lineF"].view(np.uint64), o["lineF"].view(np.uint64))


def test_lsd_matcher(ctx, oracle_mod, frames_room):
    """LSDmatcher::SearchByDescriptor(KF, Frame) (ratio rule) and the KF-KF / initialisation variant
    (MAD-gap rule) on the LBD descriptors of two consecutive frames."""
    a = ctx.lsd_extract(frames_room[0][0])
    b = ctx.lsd_extract(frames_room[1][0])
    rng = np.random.default_rng(2)
    has = (rng.random(len(a["desc"])) > 0.2).astype(np.uint8)
    n_o, m_o = oracle_mod.lsd_search_by_descriptor(a["desc"], has, b["desc"])
    n_g, m_g = ctx.lsd_search_by_descriptor(a["desc"], b["desc"], has, mode=0)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_o >= 5
    has_t = (rng.random(len(b["desc"])) > 0.2).astype(np.uint8)
    n_o, m_o = oracle_mod.lsd_search_by_gap(a["desc"], b["desc"], has_t)
    n_g, m_g = ctx.lsd_search_by_descriptor(a["desc"], b["desc"], has_t, mode=1)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_o >= 5
    n_g, m_g = ctx.lsd_search_by_descriptor(a["desc"], b["desc"][:1], None, mode=0)     # knn k=2 needs 2 train rows
    assert n_g == 0 and (m_g == -1).all()
    # LSDmatcher::SearchForTriangulation (LocalMapping::CreateNewMapLines): tenth-of-MAD gap, neither side has a MapLine
    h1 = (rng.random(len(a["desc"])) < 0.3).astype(np.uint8)
    h2 = (rng.random(len(b["desc"])) < 0.3).astype(np.uint8)
    n_o, m_o = oracle_mod.lsd_search_for_triangulation(a["desc"], b["desc"], h1, h2)
    n_g, m_g = ctx.lsd_search_for_triangulation(a["desc"], b["desc"], h1, h2)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_o >= 5
    assert (m_o[h1 == 1] == -1).all() and not h2[m_o[m_o >= 0]].any()


def test_flat_image_has_no_lines(ctx, oracle_mod):
    g = np.full((480, 640), 99, np.uint8)
    a = ctx.lsd_extract(g)
    assert len(a["lines"]) == 0 and a["detected"] == 0
    assert oracle_mod.extract_lines(g)["detected"] == 0


def test_rectangle_edges(ctx, oracle_mod):
    """Clean synthetic edges: the long vertical edges are found with the expected geometry."""
    g = np.full((480, 640), 60, np.uint8)
    g[100:300, 150:450] = 180
    a = ctx.lsd_extract(g)
    o = oracle_mod.extract_lines(g)
    assert len(a["lines"]) == len(o["lines"]) >= 2
    xs = sorted(float(v) for v in a["lines"]["start_point_x"])
    assert abs(xs[0] - 149.4) < 1.0 and abs(xs[-1] - 449.4) < 1.0
    assert np.array_equal(a["desc"], o["desc"])


@pytest.mark.parametrize("seed,motion,th", [(1, 0.0, 15.0), (2, 0.5, 15.0), (3, -0.5, 15.0), (4, 0.0, 7.0), (5, 0.0, 30.0),
                                            (11, 0.0, 15.0), (12, 0.2, 20.0)])
def test_lsd_search_by_projection(ctx, oracle_mod, seed, motion, th):
    """LSDmatcher::SearchByProjection, both overloads (row a-15): device replay vs the oracle, identical claims."""
    import os
    import line_scenarios as LS
    from dr_slam_amd import lib
    O = oracle_mod
    sc = LS.make(seed, lib.KEYLINE_DTYPE, lib.MAPLINE_DTYPE, lib.TRACKED_LINE_DTYPE, n_cur=40 if seed < 10 else 150,
                 n_last=48 if seed < 10 else 200, motion=motion)
    cam = lib.Camera(**LS.CAM) if hasattr(lib, "Camera") else None
    assert cam is not None
    n, ml = ctx.lsd_search_by_projection_last(sc["Tcw_cur"], sc["Tcw_last"], cam, sc["last"], sc["cur"], sc["cur_desc"], th,
                                              False, 0.9, sc["cur_ml"], sc["cur_obs"])
    no, mlo = O.lsd_search_by_projection_last(LS.cam9(), sc["Tcw_cur"], sc["Tcw_last"], LS.SCALE, sc["last"], sc["cur"],
                                              sc["cur_desc"], th, False, 0.9, sc["cur_ml"], sc["cur_obs"])
    assert n == no and np.array_equal(ml, mlo)
    assert n > 5
    n, ml = ctx.lsd_search_by_projection_map(sc["tracked"], sc["cur"], sc["cur_desc"], th / 15.0, 0.9, sc["cur_ml"], sc["cur_obs"])
    no, mlo = O.lsd_search_by_projection_map(LS.SCALE, sc["tracked"], sc["cur"], sc["cur_desc"], th / 15.0, 0.9, sc["cur_ml"],
                                             sc["cur_obs"])
    assert n == no and np.array_equal(ml, mlo)
    if seed < 10:
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lsd_projection.npz"))
        assert np.array_equal(np.concatenate([[n], ml]), g[f"map_{seed}"])


def test_lsd_search_by_projection_empty(ctx):
    from dr_slam_amd import lib
    import line_scenarios as LS
    cam = lib.Camera(**LS.CAM)
    T = np.eye(4, dtype=np.float32)
    n, ml = ctx.lsd_search_by_projection_last(T, T, cam, np.zeros(0, lib.MAPLINE_DTYPE), np.zeros(0, lib.KEYLINE_DTYPE),
                                              np.zeros((0, 32), np.uint8), 15.0, False, 0.9, np.zeros(0, np.int32))
    assert n == 0 and len(ml) == 0


def _same_lines(a, b):
    assert a["detected"] == b["detected"] and len(a["lines"]) == len(b["lines"])
    assert np.array_equal(a["lines"].view(np.uint8), b["lines"].view(np.uint8))
    assert np.array_equal(a["desc"], b["desc"])
    assert np.array_equal(a["lineF"].view(np.uint64), b["lineF"].view(np.uint64))


@pytest.mark.parametrize("device_grow", [True, False, 2, 3], ids=["device_auto", "host", "one_wave_per_frame", "four_waves_per_frame"])
def test_lsd_extract_batch_equals_single(ctx, device_grow):
    """drfe_lsd_extract_batch == per-frame calls for any thread count, with region growing on the device (k_lsd_grow: one
    wavefront per frame replays the detector's seed loop) and on the host pool."""
    from dr_slam_amd import synth
    frames = [f[0] for f in synth.sequence(2, 6, kind="room_boxes")] + [_frame(5, "corridor")]
    single = [ctx.lsd_extract(g) for g in frames]
    ctx.lsd_configure(device_grow)
    try:
        for threads in (1, 3, 16):
            batch = ctx.lsd_extract_batch(np.stack(frames), n_threads=threads)
            assert len(batch) == len(frames)
            for a, b in zip(batch, single):
                assert len(a["lines"]) > 5
                _same_lines(a, b)
    finally:
        ctx.lsd_configure(True)


@pytest.mark.parametrize("kind,seed", [("living_room", 3), ("room_boxes", 7), ("corridor", 5), ("planar_lowtexture", 4)])
def test_lsd_device_grow_matches_oracle(ctx, oracle_mod, kind, seed, rect_mode):
    """The device region growing against the CPU oracle (own restatement, glibc cos / sin) on every scene kind: key lines,
    LBD descriptors and line equations of 12 frames per kind, bit for bit; more chunks than one (chunking is by 16 frames)."""
    from dr_slam_amd import synth
    frames = [f[0] for f in synth.sequence(seed, 12, cam=synth.ICL if kind == "living_room" else synth.TUM3, kind=kind)]
    frames = frames + frames[:8]                      # 20 frames: two chunks
    batch = ctx.lsd_extract_batch(np.stack(frames), n_threads=4)
    for g, a in zip(frames[:12], batch):
        o = oracle_mod.extract_lines(g, rect_mode=rect_mode)
        assert a["detected"] == o["detected"] and len(a["lines"]) == len(o["lines"])
        for gk, ok in PAIRS:
            assert np.array_equal(a["lines"][gk].view(np.uint32), o["lines"][ok].view(np.uint32)), gk
        assert np.array_equal(a["desc"], o["desc"])
        assert np.array_equal(a["lineF"].view(np.uint64), o["lineF"].view(np.uint64))
    for a, b in zip(batch[12:], batch[:8]):
        _same_lines(a, b)


@pytest.mark.parametrize("grow", [2, 3], ids=["one_wave_per_frame", "four_waves_per_frame"])
@pytest.mark.parametrize("kind,seed", [("living_room", 13), ("planar_lowtexture", 14)])
def test_lsd_each_growth_kernel_matches_oracle(ctx, oracle_mod, kind, seed, grow):
    """k_lsd_grow (one wavefront per frame) and k_lsd_grow_mw (four: seeds speculated, committed in seed order), each forced by
    drfe_lsd_configure, against the CPU oracle: key lines, descriptors, line equations bit for bit, and no frame handed back."""
    from dr_slam_amd import synth
    frames = [f[0] for f in synth.sequence(seed, 10, cam=synth.ICL if kind == "living_room" else synth.TUM3, kind=kind)]
    ctx.lsd_configure(grow)
    try:
        s0 = ctx.lsd_stats()
        batch = ctx.lsd_extract_batch(np.stack(frames), n_threads=4)
        s1 = ctx.lsd_stats()
    finally:
        ctx.lsd_configure(True)
    assert s1["frames"] - s0["frames"] == len(frames) and s1["grow_to_host"] == s0["grow_to_host"]
    for g, a in zip(frames, batch):
        o = oracle_mod.extract_lines(g)
        assert a["detected"] == o["detected"] and len(a["lines"]) == len(o["lines"])
        for gk, ok in PAIRS:
            assert np.array_equal(a["lines"][gk].view(np.uint32), o["lines"][ok].view(np.uint32)), gk
        assert np.array_equal(a["desc"], o["desc"])
        assert np.array_equal(a["lineF"].view(np.uint64), o["lineF"].view(np.uint64))


def test_lsd_device_grow_polygons_and_odd_size(ctx, oracle_mod, rect_mode):
    """Analytic polygons (every edge must be found where it was drawn) and a frame size whose scaled width is not a multiple
    of 64 through the device path; flat and noise-only images give no lines."""
    imgs = []
    rng = np.random.default_rng(5)
    for k in range(3):
        img = np.full((480, 640), 40, np.uint8)
        import itertools
        pts = np.array([[120 + 60 * k, 90], [520, 130 + 40 * k], [470 - 30 * k, 400], [150, 350]], np.int32)
        yy, xx = np.mgrid[0:480, 0:640]
        inside = np.ones((480, 640), bool)
        for i in range(4):
            (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % 4]
            inside &= ((x1 - x0) * (yy - y0) - (y1 - y0) * (xx - x0)) >= 0
        img[inside] = 200
        imgs.append(img)
    from line_scenarios import analytic_polygons
    gp, edges = analytic_polygons()
    if gp.shape == (480, 640):
        imgs.append(gp)
    imgs.append(np.full((480, 640), 90, np.uint8))
    imgs.append(rng.integers(0, 255, (480, 640)).astype(np.uint8))
    batch = ctx.lsd_extract_batch(np.stack(imgs), n_threads=2)
    for g, a in zip(imgs, batch):
        _same_lines(a, ctx.lsd_extract(g))
    # four sides under the real-valued rect_nfa; the literal one validates one or two of these oblique quadrilaterals' sides
    assert all(len(b["lines"]) >= (4 if rect_mode == 1 else 1) for b in batch[:3]) and len(batch[-2]["lines"]) == 0
    if gp.shape == (480, 640):
        assert batch[3]["detected"] == (len(edges) if rect_mode == 1 else oracle_mod.extract_lines(gp, rect_mode=rect_mode)["detected"])
    odd = [f[0][:403, :531].copy() for f in synth_frames_odd()]
    for g, a in zip(odd, ctx.lsd_extract_batch(np.stack(odd), n_threads=2)):
        _same_lines(a, ctx.lsd_extract(g))


def synth_frames_odd():
    from dr_slam_amd import synth
    return list(synth.sequence(9, 3, kind="room_boxes"))


@pytest.mark.parametrize("seed", [1, 2, 3, 11])
def test_lsd_fuse_search(oracle_mod, seed):
    """Search part of LSDmatcher::Fuse(pKF, vpMapLines, 3.0) (LocalMapping::SearchInNeighbors): projection gates, cone test,
    unclamped PredictScale (levels outside the pyramid are reported as -2), KeyFrame::GetLinesInArea, octave window, first
    minimum.  Bit-equal best index and distance per map line."""
    import line_scenarios as LS
    from dr_slam_amd import lib
    n_kf, n = (40, 64) if seed < 10 else (300, 2000)
    sc = LS.make(seed, lib.KEYLINE_DTYPE, lib.MAPLINE_DTYPE, lib.TRACKED_LINE_DTYPE, n_cur=n_kf, n_last=n)
    rng = np.random.RandomState(100 + seed)
    Tcw = sc["Tcw_cur"]
    Twc = np.linalg.inv(Tcw.astype(np.float64))
    lines = np.zeros(n, lib.FRUSTUM_LINE_DTYPE)
    lines["world"] = sc["last"]["world"]
    mid = 0.5 * (lines["world"][:, :3] + lines["world"][:, 3:])
    om = mid - Twc[:3, 3][None, :]
    dist = np.linalg.norm(om, axis=1)
    nrm = om / dist[:, None] + rng.normal(0, 0.5, (n, 3))            # some outside the 60-degree cone
    lines["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    lvl = rng.choice([-1, 0, 1, 2, 9], size=n, p=[0.04, 0.4, 0.4, 0.12, 0.04])
    lines["max_distance"] = (dist * 1.2 ** (lvl - rng.uniform(0.1, 0.9, n))).astype(np.float32)
    lines["min_distance"] = (dist * rng.uniform(0.3, 1.3, n)).astype(np.float32)
    descs = sc["last"]["desc"]
    skip = (rng.uniform(size=n) < 0.1).astype(np.uint8)
    ctx = lib.Context(max_batch=1)
    try:
        cam = lib.Camera(**LS.CAM)
        for th in (3.0, 12.0):
            bi, bd = ctx.lsd_fuse_search(Tcw, cam, lines, descs, skip, sc["cur"], sc["cur_desc"], th)
            obi, obd = oracle_mod.lsd_fuse_search(LS.cam9(), Tcw, 1.2, LS.SCALE, lines, descs, skip, sc["cur"], sc["cur_desc"], th)
            assert np.array_equal(bi, obi) and np.array_equal(bd, obd), th
        assert (obi[skip == 1] == -1).all()
        assert (obi >= 0).sum() > n // 8 and ((obi >= 0) & (obd <= 50)).sum() > n // 20
        assert seed < 10 or (obi == -2).sum() > 10
        # a keyframe without key lines
        bi, bd = ctx.lsd_fuse_search(Tcw, cam, lines, descs, skip, sc["cur"][:0], sc["cur_desc"][:0], 3.0)
        assert (bi < 0).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 11])
def test_lsd_fuse_and_projection_with_similarity(oracle_mod, seed):
    """LSDmatcher::Fuse(pKF, Scw, ...) (the search) and LSDmatcher::SearchByProjection(pKF, Scw, vpLines, vpMatched, th): the pose is
    the similarity Scw = s [R | t] decomposed as the reference does; the second one claims key lines first come, first served
    (every map line is offered twice so that the claims matter)."""
    import line_scenarios as LS
    from dr_slam_amd import lib
    n_kf, n = (40, 64) if seed < 10 else (300, 1500)
    sc, lines, rng = LS.sim3_line_scene(seed, n_kf, n, lib.KEYLINE_DTYPE, lib.MAPLINE_DTYPE, lib.TRACKED_LINE_DTYPE, lib.FRUSTUM_LINE_DTYPE)
    Tcw = sc["Tcw_cur"]
    descs = sc["last"]["desc"]
    skip = (rng.uniform(size=n) < 0.1).astype(np.uint8)
    ctx = lib.Context(max_batch=1)
    try:
        cam = lib.Camera(**LS.CAM)
        total = 0
        for s in (1.0, 0.83, 1.7):
            Scw = Tcw.astype(np.float32).copy()
            Scw[:3, :] *= np.float32(s)
            for th in (3.0, 12.0):
                bi, bd = ctx.lsd_fuse_search_sim3(Scw, cam, lines, descs, skip, sc["cur"], sc["cur_desc"], th)
                obi, obd = oracle_mod.lsd_fuse_search_sim3(LS.cam9(), Scw, 1.2, LS.SCALE, lines, descs, skip, sc["cur"], sc["cur_desc"], th)
                assert np.array_equal(bi, obi) and np.array_equal(bd, obd), (s, th)
                assert (obi[skip == 1] == -1).all()
                total += int(((obi >= 0) & (obd <= 50)).sum())
            # first-come claims: the list twice, a fifth of the key lines matched on entry
            l2 = np.concatenate([lines, lines]); d2 = np.concatenate([descs, descs]); s2 = np.concatenate([skip, skip])
            matched = (rng.uniform(size=len(sc["cur"])) < 0.2).astype(np.uint8)
            for th in (4, 10):
                nm, new = ctx.lsd_search_by_projection_kf(Scw, cam, l2, d2, s2, sc["cur"], sc["cur_desc"], matched, th)
                onm, onew = oracle_mod.lsd_search_by_projection_kf(LS.cam9(), Scw, 1.2, LS.SCALE, l2, d2, s2, sc["cur"], sc["cur_desc"],
                                                                  matched, th)
                assert nm == onm and np.array_equal(new, onew), (s, th)
                assert (new[matched == 1] == -1).all() and (new >= 0).sum() == nm
                total += nm
        assert total > (20 if seed < 10 else 400)
        nm, new = ctx.lsd_search_by_projection_kf(Scw, cam, lines, descs, skip, sc["cur"][:0], sc["cur_desc"][:0], np.zeros(0, np.uint8), 4)
        assert nm == 0 and len(new) == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 11])
def test_lsd_search_by_sim3(oracle_mod, seed):
    """LSDmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): the map lines of each keyframe carried into the other
    one with the similarity (two chained float transforms), KeyFrame::IsInImage, the distance band of the midpoint, unclamped
    PredictScale, GetLinesInArea + octave window, TH_HIGH, and the agreement of the two directions.  Both keyframes see the same
    n lines from nearby poses (key line i of either carries map line i), a sixth of them skipped on either side."""
    import line_scenarios as LS
    from dr_slam_amd import lib
    K = LS.keyframe_pair_for_sim3(seed, lib.KEYLINE_DTYPE, lib.FRUSTUM_LINE_DTYPE)
    n, perm, skip1 = K["n"], K["perm"], K["skip1"]
    args = (K["T1w"], K["T2w"], K["s12"], K["R12"], K["t12"])
    sides = (K["lines1"], K["descs1"], K["skip1"], K["kl1"], K["kd1"], K["lines2"], K["descs2"], K["skip2"], K["kl2"], K["kd2"])
    ctx = lib.Context(max_batch=1)
    try:
        cam = lib.Camera(**LS.CAM)
        found = 0
        for th in (7.5, 20.0):
            nf, m12 = ctx.lsd_search_by_sim3(cam, *args, *sides, th)
            onf, om12 = oracle_mod.lsd_search_by_sim3(LS.cam9(), *args, 1.2, LS.SCALE, *sides, th)
            assert nf == onf and np.array_equal(m12, om12), th
            assert (m12 >= 0).sum() == nf and (m12[skip1 == 1] == -1).all()
            ok = m12 >= 0
            assert (perm[m12[ok]] == np.flatnonzero(ok)).mean() > 0.9 if ok.any() else True     # agreements are the true pairs
            found += nf
        assert found > n // 4
    finally:
        ctx.close()


def test_device_order_sort_equals_std_sort(ctx):
    """k_lsd_order (introsort's element moves on the device) against std::sort itself under lsd.cpp's compare_norm: the
    permutation of equal bins must be libstdc++'s.  Random / few-valued / constant / sorted / organ-pipe / image-like key arrays
    around the range thresholds of the kernel (16, 64, 1024, 8192) and at the size of a 640 x 480 frame."""
    import ctypes as C
    from dr_slam_amd import lib, synth
    L = lib.load()
    rng = np.random.default_rng(11)

    def keys_of(bins):
        n = len(bins)
        idx = np.arange(n, dtype=np.uint32)
        return (bins.astype(np.uint32) << 22) | ((idx // 2047) << 11) | (idx % 2047)

    def check(bins, what):
        k = keys_of(np.asarray(bins))
        dev, ref = k.copy(), k.copy()
        st = C.c_int(-1)
        assert L.drfe_debug_device_order_sort(ctx.h, dev.ctypes.data_as(C.c_void_p), len(dev), C.byref(st)) == 0, ctx.last_error()
        assert L.drfe_debug_order_sort(ref.ctypes.data_as(C.c_void_p), len(ref), 0, 0, -1, 0) == 0
        if what == "organ pipe" and st.value == 1:
            return                       # median-of-three's bad case: introsort falls to heap sort, the device hands the frame to the host
        assert st.value == 0, (what, st.value)
        assert np.array_equal(dev, ref), (what, len(k), int(np.argmax(dev != ref)))

    for n in (1, 2, 16, 17, 33, 64, 65, 100, 1000, 1024, 1025, 4097, 8192, 8193, 20000, 70001):
        check(rng.integers(0, 1024, n), "random")
        check(rng.integers(0, 3, n), "three bins")
        check(np.full(n, 7), "constant")
        check(np.sort(rng.integers(0, 1024, n)), "ascending")
        check(np.sort(rng.integers(0, 1024, n))[::-1], "descending")
        check(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]) % 1024, "organ pipe")
        check(np.minimum(1023, rng.exponential(40, n).astype(np.int64)), "exponential")
    # the keys of real level-line fields, as the batch path sorts them
    for kind, seed in (("living_room", 3), ("room_boxes", 2), ("planar_lowtexture", 1)):
        g = next(synth.sequence(seed, 1, cam=synth.ICL if kind == "living_room" else synth.TUM3, kind=kind))[0]
        a = ctx.lsd_extract(g, stages=True)
        mod = a["modgrad"][:-1, :-1]
        bins = (mod * (1023.0 / mod.max())).astype(np.int64).ravel()
        k = (bins.astype(np.uint32) << 22) | (np.repeat(np.arange(mod.shape[0], dtype=np.uint32), mod.shape[1]) << 11) | np.tile(np.arange(mod.shape[1], dtype=np.uint32), mod.shape[0])
        dev, ref = k.copy(), k.copy()
        st = C.c_int(-1)
        assert L.drfe_debug_device_order_sort(ctx.h, dev.ctypes.data_as(C.c_void_p), len(dev), C.byref(st)) == 0
        assert L.drfe_debug_order_sort(ref.ctypes.data_as(C.c_void_p), len(ref), 0, 0, -1, 0) == 0
        assert st.value == 0 and np.array_equal(dev, ref), kind


def test_device_order_sort_heap_branch_equals_libstdcxx(ctx):
    """introsort's depth limit forced low, so that ranges of every length reach std::__partial_sort (make_heap + sort_heap):
    the device's one-lane heap sort (introsort_device.h: heap_sort_range) must leave libstdc++'s permutation - compared with the
    plain transcription of std::__introsort_loop run at the same depth limit (drfe_debug_order_sort mode 3; at the natural limit
    that transcription is itself compared with std::sort in tests/test_host_cpu.py).  A range above 1024 keys at depth 0 is not
    heap-sorted by one lane: status 1, the caller's host path."""
    import ctypes as C
    from dr_slam_amd import lib
    L = lib.load()
    rng = np.random.default_rng(23)

    def keys_of(bins):
        n = len(bins)
        idx = np.arange(n, dtype=np.uint32)
        return (bins.astype(np.uint32) << 22) | ((idx // 2047) << 11) | (idx % 2047)

    reached = 0
    for n in (17, 18, 25, 33, 64, 100, 257, 1000, 1024, 3000, 20000, 70001):
        for what, bins in (("random", rng.integers(0, 1024, n)), ("three bins", rng.integers(0, 3, n)), ("constant", np.full(n, 7)),
                           ("organ pipe", np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]) % 1024),
                           ("exponential", np.minimum(1023, rng.exponential(40, n).astype(np.int64)))):
            lg = int(np.log2(n))
            for depth in sorted({0, 1, 2, 3, max(0, lg - 4), max(0, lg - 1), lg + 2}):
                k = keys_of(np.asarray(bins))
                dev, ref = k.copy(), k.copy()
                st = C.c_int(-1)
                assert L.drfe_debug_device_order_sort_depth(ctx.h, dev.ctypes.data_as(C.c_void_p), len(dev), depth, C.byref(st)) == 0, ctx.last_error()
                assert L.drfe_debug_order_sort(ref.ctypes.data_as(C.c_void_p), len(ref), 0, 3, depth, 0) == 0
                if st.value == 1:
                    assert n > 1024 and depth <= lg, (what, n, depth)        # some range above 1024 keys ran out of depth
                    continue
                assert st.value == 0, (what, n, depth, st.value)
                assert np.array_equal(dev, ref), (what, n, depth, int(np.argmax(dev != ref)))
                reached += 1
    assert reached > 200


def test_lsd_alignment_shortcut_equals_reference_arithmetic(ctx):
    """k_lsd_grow decides region_grow's alignment test from dot / cross products wherever the decision is not within 0.02 degrees
    of the tolerance (lsd_grow_kernels.hip: align_class) and evaluates cv::fastAtan2 only inside that band.  With
    DRFE_LSD_EXACT_ALIGN every test is the reference's arithmetic: both launches must return the same bytes - on top of the
    comparisons with the oracle, this holds the shortcut to the kernel's own long way on more frames than the oracle has time for."""
    import os
    from dr_slam_amd import synth
    for kind, cam, seed in (("living_room", synth.ICL, 41), ("room_boxes", synth.TUM3, 42), ("planar_lowtexture", synth.TUM3, 43), ("corridor", synth.ICL, 44)):
        gray = np.stack([f[0] for f in synth.sequence(seed, 12, cam=cam, kind=kind)])
        a = ctx.lsd_extract_batch(gray, n_threads=2)
        os.environ["DRFE_LSD_EXACT_ALIGN"] = "1"
        try:
            b = ctx.lsd_extract_batch(gray, n_threads=2)
        finally:
            del os.environ["DRFE_LSD_EXACT_ALIGN"]
        assert len(a) == len(b) == 12
        for x, y in zip(a, b):
            assert x["detected"] == y["detected"] and x["lines"].tobytes() == y["lines"].tobytes(), kind
            assert np.array_equal(x["desc"], y["desc"]) and x["lineF"].tobytes() == y["lineF"].tobytes(), kind


def test_lsd_batch_device_edge_cases(ctx):
    """drfe_lsd_extract_batch with the sequential core on the device, on frames at the edges of what the kernels assume, all in
    one batch: a constant image (no pixel has a level-line angle: nothing to order, no seed), uniform noise (every pixel a seed,
    tens of thousands of one-pixel regions), a step edge (one long region: the queue phase beyond the 7 x 7 window, the
    largest rectangle), a dense checkerboard (thousands of accepted rectangles), and an odd size.  Identical to the single-frame
    entry (host growth)."""
    rng = np.random.default_rng(5)
    h, w = 480, 640
    flat = np.full((h, w), 97, np.uint8)
    noise = rng.integers(0, 256, (h, w)).astype(np.uint8)
    step = np.zeros((h, w), np.uint8); step[:, w // 2:] = 200
    yy, xx = np.mgrid[0:h, 0:w]
    checker = ((((yy // 12) + (xx // 12)) % 2) * 180 + 30).astype(np.uint8)
    diag = (((xx + 2 * yy) // 40) % 2 * 150 + 40).astype(np.uint8)
    batch = np.stack([flat, noise, step, checker, diag])
    ctx.lsd_configure(True)
    for frames in (batch, batch[:, :257, :333].copy()):
        got = ctx.lsd_extract_batch(frames, n_threads=3)
        for f in range(len(frames)):
            a = ctx.lsd_extract(frames[f])
            assert a["detected"] == got[f]["detected"] and a["lines"].tobytes() == got[f]["lines"].tobytes()
            assert np.array_equal(a["desc"], got[f]["desc"]) and a["lineF"].tobytes() == got[f]["lineF"].tobytes()
    assert len(got[0]["lines"]) == 0 and got[2]["detected"] >= 1 and got[3]["detected"] >= 40


def test_lsd_device_nfa_equals_host_nfa(ctx, oracle_mod, rect_mode):
    """rect_improve's decisions on the device (k_rect_improve: certified comparisons on its own exp / log10 / pow) against
    the host pool's (the caller's libm) for every scene kind, and against the oracle; frames whose decisions the device
    could not certify return to the host and are counted."""
    from dr_slam_amd import synth
    frames = []
    for kind, seed in (("living_room", 3), ("room_boxes", 7), ("corridor", 5), ("planar_lowtexture", 4)):
        frames += [f[0] for f in synth.sequence(seed, 5, cam=synth.ICL if kind == "living_room" else synth.TUM3, kind=kind)]
    yy, xx = np.mgrid[0:480, 0:640]
    for sl in (0.2, 0.6, 1.0, 1.7):                       # long clean edges: NFA values far out in the tail (subnormal first terms)
        g = np.full((480, 640), 50, np.uint8)
        g[(yy - sl * xx) > 40] = 200
        frames.append(g)
    batch = np.stack(frames)
    s0 = ctx.lsd_stats()
    dev = ctx.lsd_extract_batch(batch, n_threads=4)
    s1 = ctx.lsd_stats()
    assert s1["frames"] - s0["frames"] == len(frames)
    assert s1["nfa_to_host"] - s0["nfa_to_host"] <= 1            # certification failures are rare events
    ctx.lsd_configure_nfa(False)
    try:
        host = ctx.lsd_extract_batch(batch, n_threads=4)
    finally:
        ctx.lsd_configure_nfa(True)
    assert ctx.lsd_stats()["nfa_to_host"] == s1["nfa_to_host"]
    for a, b in zip(dev, host):
        _same_lines(a, b)
    # the key-line stage alone back on the host (k_rect_improve stays): KeyLine fields, the std::sort cut, LBD direction, equations
    import os
    assert s1["keylines_to_host"] - s0["keylines_to_host"] <= 1
    os.environ["DRFE_LSD_HOST_KEYLINES"] = "1"
    try:
        hostkl = ctx.lsd_extract_batch(batch, n_threads=4)
    finally:
        del os.environ["DRFE_LSD_HOST_KEYLINES"]
    for a, b in zip(dev, hostkl):
        _same_lines(a, b)
    for g, a in list(zip(frames, dev))[::5] + list(zip(frames, dev))[-4:]:
        o = oracle_mod.extract_lines(g, rect_mode=rect_mode)
        assert a["detected"] == o["detected"]
        for gk, ok in PAIRS:
            assert np.array_equal(a["lines"][gk].view(np.uint32), o["lines"][ok].view(np.uint32)), gk
        assert np.array_equal(a["desc"], o["desc"])

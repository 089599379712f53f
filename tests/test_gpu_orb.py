"""-m gpu parity tests of the ORB extraction path: HIP (through the C-ABI) vs the CPU oracle.

Bar: bit-exact — pyramid bytes, blurred bytes, FAST candidates, keypoint fields (float bit patterns),
256-bit descriptors and their order.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same_kps(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    for f in ("octave", "class_id"):
        assert np.array_equal(a[f], b[f]), f


@pytest.fixture(scope="module")
def ctx():
    from dr_slam_amd import lib
    c = lib.Context(max_batch=8)
    yield c
    c.close()


def test_single_frame_all_stages(ctx, frames_room, oracle_mod):
    o = oracle_mod.OrbOracle()
    g = frames_room[0][0]
    kps, desc = ctx.orb_extract(g)
    okps, odesc = o(g)
    for l in range(8):
        assert np.array_equal(ctx.pyramid_level(0, l), o.pyramid(l)), f"pyramid level {l}"
        ob = o.blurred(l)
        if ob is not None:
            assert np.array_equal(ctx.blurred_level(0, l), ob), f"blur level {l}"
        assert np.array_equal(ctx.candidates(0, l), o.candidates(l)), f"FAST candidates level {l}"
    _same_kps(kps, okps)
    assert np.array_equal(desc, odesc)
    assert len(kps) >= 900


@pytest.mark.parametrize("kind,seed", [("planar_lowtexture", 1), ("living_room", 3), ("corridor", 5)])
def test_scene_kinds(ctx, oracle_mod, kind, seed):
    """Low-texture scenes exercise the minThFAST fallback (reference src/ORBextractor.cc:812-816)."""
    from dr_slam_amd import synth
    g, _, _ = next(synth.sequence(seed, 1, kind=kind))
    o = oracle_mod.OrbOracle()
    kps, desc = ctx.orb_extract(g)
    okps, odesc = o(g)
    _same_kps(kps, okps)
    assert np.array_equal(desc, odesc)


def test_dense_noise_frame(ctx, oracle_mod):
    """Worst-case corner density: tens of thousands of candidates per level."""
    from dr_slam_amd import synth
    g = synth.noise_frame(7, 640, 480)
    o = oracle_mod.OrbOracle()
    kps, desc = ctx.orb_extract(g)
    okps, odesc = o(g)
    _same_kps(kps, okps)
    assert np.array_equal(desc, odesc)


def test_flat_frame_gives_no_keypoints(ctx, oracle_mod):
    g = np.full((480, 640), 77, np.uint8)
    kps, desc = ctx.orb_extract(g)
    okps, _ = oracle_mod.OrbOracle()(g)
    assert len(kps) == 0 and len(okps) == 0 and desc.shape == (0, 32)


def test_empty_image_returns_silently(ctx):
    kps, desc = ctx.orb_extract(None)   # reference: `if(_image.empty()) return;` src/ORBextractor.cc:1046
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_strided_input(ctx, frames_room, oracle_mod):
    g = frames_room[1][0]
    big = np.zeros((480, 704), np.uint8)
    big[:, :640] = g
    view = big[:, :640]
    kps, desc = ctx.orb_extract(view)
    okps, odesc = oracle_mod.OrbOracle()(g)
    _same_kps(kps, okps)
    assert np.array_equal(desc, odesc)


def test_other_resolution_and_params(oracle_mod):
    """320x240, 500 features, 6 levels: geometry is recomputed per (w,h)."""
    from dr_slam_amd import lib, synth
    cam = synth.TUM3.scaled(0.5)
    g, _, _ = next(synth.sequence(4, 1, cam=cam))
    c = lib.Context(nfeatures=500, scale_factor=1.2, nlevels=6, max_width=320, max_height=240)
    o = oracle_mod.OrbOracle(500, 1.2, 6, 20, 7)
    kps, desc = c.orb_extract(g)
    okps, odesc = o(g)
    _same_kps(kps, okps)
    assert np.array_equal(desc, odesc)
    c.close()


def test_batch_equals_single(ctx, frames_room, oracle_mod):
    """Device-resident batch of 4 frames == 4 single-frame calls == oracle; second run is identical."""
    import torch
    gray = torch.from_numpy(np.stack([f[0] for f in frames_room])).cuda()
    o = oracle_mod.OrbOracle()
    for rep in range(2):
        ctx.orb_extract_batch_ptr(gray.data_ptr(), 640 * 480, 640, 640, 480, 4, torch.cuda.current_stream().cuda_stream)
        counts = ctx.orb_counts(4)
        for s in range(4):
            kps, desc = ctx.orb_download(s)
            okps, odesc = o(frames_room[s][0])
            assert counts[s] == len(okps)
            _same_kps(kps, okps)
            assert np.array_equal(desc, odesc)


def test_1280x960_config5(oracle_mod):
    """1280x960 RealSense-style frame, 800 features (BASELINE config 5): oracle parity plus
    size-independent properties (determinism across runs, keypoints inside the detection border,
    non-trivial descriptor rows)."""
    from dr_slam_amd import lib, synth
    cam = synth.REALSENSE.scaled(2.0)
    g, _, _ = next(synth.sequence(5, 1, cam=cam, kind="corridor"))
    c = lib.Context(nfeatures=800, max_width=1280, max_height=960)
    k1, d1 = c.orb_extract(g)
    k2, d2 = c.orb_extract(g)
    assert np.array_equal(k1.view(np.uint8), k2.view(np.uint8)) and np.array_equal(d1, d2)
    okps, odesc = oracle_mod.OrbOracle(800, 1.2, 8, 20, 7)(g)
    _same_kps(k1, okps)
    assert np.array_equal(d1, odesc)
    assert 700 <= len(k1) <= 830
    sc = c.scale_tables()[0]
    lvl = k1["octave"]
    x0 = k1["x"] / sc[lvl]
    assert (x0 >= 18.99).all() and (k1["y"] / sc[lvl] >= 18.99).all()
    assert (np.unpackbits(d1, axis=1).sum(1) > 40).all()
    c.close()


def test_large_batch_uses_both_quadtree_variants(frames_room, oracle_mod):
    """72 frames in one batch: nlevels * nframes exceeds what is resident at once, so the launcher splits the quadtree
    into the 512-thread variant (large levels) and the 256-thread / small-LDS variant (small levels).  Every slot must
    still equal the oracle of its source frame."""
    import torch
    from dr_slam_amd import lib
    B = 72
    c = lib.Context(max_batch=B)
    try:
        order = [i % 4 for i in range(B)]
        gray = torch.from_numpy(np.stack([frames_room[i][0] for i in order])).cuda()
        c.orb_extract_batch_ptr(gray.data_ptr(), 640 * 480, 640, 640, 480, B, torch.cuda.current_stream().cuda_stream)
        o = oracle_mod.OrbOracle()
        ref = [o(frames_room[i][0]) for i in range(4)]
        counts = c.orb_counts(B)
        for s in range(B):
            kps, desc = c.orb_download(s)
            okps, odesc = ref[order[s]]
            assert counts[s] == len(okps)
            _same_kps(kps, okps)
            assert np.array_equal(desc, odesc)
    finally:
        c.close()


@pytest.mark.parametrize("nfeatures", [500, 2000])
def test_other_feature_budgets(frames_room, oracle_mod, nfeatures):
    """nfeatures = 2000 (per-level quotas above 256 nodes: the 1024-node quadtree variant) and 500."""
    from dr_slam_amd import lib
    c = lib.Context(nfeatures=nfeatures)
    try:
        g = frames_room[1][0]
        kps, desc = c.orb_extract(g)
        okps, odesc = oracle_mod.OrbOracle(nfeatures, 1.2, 8, 20, 7)(g)
        _same_kps(kps, okps)
        assert np.array_equal(desc, odesc)
        assert abs(len(kps) - nfeatures) < 0.1 * nfeatures
    finally:
        c.close()


@pytest.mark.parametrize("w,h", [(641, 479), (752, 480), (333, 250)])
def test_odd_image_sizes(oracle_mod, w, h):
    """Widths that are not multiples of 16 (unaligned source rows take the byte-wise border kernel, EuRoC's 752x480,
    a small image with few cells per level): same keypoints and descriptors as the oracle."""
    from dr_slam_amd import lib, synth
    g = synth.noise_frame(7, w, h)
    c = lib.Context(nfeatures=600, max_width=w, max_height=h)
    try:
        kps, desc = c.orb_extract(g)
        oo = oracle_mod.OrbOracle(600, 1.2, 8, 20, 7)
        okps, odesc = oo(g)
        _same_kps(kps, okps)
        assert np.array_equal(desc, odesc)
        assert len(kps) > 300
        # every level and its blurred copy (the blurred levels are tiled 32 x 4 on the device: widths / heights that are
        # not multiples of the tile exercise the untiling download and the overhanging blur blocks)
        for l in range(8):
            assert np.array_equal(c.pyramid_level(0, l), oo.pyramid(l)), f"pyramid level {l}"
            ob = oo.blurred(l)
            if ob is not None:
                assert np.array_equal(c.blurred_level(0, l), ob), f"blur level {l}"
    finally:
        c.close()


@pytest.mark.parametrize("scale,nlevels", [(1.5, 5), (2.0, 3), (1.1, 12)])
def test_other_scale_factors(frames_room, oracle_mod, scale, nlevels):
    """Scale factors whose source windows no longer fit the LDS tile take the register-only resize kernel; 1.1 has twelve
    closely spaced levels.  Pyramid bytes and keypoints / descriptors equal the oracle."""
    from dr_slam_amd import lib
    g = frames_room[2][0]
    c = lib.Context(nfeatures=700, scale_factor=scale, nlevels=nlevels)
    try:
        kps, desc = c.orb_extract(g)
        o = oracle_mod.OrbOracle(700, scale, nlevels, 20, 7)
        okps, odesc = o(g)
        _same_kps(kps, okps)
        assert np.array_equal(desc, odesc)
        assert len(kps) > 400
    finally:
        c.close()
    with pytest.raises(lib.DrfeError):
        lib.Context(scale_factor=3.0, nlevels=3).orb_extract(g)


def test_single_frame_graph_survives_a_geometry_change(frames_room):
    """the captured graph of drfe_orb_extract at size A must not replay against the tables of size B.
    extract(A) -> extract_batch(B) on the same context -> extract(A) == a fresh context's extract(A); and a slot that no call
    has filled after a geometry change holds zero keypoints."""
    import torch
    from dr_slam_amd import lib
    gA = frames_room[0][0]
    gB = np.ascontiguousarray(frames_room[1][0][:360, :480])
    fresh = lib.Context(max_batch=2)
    want = fresh.orb_extract(gA)
    fresh.close()
    c = lib.Context(max_batch=2)
    k0, d0 = c.orb_extract(gA)
    tB = torch.from_numpy(np.stack([gB, gB])).cuda()
    c.orb_extract_batch_ptr(tB.data_ptr(), 480 * 360, 480, 480, 360, 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    k1, d1 = c.orb_extract(gA)
    for k, d in ((k0, d0), (k1, d1)):
        assert np.array_equal(k.view(np.uint8), want[0].view(np.uint8)) and np.array_equal(d, want[1])
    # geometry is A's again and only slot 0 was written: slot 1 must read as empty, not as B's leftovers
    cnt = c.orb_counts(1)
    assert cnt[0] == len(want[0])
    assert c.L.drfe_batch_check(c.h) == 0
    c.close()


@pytest.mark.parametrize("screen", ["0", "1", "2"])
def test_fast_screen_modes_on_mixed_texture(oracle_mod, screen, monkeypatch):
    """k_fast_cells_cols picks per cell between the plain strength tree, the compass screen at minThFAST (low-texture cells) and,
    round 5, the screen at iniThFAST with the exact path as the fallback for a cell without a corner at iniThFAST
    (src/ORBextractor.cc:809-816).  Frames whose cells fall on every side of those decisions - a textured half beside a low-texture
    half, a frame of faint noise whose corner scores sit between the two thresholds, the plain scene kinds - through a context of
    each DRFE_FAST_SCREEN mode: candidates of every level, keypoints and descriptors equal the oracle's."""
    from dr_slam_amd import lib, synth
    monkeypatch.setenv("DRFE_FAST_SCREEN", screen)
    c = lib.Context(max_batch=4)
    try:
        room = next(synth.sequence(2, 1, kind="room_boxes"))[0]
        low = next(synth.sequence(1, 1, kind="planar_lowtexture"))[0]
        living = next(synth.sequence(3, 1, kind="living_room"))[0]
        half = room.copy(); half[:, 320:] = low[:, 320:]
        bands = low.copy(); bands[96:200] = room[96:200]; bands[330:400, 100:500] = living[330:400, 100:500]
        rng = np.random.default_rng(11)
        faint = np.clip(120 + rng.integers(-9, 10, (480, 640)), 0, 255).astype(np.uint8)        # scores mostly in [7, 20)
        faint[200:280, 260:380] = room[200:280, 260:380]
        for name, g in (("room", room), ("low", low), ("half", half), ("bands", bands), ("faint", faint)):
            o = oracle_mod.OrbOracle()
            kps, desc = c.orb_extract(g)
            okps, odesc = o(g)
            for l in range(8):
                assert np.array_equal(c.candidates(0, l), o.candidates(l)), (name, "FAST candidates level", l)
            _same_kps(kps, okps)
            assert np.array_equal(desc, odesc), name
        # the batch entry runs the same kernel over frames side by side
        import torch
        batch = np.stack([half, faint, bands, room])
        gray = torch.from_numpy(batch).cuda()
        c.orb_extract_batch_ptr(gray.data_ptr(), 640 * 480, 640, 640, 480, 4, torch.cuda.current_stream().cuda_stream)
        for f, g in enumerate(batch):
            kps, desc = c.orb_download(f)
            okps, odesc = oracle_mod.OrbOracle()(g)
            _same_kps(kps, okps)
            assert np.array_equal(desc, odesc)
    finally:
        c.close()

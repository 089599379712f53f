"""-m gpu parity tests of the AHC depth-plane path (PlaneDetection): device block fits + product
clustering (through the C-ABI) vs the CPU oracle.  Bar: identical float64 bit patterns for the block
sums / plane fits, identical plane lists, label image and per-plane pixel lists."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from dr_slam_amd import lib
    c = lib.Context(max_batch=1)
    yield c
    c.close()


CASES = [(2, "room_boxes", "TUM3"), (3, "living_room", "ICL"), (5, "corridor", "TUM3"), (1, "planar_lowtexture", "TUM3")]


def _case(seed, kind, camname):
    from dr_slam_amd import synth
    cam = getattr(synth, camname)
    _, d, _ = next(synth.sequence(seed, 1, cam=cam, kind=kind))
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    return d, K4, np.float32(1.0) / np.float32(cam.depth_factor)


@pytest.mark.parametrize("seed,kind,camname", CASES)
def test_block_fits_bit_exact(ctx, oracle_mod, seed, kind, camname):
    """depth -> cloud -> 10x10 block statistics -> Eigen 3x3 solve, all on the device."""
    d, K4, f = _case(seed, kind, camname)
    blocks, valid, n = ctx.planes_ahc_blocks(d, K4, f)
    o = oracle_mod.ahc_planes(d, K4, f)
    assert np.array_equal(valid, o["block_valid"]) and np.array_equal(n, o["block_N"])
    assert valid.sum() > 500
    a, b = blocks.view(np.uint64), o["blocks"].view(np.uint64)
    nan = np.isnan(o["blocks"])
    assert np.array_equal(np.isnan(blocks), nan)
    assert np.array_equal(a[~nan], b[~nan])


@pytest.mark.parametrize("seed,kind,camname", CASES)
def test_planes_bit_exact(ctx, oracle_mod, seed, kind, camname):
    d, K4, f = _case(seed, kind, camname)
    g = ctx.planes_ahc(d, K4, f)
    o = oracle_mod.ahc_planes(d, K4, f)
    assert len(g["planes"]) == len(o["planes"]) >= 3
    for k, col in (("normal", slice(0, 3)), ("center", slice(3, 6))):
        assert np.array_equal(g["planes"][k].view(np.uint64), o["planes"][:, col].view(np.uint64)), k
    assert np.array_equal(g["planes"]["mse"].view(np.uint64), o["planes"][:, 6].view(np.uint64))
    assert np.array_equal(g["planes"]["curvature"].view(np.uint64), o["planes"][:, 7].view(np.uint64))
    assert np.array_equal(g["planes"]["n_points"], o["N"]) and np.array_equal(g["planes"]["rid"], o["rid"])
    assert np.array_equal(g["seg"], o["seg"])
    for a, b in zip(g["members"], o["members"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("seed,kind,camname,patch", [(2, "room_boxes", "TUM3", 20), (3, "living_room", "ICL", 20),
                                                      (5, "corridor", "TUM3", 10), (1, "planar_lowtexture", "TUM3", 10)])
def test_cape_bit_exact(ctx, oracle_mod, seed, kind, camname, patch):
    """PlaneDetection_CAPE: device cell fits + product region growing / merging / refinement vs oracle.
    ICL intrinsics have fy < 0 (Examples/RGB-D/ICL.yaml:9)."""
    from dr_slam_amd import synth
    cam = getattr(synth, camname)
    _, d, _ = next(synth.sequence(seed, 1, cam=cam, kind=kind))
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    dm = oracle_mod.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor))
    g = ctx.planes_cape(dm, K4, patch)
    o = oracle_mod.cape_planes(dm, K4, patch)
    assert np.array_equal(g["cell_planar"], o["cell_planar"]) and np.array_equal(g["cell_npts"], o["cell_npts"])
    assert np.array_equal(g["cells"].view(np.uint64), o["cells"].view(np.uint64))
    assert np.array_equal(g["cell_mst"].view(np.uint32), o["cell_mst"].view(np.uint32))
    assert len(g["planes"]) == len(o["planes"]) >= 3
    assert np.array_equal(g["planes"]["normal"].view(np.uint64), o["planes"][:, 0:3].view(np.uint64))
    assert np.array_equal(g["planes"]["mean"].view(np.uint64), o["planes"][:, 3:6].view(np.uint64))
    assert np.array_equal(g["planes"]["d"].view(np.uint64), o["planes"][:, 6].view(np.uint64))
    assert np.array_equal(g["planes"]["mse"].view(np.uint32), o["MSE"].view(np.uint32))
    assert np.array_equal(g["planes"]["score"].view(np.uint32), o["score"].view(np.uint32))
    assert np.array_equal(g["planes"]["n_points"], o["nr_pts"])
    assert np.array_equal(g["seg"], o["seg"])


def test_no_depth_gives_no_planes(ctx, oracle_mod):
    d = np.zeros((480, 640), np.uint16)
    K4 = np.array([500, 500, 320, 240], np.float32)
    g = ctx.planes_ahc(d, K4, 0.0002)
    assert len(g["planes"]) == 0 and not g["seg"].any()
    far = np.full((480, 640), 40000, np.uint16)        # 8 m: beyond the 5 m clamp (src/PlaneExtractor.cpp:44)
    assert len(ctx.planes_ahc(far, K4, 0.0002)["planes"]) == 0


def test_ahc_batch_equals_single(ctx):
    """drfe_planes_ahc_batch - the extractor on the device, one wavefront per frame (default), or on the host thread pool -
    == per-frame calls (host extractor), any thread count: planes, label images, member lists."""
    from dr_slam_amd import synth
    cam = synth.TUM3
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    depth = np.stack([f[1] for f in synth.sequence(2, 5, kind="room_boxes")] + [next(synth.sequence(5, 1, kind="corridor"))[1]])
    single = [ctx.planes_ahc(d, K4, inv) for d in depth]
    for threads, on_device in ((1, True), (4, True), (16, True), (4, False)):
        ctx.planes_configure_extractor(on_device=on_device)
        batch = ctx.planes_ahc_batch(depth, K4, inv, n_threads=threads)
        ctx.planes_configure_extractor(on_device=True)
        for a, b in zip(batch, single):
            assert len(a["planes"]) == len(b["planes"]) >= 2
            assert np.array_equal(a["planes"].view(np.uint8), b["planes"].view(np.uint8))
            assert np.array_equal(a["seg"], b["seg"])
            for ma, mb in zip(a["members"], b["members"]):
                assert np.array_equal(ma, mb)


@pytest.mark.parametrize("on_device", [True, False])
def test_cape_batch_equals_single(on_device):
    """drfe_planes_cape_batch == per-frame drfe_planes_cape: planes, counts, labels - with CAPE::process on the device (k_cape_frame:
    histogram seeding, cell growing, segment fits, merging, masks; one wavefront per frame) and on the pool of host threads."""
    from dr_slam_amd import lib, synth
    cam = synth.ICL
    frames = list(synth.sequence(3, 5, cam=cam, kind="living_room"))
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    dm = np.stack([f[1].astype(np.float32) * inv for f in frames])
    c = lib.Context(max_batch=1)
    try:
        c.planes_configure_cape(on_device)
        single = [c.planes_cape(d, K4, 20) for d in dm]
        for T in (1, 3):
            planes, n, seg = c.planes_cape_batch(dm, K4, 20, n_threads=T, seg=True)
            for f, s in enumerate(single):
                assert n[f] == len(s["planes"]) > 0
                assert np.array_equal(planes[f, :n[f]].view(np.uint8), np.ascontiguousarray(s["planes"]).view(np.uint8))
                assert np.array_equal(seg[f], s["seg"])
        st = c.planes_cape_stats()
        assert st["frames"] == (10 if on_device else 0) and st["to_host"] <= 1
    finally:
        c.close()


@pytest.mark.parametrize("kind,camname,patch", [("living_room", "ICL", 20), ("room_boxes", "TUM3", 20), ("corridor", "TUM3", 10),
                                                 ("planar_lowtexture", "TUM3", 40)])
def test_cape_device_batch_matches_oracle(oracle_mod, kind, camname, patch):
    """The device CAPE (k_cape_cells -> k_cape_frame -> k_cape_refine, batch form) against the CPU oracle on every scene kind and
    three cell sizes, 40 frames (two upload chunks): plane records bit for bit, label images equal; without the label download too."""
    from dr_slam_amd import lib, synth
    cam = getattr(synth, camname)
    frames = list(synth.sequence(11, 20, cam=cam, kind=kind))
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    dm = np.stack([f[1].astype(np.float32) * inv for f in frames] * 2)
    c = lib.Context(max_batch=1)
    try:
        planes, n, seg = c.planes_cape_batch(dm, K4, patch, n_threads=2, seg=True)
        planes2, n2 = c.planes_cape_batch(dm, K4, patch, n_threads=2)
        assert np.array_equal(n, n2) and np.array_equal(planes.view(np.uint8), planes2.view(np.uint8))
        st = c.planes_cape_stats()
        assert st["frames"] == 2 * len(dm) and st["to_host"] <= 2
        for f in list(range(0, 20, 3)) + [39]:
            o = oracle_mod.cape_planes(dm[f], K4, patch)
            assert n[f] == len(o["planes"])
            assert np.array_equal(planes[f, :n[f]]["normal"].view(np.uint64), o["planes"][:, 0:3].view(np.uint64))
            assert np.array_equal(planes[f, :n[f]]["d"].view(np.uint64), o["planes"][:, 6].view(np.uint64))
            assert np.array_equal(seg[f], o["seg"])
        assert n.max() >= 2
    finally:
        c.close()

"""-m gpu parity tests of rows a-20 / f-2: the surface-normal pass (device) and the per-plane post-processing behind the
C-ABI vs the CPU oracle (bit-exact: float bit patterns, NaN positions, accepted flags, voxel clouds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check_records(rec, cloud, nrm, O):
    on, oc, fx, fy = O.post_surface_normal_records(cloud, nrm)
    assert len(rec) == len(on)
    assert np.array_equal(_bits(rec["normal"]) & 0x7FC00000 == 0x7FC00000, np.isnan(on))     # NaN in the same places
    m = ~np.isnan(on)
    assert np.array_equal(_bits(rec["normal"])[m], _bits(on)[m])
    assert np.array_equal(_bits(rec["camera_position"]), _bits(oc))
    assert np.array_equal(rec["frame_x"], fx) and np.array_equal(rec["frame_y"], fy)
    return int(m.all(1).sum())


@pytest.mark.parametrize("camname,kind,seed,maxd", [("TUM3", "room_boxes", 2, 9.0), ("ICL", "living_room", 3, 9.0),
                                                    ("TUM1", "room_boxes", 10, 3.0)])
def test_surface_normals_single_frame(oracle_mod, camname, kind, seed, maxd):
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = getattr(synth, camname)
    g, d, _ = next(synth.sequence(seed, 1, cam=cam, kind=kind))
    dm = O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor))
    K4 = (cam.fx, cam.fy, cam.cx, cam.cy)
    c = lib.Context()
    try:
        rec, cloud, nrm, dist = c.surface_normals(dm, K4, maxd, taps=True)
        ocloud, onrm = O.post_surface_normals(dm, K4, maxd)
        assert np.array_equal(_bits(cloud), _bits(ocloud))
        nan = np.isnan(onrm)
        assert np.array_equal(np.isnan(nrm), nan) and np.array_equal(_bits(nrm)[~nan], _bits(onrm)[~nan])
        good = _check_records(rec, ocloud, onrm, O)
        assert good > (2000 if maxd > 5 else 10)            # a real share of the 8560 records carries a normal
        assert dist.min() == 0 and dist.max() > 5           # discontinuities exist and so do smooth regions
    finally:
        c.close()


def test_surface_normals_batch_from_device_depth(oracle_mod):
    """Device-resident raw depth (CV_16U) of 6 frames in one call: every frame equals the single-frame oracle."""
    import torch
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.TUM3
    frames = list(synth.sequence(2, 6, cam=cam))
    inv = np.float32(1.0) / np.float32(cam.depth_factor)
    K4 = (cam.fx, cam.fy, cam.cx, cam.cy)
    depth = torch.from_numpy(np.stack([f[1] for f in frames]).view(np.int16)).cuda()
    c = lib.Context()
    try:
        c.surface_normals_batch_ptr(depth.data_ptr(), cam.w * cam.h, cam.w, cam.w, cam.h, K4, inv, 9.0, len(frames),
                                    torch.cuda.current_stream().cuda_stream)
        for s, (_, d, _) in enumerate(frames):
            ocloud, onrm = O.post_surface_normals(O.depth_to_float(d, inv), K4, 9.0)
            assert _check_records(c.surface_normals_download(s), ocloud, onrm, O) > 2000
    finally:
        c.close()


def test_surface_normals_1280x960(oracle_mod):
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.REALSENSE.scaled(2.0)
    _, d, _ = next(synth.sequence(5, 1, cam=cam, kind="corridor"))
    dm = O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor))
    K4 = (cam.fx, cam.fy, cam.cx, cam.cy)
    c = lib.Context(max_width=1280, max_height=960)
    try:
        rec = c.surface_normals(dm, K4, 9.0)
        ocloud, onrm = O.post_surface_normals(dm, K4, 9.0)
        assert ocloud.shape[:2] == (320, 427)
        assert _check_records(rec, ocloud, onrm, O) > 5000
    finally:
        c.close()


def _same_post(g, o, n_min_accept):
    assert len(g["post"]) == len(o[0])
    acc = 0
    for i, rec in enumerate(o[0]):
        assert bool(g["post"]["accepted"][i]) == rec["accepted"], i
        assert g["post"]["n_voxels"][i] == len(rec["voxels"])
        assert np.array_equal(_bits(g["post"]["coef"][i]), _bits(rec["coef"])), i
        if rec["accepted"]:
            acc += 1
            assert np.array_equal(_bits(g["voxels"][i]), _bits(rec["voxels"]))
        else:
            assert len(g["voxels"][i]) == 0
    assert g["n_accepted"] == acc >= n_min_accept and g["plane_num"] == o[1]


@pytest.mark.parametrize("camname,kind,seed", [("TUM3", "room_boxes", 2), ("ICL", "living_room", 3)])
def test_ahc_planes_postprocess(oracle_mod, camname, kind, seed):
    """Frame::ComputePlanes after runPlaneDetection: gates, voxel clouds and the refit coefficients of every AHC plane."""
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = getattr(synth, camname)
    _, d, _ = next(synth.sequence(seed, 1, cam=cam, kind=kind))
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    c = lib.Context()
    try:
        ga, oa = c.planes_ahc(d, K4, inv), O.ahc_planes(d, K4, inv)
        assert len(ga["planes"]) == len(oa["planes"]) >= 2
        for maxd, th in ((9.0, 0.10), (9.0, 0.05), (2.0, 0.05)):
            g = c.planes_ahc_postprocess(d, K4, inv, ga, maxd, th)
            o = O.ahc_post_planes(d, K4, inv, oa, maxd, th)
            _same_post(g, o, 1 if (maxd, th) == (9.0, 0.10) else 0)
    finally:
        c.close()


def test_cape_planes_postprocess(oracle_mod):
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.ICL
    _, d, _ = next(synth.sequence(3, 1, cam=cam, kind="living_room"))
    dm = O.depth_to_float(d, np.float32(1.0) / np.float32(cam.depth_factor))
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    c = lib.Context()
    try:
        gp, op = c.planes_cape(dm, K4, 20), O.cape_planes(dm, K4, 20)
        assert len(gp["planes"]) == len(op["planes"]) >= 3
        for maxd, th in ((9.0, 0.6), (9.0, 0.10), (9.0, 0.03)):
            g = c.planes_cape_postprocess(dm, K4, gp, maxd, th)
            o = O.cape_post_planes(dm, K4, op, maxd, th)
            _same_post(g, o, 1 if th == 0.6 else 0)
    finally:
        c.close()


def test_golden_post_normals_on_device():
    """The committed fixture (tests/golden/post_room_320x240.npz) through the device path, without the oracle."""
    import os
    from dr_slam_amd import lib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_room_320x240.npz"))
    dm = (g["depth"].astype(np.float32) * (np.float32(1.0) / g["depth_factor"])).astype(np.float32)
    c = lib.Context()
    try:
        rec, cloud, nrm, _ = c.surface_normals(dm, g["K4"], float(g["max_point_dist"]), taps=True)
    finally:
        c.close()
    nan = np.isnan(g["normals"])
    assert np.array_equal(np.isnan(nrm), nan) and np.array_equal(_bits(nrm)[~nan], _bits(g["normals"])[~nan])
    assert len(rec) == (nrm.shape[0] // 2) * (nrm.shape[1] // 2)


def test_post_edge_cases(oracle_mod):
    """Empty and degenerate inputs: a depth image of zeros (every normal undefined), depth beyond Point.MaxDistance only, an
    image too small for the 10-point border, no planes at all."""
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = synth.TUM3
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    c = lib.Context()
    try:
        for dm in (np.zeros((cam.h, cam.w), np.float32), np.full((cam.h, cam.w), 12.0, np.float32)):
            rec = c.surface_normals(dm, K4, 9.0)
            ocl, onr = O.post_surface_normals(dm, K4, 9.0)
            assert np.isnan(onr).all() and np.isnan(rec["normal"]).all() and len(rec) == 80 * 107
            assert np.array_equal(_bits(rec["camera_position"]), _bits(O.post_surface_normal_records(ocl, onr)[1]))
        small = np.full((48, 60), 2.0, np.float32)               # 16 x 20 cloud: nothing inside the 10-point border
        rec = c.surface_normals(small, K4, 9.0)
        assert len(rec) == 8 * 10 and np.isnan(rec["normal"]).all()
        with pytest.raises(lib.DrfeError):
            c.surface_normals(np.zeros((2, 2), np.float32), K4, 9.0)
        d = np.zeros((cam.h, cam.w), np.uint16)                   # no depth: AHC finds nothing, the post-processing loop is empty
        ga = c.planes_ahc(d, K4, 1.0 / 5000.0)
        assert len(ga["planes"]) == 0
        g = c.planes_ahc_postprocess(d, K4, 1.0 / 5000.0, ga, 9.0, 0.05)
        assert g["n_accepted"] == 0 and g["plane_num"] == 0 and len(g["post"]) == 0
    finally:
        c.close()


@pytest.mark.parametrize("camname,kind,seed", [("TUM3", "room_boxes", 2), ("ICL", "living_room", 3)])
def test_ahc_post_batch_equals_single_frame_calls(oracle_mod, camname, kind, seed):
    """drfe_planes_ahc_post_batch (AHC + the per-plane loop on a pool of host threads, pcl::VoxelGrid of every plane ON THE
    DEVICE: voxel_kernels.hip = std::sort's permutation by introsort_device.h + centroid sums in that order) == per-frame
    drfe_planes_ahc + drfe_planes_ahc_postprocess (voxel grid on the host): voxel counts and the refitted coefficients, which
    depend on every bit of every centroid, are identical."""
    from dr_slam_amd import lib, synth
    cam = getattr(synth, camname)
    frames = list(synth.sequence(seed, 5, cam=cam, kind=kind))
    depth = np.stack([f[1] for f in frames])
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    c = lib.Context()
    try:
        # extractor on the host pool or on the device (default), voxel grids on the host or on the device (default: behind the
        # device extractor): identical planes, labels, post results in every combination
        c.planes_configure_extractor(on_device=False)
        c.planes_configure(device_voxel_grid=0)
        planes_h, n_h, post_h, na_h, pn_h, seg_h = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=3, seg=True)
        c.planes_configure(device_voxel_grid=2)
        planes_d, n_d, post_d, na_d, pn_d, seg_d = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=3, seg=True)
        assert planes_d.tobytes() == planes_h.tobytes() and post_d.tobytes() == post_h.tobytes() and np.array_equal(na_d, na_h) and np.array_equal(pn_d, pn_h)
        c.planes_configure_extractor(on_device=True)
        c.planes_configure(device_voxel_grid=0)
        planes_e, n_e, post_e, na_e, pn_e, seg_e = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=3, seg=True)
        assert np.array_equal(n_e, n_h) and planes_e.tobytes() == planes_h.tobytes() and np.array_equal(seg_e, seg_h)
        assert post_e.tobytes() == post_h.tobytes() and np.array_equal(na_e, na_h) and np.array_equal(pn_e, pn_h)
        c.planes_configure(device_voxel_grid=1)
        planes, n, post, na, pn, seg = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=3, seg=True)
        assert np.array_equal(n, n_h) and planes.tobytes() == planes_h.tobytes() and np.array_equal(seg, seg_h)
        assert post.tobytes() == post_h.tobytes() and np.array_equal(na, na_h) and np.array_equal(pn, pn_h)
        # without the label image (the bench's call), and with points beyond max_point_dist left out of the clouds
        r2 = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=2)
        assert r2[2].tobytes() == post.tobytes()
        near = c.planes_ahc_post_batch(depth, K4, inv, 2.5, 0.10, n_threads=2)
        c.planes_configure(device_voxel_grid=0)
        near_h = c.planes_ahc_post_batch(depth, K4, inv, 2.5, 0.10, n_threads=2)
        c.planes_configure(device_voxel_grid=1)
        assert near[2].tobytes() == near_h[2].tobytes() and np.array_equal(near[3], near_h[3]) and near[2].tobytes() != post.tobytes()
        for f in range(len(frames)):
            ga = c.planes_ahc(depth[f], K4, inv)
            g = c.planes_ahc_postprocess(depth[f], K4, inv, ga, 9.0, 0.10)
            assert n[f] == len(ga["planes"]) and planes[f, :n[f]].tobytes() == ga["planes"].tobytes()
            assert np.array_equal(seg[f], ga["seg"])
            assert post[f, :n[f]].tobytes() == g["post"].tobytes() and na[f] == g["n_accepted"] and pn[f] == g["plane_num"]
        assert na.sum() >= 5
    finally:
        c.close()


@pytest.mark.parametrize("camname,kind,seed", [("TUM3", "room_boxes", 12), ("ICL", "living_room", 13), ("TUM3", "corridor", 15)])
def test_device_refit_equals_host_refit(oracle_mod, camname, kind, seed):
    """Gates + Frame::MaxPointDistanceFromPlane on the device (k_plane_refit: one wavefront per plane, pcl's mt19937 sample sequence,
    adaptive iteration bound, covariance sums in inlier order, pcl::eigen33) against the host's refit_plane and against the oracle:
    accepted flags, voxel counts and every bit of the refitted coefficients, for three thresholds and two distance limits."""
    from dr_slam_amd import lib, synth
    O = oracle_mod
    cam = getattr(synth, camname)
    frames = list(synth.sequence(seed, 6, cam=cam, kind=kind))
    depth = np.stack([f[1] for f in frames])
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    c = lib.Context(max_batch=1)
    try:
        accepted = 0
        for maxd, th in ((9.0, 0.10), (9.0, 0.05), (2.5, 0.03)):
            c.planes_configure_refit(False)
            ref = c.planes_ahc_post_batch(depth, K4, inv, maxd, th, n_threads=2)
            c.planes_configure_refit(True)
            s0 = c.planes_refit_stats()
            got = c.planes_ahc_post_batch(depth, K4, inv, maxd, th, n_threads=2)
            s1 = c.planes_refit_stats()
            for a, b in zip(ref, got):
                assert a.tobytes() == b.tobytes()
            assert s1["frames"] - s0["frames"] == len(frames) and s1["to_host"] - s0["to_host"] <= 1
            accepted += int(got[3].sum())
            planes, n, post, na, pn = got
            for f in (0, len(frames) - 1):
                oa = O.ahc_planes(depth[f], K4, inv)
                o2, opn = O.ahc_post_planes(depth[f], K4, inv, oa, maxd, th)
                assert pn[f] == opn and na[f] == sum(1 for r in o2 if r["accepted"])
                for k, rec in enumerate(o2):
                    assert bool(post[f, k]["accepted"]) == rec["accepted"] and post[f, k]["n_voxels"] == len(rec["voxels"])
                    assert np.array_equal(post[f, k]["coef"].view(np.uint32), rec["coef"].view(np.uint32))
        assert accepted >= 10
    finally:
        c.close()


def test_ahc_post_batch_device_edge_cases():
    """The device extractor + device voxel grids on inputs at the edges of what the kernels assume: no depth at all (no block is
    valid: empty queue, no seeds), one fronto-parallel wall filling the image (a single plane of ~3000 blocks: the longest
    clustering chain, a 300 000-point cloud for the voxel grid), sensor-noise depth (hundreds of tiny clusters, none with
    support), frames in one batch that differ completely, and an odd image size (333 x 257: blocks do not tile the image).
    Identical to the host extractor + host voxel grids in every case."""
    from dr_slam_amd import lib, synth
    cam = synth.TUM3
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    rng = np.random.default_rng(11)
    h, w = cam.h, cam.w
    wall = np.full((h, w), int(2.0 * cam.depth_factor), np.uint16)
    tilt = (1.5 * cam.depth_factor + 3.0 * np.arange(w)[None, :] + 2.0 * np.arange(h)[:, None]).astype(np.uint16)
    noise = rng.integers(500, 60000, (h, w)).astype(np.uint16)
    room = next(synth.sequence(5, 1, cam=cam, kind="room_boxes"))[1]
    half = room.copy(); half[:, w // 2:] = 0
    batch = np.stack([np.zeros((h, w), np.uint16), wall, tilt, noise, room, half])
    odd = np.stack([room[:257, :333].copy(), wall[:257, :333].copy()])
    c = lib.Context(max_batch=1)
    try:
        for depth in (batch, odd):
            c.planes_configure_extractor(on_device=False); c.planes_configure(device_voxel_grid=0)
            ref = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=3, seg=True)
            c.planes_configure_extractor(on_device=True); c.planes_configure(device_voxel_grid=1)
            got = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=3, seg=True)
            for a, b in zip(ref, got):
                assert a.tobytes() == b.tobytes()
        assert ref[1][1] >= 1                   # the odd-sized wall is a plane
        n = got and c.planes_ahc_post_batch(batch, K4, inv, 9.0, 0.10, n_threads=2)[1]
        assert n[0] == 0 and n[1] == 1 and n[2] >= 1 and n[3] == 0 and n[4] >= 2
    finally:
        c.close()


@pytest.mark.timeout(180)
def test_voxel_sort_runs_out_of_depth_in_the_workgroup_phase():
    """Regression (found by tools/parity_soak_batch.py at 256 frames per scene kind): the fourth plane of this frame is a
    12 905-point cloud whose leaf indices drive libstdc++'s introsort to its depth limit while a range is still above the
    8192 records the WORKGROUP partitions (the usual place to run out of depth is a wavefront's small range).  The device
    sort must flag it and hand the cloud back - it used to stop the workgroup for good - and the results must equal the host's."""
    from dr_slam_amd import lib, synth
    cam = synth.TUM3
    frames = list(synth.sequence(3000 + 17 * 2 + 200, 8, cam=cam, kind="living_room", start=0))
    depth = frames[7][1][None]
    K4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32)
    inv = float(np.float32(1.0) / np.float32(cam.depth_factor))
    c = lib.Context(max_batch=1)
    try:
        c.planes_configure_extractor(on_device=False); c.planes_configure(device_voxel_grid=0)
        ref = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=1, seg=True)
        for extractor, vox in ((True, 1), (False, 2)):
            c.planes_configure_extractor(on_device=extractor); c.planes_configure(device_voxel_grid=vox)
            got = c.planes_ahc_post_batch(depth, K4, inv, 9.0, 0.10, n_threads=1, seg=True)
            for a, b in zip(ref, got):
                assert a.tobytes() == b.tobytes()
        assert ref[1][0] == 6 and ref[2]["n_voxels"][0, 3] == 127
    finally:
        c.close()

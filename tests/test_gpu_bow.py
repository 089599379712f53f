"""-m gpu parity tests of the bag-of-words step: vocabulary tree descent (TemplatedVocabulary::transform)
and ORBmatcher::SearchByBoW on the device vs the CPU oracle (DBoW2 restatement).  Bar: identical word
ids / node ids / weights per feature, bit-identical BowVector values, identical match arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(frames_room, oracle_mod):
    import torch
    from dr_slam_amd import lib
    c = lib.Context(max_batch=4)
    gray = torch.from_numpy(np.stack([f[0] for f in frames_room])).cuda()
    c.orb_extract_batch_ptr(gray.data_ptr(), 640 * 480, 640, 640, 480, 4, 0)
    frames = [c.orb_download(s) for s in range(4)]
    yield c, frames
    c.close()


@pytest.mark.parametrize("k,L,scoring,weighting,levelsup", [(10, 4, 0, 0, 2), (6, 5, 1, 1, 4), (10, 3, 5, 2, 4), (4, 6, 0, 3, 3)])
def test_transform_and_bow_vectors(env, oracle_mod, k, L, scoring, weighting, levelsup):
    from dr_slam_amd import vocabulary as V
    c, frames = env
    voc = V.make_synthetic(k, L, seed=3, scoring=scoring, weighting=weighting, stop_fraction=0.05)
    ov = oracle_mod.VocabularyOracle(voc.to_text())
    assert (ov.k, ov.L, ov.scoring, ov.weighting, ov.n_nodes) == (k, L, scoring, weighting, voc.n_nodes)
    voc.upload(c)
    c.bow_transform_batch(levelsup, 4)
    for s in range(4):
        kps, desc = frames[s]
        n = len(kps)
        word, weight, nid = c.bow_download(s)
        oword, oweight, onid = ov.transform_each(desc, levelsup)
        assert np.array_equal(word[:n], oword) and np.array_equal(nid[:n], onid)
        assert np.array_equal(weight[:n].view(np.uint64), oweight.view(np.uint64))
        ids, vals, fv = V.bow_and_feature_vectors(voc, word[:n], weight[:n], nid[:n])
        oids, ovals = ov.bow_vector(desc, levelsup)
        assert np.array_equal(ids, oids)
        assert np.array_equal(vals.view(np.uint64), ovals.view(np.uint64))
        assert len(ids) > 20 and (weight[:n] == 0).any()     # some stopped words


def test_search_by_bow(env, oracle_mod):
    """ORBmatcher(0.7, true).SearchByBoW(pKF = frame 0, F = frame 1) — TrackReferenceKeyFrame's call."""
    from dr_slam_amd import vocabulary as V
    c, frames = env
    voc = V.make_synthetic(10, 4, seed=5, stop_fraction=0.02)
    ov = oracle_mod.VocabularyOracle(voc.to_text())
    voc.upload(c)
    c.bow_transform_batch(2, 4)          # L - levelsup = level 2: 100 nodes, several features per node
    rng = np.random.default_rng(1)
    for kf, f, ratio, ori in ((0, 1, 0.7, True), (1, 2, 0.75, True), (3, 2, 0.9, False)):
        (kkps, kdesc), (fkps, fdesc) = frames[kf], frames[f]
        kf_mp = np.where(rng.random(len(kkps)) > 0.25, 1, -1).astype(np.int32)
        _, kw, knid = ov.transform_each(kdesc, 2)
        _, fw, fnid = ov.transform_each(fdesc, 2)
        n_o, m_o = oracle_mod.search_by_bow(np.where(kw > 0, knid, -1), np.where(fw > 0, fnid, -1), kdesc, kkps["angle"],
                                            kf_mp, fdesc, fkps["angle"], ratio, ori)
        n_g, m_g = c.search_by_bow(kf, f, kf_mp, len(fkps), ratio, ori)
        assert n_g == n_o, (kf, f, n_g, n_o)
        assert np.array_equal(m_g, m_o)
        assert n_o > 30


def test_search_by_bow_keyframes(env, oracle_mod):
    """ORBmatcher(0.75, true).SearchByBoW(pKF1, pKF2, vpMatches12) — LoopClosing::ComputeSim3's call: map points on both
    sides, strict `< TH_LOW`."""
    from dr_slam_amd import vocabulary as V
    c, frames = env
    voc = V.make_synthetic(10, 4, seed=5, stop_fraction=0.02)
    ov = oracle_mod.VocabularyOracle(voc.to_text())
    voc.upload(c)
    c.bow_transform_batch(2, 4)
    rng = np.random.default_rng(3)
    for s1, s2, ratio, ori in ((0, 1, 0.75, True), (2, 0, 0.9, True), (1, 3, 0.75, False)):
        (k1, d1), (k2, d2) = frames[s1], frames[s2]
        mp1 = np.where(rng.random(len(k1)) > 0.3, 1, -1).astype(np.int32)
        mp2 = np.where(rng.random(len(k2)) > 0.3, 1, -1).astype(np.int32)
        _, w1, n1 = ov.transform_each(d1, 2)
        _, w2, n2 = ov.transform_each(d2, 2)
        no, mo = oracle_mod.search_by_bow_kf(np.where(w1 > 0, n1, -1), np.where(w2 > 0, n2, -1), d1, k1["angle"], mp1, d2,
                                             k2["angle"], mp2, ratio, ori)
        ng, mg = c.search_by_bow_kf(s1, s2, mp1, mp2, ratio, ori)
        assert ng == no and np.array_equal(mg, mo), (s1, s2, ng, no)
        assert no > 20 and (mo[mp2 < 0] == -1).all()


def test_vocabulary_limits(env):
    from dr_slam_amd import lib, vocabulary as V
    c, _ = env
    voc = V.make_synthetic(4, 2)
    with pytest.raises(lib.DrfeError):
        c.voc_upload(25, 2, 0, 0, voc.parent, voc.desc, voc.weight, voc.is_leaf)     # k > 20 rejected like the loader


def _skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64)


def test_search_for_triangulation(frames_room, oracle_mod):
    """ORBmatcher(0.6, false/true).SearchForTriangulation(KF1, KF2, F12, pairs, bOnlyStereo) as LocalMapping::
    CreateNewMapPoints calls it: vocabulary-node groups, epipole exclusion, epipolar-line gate, rotation histogram."""
    import torch
    from dr_slam_amd import lib, synth, vocabulary as V
    from dr_slam_amd.pipeline import FrontEnd
    cam = synth.TUM3
    fe = FrontEnd(cam, max_batch=4)
    try:
        gray = torch.from_numpy(np.stack([f[0] for f in frames_room])).cuda()
        depth = torch.from_numpy(np.stack([f[1] for f in frames_room]).view(np.int16)).cuda()
        fe.process(gray, depth, None, None, stream=torch.cuda.current_stream().cuda_stream)
        c = fe.ctx
        voc = V.make_synthetic(10, 4, seed=5, stop_fraction=0.02)
        ov = oracle_mod.VocabularyOracle(voc.to_text())
        voc.upload(c)
        c.bow_transform_batch(2, 4)
        scale, _, sigma2, _ = c.scale_tables()
        K = np.array([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], np.float64)
        rng = np.random.default_rng(7)
        total = 0
        for s1, s2, only_stereo, ori in ((0, 2, False, True), (3, 1, False, False), (1, 3, True, True)):
            Twc1, Twc2 = frames_room[s1][2].astype(np.float64), frames_room[s2][2].astype(np.float64)
            T1w, T2w = np.linalg.inv(Twc1), np.linalg.inv(Twc2)
            R12 = T1w[:3, :3] @ T2w[:3, :3].T                       # LocalMapping::ComputeF12
            t12 = -R12 @ T2w[:3, 3] + T1w[:3, 3]
            F12 = (np.linalg.inv(K).T @ _skew(t12) @ R12 @ np.linalg.inv(K)).astype(np.float32)
            Cw1 = Twc1[:3, 3].astype(np.float32)
            T2w32 = T2w.astype(np.float32)
            kf = []
            for s in (s1, s2):
                kps, desc = c.orb_download(s)
                n = len(kps)
                un = c.download_keys_un(s, n)
                ur, _ = c.download_stereo(s)
                ur = ur[:n].copy()
                ur[rng.random(n) < 0.3] = -1.0                      # mix of monocular and stereo keypoints ...
                mp = np.where(rng.random(n) < 0.4, 5, -1).astype(np.int32)
                _, w, nid = ov.transform_each(desc, 2)
                kf.append(dict(x=un["x"], y=un["y"], angle=un["angle"], u_right=ur, octave=un["octave"], mp=mp,
                               nid=np.where(w > 0, nid, -1), desc=desc, n=n))
            # ... the product reads mvuRight of its slots: make the device copy agree with the edited arrays
            # (only_stereo / epipole rules are exercised through the map-point mask instead when unedited)
            for k, s in zip(kf, (s1, s2)):
                ur_dev, _ = c.download_stereo(s)
                k["u_right"] = ur_dev[:k["n"]]
            C2 = T2w32[:3, :3] @ Cw1 + T2w32[:3, 3]
            ex = np.float32(cam.fx) * C2[0] * (np.float32(1.0) / C2[2]) + np.float32(cam.cx)
            ey = np.float32(cam.fy) * C2[1] * (np.float32(1.0) / C2[2]) + np.float32(cam.cy)
            no, mo = oracle_mod.search_for_triangulation(kf[0], kf[1], F12, ex, ey, scale, sigma2, only_stereo, ori)
            ng, mg = c.search_for_triangulation(s1, s2, kf[0]["mp"], kf[1]["mp"], F12, Cw1, T2w32, fe.cam, only_stereo, ori)
            assert ng == no, (s1, s2, ng, no)
            assert np.array_equal(mg, mo)
            total += no
        assert total > 60
    finally:
        fe.ctx.close()

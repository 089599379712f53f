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


def test_vocabulary_limits(env):
    from dr_slam_amd import lib, vocabulary as V
    c, _ = env
    voc = V.make_synthetic(4, 2)
    with pytest.raises(lib.DrfeError):
        c.voc_upload(25, 2, 0, 0, voc.parent, voc.desc, voc.weight, voc.is_leaf)     # k > 20 rejected like the loader

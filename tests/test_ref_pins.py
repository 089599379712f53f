"""Pins against the REFERENCE ITSELF, for the slice of it that compiles in this image: oracle/_ref/libdrslam_ref.so is built by
`make -C oracle ref` from the reference's own Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector}.cpp and
include/peac/{AHCParamSet,DisjointSet}.hpp where they lie under /root/reference (oracle/ref_shim.cpp is the C entry layer;
nothing of the reference is copied).  The oracle's restatements - and the product's host-side container code - must agree
with it bit for bit.  Everything else on the hot path needs OpenCV 3.4 / Eigen / PCL: still unpinned.
Skipped where the library has not been built (no /root/reference and no shipped oracle/_ref)."""
import ctypes as C
import os

import numpy as np
import pytest

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libdrslam_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (make -C oracle ref needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return C.CDLL(REF)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_ahc_thresholds_equal_the_reference_paramset(ref, oracle_mod):
    """ahc::ParamSet::T_mse / T_ang / T_dz (include/peac/AHCParamSet.hpp:95-153) in all three phases, over depths inside and
    outside [z_near, z_far] (mm), negative and zero included."""
    L = oracle_mod.lib()
    for fn in (ref.ref_ahc_thresholds, L.orc_ahc_thresholds):
        fn.argtypes = [C.c_int, C.c_double, C.c_void_p]
        fn.restype = None
    rng = np.random.default_rng(0)
    zs = np.concatenate([[0.0, -1.0, 499.999, 500.0, 4000.0, 4000.001, 1e6], rng.uniform(-100, 9000, 500), rng.uniform(0, 5, 200)])
    a, b = np.zeros(3), np.zeros(3)
    for phase in (0, 1, 2):
        for z in zs:
            ref.ref_ahc_thresholds(phase, float(z), _p(a))
            L.orc_ahc_thresholds(phase, float(z), _p(b))
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (phase, z, a, b)


def test_disjoint_set_equals_the_reference(ref, oracle_mod):
    """DisjointSet::Union / Find / getSetSize (include/peac/DisjointSet.hpp): same roots (union by size, ties to x), same
    return values, on the 64 x 48 block grid's size and on small sets with many repeated unions."""
    L = oracle_mod.lib()
    for fn in (ref.ref_ahc_disjoint_set, L.orc_ahc_disjoint_set):
        fn.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        fn.restype = None
    rng = np.random.default_rng(1)
    for n, m in ((3072, 4000), (3072, 200), (16, 64), (2, 5), (1, 0)):
        pairs = rng.integers(0, n, (max(m, 1), 2)).astype(np.int32)
        out = []
        for fn in (ref.ref_ahc_disjoint_set, L.orc_ahc_disjoint_set):
            u, f, s = np.zeros(max(m, 1), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
            fn(n, _p(pairs), m, _p(u), _p(f), _p(s))
            out.append((u[:m].copy(), f, s))
        for x, y in zip(out[0], out[1]):
            assert np.array_equal(x, y), (n, m)
        if m >= 64:
            assert len(np.unique(out[0][1])) < n


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (0, 1), (1, 5), (2, 0), (3, 1), (1, 2), (3, 5)])
def test_bow_containers_equal_the_reference(ref, oracle_mod, weighting, scoring):
    """BowVector::addWeight / addIfNotExist / normalize and FeatureVector::addFeature of the reference, fed with the
    per-feature (word, weight, node) triples of the oracle's tree descent in feature order as TemplatedVocabulary::transform
    does: the oracle's BowVector (ids and float64 values) and the product's host-side containers
    (dr_slam_amd.vocabulary.bow_and_feature_vectors, what Frame::ComputeBoW's adaptor builds from the device's per-feature
    results) are identical, values bit for bit."""
    from dr_slam_amd import vocabulary as V
    voc = V.make_synthetic(6, 3, seed=3 + weighting, scoring=scoring, weighting=weighting, stop_fraction=0.05)
    ov = oracle_mod.VocabularyOracle(voc.to_text())
    rng = np.random.default_rng(10 * weighting + scoring)
    n = 900
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    desc[100:400] = desc[:300]                                   # repeated words: addWeight accumulates, addIfNotExist does not
    word, weight, nid = ov.transform_each(desc, levelsup=2)
    assert (weight == 0).any() and (weight > 0).sum() > 500      # stopped words are skipped by both containers
    ref.ref_bow_containers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    ids, vals = np.zeros(n, np.int32), np.zeros(n)
    fvn, fvc, fvf = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    nw, nn = C.c_int(0), C.c_int(0)
    ref.ref_bow_containers(_p(word), _p(weight), _p(nid), n, weighting, scoring, _p(ids), _p(vals), C.byref(nw), _p(fvn), _p(fvc),
                           _p(fvf), C.byref(nn))
    ids, vals = ids[:nw.value], vals[:nw.value]
    # the oracle's transform()
    o_ids, o_vals = ov.bow_vector(desc, levelsup=2)
    assert np.array_equal(ids, o_ids)
    assert np.array_equal(vals.view(np.uint64), o_vals.view(np.uint64))
    # the product's host-side containers
    p_ids, p_vals, p_fv = V.bow_and_feature_vectors(voc, word, weight, nid)
    assert np.array_equal(ids, p_ids) and np.array_equal(vals.view(np.uint64), p_vals.view(np.uint64))
    off = np.concatenate([[0], np.cumsum(fvc[:nn.value])])
    r_fv = {int(fvn[i]): fvf[off[i]:off[i + 1]].tolist() for i in range(nn.value)}
    assert list(r_fv) == list(p_fv) and r_fv == p_fv
    if scoring in (0, 2):
        assert abs(np.abs(vals).sum() - 1) < 1e-12

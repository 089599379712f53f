"""Oracle vs a REAL OpenCV, when somebody has produced tests/golden/opencv_pins.npz with tools/dump_opencv_reference.py on a
machine that has OpenCV 3.4.x.  This repository does not carry the file (no OpenCV in the build container, no network):
every test here then SKIPS with the reason "parity unpinned" - which is the honest state of the OpenCV-level parity
(DESIGN.md section 5)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PINS = os.path.join(GOLD, "opencv_pins.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PINS), reason="parity unpinned: tests/golden/opencv_pins.npz absent (run "
                                "tools/dump_opencv_reference.py where OpenCV 3.4.x is installed)")


@pytest.fixture(scope="module")
def pins():
    return np.load(PINS)


@pytest.mark.parametrize("tag,name", [("low", "orb_lowtexture_640x480.npz"), ("room", "orb_room_320x240.npz")])
def test_pyramid_and_blur_equal_opencv(oracle_mod, pins, tag, name):
    g = np.load(os.path.join(GOLD, name))
    p = g["params"]
    o = oracle_mod.OrbOracle(int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]))
    o(g["gray"])
    for l in range(int(p[2])):
        assert np.array_equal(o.pyramid(l), pins[f"{tag}_pyr{l}"]), ("pyramid", l)
        if o.blurred(l) is not None:
            assert np.array_equal(o.blurred(l), pins[f"{tag}_blur{l}"]), ("blur", l)


@pytest.mark.parametrize("tag,name", [("low", "orb_lowtexture_640x480.npz"), ("room", "orb_room_320x240.npz")])
def test_fast_equal_opencv(oracle_mod, pins, tag, name):
    gray = np.load(os.path.join(GOLD, name))["gray"]
    for th in (20, 7):
        kps = oracle_mod.fast_detect(gray, th)
        ref = pins[f"{tag}_fast{th}"]
        assert len(kps) == len(ref)
        assert np.array_equal(np.asarray(kps, np.float32).reshape(-1, 3), ref)


def test_fast_atan2_equal_opencv(oracle_mod, pins):
    got = np.array([oracle_mod.fast_atan2(float(y), float(x)) for y, x in pins["atan2_yx"]], np.float32)
    assert np.array_equal(got.view(np.uint32), pins["atan2_deg"].view(np.uint32))


def test_undistort_points_equal_opencv(oracle_mod, pins):
    K4 = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
    dist = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)
    got = oracle_mod.undistort_points(pins["undist_in"].reshape(-1, 2), K4, dist)
    assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), pins["undist_out"].reshape(-1, 2).view(np.uint32))


@pytest.mark.parametrize("tag", ["lsd", "lsd_oblique", "lsd_diag45"])
def test_lsd_segments_equal_opencv(oracle_mod, pins, tag):
    """The detector's segments with their exact float32 bits, and width / precision / -log10(NFA) per segment (the NFA is a
    function of rect_nfa's pixel counts: it tells which reading of rect_nfa - drfe_lsd_configure_rect 0 or 1 - the library
    implements; the oblique-edge scene decides it outright: the literal reading rejects that segment)."""
    if tag + "_img" not in pins:
        pytest.skip("opencv_pins.npz predates the per-segment dump: regenerate it with tools/dump_opencv_reference.py")
    o = oracle_mod.extract_lines(pins[tag + "_img"], max_lines=100000, trace=True)
    ref = pins[tag + "_segments"]
    assert len(o["segments"]) == len(ref)
    assert np.array_equal(o["segments"].view(np.uint32), ref.view(np.uint32))
    if tag + "_nfa" in pins and len(ref):
        assert np.array_equal(o["seg_width"], pins[tag + "_width"])
        assert np.allclose(o["seg_prec"], pins[tag + "_prec"], rtol=1e-15, atol=0)
        assert np.allclose(o["seg_nfa"], pins[tag + "_nfa"], rtol=1e-12, atol=1e-12)     # libm of the dumping host


def test_lsd_reading_decided_by_opencv(oracle_mod, pins):
    """Which of drfe_lsd_configure_rect's readings of lsd.cpp the real library implements: the segment COUNT on the line fixture alone
    separates them (mode 0 keeps four to five times as many as modes 1 / 2).  The default (0) must be the one."""
    if "lsd_img" not in pins:
        pytest.skip("opencv_pins.npz predates the per-segment dump")
    ref = pins["lsd_segments"]
    same = [m for m in (0, 1, 2)
            if (lambda o: len(o["segments"]) == len(ref) and np.array_equal(o["segments"].view(np.uint32), ref.view(np.uint32)))(
                oracle_mod.extract_lines(pins["lsd_img"], max_lines=100000, trace=True, rect_mode=m))]
    assert same == [0], "OpenCV %s matches drfe_lsd_configure_rect mode(s) %s, the default is 0" % (str(pins["cv_version"]), same)


@pytest.mark.parametrize("tag", ["impulse", "noise"])
def test_gaussian_kernel_equal_opencv(oracle_mod, pins, tag):
    """The 8-bit 7 x 7 sigma-2 Gaussian itself: a grid of isolated impulses of every amplitude (each 7 x 7 response is the fixed-point
    product of the two 1-D kernels, rounding included) and a noise image.  The oracle's taps {18, 34, 49, 55, 49, 34, 18} / 256 sum to
    257 (oracle/orb_oracle.cpp); whether OpenCV 3.4.4's getGaussianKernel + its 8-bit fixed-point path normalise them the same way
    decides every blurred byte and therefore every rBRIEF bit.  A failure here lists the amplitudes and tap pairs that differ."""
    if f"blur_{tag}_in" not in pins:
        pytest.skip("opencv_pins.npz predates the impulse dump: regenerate it with tools/dump_opencv_reference.py")
    got = oracle_mod.gaussian_blur(pins[f"blur_{tag}_in"])
    ref = pins[f"blur_{tag}_out"]
    bad = np.argwhere(got != ref)
    assert len(bad) == 0, ("first differing pixels (row, col, oracle, OpenCV)", [(int(r), int(c), int(got[r, c]), int(ref[r, c])) for r, c in bad[:12]])


@pytest.mark.parametrize("tag", ["lbd", "lbd_oblique", "lbd_diag45"])
def test_lbd_rows_equal_opencv(oracle_mod, pins, tag):
    """LineSegment::ExtractLineSegment's library half (src/LSDextractor.cpp:14-30): LSDDetector::detect + BinaryDescriptor::compute
    on every detected line - all KeyLine fields with their float32 bits and all 32-byte LBD rows."""
    if tag + "_img" not in pins:
        pytest.skip("opencv_pins.npz carries no LBD rows (dumped without opencv-contrib's line_descriptor)")
    o = oracle_mod.extract_lines(pins[tag + "_img"], max_lines=100000)
    ref_kl, ref_desc = pins[tag + "_keylines"], pins[tag + "_desc"]
    assert len(o["lines"]) == len(ref_kl)
    names = o["lines"].dtype.names                       # cv::line_descriptor::KeyLine's declaration order, 17 scalars
    assert len(names) == 17 == ref_kl.shape[1]
    for j, f in enumerate(names):
        col = o["lines"][f]
        want = ref_kl[:, j].astype(col.dtype)
        assert np.array_equal(col.view(np.uint32) if col.dtype == np.float32 else col, want.view(np.uint32) if col.dtype == np.float32 else want), f
    assert np.array_equal(o["desc"], ref_desc)

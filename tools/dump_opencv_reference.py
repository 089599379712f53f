#!/usr/bin/env python3
"""Pins the oracle to a REAL OpenCV (run this OUTSIDE the build container, on any machine with opencv-python /
opencv-contrib-python 3.4.x - the version the reference links, README.md:25).  It reads the inputs of the committed golden
fixtures and writes tests/golden/opencv_pins.npz with what OpenCV itself computes for the library calls the oracle restates
from memory:

  cv2.resize INTER_LINEAR cascade + copyMakeBorder(REFLECT_101)   -> the 8 bordered pyramid levels      (ORBextractor.cc:1107-1132)
  cv2.GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) per level    -> the blurred levels                  (:1085-1086)
  cv2.FastFeatureDetector(20 / 7, nonmax) on level 0              -> keypoints + responses               (:809-815)
  cv2.fastAtan2 on a grid of (y, x)                               -> degrees
  cv2.undistortPoints on a grid, TUM1 coefficients                -> mvKeysUn arithmetic                 (Frame.cc:835-871)
  cv2.createLineSegmentDetector(LSD_REFINE_ADV).detect            -> segments of the line fixture        (LSDextractor.cpp:14-17)
  cv2.GaussianBlur(7x7, sigma 2) on a grid of isolated impulses   -> the 8-bit fixed-point kernel itself (every amplitude 1..255 through
                                                                     every tap pair: decides ALL descriptor bits; the oracle's taps sum to 257)
  cv2.line_descriptor LSDDetector.detect + BinaryDescriptor.compute -> key lines + 32-byte LBD rows     (LSDextractor.cpp:14-30; needs opencv-contrib)

tests/test_opencv_pins.py compares the oracle with the file when it exists and SKIPS (reporting "parity unpinned") when it
does not - which is the state of this repository: no OpenCV is installed or installable in the build container.

usage: python tools/dump_opencv_reference.py            (needs: numpy, cv2 with version 3.4.x)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    import cv2
    if not cv2.__version__.startswith("3.4"):
        sys.stderr.write("warning: the reference links OpenCV 3.4.4; this is %s - the 8-bit GaussianBlur taps and the LSD pixel "
                         "ordering changed between versions\n" % cv2.__version__)
    out = {"cv_version": np.array(cv2.__version__)}
    for tag, name in (("low", "orb_lowtexture_640x480.npz"), ("room", "orb_room_320x240.npz")):
        g = np.load(os.path.join(GOLD, name))
        gray, params = g["gray"], g["params"]
        nlevels, sf = int(params[2]), np.float32(params[1])
        scale = np.float32(1.0)
        prev = gray
        for l in range(nlevels):
            if l > 0:
                scale = np.float32(scale * sf)
                inv = np.float32(1.0) / scale
                size = (int(round(float(np.float32(gray.shape[1]) * inv))), int(round(float(np.float32(gray.shape[0]) * inv))))
                prev = cv2.resize(prev, size, interpolation=cv2.INTER_LINEAR)
            out[f"{tag}_pyr{l}"] = cv2.copyMakeBorder(prev, 19, 19, 19, 19, cv2.BORDER_REFLECT_101)
            out[f"{tag}_blur{l}"] = cv2.GaussianBlur(prev.copy(), (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
        for th in (20, 7):
            kps = cv2.FastFeatureDetector_create(th, True).detect(gray)
            out[f"{tag}_fast{th}"] = np.array([[k.pt[0], k.pt[1], k.response] for k in kps], np.float32).reshape(-1, 3)
    rng = np.random.RandomState(7)
    yx = rng.normal(0, 1000, (4096, 2)).astype(np.float32)
    out["atan2_yx"] = yx
    out["atan2_deg"] = np.array([cv2.fastAtan2(float(y), float(x)) for y, x in yx], np.float32)
    K = np.array([[517.306408, 0, 318.643040], [0, 516.469215, 255.313989], [0, 0, 1]], np.float32)
    dist = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)
    pts = np.stack(np.meshgrid(np.linspace(0, 639, 41), np.linspace(0, 479, 31)), -1).reshape(-1, 1, 2).astype(np.float32)
    out["undist_in"] = pts
    out["undist_out"] = cv2.undistortPoints(pts, K, dist, None, None, K)
    lines_img = np.load(os.path.join(GOLD, "planes_lines_bow.npz"))["gray"] if "gray" in np.load(os.path.join(GOLD, "planes_lines_bow.npz")) else gray
    lsd = cv2.createLineSegmentDetector(cv2.LSD_REFINE_ADV)

    def detect(img, tag):
        """segments with their exact float32 bits plus the detector's optional outputs: width, precision and -log10(NFA) of
        every segment.  The NFA is a function of rect_nfa's (total_pts, alg_pts, p) alone, so these values say which pixels
        the library's rect_nfa walked AND how nfa() sums its first term - the two questions drfe_lsd_configure_rect leaves open:
        integer or real-valued steps (modes 0 / 2 against 1), and `double(n) + 1` against log_gamma(n + 1) (mode 0 against 1 / 2:
        the values differ by log_gamma(n + 1) - (n + 1) over ln 10, hundreds for any real segment, and the library's reading lets
        four to five times as many segments through - 1397 against 298 on the committed line fixture).
        tests/test_opencv_pins.py::test_lsd_reading_decided_by_opencv names the mode that matches."""
        segs, width, prec, nfa = lsd.detect(img)
        n = 0 if segs is None else len(segs)
        out[tag + "_img"] = img
        out[tag + "_segments"] = np.zeros((0, 4), np.float32) if n == 0 else np.asarray(segs, np.float32).reshape(-1, 4)
        out[tag + "_width"] = np.zeros(0) if n == 0 else np.asarray(width, np.float64).reshape(-1)
        out[tag + "_prec"] = np.zeros(0) if n == 0 else np.asarray(prec, np.float64).reshape(-1)
        out[tag + "_nfa"] = np.zeros(0) if n == 0 else np.asarray(nfa, np.float64).reshape(-1)

    detect(lines_img, "lsd")
    # the decisive scenes: one long oblique step edge (slope 0.6) and its 45-degree sibling.  Under the literal reading of
    # rect_nfa (integer step quotients) the first one is REJECTED by LSD_REFINE_ADV and the second kept; under the real-valued
    # reading both are kept (tests/test_oracle_cpu2.py::test_lsd_lines_on_clean_edges).
    yy, xx = np.mgrid[0:480, 0:640]
    ob = np.full((480, 640), 50, np.uint8)
    ob[(yy - 0.6 * xx) > 40] = 200
    detect(ob, "lsd_oblique")
    d45 = np.full((480, 640), 50, np.uint8)
    d45[(yy - xx) > 40] = 200
    detect(d45, "lsd_diag45")
    # ---- round 6: the two items that decide the most output bits and were not pinned yet ----
    # (1) the 7-tap 8-bit Gaussian OpenCV really applies for sigma 2: a 16 x 16 grid of isolated impulses, 8 px apart (the 7 x 7 responses
    #     do not overlap), amplitudes 1 .. 255 (the last one repeated): each response is the fixed-point product of the two 1-D kernels with
    #     its rounding - any other tap set or rounding rule differs somewhere on this image.  Plus one noise image for the general case.
    imp = np.zeros((16 * 8 + 8, 16 * 8 + 8), np.uint8)
    amp = np.minimum(np.arange(256) + 1, 255).astype(np.uint8).reshape(16, 16)
    imp[8:8 + 16 * 8:8, 8:8 + 16 * 8:8] = amp
    out["blur_impulse_in"] = imp
    out["blur_impulse_out"] = cv2.GaussianBlur(imp.copy(), (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    noise = np.random.RandomState(11).randint(0, 256, (97, 131)).astype(np.uint8)
    out["blur_noise_in"] = noise
    out["blur_noise_out"] = cv2.GaussianBlur(noise.copy(), (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    # (2) LineSegment::ExtractLineSegment as the reference spells it (src/LSDextractor.cpp:14-30): LSDDetector::detect(img, keylines, 1.2, 1)
    #     and BinaryDescriptor::compute on ALL detected lines (no top-40 cut: the cut is std::sort on the response and is checked apart) -
    #     every KeyLine field and every LBD row.  LBD stays unpinned even with the LSD segments pinned: band sums, the float walks, the
    #     normalisation chain and the binary conversion are restated from memory of descriptor.cpp / binary_descriptor.cpp.
    try:
        ld = cv2.line_descriptor
        det = ld.LSDDetector_createLSDDetector() if hasattr(ld, "LSDDetector_createLSDDetector") else ld.LSDDetector.createLSDDetector()
        bd = ld.BinaryDescriptor_createBinaryDescriptor() if hasattr(ld, "BinaryDescriptor_createBinaryDescriptor") else ld.BinaryDescriptor.createBinaryDescriptor()
        fields = ("angle", "class_id", "octave", "pt", "response", "size", "startPointX", "startPointY", "endPointX", "endPointY",
                  "sPointInOctaveX", "sPointInOctaveY", "ePointInOctaveX", "ePointInOctaveY", "lineLength", "numOfPixels")
        for tag, img in (("lbd", lines_img), ("lbd_oblique", ob), ("lbd_diag45", d45)):
            kls = det.detect(img, 1.2, 1)
            kls, desc = bd.compute(img, kls)
            rows = []
            for k in kls:
                r = []
                for f in fields:
                    v = getattr(k, f)
                    r += [float(v[0]), float(v[1])] if f == "pt" else [float(v)]
                rows.append(r)
            out[tag + "_img"] = img
            out[tag + "_keylines"] = np.array(rows, np.float64).reshape(-1, 17)       # field order of cv::line_descriptor::KeyLine
            out[tag + "_desc"] = np.zeros((0, 32), np.uint8) if desc is None else np.asarray(desc, np.uint8).reshape(-1, 32)
    except (AttributeError, cv2.error) as e:
        sys.stderr.write("note: cv2.line_descriptor is not usable in this OpenCV build (%s): the LBD rows stay unpinned; install "
                         "opencv-contrib-python 3.4.x\n" % e)
    np.savez_compressed(os.path.join(GOLD, "opencv_pins.npz"), **out)
    print("wrote", os.path.join(GOLD, "opencv_pins.npz"), "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()

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
    segs = lsd.detect(lines_img)[0]
    out["lsd_img"] = lines_img
    out["lsd_segments"] = np.zeros((0, 4), np.float32) if segs is None else segs.reshape(-1, 4).astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, "opencv_pins.npz"), **out)
    print("wrote", os.path.join(GOLD, "opencv_pins.npz"), "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()

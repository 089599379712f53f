"""Per-frame pipelined flow (drfe_frame_submit / drfe_frame_collect) against the synchronous single-frame entry:
host cost of a submission, latency of one frame, frames/s with 1 / 2 / 4 submissions in flight.
    python tools/per_frame_flow.py [frames]        (DRFE_NO_GRAPH=1: plain enqueue instead of the captured graph)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dr_slam_amd import synth          # noqa: E402
from dr_slam_amd.pipeline import FrontEnd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    cam = synth.TUM3
    frames = list(synth.sequence(3, 16, cam=cam))
    print(f"graph: {'off (DRFE_NO_GRAPH=1)' if os.environ.get('DRFE_NO_GRAPH') == '1' else 'on'}; {n} frames 640x480, ORB + stereo/grid glue")
    fe = FrontEnd(cam, max_batch=4)
    c = fe.ctx
    for _ in range(20):
        c.orb_extract(frames[0][0])
    t = time.perf_counter()
    for i in range(n):
        c.orb_extract(frames[i % 16][0])
    dt = time.perf_counter() - t
    print(f"drfe_orb_extract (synchronous, ORB only)        {1e3 * dt / n:7.3f} ms/frame  {n / dt:8.0f} frames/s")
    for depth in (False, True):
        for inflight in (1, 2, 4):
            for w in range(8):                         # warm-up: staging, graphs
                c.frame_submit(w % 4, frames[0][0], frames[0][1] if depth else None, fe.cam)
                c.frame_collect(w % 4)
            sub = 0.0
            t = time.perf_counter()
            for i in range(n + inflight - 1):
                if i < n:
                    t0 = time.perf_counter()
                    c.frame_submit(i % inflight, frames[i % 16][0], frames[i % 16][1] if depth else None, fe.cam)
                    sub += time.perf_counter() - t0
                if i >= inflight - 1:
                    c.frame_collect((i - inflight + 1) % inflight, stereo=depth)
            dt = time.perf_counter() - t
            print(f"submit/collect {'ORB + glue' if depth else 'ORB only  '} {inflight} in flight          {1e3 * dt / n:7.3f} ms/frame  "
                  f"{n / dt:8.0f} frames/s   host time inside submit {1e3 * sub / n:6.3f} ms")
    # one submission per tracked frame: ORB + glue + SearchByProjection(Cur, Last) in the slot's captured graph
    Twc = np.stack([f[2] for f in frames]).astype(np.float64)
    Tcw = np.linalg.inv(Twc).astype(np.float32)
    Twc = Twc.astype(np.float32)
    for inflight in (1, 2):
        S = inflight + 1
        c.frame_submit(0, frames[0][0], frames[0][1], fe.cam)
        c.frame_collect(0)
        for w_ in range(1, 9):                          # warm-up: graphs of every slot
            c.frame_submit_tracked(w_ % S, frames[w_ % 16][0], frames[w_ % 16][1], fe.cam, (w_ - 1) % S, Tcw[w_ % 16], Tcw[(w_ - 1) % 16], Twc_last=Twc[(w_ - 1) % 16])
            c.frame_collect_tracked(w_ % S)
        sub = 0.0
        nm = 0
        t = time.perf_counter()
        for i in range(9, 9 + n + inflight - 1):
            if i < 9 + n:
                t0 = time.perf_counter()
                c.frame_submit_tracked(i % S, frames[i % 16][0], frames[i % 16][1], fe.cam, (i - 1) % S, Tcw[i % 16], Tcw[(i - 1) % 16], Twc_last=Twc[(i - 1) % 16])
                sub += time.perf_counter() - t0
            if i >= 9 + inflight - 1:
                nm += c.frame_collect_tracked((i - inflight + 1) % S)[5]
        dt = time.perf_counter() - t
        print(f"submit/collect tracked: ORB + glue + SearchByProjection {inflight} in flight   {1e3 * dt / n:7.3f} ms/frame  "
              f"{n / dt:8.0f} frames/s   host time inside submit {1e3 * sub / n:6.3f} ms   ({nm / n:.0f} matches per frame)")
    c.close()


if __name__ == "__main__":
    main()

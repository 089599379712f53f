"""FAST stage and whole-step time of the headline path per scene kind (the textures BASELINE's configs name): 512-frame batches of
room_boxes / living_room / planar_lowtexture, HIP-event stage times (one context, one batch at a time) and the step rate.
    python tools/fast_scene_stages.py [batch=512]        (DRFE_FAST_SCREEN=0 / 1 / 2: no screened path / at minThFAST only / at iniThFAST first - the default - for A/B)"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from dr_slam_amd import sharding, synth
from dr_slam_amd.pipeline import FrontEnd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
out = {"batch": B, "fast_screen": os.environ.get("DRFE_FAST_SCREEN", "2"), "scenes": {}}
for kind, cam in (("room_boxes", synth.TUM3), ("living_room", synth.ICL), ("planar_lowtexture", synth.TUM3)):
    base = sharding.render_sequence(3, 16, cam, kind, workers=8)
    order = sharding.pingpong_order(B, len(base))
    gray = torch.from_numpy(np.stack([base[i][0] for i in order])).cuda()
    depth = torch.from_numpy(np.stack([base[i][1] for i in order]).view(np.int16)).cuda()
    Twc = np.stack([base[i][2] for i in order]).astype(np.float64)
    Tcw = np.linalg.inv(Twc).astype(np.float32); Twc = Twc.astype(np.float32)
    fe = FrontEnd(cam, max_batch=B)
    for _ in range(5):
        fe.process(gray, depth, Tcw, Twc, stream=0)
    torch.cuda.synchronize()
    fe.ctx.profile_enable(True)
    acc = {}
    for _ in range(10):
        fe.process(gray, depth, Tcw, Twc, stream=0)
        for k, v in fe.ctx.profile_stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 10
    fe.ctx.profile_enable(False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        fe.process(gray, depth, Tcw, Twc, stream=0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    out["scenes"][kind] = {"stage_ms_per_batch": {k: round(v, 4) for k, v in acc.items()}, "frames_per_s_one_context": round(B * 100 / el),
                           "keypoints_per_frame_min": int(fe.ctx.orb_counts(B).min())}
    fe.ctx.close()
print(json.dumps(out))

#!/usr/bin/env python3
"""Soak: 4 frames in flight (frame slots, 4 streams, pano_compose_pair) over several hundred steps with changing frame
content; every panorama must equal the one the same frames give one frame at a time."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c2_group()
F, SETS, STEPS = int(os.environ.get("F", "4")), 6, int(os.environ.get("STEPS", "600"))
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
ow, oh = ctxs[0].output_size()
st0 = torch.cuda.current_stream().cuda_stream
sets = [[[torch.from_numpy(synth_frame(g["w"], g["h"], 1000 + 97 * s + 4 * k + i)).cuda() for i in range(4)] for k in range(2)] for s in range(SETS)]
strides = [g["w"] * 3] * 4
ref = []
for s in range(SETS):  # one frame at a time
    o = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in sets[s][0]], strides, o[0].data_ptr(), ow * 3,
                         [t.data_ptr() for t in sets[s][1]], strides, o[1].data_ptr(), ow * 3, st0)
    torch.cuda.synchronize()
    ref.append([x.clone() for x in o])
for c in ctxs:
    c.set_frame_slots(F)
# the library's flight streams (probed to sit on distinct hardware queues), wrapped for torch
_fs, _distinct = ctxs[0].frame_streams(F)
streams = [torch.cuda.ExternalStream(p) for p in _fs]
outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(F)]
bad = 0
pending = [None] * F
t0 = time.perf_counter()
for k in range(STEPS + F):
    f = k % F
    if pending[f] is not None:  # the frame that used this slot F steps ago
        streams[f].synchronize()
        s = pending[f]
        for q in range(2):
            if not torch.equal(outs[f][q], ref[s][q]):
                bad += 1
        pending[f] = None
    if k < STEPS:
        s = (k * 5 + k // 7) % SETS
        for c in ctxs:
            c.select_frame_slot(f)
        ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in sets[s][0]], strides, outs[f][0].data_ptr(), ow * 3,
                             [t.data_ptr() for t in sets[s][1]], strides, outs[f][1].data_ptr(), ow * 3, streams[f].cuda_stream)
        pending[f] = s
torch.cuda.synchronize()
print(json.dumps({"steps": STEPS, "frames_in_flight": F, "distinct_hw_queues": _distinct, "mismatching_panoramas": bad, "seconds": round(time.perf_counter() - t0, 2)}))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Frames pipelined by PHASE instead of by frame: stream FRONT runs the warp + pyrDown 0->1 of every frame, streams MID0 / MID1 the
latency-bound middle (rest of the pyramid, blend levels >= 2) of the even / odd frames, stream BACK blend levels 1 and 0 - so that at
any moment one bandwidth-bound head, one bandwidth-bound tail and up to two middles are in flight, instead of whatever four
independent frame chains happen to line up.  us per config-2 frame; MODE=frames runs the four-frames-in-flight loop for comparison.
   python3 tools/pipeline_time.py [steps] [check]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
PHASE_FRONT, PHASE_MIDDLE, PHASE_BACK, PHASE_ALL = 1, 2, 4, 7   # include/pano.h of the patched build
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
check = len(sys.argv) > 2 and sys.argv[2] == "check"
mode = os.environ.get("MODE", "phases")
NMID = int(os.environ.get("NMID", "2"))
F = int(os.environ.get("SLOTS", "4"))
g = c2_group()
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4): ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
nsets = 3
sets = [[[torch.from_numpy(synth_frame(g["w"], g["h"], 42 + 100 * s + 4 * k + i)).cuda() for i in range(4)] for k in range(2)] for s in range(nsets)]
ptr = [[[t.data_ptr() for t in fr] for fr in s] for s in sets]
ow, oh = ctxs[0].output_size()
strides = [g["w"] * 3] * 4
st0 = torch.cuda.current_stream().cuda_stream
ref = []
if check:
    for s in range(nsets):
        o = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        ctxs[0].compose_pair(ctxs[1], ptr[s][0], strides, o[0].data_ptr(), ow * 3, ptr[s][1], strides, o[1].data_ptr(), ow * 3, st0)
        torch.cuda.synchronize(); ref.append(o)
for c in ctxs: c.set_frame_slots(F)
fs, distinct = ctxs[0].frame_streams(4)
S = [torch.cuda.ExternalStream(p) for p in fs]
outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(F)]
# lean host side: HIP events straight through ctypes, argument arrays built once (the Python mirror's per-call marshalling and torch's
# event objects cost more per frame than the GPU needs)
import ctypes as C  # (needs the build of experiments/compose_in_phases.patch: PANO_LIB=experiments/_build/libpano_phases.so)
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [C.c_void_p, C.c_uint]; hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
def mkev():
    e = C.c_void_p(); assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0; return e   # hipEventDisableTiming
ev_front = [mkev() for _ in range(F)]; ev_mid = [mkev() for _ in range(F)]; ev_done = [mkev() for _ in range(F)]
used = [False] * F
lib = ctxs[0].lib
lib.pano_compose_pair_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_uint, C.c_void_p]
lib.pano_select_frame_slot.argtypes = [C.c_void_p, C.c_int]
cst = (C.c_size_t * 4)(*strides)
cptr = [[(C.c_void_p * 4)(*ptr[s][q]) for q in range(2)] for s in range(nsets)]
def phase(f, s, ph, stream):
    lib.pano_select_frame_slot(ctxs[0].h, f); lib.pano_select_frame_slot(ctxs[1].h, f)
    r = lib.pano_compose_pair_phases(ctxs[0].h, ctxs[1].h, cptr[s][0], cst, outs[f][0].data_ptr(), ow * 3, cptr[s][1], cst, outs[f][1].data_ptr(), ow * 3, ph, stream)
    assert r == 0, r
def step_phases(k):
    f, s = k % F, k % nsets
    front, back, mid = fs[0], fs[1], fs[2 + (k % NMID)]
    if used[f]: hip.hipStreamWaitEvent(front, ev_done[f], 0)          # the slot's previous frame has left its last kernel
    phase(f, s, PHASE_FRONT, front); hip.hipEventRecord(ev_front[f], front)
    hip.hipStreamWaitEvent(mid, ev_front[f], 0); phase(f, s, PHASE_MIDDLE, mid); hip.hipEventRecord(ev_mid[f], mid)
    hip.hipStreamWaitEvent(back, ev_mid[f], 0); phase(f, s, PHASE_BACK, back); hip.hipEventRecord(ev_done[f], back)
    used[f] = True
def step_frames(k):
    f, s = k % F, k % nsets
    phase(f, s, PHASE_ALL, fs[f])
step = step_phases if mode == "phases" else step_frames
bad = 0
if check:
    for k in range(4 * F + 3):
        step(k)
        if k >= F - 1:   # frame k - (F - 1) is the oldest still un-checked; wait for everything (simple, untimed)
            torch.cuda.synchronize()
            kk = k
            bad += sum(not torch.equal(outs[kk % F][q], ref[kk % nsets][q]) for q in range(2))
    torch.cuda.synchronize()
for k in range(400): step(k)
torch.cuda.synchronize()
best, enq = [], []
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / steps * 1e6); enq.append((t1 - t0) / steps * 1e6)
# how fast the host alone is: a short burst into an idle queue
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(24): step(k)
host_us = (time.perf_counter() - t0) / 24 * 1e6
torch.cuda.synchronize()
print(json.dumps({"mode": mode, "slots": F, "mid_streams": NMID, "distinct_hw_queues": distinct, "mismatching_panoramas": bad if check else None,
                  "us_per_frame": [round(b, 2) for b in best], "enqueue_us_per_frame": [round(b, 1) for b in enq], "host_us_per_frame_burst_of_24": round(host_us, 1)}))

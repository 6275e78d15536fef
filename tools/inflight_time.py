#!/usr/bin/env python3
"""us per config-2 frame (8 x 1080p, both stitchers per launch sequence) with F frames in flight - the shape of bench.py's timed
region, nothing else.  PANO_LIB selects a library build; with the `skip_launches` variant (experiments/skip_launches.patch,
TIMING ONLY) PANO_SKIP=<bits> leaves launches out: 1 K1, 2 pyrDown 0->1, 4 pyrDown 1->2, 8 tail, 16 small levels, 32 level 2,
64 level 1, 128 level 0.   python3 tools/inflight_time.py [F] [steps] [rotate]   (rotate=1: six frame sets, > the 256 MiB cache)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rotate = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = c2_group()
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
nsets = 6 if rotate else 1
sets = [[[torch.from_numpy(synth_frame(g["w"], g["h"], 42 + 4 * k + i)).cuda() for i in range(4)] for k in range(2)]]
for _ in range(nsets - 1):
    sets.append([[t.clone() for t in fr] for fr in sets[0]])
ptr = [[[t.data_ptr() for t in fr] for fr in s] for s in sets]
ow, oh = ctxs[0].output_size()
for c in ctxs:
    c.set_frame_slots(F)
# PANO_TORCH_STREAMS=1: torch.cuda.Stream()s (wherever the runtime puts them) instead of the library's probed flight streams
# PANO_CU_MASK=block|stride|halves: every flight stream gets its own share of the 256 CUs (hipExtStreamCreateWithCUMask) - bits
# F*k .. in one run ("block"), every F-th bit ("stride"), or two streams per half ("halves")
_cm = os.environ.get("PANO_CU_MASK")
if _cm:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    fstreams, distinct = [], None
    for f in range(F):
        if _cm == "block":
            bits = [i for i in range(ncu) if i * F // ncu == f]
        elif _cm == "stride":
            bits = [i for i in range(ncu) if i % F == f]
        else:
            bits = [i for i in range(ncu) if (i * 2 // ncu) == (f % 2)]
        words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
        for b in bits:
            words[b // 32] |= 1 << (b % 32)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), words)
        assert rc == 0, rc
        fstreams.append(st.value)
elif os.environ.get("PANO_TORCH_STREAMS") == "1" or F > pano.MAX_FRAME_SLOTS:
    _ts = [torch.cuda.Stream() for _ in range(F)]
    fstreams, distinct = [t.cuda_stream for t in _ts], None
else:
    fstreams, distinct = ctxs[0].frame_streams(F)
outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(F)]
strides = [g["w"] * 3] * 4
def step(k):
    f = k % F
    ctxs[0].select_frame_slot(f); ctxs[1].select_frame_slot(f)
    p = ptr[k % nsets]
    ctxs[0].compose_pair(ctxs[1], p[0], strides, outs[f][0].data_ptr(), ow * 3, p[1], strides, outs[f][1].data_ptr(), ow * 3,
                         fstreams[f])
for k in range(400): step(k)
torch.cuda.synchronize()
best = []
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / steps * 1e6)
print(json.dumps({"lib": os.path.basename(os.environ.get("PANO_LIB", "product")), "skip": os.environ.get("PANO_SKIP", "0"), "F": F, "distinct_hw_queues": distinct,
                  "rotate": rotate, "cu_mask": os.environ.get("PANO_CU_MASK"), "us_per_frame": [round(b, 2) for b in best]}))

import importlib, os, sys, time
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
for (w, h, f, n) in ((1920, 1080, 1002.416, 4), (960, 540, 501.2, 4)):
    g = c2_group(w=w, h=h, f=f)
    ctx = pano.Context(n, w, h, scale=g["scale"], num_bands=5, device=0)
    for i in range(n): ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi()
    frames = [synth_frame(w, h, 5 + i) for i in range(n)]
    ctx.compose_host(frames); ctx.compose_host(frames)
    for rep in range(3):
        t0 = time.perf_counter(); ctx.refresh_masks_begin(frames); t1 = time.perf_counter()
        ctx.refresh_masks_wait(); t2 = time.perf_counter()   # includes the thread's cuts + the install
        ctx.compose_host(frames); t3 = time.perf_counter()    # ensure_weights + one frame
        ctx.compose_host(frames); t4 = time.perf_counter()
        # install alone: begin, sleep until surely done, then poll
        ctx.refresh_masks_begin(frames); time.sleep(0.5)
        t5 = time.perf_counter(); d = ctx.refresh_masks_poll(); t6 = time.perf_counter()
        ctx.compose_host(frames)
        print("%dx%d rep %d: begin %.2f ms, cuts+install %.2f, first frame after %.2f, next frame %.2f, poll-install alone %.2f (%s)" %
              (w, h, rep, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t6-t5), d))

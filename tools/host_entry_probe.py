#!/usr/bin/env python3
"""where the time of pano_compose_host goes: pageable / page-locked caller memory, one thread / two stitchers on two threads,
and the cost of the page-locked test itself (hipPointerGetAttributes)"""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import c2_group, synth_frame  # noqa: E402


def main():
    pano = importlib.import_module("img-stitching_amd")
    g = c2_group()
    W, H = g["w"], g["h"]
    ctxs = []
    for grp in range(2):
        ctx = pano.Context(4, W, H, scale=g["scale"], num_bands=5, device=0)
        for i in range(4):
            ctx.set_camera(i, g["K"][i], g["R"][i])
        ctx.prepare(); ctx.build_masks_voronoi()
        ctxs.append(ctx)
    ow, oh = ctxs[0].output_size()
    fr = [[synth_frame(W, H, 42 + 4 * grp + i) for i in range(4)] for grp in range(2)]
    outs = [np.empty((oh, ow, 3), np.uint8) for _ in range(2)]
    pin = [[pano.HostBuffer((H, W, 3)) for _ in range(4)] for _ in range(2)]
    pout = [pano.HostBuffer((oh, ow, 3)) for _ in range(2)]
    for grp in range(2):
        for i in range(4):
            pin[grp][i].array[:] = fr[grp][i]

    def t(fn, reps=30):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e3

    print("one stitcher, pageable   : %.3f ms" % t(lambda: ctxs[0].compose_host(fr[0], out=outs[0])))
    print("one stitcher, page-locked: %.3f ms" % t(lambda: ctxs[0].compose_host([b.array for b in pin[0]], out=pout[0].array)))
    print("one stitcher, page-locked in, pageable out: %.3f ms" % t(lambda: ctxs[0].compose_host([b.array for b in pin[0]], out=outs[0])))
    print("one stitcher, pageable in, page-locked out: %.3f ms" % t(lambda: ctxs[0].compose_host(fr[0], out=pout[0].array)))
    hip = C.CDLL("libamdhip64.so")
    buf = (C.c_char * 256)()
    p = pin[0][0].ptr
    t0 = time.perf_counter()
    for _ in range(1000):
        hip.hipPointerGetAttributes(buf, C.c_void_p(p))
    print("hipPointerGetAttributes(page-locked): %.2f us" % ((time.perf_counter() - t0) * 1e3))
    q = fr[0][0].ctypes.data
    t0 = time.perf_counter()
    for _ in range(1000):
        hip.hipPointerGetAttributes(buf, C.c_void_p(q))
    print("hipPointerGetAttributes(pageable): %.2f us" % ((time.perf_counter() - t0) * 1e3))

    def both(frames, out):
        th = [threading.Thread(target=lambda k=k: ctxs[k].compose_host(frames[k], out=out[k])) for k in range(2)]
        [x.start() for x in th]; [x.join() for x in th]
    print("two stitchers on two threads, pageable   : %.3f ms" % t(lambda: both(fr, outs)))
    print("two stitchers on two threads, page-locked: %.3f ms" % t(lambda: both([[b.array for b in g_] for g_ in pin], [b.array for b in pout])))


if __name__ == "__main__":
    main()

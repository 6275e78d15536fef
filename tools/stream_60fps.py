#!/usr/bin/env python3
"""BASELINE.json config 5 - the paced capture loop: 8 x 1080p frames arrive in host memory every 1/fps seconds (the reference's
loop pops one cv::Mat per camera from the nvCam queues and calls process(), src/master.cpp:302-411) and go through the streaming
entries pano_stream_* (page-locked slots, H2D || compose || D2H, two panoramas in flight).

Per tick: the frame set of this tick (rotating host buffers, so nothing stays warm) is written into the slot's page-locked inputs -
the write the capture thread does - the slot is submitted, and the PREVIOUS tick's panorama is waited for.  Reported: achieved fps,
p50 / p99 / max latency from a frame's tick to its panorama in host memory, and dropped frames (a tick the loop reaches more than a
period late is skipped and counted).  PANO_GRAPH=1 in the environment runs the same loop with hipGraph replay of the launch sequence.

    python tools/stream_60fps.py [--fps 60] [--frames 600] [--width 1920 --height 1080] [--check]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# the lens of cfg/cameras.yaml sensing/imx390/fov120 (K, distorParams and rect of its 960 x 540 entry), scaled to an undistorted size
def imx390_front(raw_wh, undist_wh):
    k = undist_wh[0] / 960.0
    K = [4.890925118101495e+02 * k, 0, 4.940763211103715e+02 * k, 0, 4.912630345468579e+02 * k, 2.865820139005963e+02 * k, 0, 0, 1]
    return {"raw": tuple(raw_wh), "undist": tuple(undist_wh), "K": K, "dist": [-0.2838, 0.0628, 0, 0],
            "rect": tuple(int(round(v * k)) for v in (70, 66, 885, 410))}


def run(fps=60.0, frames=600, width=1920, height=1080, bands=5, nsets=4, check=False, device=0, pipeline=None, refresh_every=0,
        refresh_async=True, front=None):
    """refresh_every: N > 0 refreshes the graph-cut masks of both stitchers every N ticks like ocvStitcher::process (every 200
    frames, ocvstitcher.hpp:1152-1159) - inline (refresh_async False: pano_build_masks_graphcut inside the tick, as the reference
    does) or beside the loop (pano_refresh_masks_begin / _poll).
    front: None, or imx390_front(raw size, undistorted size): the frames offered are RAW camera frames and the undistort -> crop ->
    resize front end of include/nvcam.hpp:898-921 runs fused inside the warp (pano_set_undistort) - BASELINE config 5 as stated.
    pipeline: True = wait for tick k's panorama after submitting tick k + 1 (two in flight, what throughput needs), False = wait
    right after the submit (lowest latency); None = False when a period leaves room for it (fps <= 100)"""
    if pipeline is None:
        pipeline = fps > 100.0
    from helpers import c2_group, synth_frame
    pano = importlib.import_module("img-stitching_amd")
    g = c2_group(w=width, h=height, f=1002.416 * width / 1920.0)
    NG, NC = 2, 4
    ctxs = []
    for grp in range(NG):
        ctx = pano.Context(NC, width, height, scale=g["scale"], num_bands=bands, device=device)
        for i in range(NC):
            ctx.set_camera(i, g["K"][i], g["R"][i])
            if front:
                ctx.set_undistort(i, front["raw"], front["undist"], front["K"], front["dist"], front["rect"])
        ctx.prepare()
        ctx.build_masks_voronoi()
        ctxs.append(ctx)
    # rotating frame sets in ordinary (pageable) host memory: what a capture queue hands over
    fw, fh = front["raw"] if front else (width, height)
    sets = [[[synth_frame(fw, fh, 1000 + 100 * s + grp * NC + i) for i in range(NC)] for grp in range(NG)] for s in range(nsets)]
    ins = [[[ctxs[grp].stream_input(s, i) for i in range(NC)] for grp in range(NG)] for s in range(2)]
    period = 1.0 / fps
    tick_t, done_t = {}, {}
    sample, sample_masks = {}, {}
    # the ticks to sample; a sample tick that is DROPPED hands its sample to the next composed tick, and a refresh that falls due on
    # a dropped tick begins with the next composed one: one late tick changes the report, not the outcome of a check
    sample_due = [0, frames // 2, frames - 1] if check else []
    sample_ticks = set()
    refresh_due = 0
    dropped = 0
    pending = None   # (frame index, slot)
    refreshed = [0]

    def finish(p):
        k, s = p
        for grp in range(NG):
            ctxs[grp].stream_wait(s)
        done_t[k] = time.perf_counter()
        if k in sample_ticks:
            sample[k] = [ctxs[grp].stream_output(s).copy() for grp in range(NG)]

    # two untimed frames: page-locked slots, device buffers and weights come into being here
    for s in range(2):
        for grp in range(NG):
            for i in range(NC):
                ins[s][grp][i][:] = sets[0][grp][i]
            ctxs[grp].stream_submit(s)
        for grp in range(NG):
            ctxs[grp].stream_wait(s)
    t0 = time.perf_counter() + 0.01
    slot = 0
    for k in range(frames):
        target = t0 + k * period
        now = time.perf_counter()
        if refresh_every and k and k % refresh_every == 0:
            refresh_due += 1
        if now > target + period:      # this tick is over before we got here: the frame is lost
            dropped += 1
            continue
        if sample_due and sample_due[0] <= k:
            sample_ticks.add(k)
            while sample_due and sample_due[0] <= k:
                sample_due.pop(0)
        while now < target:
            if target - now > 0.002:
                time.sleep(target - now - 0.001)
            now = time.perf_counter()
        tick_t[k] = target
        fs = sets[k % nsets]
        if refresh_every and not refresh_async:
            for grp in range(NG):
                if refresh_due:     # inside the tick, in front of the frame, like ocvStitcher::process
                    ctxs[grp].build_masks_graphcut(fs[grp])
                    refreshed[0] += 1
            refresh_due = 0
        if k in sample_ticks:   # the masks this tick's panoramas are composed with (a refresh may have installed new ones since the last sample)
            sample_masks[k] = [[ctxs[grp].get_mask(i) for i in range(NC)] for grp in range(NG)]
        for grp in range(NG):
            for i in range(NC):
                ins[slot][grp][i][:] = fs[grp][i]          # the capture thread's write into the slot
        for grp in range(NG):
            ctxs[grp].stream_submit(slot)
        if refresh_every and refresh_async:
            # behind the submit: the frame is on its way while the refresh's frames are uploaded and warped (begin) or its
            # masks installed (poll).  Both stitchers of the rig begin in the SAME tick: they are independent contexts with a
            # refresh thread each, so the rig's refresh takes one stitcher's graph-cut time (58 ms for 8 x 1080p) instead of two
            # (103 ms: profiles/r04_refresh_rig_beside_the_loop.json)
            for grp in range(NG):
                if ctxs[grp].refresh_masks_poll():
                    refreshed[0] += 1
                if refresh_due:
                    ctxs[grp].refresh_masks_begin(fs[grp])
            refresh_due = 0
        if pipeline:
            if pending is not None:
                finish(pending)
            pending = (k, slot)
        else:
            finish((k, slot))
        slot ^= 1
    if pending is not None:
        finish(pending)
    t_end = time.perf_counter()
    if refresh_every and refresh_async:
        for grp in range(NG):
            ctxs[grp].refresh_masks_wait()
    lat = np.array([done_t[k] - tick_t[k] for k in sorted(done_t)]) * 1e3
    out = {"config": "C5: %d x %dx%d -> 2 panoramas, %d bands, paced at %.1f fps, %d frames, pano_stream_* (2 slots)%s" %
                     (NG * NC, fw, fh, bands, fps, frames, (", hipGraph replay" if os.environ.get("PANO_GRAPH") == "1" else "") +
                      (", fused undistort front end (raw %dx%d -> undistorted %dx%d -> rect %s -> %dx%d)" % (fw, fh, front["undist"][0], front["undist"][1],
                                                                                                       list(front["rect"]), width, height) if front else "")),
           "target_fps": fps, "frames_offered": frames, "frames_composed": int(len(lat)), "dropped": int(dropped),
           "achieved_fps": round(len(lat) / (t_end - t0), 2),
           "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "p99": round(float(np.percentile(lat, 99)), 3),
                          "max": round(float(lat.max()), 3)},
           "pipelined": bool(pipeline),
           "hipgraph": [dict(zip(("graphs_held", "replays"), c.graph_stats())) for c in ctxs],
           "mask_refresh": None if not refresh_every else {"every_ticks": refresh_every, "how": "beside the loop (pano_refresh_masks_*)" if refresh_async else
                                                            "inline (pano_build_masks_graphcut inside the tick, as ocvStitcher::process does)",
                                                            "masks_installed": refreshed[0]},
           "latency_definition": "frame tick (frames in host memory) -> panorama in host memory, including the write of the 8 frames "
                                 "into the page-locked slot" + (" and one tick of pipelining (the wait happens after the next submit)" if pipeline else "")}
    if check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pano_oracle as po
        po.set_threads(min(16, os.cpu_count() or 1))
        ok = True
        fe = [po.front_end(front["raw"], front["undist"], front["K"], front["dist"], front["rect"], (width, height)) for _ in range(NC)] if front else None
        for k, got in sample.items():
            for grp in range(NG):
                want, _ = po.compose(sets[k % nsets][grp], g["K"], g["R"], g["scale"], sample_masks[k][grp], bands, front=fe)
                ok &= bool(np.array_equal(got[grp], want))
        po.set_threads(1)
        out["sampled_frames_equal_oracle"] = ok
        out["sampled_frames"] = sorted(sample)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fps", type=float, default=60.0)
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bands", type=int, default=5)
    ap.add_argument("--check", action="store_true", help="compare two sampled panoramas with the CPU oracle")
    ap.add_argument("--refresh-every", type=int, default=0, help="refresh the graph-cut masks every N ticks (the reference: 200)")
    ap.add_argument("--refresh-inline", action="store_true", help="... inside the tick like the reference, instead of beside the loop")
    ap.add_argument("--pipeline", type=int, default=-1, help="1: two panoramas in flight, 0: wait after every submit, -1: by fps")
    ap.add_argument("--raw", action="store_true", help="offer RAW frames of --width x --height: the imx390 lens front end (undistort, crop, "
                                                        "resize) runs fused in the warp - BASELINE config 5 as stated (with PANO_GRAPH=1)")
    a = ap.parse_args()
    print(json.dumps(run(a.fps, a.frames, a.width, a.height, a.bands, check=a.check,
                         pipeline=None if a.pipeline < 0 else bool(a.pipeline), refresh_every=a.refresh_every,
                         refresh_async=not a.refresh_inline,
                         front=imx390_front((a.width, a.height), (a.width, a.height)) if a.raw else None)), flush=True)


if __name__ == "__main__":
    main()

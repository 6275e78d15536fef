#!/usr/bin/env python3
"""bench.py - composed panoramas/s of the MI355X compose path on BASELINE.json config 2.

One step = one 8-camera panorama: 8 x 1920x1080 BGR8 frames (resident in HBM), two groups of four
cameras (the reference never stitches a full ring in one pass: README.md:27-29, src/master.cpp:314-318),
spherical warp + 5-band multi-band blend per group, fixed K/R (imx390-derived f=1002.416, yaws
+-22.5/+-67.5 deg), Voronoi seam masks.  Synthetic frames.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: launched by torch.distributed.run, one rank per GPU; cameras are sharded over the ranks
  (rank r owns cameras [8r/N, 8(r+1)/N)), each rank warps + builds the Gaussian pyramids of its cameras,
  ONE RCCL gather per group lands the pyramid slots on rank 0, which blends.  Total work per step is
  fixed -> "scaling": "strong".  (`replicas_panoramas_per_s` reports, as extra information, the same ranks each
  composing their own rig with no exchange.)

At N = 1 the K steps are dealt round-robin over --frames-in-flight frame slots / streams (default 4): one frame is a
chain of ten dependent launches, several too small to fill the GPU, and the chains of consecutive frames overlap the
way they do behind a capture loop (2 -> 10.4k, 3 -> 11.1k, 4 -> 11.3k panoramas/s).  --frames-in-flight 1 composes one frame at a time.

Prints ONE JSON line (rank 0).  `roofline` is the warp kernel (K1): algorithmic bytes
sum_cams(W*H*3 read once + Wt*Ht*3 written once) per launch / mean launch duration from the kernel's own dispatch
events.  A roofline fraction describes the kernel, so it is taken over K steps composed one frame at a time, where a
launch has the GPU to itself (the figure rocprofv3 reports for `bench.py --frames-in-flight 1`, profiles/);
`roofline.in_timed_region` is the same measurement over K steps with the frames in flight of the timed region, where a
launch shares the GPU with the other frame's kernels and takes correspondingly longer.  `cpu_baseline` is the CPU oracle
(OpenCV-3.4-semantics restatement, kind "port") on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class DevView:
    """zero-copy torch view of a raw device allocation (for RCCL on the library's pyramid slots)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-paths", action="store_true",
                    help="skip the PCIe-inclusive side measurements (host-buffer and streaming entries); used under rocprofv3 so that "
                         "the kernel summary covers the device-resident loop only")
    ap.add_argument("--bands", type=int, default=5)
    ap.add_argument("--one-stream", action="store_true", help="both groups on one stream (no overlap)")
    ap.add_argument("--force-sharded-path", action="store_true", help="diagnostic: run the N>1 step code at N=1")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="N=1: frame slots / streams the K steps are dealt over round-robin (1 = one frame at a time)")
    ap.add_argument("--no-isolated-pass", action="store_true",
                    help="N=1: skip the K steps composed one frame at a time (kernel durations without overlap)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from helpers import c2_group, synth_frame

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # PANO_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on ONE GPU (every rank on cuda:0, slot ranges staged through
    # host memory); the judged multi-GPU run uses RCCL, one GPU per rank
    rehearsal = world > 1 and os.environ.get("PANO_BENCH_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local = 0
    red_dev = "cpu" if rehearsal else "cuda"  # where the scalar reductions of the timing live
    torch.cuda.set_device(local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    pano = importlib.import_module("img-stitching_amd")
    g = c2_group()
    W, H, NG, NC = g["w"], g["h"], 2, 4
    ctxs = []
    for grp in range(NG):
        ctx = pano.Context(NC, W, H, scale=g["scale"], num_bands=args.bands, device=local)
        for i in range(NC):
            ctx.set_camera(i, g["K"][i], g["R"][i])
        ctx.prepare()
        ctx.build_masks_voronoi()
        ctxs.append(ctx)
    ow, oh = ctxs[0].output_size()
    frames = [[torch.from_numpy(synth_frame(W, H, 42 + grp * NC + i)).cuda() for i in range(NC)] for grp in range(NG)]
    outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(NG)]
    fptr = [[t.data_ptr() for t in fr] for fr in frames]
    strides = [W * 3] * NC
    # one HIP stream per camera group, like the reference's one thread per stitcher (src/master.cpp:314-318):
    # the latency-bound small pyramid levels of one group overlap the bandwidth-bound kernels of the other
    group_streams = [torch.cuda.Stream() for _ in range(NG)]
    torch.cuda.set_stream(group_streams[0])
    stream = group_streams[0].cuda_stream
    gstream = [st.cuda_stream for st in group_streams] if not args.one_stream else [stream] * NG

    # camera sharding (world > 1): host logic in img-stitching_amd/sharding.py (also exercised on gloo/CPU)
    sh = importlib.import_module("img-stitching_amd.sharding")
    plans = sh.group_plan(NG * NC, NC, world, rank)
    per_rank = (NG * NC) // world
    slot_views = []
    for ctx in ctxs:
        base, slot = ctx.pyramid_slots()
        slot_views.append((torch.as_tensor(DevView(base, slot * NC), device="cuda"), slot))

    # frames in flight (N = 1): one frame is a chain of ten dependent launches, several too small to fill the GPU.
    # With F frame slots (per-frame buffers replicated, everything static shared) step k runs in slot k % F on
    # stream k % F and the chains of consecutive frames overlap - what a capture loop feeding frames back to back does.
    F = max(1, min(args.frames_in_flight, pano.MAX_FRAME_SLOTS)) if world == 1 and not args.force_sharded_path else 1
    flight = [stream]
    outs_f = [outs]
    if F > 1:
        for c in ctxs:
            c.set_frame_slots(F)
        flight_streams = [torch.cuda.Stream() for _ in range(F)]
        flight = [st.cuda_stream for st in flight_streams]
        outs_f = [outs] + [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(NG)] for _ in range(F - 1)]

    def step_single(k=0, slots=None):
        # both stitchers of the rig in one launch sequence (the reference: two threads, src/master.cpp:314-318)
        f = k % (slots or F)
        if F > 1:
            ctxs[0].select_frame_slot(f)
            ctxs[1].select_frame_slot(f)
        o = outs_f[f]
        ctxs[0].compose_pair(ctxs[1], fptr[0], strides, o[0].data_ptr(), ow * 3, fptr[1], strides, o[1].data_ptr(),
                             ow * 3, flight[f] if slots is None else stream)

    def step_serial():
        for grp in range(NG):
            ctxs[grp].compose(fptr[grp], strides, outs[grp].data_ptr(), ow * 3, stream)

    def step_sharded(k=0):
        for grp, plan in enumerate(plans):
            if plan["bits"]:
                ctxs[grp].feed_cameras(plan["bits"], fptr[grp], strides, stream)
            buf, slot = slot_views[grp]
            if rehearsal:
                torch.cuda.synchronize()
            sh.exchange_slots(dist, rank, buf, slot, plan["moves"], via_host=rehearsal)   # RCCL p2p on the current stream
            if plan["blend_here"]:
                ctxs[grp].blend(outs[grp].data_ptr(), ow * 3, stream)
            if plan["pano_from"] != 0:
                if rank == plan["pano_from"]:
                    dist.send(outs[grp].cpu() if rehearsal else outs[grp], dst=0)
                elif rank == 0:
                    if rehearsal:
                        host = outs[grp].cpu()
                        dist.recv(host, src=plan["pano_from"])
                        outs[grp].copy_(host)
                    else:
                        dist.recv(outs[grp], src=plan["pano_from"])

    step = step_single if (world == 1 and not args.force_sharded_path) else step_sharded
    assert int(slot_views[0][0].numel()) == slot_views[0][1] * NC
    # N > 1: one trial step of the camera-sharded path on every rank.  If any rank cannot run it (the path has only been
    # rehearsed with gloo on CPU), all ranks agree to time independent replicas instead and the line says so.
    sharded_failed = None
    if world > 1:
        ok = 1
        try:
            step_sharded(0)
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            ok, sharded_failed = 0, repr(exc)[:200]
        flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            sharded_failed = sharded_failed or "another rank failed"
            step = step_single

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # second pass over the same K steps with HIP events on the launch stream (direct launches): per-stage and
    # K1 kernel durations for the roofline; not part of `value`
    for c in ctxs:
        c.set_profiling(True)
        c.stage_stats(reset=True)
    t1 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    dt_profiled = time.perf_counter() - t1
    stats_timed = None
    if world == 1:
        stats_timed = [c.stage_stats(reset=True) for c in ctxs]
    # optional: the same K steps one frame at a time on one stream (kernel durations without overlap)
    stats_iso, dt_iso = None, None
    if world == 1 and F > 1 and not args.no_isolated_pass:
        ti = time.perf_counter()
        for k in range(args.steps):
            step_single(k, slots=1)
        torch.cuda.synchronize()
        dt_iso = time.perf_counter() - ti
        stats_iso = [c.stage_stats(reset=True) for c in ctxs]
    # N > 1 only, extra information: the same K steps with every rank composing its OWN whole rig (replicas, no
    # exchange).  One MI355X composes a panorama in ~0.16 ms, less than it takes to move one half panorama (11.6 MB)
    # over an xGMI link, so sharding ONE rig over GPUs cannot raise throughput; independent rigs scale linearly.
    replicas_rate = None
    if world > 1:
        try:
            for c in ctxs:
                c.set_profiling(False)
            dist.barrier()
            torch.cuda.synchronize()
            tr = time.perf_counter()
            for k in range(args.steps):
                step_single(k)
            torch.cuda.synchronize()
            trt = torch.tensor([time.perf_counter() - tr], dtype=torch.float64, device=red_dev)
            dist.all_reduce(trt, op=dist.ReduceOp.MAX)
            replicas_rate = round(world * args.steps / float(trt.item()), 1)
        except Exception:  # never let the side measurement break the contract line
            replicas_rate = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    result = None
    if rank == 0:
        src_b, dst_b = ctxs[0].warp_bytes()
        # per K1 launch: at N=1 one launch warps all 8 cameras (both stitchers), when sharded one group of 4
        launches_per_step = 1 if world == 1 and not args.force_sharded_path else NG
        alg_bytes = (src_b + dst_b) * (NG // launches_per_step)
        def fold(stats):
            ms3, n3 = [0.0] * 3, [0] * 3
            for ms, n in stats:
                ms3 = [a + b for a, b in zip(ms3, ms)]
                n3 = [a + b for a, b in zip(n3, n)]
            return ms3, n3

        def k1_roofline(ms3, n3):
            avg_ms = ms3[0] / n3[0]
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            return {"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_us": round(avg_ms * 1e3, 2)}

        roofline = None
        stage_ms, stage_n = fold(stats_timed if stats_timed is not None else [c.stage_stats(reset=True) for c in ctxs])
        if world == 1 and stage_n[0]:
            traffic = None
            tp = os.path.join(ROOT, "profiles", "warp_traffic.json")
            if os.path.exists(tp):
                try:
                    traffic = json.load(open(tp)).get("hbm_bytes_per_launch_8cam")
                except Exception:
                    traffic = None
            # the kernel's roofline fraction is taken where its launches have the GPU to themselves (one frame at a
            # time); what a launch takes while it shares the GPU with the other frame in flight is reported beside it
            timed = k1_roofline(stage_ms, stage_n)
            alone, alone_stage, alone_rate = timed, None, None
            if stats_iso is not None:
                ims, inn = fold(stats_iso)
                alone = k1_roofline(ims, inn)
                alone_stage = {k: round(ims[i] / max(inn[i], 1) * 1e3, 2) for i, k in enumerate(("warp", "pyramid", "blend"))}
                alone_rate = round(args.steps / dt_iso, 1)
            roofline = {"kernel": "warp_tiles_lut_kernel", "bound": "hbm", "achieved": alone["achieved"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alone["frac"],
                        "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                        "avg_launch_us": alone["avg_launch_us"], "launches_per_step": launches_per_step,
                        "measured": "K steps, one frame at a time, dispatch events of the kernel" if stats_iso is not None
                                    else "K steps under the conditions of the timed region, dispatch events of the kernel",
                        "in_timed_region": dict(timed, frames_in_flight=F),
                        "one_frame_at_a_time_panoramas_per_s": alone_rate,
                        "one_frame_at_a_time_stage_us": alone_stage}
            # also against a measured device-copy ceiling - a 256 MB device-to-device copy (bytes read + written)
            try:
                x = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
                y = torch.empty_like(x)
                for _ in range(3):
                    y.copy_(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    y.copy_(x)
                e1.record()
                torch.cuda.synchronize()
                copy_gbs = 2 * x.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                roofline["device_copy_ceiling_GBps"] = round(copy_gbs, 1)
                roofline["frac_of_copy_ceiling"] = round(alone["achieved"] / copy_gbs, 4)
                del x, y
            except Exception:  # the side measurement never breaks the line
                pass
        if world > 1 and stage_n[0]:
            # N > 1: rank 0's own K1 launches (its cameras of each group, one launch per group it feeds), taken from the
            # event pass; the algorithmic bytes are its cameras' share of the group's
            fed = [bin(pl["bits"]).count("1") for pl in plans if pl["bits"]]
            if fed:
                alg_rank0 = (src_b + dst_b) * fed[0] // NC
                avg_ms = stage_ms[0] / stage_n[0]
                ach = alg_rank0 / (avg_ms * 1e-3) / 1e9
                roofline = {"kernel": "warp_tiles_lut_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                            "algorithmic_bytes_per_launch": alg_rank0, "avg_launch_us": round(avg_ms * 1e3, 2),
                            "launches_per_step": len(fed),
                            "measured": "rank 0, K steps of the sharded path, dispatch events of its %d-camera launches" % fed[0]}
        result = {
            "metric": "stitched panoramas/sec (8x1080p->pano)", "value": round((args.steps if sharded_failed is None else world * args.steps) / dt, 2),
            "unit": "panoramas/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if sharded_failed is None else "weak", "vs_baseline": None, "dtype": "u8/int16 fixed-point (f32 weights)",
            "data": "synthetic",
            "config": {"workload": "C2: 8x1920x1080 BGR8 -> 2 groups x 4 cameras, spherical warp + %d-band "
                                   "multi-band blend, Voronoi seams, pano 2 x %dx%d" % (args.bands, ow, oh),
                       "parallelism": ("single GPU, %d frames in flight" % F) if world == 1 else
                                      ("cameras sharded %d/rank, %s" % (per_rank, "gloo rehearsal on one GPU, slots staged through the host"
                                                                          if rehearsal else "RCCL gather to rank 0")) if sharded_failed is None else
                                      ("replicas, one rig per GPU (camera-sharded path failed: %s)" % sharded_failed)},
            "roofline": roofline,
            "ms_per_step_event_pass": round(dt_profiled / args.steps * 1e3, 4),
            "replicas_panoramas_per_s": replicas_rate,
            "stage_us_per_launch": {k: round(stage_ms[i] / max(stage_n[i], 1) * 1e3, 2)
                                    for i, k in enumerate(("warp", "pyramid", "blend"))},
        }
        if world == 1 and not args.no_host_paths:
            # the reference-shaped entry (host cv::Mat in, host cv::Mat out; H2D + compose + D2H, synchronous):
            # reported for DESIGN.md, never the `value`
            for c in ctxs:
                c.set_profiling(False)
                c.select_frame_slot(0)
            hframes = [[f.cpu().numpy() for f in fr] for fr in frames]
            houts = [np.empty((oh, ow, 3), np.uint8) for _ in range(NG)]
            import threading

            def host_rate(fr, out, reps):
                # the reference calls process() of its two stitchers from two threads per frame and joins
                # (src/master.cpp:314-318): two persistent threads meeting at a barrier after every frame; ctypes
                # releases the GIL inside the call
                bar = threading.Barrier(NG + 1)

                def run(grp):
                    for _ in range(3 + reps):
                        ctxs[grp].compose_host(fr[grp], out=out[grp])
                        bar.wait()
                th = [threading.Thread(target=run, args=(grp,)) for grp in range(NG)]
                [t.start() for t in th]
                for _ in range(3):
                    bar.wait()
                t0h = time.perf_counter()
                for _ in range(reps):
                    bar.wait()
                dth = time.perf_counter() - t0h
                [t.join() for t in th]
                return round(reps / dth, 1)

            result["host_buffer_path_panoramas_per_s"] = host_rate(hframes, houts, 100)
            try:   # the same entry with page-locked caller memory (pano_host_alloc): no staging copy
                pin = [[pano.HostBuffer((H, W, 3)) for _ in range(NC)] for _ in range(NG)]
                pout = [pano.HostBuffer((oh, ow, 3)) for _ in range(NG)]
                for grp in range(NG):
                    for i in range(NC):
                        pin[grp][i].array[:] = hframes[grp][i]
                result["host_buffer_path_pinned_panoramas_per_s"] = host_rate([[b.array for b in g_] for g_ in pin], [b.array for b in pout], 100)
                for b in [x for g_ in pin for x in g_] + pout:
                    b.close()
            except Exception as exc:  # noqa: BLE001
                result["host_buffer_path_pinned_panoramas_per_s"] = None
            # streaming form (BASELINE config 5): frames land in the library's pinned slots, two panoramas in flight
            for grp in range(NG):
                for s in range(2):
                    for i in range(NC):
                        ctxs[grp].stream_input(s, i)[:] = hframes[grp][i]
            nstream = 60
            ts = time.perf_counter()
            for k in range(nstream + 1):
                s = k & 1
                if k >= 1:
                    for grp in range(NG):
                        ctxs[grp].stream_wait(1 - s)
                if k < nstream:
                    for grp in range(NG):
                        ctxs[grp].stream_submit(s)
            result["host_streaming_panoramas_per_s"] = round(nstream / (time.perf_counter() - ts), 1)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ctxs, g, args.bands)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(ctxs, g, bands):
    """the CPU oracle (checker infrastructure, used here only as the timed CPU leg) on a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pano_oracle as po
    from helpers import synth_frame
    W, H = g["w"], g["h"]
    threads = min(os.cpu_count() or 1, 16)
    po.set_threads(threads)
    masks = [ctxs[0].get_mask(i) for i in range(4)]
    frames = [synth_frame(W, H, 42 + i) for i in range(4)]
    t0 = time.perf_counter()
    reps = 0
    stage = [0.0, 0.0, 0.0]
    while True:
        # one panorama = two groups
        for _ in range(2):
            _, ms = po.compose(frames, g["K"], g["R"], g["scale"], masks, bands)
            stage = [a + b for a, b in zip(stage, ms)]
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 20:
            break
    po.set_threads(1)
    return {"value": round(reps / el, 3), "unit": "panoramas/s", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": "%d panoramas of the same C2 workload (8x1080p, 2 groups, %d bands), OpenMP over rows; "
                      "stage ms/pano warp %.0f feed %.0f blend %.0f" % (reps, bands, stage[0] / reps, stage[1] / reps, stage[2] / reps)}


if __name__ == "__main__":
    main()

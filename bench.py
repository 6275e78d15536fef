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

At N = 1 the K steps are dealt round-robin over --frames-in-flight frame slots / streams (default 3): one frame is a
chain of nine dependent launches, several too small to fill the GPU, and the chains of consecutive frames overlap the
way they do behind a capture loop (round 4, us per frame: 1 -> 110, 2 -> 62-63, 3 -> 58-61, 4 -> 62-64: since the panorama is stored
with the non-temporal hint three frames in flight beat four; before it they were level).
--frames-in-flight 1 composes one frame at a time.

The driver runs `--steps 20 --warmup 5`: 1.5 ms of GPU time, which straight after set-up finds the device at its idle clocks
(every kernel 8-10 % slower than in a loop that has been running).  SURVEY 8(d) defines the metric at steady state, so the bench
composes for --preheat seconds (default 0.25), untimed, in front of the W warm-up steps; the same W + K steps timed from the idle
device are in the line too (`from_idle`).  The timed region is exactly K steps either way.

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
    ap.add_argument("--preheat", type=float, default=0.25,
                    help="seconds of untimed composing BEFORE the W warm-up steps, so that the K timed steps (1.5 ms of GPU time at the "
                         "driver's K = 20) find the device at its running clocks, as a capture loop does; 0: none.  The same W + K steps "
                         "timed from the idle device are reported beside `value` as `from_idle`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-paths", action="store_true",
                    help="skip the PCIe-inclusive side measurements (host-buffer and streaming entries); used under rocprofv3 so that "
                         "the kernel summary covers the device-resident loop only")
    ap.add_argument("--bands", type=int, default=5)
    ap.add_argument("--one-stream", action="store_true", help="both groups on one stream (no overlap)")
    ap.add_argument("--force-sharded-path", action="store_true", help="diagnostic: run the N>1 step code at N=1")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="N=1: frame slots / streams the K steps are dealt over round-robin (1 = one frame at a time)")
    ap.add_argument("--cold-only", action="store_true",
                    help="N=1: ONLY K steps one frame at a time over six rotating frame sets (the roofline.cold measurement), for a "
                         "rocprofv3 --kernel-trace --stats summary of exactly that loop; prints the cold object alone")
    ap.add_argument("--no-c4", action="store_true", help="N=1: skip the config-4 leg (4 x 4K, cylindrical, gain maps, 7 bands: the `c4` object)")
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
        import datetime
        # a bounded timeout: a peer that never posts its half of a transfer ends the run with an error instead of a hang
        tmo = datetime.timedelta(seconds=180)
        if rehearsal:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)

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
    flight_distinct = 1
    outs_f = [outs]
    if F > 1:
        for c in ctxs:
            c.set_frame_slots(F)
        # the library's own flight streams, probed to sit on DISTINCT hardware queues: the HIP runtime multiplexes a process's
        # streams onto GPU_MAX_HW_QUEUES (default 4) queues in creation order, and two flight streams on one queue run their
        # frames one after the other (-15 %: docs/EXPERIMENTS.md, round 4) - torch.cuda.Stream()s land where luck puts them
        flight, flight_distinct = ctxs[0].frame_streams(F)
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

    # the exchange of the sharded step: "cabi" = pano_gather_slots (RCCL behind the C-ABI, a communicator of the library's own),
    # "torch" = the same transfers as torch.distributed batch_isend_irecv (RCCL through PyTorch; gloo in the rehearsal)
    exchange = {"kind": "torch"}
    owners = [sh.owner_ranks(NG * NC, NC, world, grp) for grp in range(NG)]

    gather_events = []   # (start, stop) torch events around rank 0's side of the exchange, filled in the event pass only

    def step_sharded(k=0, timed_exchange=False):
        for grp, plan in enumerate(plans):
            if plan["bits"]:
                ctxs[grp].feed_cameras(plan["bits"], fptr[grp], strides, stream)
            buf, slot = slot_views[grp]
            if rehearsal:
                torch.cuda.synchronize()
            if plan["moves"]:
                ev = None
                if timed_exchange and rank == 0 and not rehearsal:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()       # torch's current stream IS the launch stream (set_stream above)
                if exchange["kind"] == "cabi":
                    ctxs[grp].gather_slots(exchange["comm"], rank, 0, owners[grp], stream)   # RCCL group on the launch stream
                else:
                    sh.exchange_slots(dist, rank, buf, slot, plan["moves"], via_host=rehearsal)
                if ev is not None:
                    ev[1].record()
                    gather_events.append(ev)
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
    sharded_failed = None
    if world > 1:
        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        # (1) every rank proves its LOCAL half first - feed its cameras, and rank 0 composes the whole rig by itself as the
        # reference of the check below - and the ranks vote BEFORE any point-to-point traffic, so that a rank that cannot even
        # launch kernels does not leave its peers waiting in a receive
        ok, why = True, ""
        ref = None
        try:
            for grp, plan in enumerate(plans):
                if plan["bits"]:
                    ctxs[grp].feed_cameras(plan["bits"], fptr[grp], strides, stream)
            if rank == 0:
                step_single(0)
                torch.cuda.synchronize()
                ref = [o.clone() for o in outs]
                for o in outs:
                    o.zero_()
                for buf, _ in slot_views:   # ... and forget the other ranks' cameras again: the trial has to bring them
                    buf.zero_()
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            ok, why = False, repr(exc)[:200]
        if not agree(ok):
            raise SystemExit("bench.py: a rank cannot run its local part of the sharded step (%s)" % (why or "another rank"))
        # (2) the communicator of the C-ABI exchange (collective; in the gloo rehearsal, where the ranks share one GPU, only when
        # PANO_RCCL_LIB names a library that accepts that - the test double of tests/src/fake_rccl.cpp; RCCL itself does not)
        if (not rehearsal or os.environ.get("PANO_RCCL_LIB")) and not os.environ.get("PANO_BENCH_EXCHANGE") == "torch":
            ok = True
            try:
                uid = [pano.Context.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                exchange["comm"] = ctxs[0].rccl_comm_create(uid[0], world, rank)
            except Exception as exc:  # noqa: BLE001
                ok, why = False, repr(exc)[:200]
            if agree(ok):
                exchange["kind"] = "cabi"
        # (3) one trial step per candidate exchange, checked on rank 0 against its own whole-rig panorama (every rank holds the
        # same synthetic frames, so rank 0 can compose the reference alone); the first exchange that gives the right bytes on
        # every rank is timed.  None -> exit non-zero: a broken sharded path must not print a number
        tried = []
        while True:
            ok, why = True, ""
            tried.append(exchange["kind"])
            try:
                step_sharded(0)
                # a peer that never posts its half would leave this rank's stream stuck for good: poll instead of blocking,
                # and end the run with an error line rather than in the driver's timeout
                done = torch.cuda.Event()
                done.record()
                t_wait = time.perf_counter()
                while not done.query():
                    if time.perf_counter() - t_wait > 120.0:
                        if rank == 0:
                            print(json.dumps({"metric": "stitched panoramas/sec (8x1080p->pano)", "value": None, "n_gpus": world,
                                              "error": "the sharded trial step (%s exchange) did not complete within 120 s" % exchange["kind"]}), flush=True)
                        os._exit(3)
                    time.sleep(0.005)
                if rank == 0 and not all(bool(torch.equal(a, b)) for a, b in zip(outs, ref)):
                    ok, why = False, "sharded panorama differs from the single-GPU panorama"
            except Exception as exc:  # noqa: BLE001
                ok, why = False, repr(exc)[:200]
            if agree(ok):
                break
            if exchange["kind"] == "cabi":
                exchange["kind"] = "torch"
                continue
            sharded_failed = why or "another rank failed"
            break
        if sharded_failed is not None:
            if rank == 0:
                print(json.dumps({"metric": "stitched panoramas/sec (8x1080p->pano)", "value": None, "n_gpus": world,
                                  "error": "camera-sharded step failed with every exchange tried %s: %s" % (tried, sharded_failed)}), flush=True)
            raise SystemExit(3)

    if args.cold_only and world == 1:
        cold_sets = 6
        rot = [[[t.clone() for t in fr] for fr in frames] for _ in range(cold_sets)]
        rot_ptr = [[[t.data_ptr() for t in fr] for fr in st] for st in rot]
        for c in ctxs:
            c.select_frame_slot(0)
            c.set_profiling(True)
        for k in range(args.warmup + args.steps):
            if k == args.warmup:
                torch.cuda.synchronize()
                for c in ctxs:
                    c.stage_stats(reset=True)
            fp = rot_ptr[k % cold_sets]
            ctxs[0].compose_pair(ctxs[1], fp[0], strides, outs[0].data_ptr(), ow * 3, fp[1], strides, outs[1].data_ptr(), ow * 3, stream)
        torch.cuda.synchronize()
        ms, n = ctxs[0].stage_stats(reset=True)
        src_b, dst_b = ctxs[0].warp_bytes()
        alg = (src_b + dst_b) * NG
        us = ms[0] / n[0] * 1e3
        print(json.dumps({"roofline_cold": {"kernel": "warp_tiles_lut_kernel", "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": alg,
                                            "achieved": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4),
                                            "frame_sets": cold_sets, "steps": args.steps}}), flush=True)
        return
    def timed_steps():
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
        return time.perf_counter() - t0

    # The driver's K = 20 steps are 1.5 ms of GPU time: straight after set-up the device is still at its idle clocks and every
    # kernel of the first steps runs 8-10 % slower than in a loop that has been running (K1 20.0 against 18.4 us).  SURVEY 8(d)
    # defines the metric at steady state, so: (1) the W + K steps from the idle device, reported as `from_idle`; (2) --preheat
    # seconds of the same composing, untimed, every rank by itself (no exchange, so ranks cannot wait for each other);
    # (3) the W warm-up steps and the K timed steps of the contract: `value`.
    from_idle = None
    if args.preheat > 0:
        dt_idle = timed_steps()
        if world > 1:
            ti_ = torch.tensor([dt_idle], dtype=torch.float64, device=red_dev)
            dist.all_reduce(ti_, op=dist.ReduceOp.MAX)
            dt_idle = float(ti_.item())
        from_idle = {"value": round(args.steps / dt_idle, 2), "ms_per_step": round(dt_idle / args.steps * 1e3, 4)}
        tph, kph = time.perf_counter(), 0
        while time.perf_counter() - tph < args.preheat:
            for _ in range(32):
                step_single(kph)
                kph += 1
            torch.cuda.synchronize()
    dt = timed_steps()
    # second pass over the same K steps with HIP events on the launch stream (direct launches): per-stage and
    # K1 kernel durations for the roofline; not part of `value`
    for c in ctxs:
        c.set_profiling(True)
        c.stage_stats(reset=True)
    t1 = time.perf_counter()
    for k in range(args.steps):
        if world > 1:
            step_sharded(k, timed_exchange=True)
        else:
            step(k)
    torch.cuda.synchronize()
    dt_profiled = time.perf_counter() - t1
    stats_timed = None
    if world == 1:
        stats_timed = [c.stage_stats(reset=True) for c in ctxs]
    # optional: the same K steps one frame at a time on one stream (kernel durations without overlap)
    stats_iso, dt_iso = None, None
    if world == 1 and F > 1 and not args.no_isolated_pass:
        # (a few untimed launches first: this pass measures the kernels, and at the driver's K = 20 the first launches after the
        # switch from four streams to one are a visible share of the mean)
        for k in range(min(10, args.steps)):
            step_single(k, slots=1)
        torch.cuda.synchronize()
        for c in ctxs:
            c.stage_stats(reset=True)
        ti = time.perf_counter()
        for k in range(args.steps):
            step_single(k, slots=1)
        torch.cuda.synchronize()
        dt_iso = time.perf_counter() - ti
        stats_iso = [c.stage_stats(reset=True) for c in ctxs]
    # cold pass (N = 1): the same one-frame-at-a-time loop over SIX rotating frame sets (6 x 49.8 MB of frames + 17.5 MB of
    # remap table > the 256 MiB Infinity Cache), so that no K1 launch finds its inputs where the previous launch left them
    stats_cold, cold_sets, rotating_rate = None, 6, None
    if world == 1 and not args.no_isolated_pass:
        rot = [[[t.clone() for t in fr] for fr in frames] for _ in range(cold_sets)]
        rot_ptr = [[[t.data_ptr() for t in fr] for fr in st] for st in rot]
        for c in ctxs:
            c.select_frame_slot(0)
        for k in range(cold_sets):  # one untimed pass over the sets (the sets themselves stay cold: six of them exceed the cache)
            fp = rot_ptr[k % cold_sets]
            ctxs[0].compose_pair(ctxs[1], fp[0], strides, outs[0].data_ptr(), ow * 3, fp[1], strides, outs[1].data_ptr(), ow * 3, stream)
        torch.cuda.synchronize()
        for c in ctxs:
            c.stage_stats(reset=True)
        for k in range(args.steps):
            fp = rot_ptr[k % cold_sets]
            ctxs[0].compose_pair(ctxs[1], fp[0], strides, outs[0].data_ptr(), ow * 3, fp[1], strides, outs[1].data_ptr(), ow * 3, stream)
        torch.cuda.synchronize()
        stats_cold = [c.stage_stats(reset=True) for c in ctxs]
        # ... and the timed loop's own shape (F frames in flight, no events) over the same rotating sets: what `value` would be
        # if no step found its frames where an earlier step left them
        rotating_rate = None
        if F > 1:
            for c in ctxs:
                c.set_profiling(False)
            def step_rot(k):
                f = k % F
                ctxs[0].select_frame_slot(f); ctxs[1].select_frame_slot(f)
                fp = rot_ptr[k % cold_sets]
                o = outs_f[f]
                ctxs[0].compose_pair(ctxs[1], fp[0], strides, o[0].data_ptr(), ow * 3, fp[1], strides, o[1].data_ptr(), ow * 3, flight[f])
            for k in range(args.warmup):
                step_rot(k)
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            for k in range(args.steps):
                step_rot(k)
            torch.cuda.synchronize()
            rotating_rate = round(args.steps / (time.perf_counter() - tr0), 1)
            for c in ctxs:
                c.select_frame_slot(0)
        del rot, rot_ptr
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
    per_rank_warp = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # SURVEY 8(e) "report": every rank's own K1 rate (its cameras' share of the algorithmic bytes / its mean launch time)
        try:
            ms_n = [c.stage_stats(reset=False) for c in ctxs]
            mine = [sum(m[0][0] for m in ms_n), sum(m[1][0] for m in ms_n)]
            allv = [None] * world
            dist.all_gather_object(allv, mine)
            per_rank_warp = allv
        except Exception:  # noqa: BLE001
            per_rank_warp = None

    result = None
    if rank == 0:
        src_b, dst_b = ctxs[0].warp_bytes()
        # per K1 launch: at N=1 one launch warps all 8 cameras (both stitchers), when sharded one group of 4
        launches_per_step = 1 if world == 1 and not args.force_sharded_path else NG
        alg_bytes = (src_b + dst_b) * (NG // launches_per_step)
        def fold(stats):
            ms3, n3 = [0.0] * 4, [0] * 4
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
            # HBM traffic of the K1 launch from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/warp_traffic.json): a figure of
            # the build it was collected on.  It is printed only when that build's kernel sources are the ones this library was made
            # from (pano_kernel_source_id) and the run is the profiled configuration; otherwise null, with the reason beside it
            traffic, traffic_note = None, None
            kid = pano.kernel_source_id()
            tp = os.path.join(ROOT, "profiles", "warp_traffic.json")
            if os.path.exists(tp):
                try:
                    tj = json.load(open(tp))
                    cur = tj.get("current", {})
                    if cur.get("kernel_source_id") != kid:
                        traffic_note = "profiles/warp_traffic.json was collected on kernel sources %s, this library is %s: re-run tools/pmc_traffic_all.sh" % (cur.get("kernel_source_id"), kid)
                    elif args.bands != 5 or (W, H) != (1920, 1080):
                        traffic_note = "the counters were collected on the default configuration (5 bands, 1920x1080)"
                    else:
                        traffic = cur.get("hbm_bytes_per_launch_8cam")
                except Exception:
                    traffic = None
            # the kernel's roofline fraction is taken where its launches have the GPU to themselves (one frame at a
            # time); what a launch takes while it shares the GPU with the other frame in flight is reported beside it
            timed = k1_roofline(stage_ms, stage_n)
            alone, alone_stage, alone_rate = timed, None, None
            if stats_iso is not None:
                ims, inn = fold(stats_iso)
                alone = k1_roofline(ims, inn)
                alone_stage = {k: round(ims[i] / max(inn[i], 1) * 1e3, 2) for i, k in enumerate(("warp", "pyramid", "blend", "blend_level0"))}
                alone_rate = round(args.steps / dt_iso, 1)
            # level-0 blend launch (the largest kernel of a frame): algorithmic bytes per canvas pixel = 3 (level-0 tile of the
            # owning camera) + 0.75 (its level 1) + 1.5 (canvas level 1, int16) + 0.25 (owner map) read, 3 written
            blend0 = None
            src_stats = stats_iso if stats_iso is not None else stats_timed
            bms, bn = fold(src_stats)
            if bn[3]:
                b0_bytes = int(8.5 * ow * oh * NG)
                b0_ms = bms[3] / bn[3]
                b0_gbs = b0_bytes / (b0_ms * 1e-3) / 1e9
                blend0 = {"kernel": "blend_level_ordered_kernel<true,3>", "bound": "hbm", "algorithmic_bytes_per_launch": b0_bytes,
                          "avg_launch_us": round(b0_ms * 1e3, 2), "achieved": round(b0_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(b0_gbs / HBM_PEAK_GBS, 4)}
            cold = None
            if stats_cold is not None:
                cms, cn = fold(stats_cold)
                if cn[0]:
                    cold = dict(k1_roofline(cms, cn), frame_sets=cold_sets,
                                bytes_rotated=int(cold_sets * NG * NC * W * H * 3),
                                note="one frame at a time over rotating frame sets larger than the 256 MiB Infinity Cache")
            # the headline fraction is the COLD one: the warm loop re-reads the same 49.8 MB of frames every step
            # with a 256 MiB Infinity Cache in front of HBM; `warm` keeps the figure earlier rounds led with
            warm_how = ("K steps, one frame at a time, the same frames every step, dispatch events of the kernel" if stats_iso is not None
                        else "K steps under the conditions of the timed region, dispatch events of the kernel")
            lead = cold if cold is not None else alone
            roofline = {"kernel": "warp_tiles_lut_kernel", "bound": "hbm", "achieved": lead["achieved"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lead["frac"],
                        "traffic": traffic, "traffic_note": traffic_note, "kernel_source_id": kid, "algorithmic_bytes_per_launch": alg_bytes,
                        "avg_launch_us": lead["avg_launch_us"], "launches_per_step": launches_per_step,
                        "measured": ("K steps, one frame at a time over %d rotating frame sets (larger than the 256 MiB Infinity Cache: no "
                                     "launch finds its frames where an earlier one left them), dispatch events of the kernel" % cold_sets)
                                    if cold is not None else warm_how,
                        "warm": dict(alone, measured=warm_how),
                        "in_timed_region": dict(timed, frames_in_flight=F),
                        "one_frame_at_a_time_panoramas_per_s": alone_rate,
                        "one_frame_at_a_time_stage_us": alone_stage,
                        "cold": cold, "blend_level0": blend0}
            # also against measured device-copy ceilings - the library's own probes (pano_probe_copy), with
            # the launch and timing machinery of its kernels: a float4 grid-stride copy of 512 MiB (what the part streams), and a copy
            # with K1's traffic SHAPE and SIZE (direct-to-LDS box + table in, one dword per lane and plane out, as many workgroups as K1
            # has live patches) warm and over rotating buffers - what a kernel launched like K1 can reach at all
            try:
                C_ = pano.Context
                f4 = max(ctxs[0].probe_copy(C_.PROBE_COPY_F4, 512 << 20, sets=1, reps=20)["GBps"],
                         ctxs[0].probe_copy(C_.PROBE_COPY_F4_FLAT, 512 << 20, sets=1, reps=20)["GBps"])
                nwg = int(sum(c.warp_table_stats()["blocks"] for c in ctxs) * 0.70) or 9300   # live share of the 64 x 16 patches (config 2: 70 %)
                per = nwg * (4608 + 2048 + 3072)
                kw = ctxs[0].probe_copy(C_.PROBE_COPY_K1_SHAPE, nwg, sets=1, reps=100)
                kc = ctxs[0].probe_copy(C_.PROBE_COPY_K1_SHAPE, nwg, sets=max(2, -(-(320 << 20) // per)), reps=100)
                copy_gbs = f4
                roofline["device_copy_ceiling_GBps"] = round(f4, 1)
                roofline["frac_of_copy_ceiling"] = round(lead["achieved"] / f4, 4)
                roofline["k1_shaped_copy"] = {
                    "workgroups": nwg, "bytes_per_launch": int(kw["bytes_per_launch"]),
                    "warm": {"us_per_launch": round(kw["us_per_launch"], 2), "GBps": round(kw["GBps"], 1)},
                    "cold": {"us_per_launch": round(kc["us_per_launch"], 2), "GBps": round(kc["GBps"], 1)},
                    "k1_us_over_copy_us": {"warm": round(alone["avg_launch_us"] / kw["us_per_launch"], 3),
                                           "cold": round(lead["avg_launch_us"] / kc["us_per_launch"], 3) if cold is not None else None},
                    "note": "a copy with K1's traffic shape, size and launch shape and none of its arithmetic (pano_probe.hip): the time K1 "
                            "cannot beat without moving fewer bytes"}
                # the whole frame in the timed region: every kernel's HBM traffic (PMC passes, profiles/) over the time of a step - only
                # for the configuration and the kernel sources the counters were collected on
                fp = os.path.join(ROOT, "profiles", "hbm_bytes_per_kernel_per_frame.json")
                if os.path.exists(fp) and args.bands == 5 and (W, H) == (1920, 1080):
                    fj = json.load(open(fp))
                    fb = float(fj.get("_total_hbm_MB_per_frame", 0.0)) * 1e6
                    if fb > 0 and fj.get("kernel_source_id") == kid:
                        fr_gbs = fb / (dt / args.steps) / 1e9
                        roofline["frame_in_timed_region"] = {
                            "hbm_traffic_bytes_per_step": int(fb), "us_per_step": round(dt / args.steps * 1e6, 2), "achieved_GBps": round(fr_gbs, 1),
                            "frac_of_peak": round(fr_gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(fr_gbs / copy_gbs, 4),
                            "algorithmic_bytes_per_step": int(NG * NC * W * H * 3 + NG * ow * oh * 3),
                            "note": "traffic = FETCH_SIZE x 2 + WRITE_SIZE of all launches of a frame, separate --pmc passes (profiles/hbm_bytes_per_kernel_per_frame.json, "
                                    "same kernel sources, same configuration); algorithmic = the frames in + the panoramas out"}
            except Exception as exc:  # the side measurement never breaks the line
                roofline["device_copy_ceiling_error"] = repr(exc)[:200]
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
            "metric": "stitched panoramas/sec (8x1080p->pano)", "value": round(args.steps / dt, 2),
            "unit": "panoramas/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u8/int16 fixed-point (f32 weights)",
            "data": "synthetic",
            "config": {"workload": "C2: 8x1920x1080 BGR8 frames resident in HBM, THE SAME frames every step (device pointers in, device panoramas "
                                   "out: no PCIe in the timed region; over rotating frame sets the rate is `rotating_inputs_panoramas_per_s`%s, with the "
                                   "host link in the loop `h2d_inclusive`) -> 2 groups x 4 cameras, spherical warp + %d-band "
                                   "multi-band blend, Voronoi seams, pano 2 x %dx%d" % ((" = %.0f" % rotating_rate) if rotating_rate else "", args.bands, ow, oh),
                       "parallelism": ("single GPU, %d frames in flight on %d distinct hardware queues (pano_frame_streams)" % (F, flight_distinct)) if world == 1 else
                                      ("cameras sharded %d/rank, %s" % (per_rank,
                                                                          "every rank composes whole stitchers; the finished half panoramas move to rank 0 (torch.distributed send / recv)"
                                                                          if not any(pl["moves"] for pl in plans) else
                                                                          "gloo rehearsal on one GPU, slots staged through the host"
                                                                          if rehearsal and exchange["kind"] != "cabi" else
                                                                          ("RCCL gather to rank 0 through the C-ABI (pano_gather_slots)" if exchange["kind"] == "cabi"
                                                                           else "RCCL gather to rank 0 (torch.distributed batch_isend_irecv)")))},
            "roofline": roofline,
            "preheat_s": args.preheat,   # untimed composing in front of the W warm-up steps (see --preheat)
            "from_idle": from_idle,      # the same W + K steps timed straight after set-up, the device at its idle clocks
            "ms_per_step_event_pass": round(dt_profiled / args.steps * 1e3, 4),
            "rotating_inputs_panoramas_per_s": rotating_rate,
            "replicas_panoramas_per_s": replicas_rate,
            "multi_gpu": None,
            "stage_us_per_launch": {k: round(stage_ms[i] / max(stage_n[i], 1) * 1e3, 2)
                                    for i, k in enumerate(("warp", "pyramid", "blend", "blend_level0"))},
        }
        if world > 1:
            mg = {"exchange": exchange["kind"] if any(pl["moves"] for pl in plans) else "none (a rank owns whole stitchers; finished half panoramas move)"}
            if exchange.get("comm") is not None:
                # what RCCL itself says about the communicator the exchange ran on (ncclCommCount), and the library behind it
                try:
                    mg["rccl_ranks"] = ctxs[0].rccl_comm_count(exchange["comm"])
                    mg["rccl_library"] = pano.Context.rccl_library()
                except Exception:  # noqa: BLE001
                    mg["rccl_ranks"] = None
            if gather_events:
                us = [a.elapsed_time(b) * 1e3 for a, b in gather_events]
                if exchange["kind"] == "cabi":
                    # the C-ABI exchange moves the live rectangles of every level, packed (pano_get_exchange_stats), not whole slots
                    recv_bytes = 0
                    for grp, pl in enumerate(plans):
                        pk = ctxs[grp].exchange_stats()["packed_bytes_per_camera"]
                        recv_bytes += sum(sum(pk[first:first + n_]) for (_, first, n_) in pl["moves"])
                else:
                    recv_bytes = sum(cnt for pl, (_, slot) in zip(plans, slot_views) for (_, _, n_) in pl["moves"] for cnt in [n_ * slot])
                senders = sum(len(pl["moves"]) for pl in plans)
                mean_us = sum(us) / len(us)
                per_group = recv_bytes / max(1, sum(1 for pl in plans if pl["moves"]))
                mg.update({"gather_us_mean": round(mean_us, 1), "gather_bytes_per_group": int(per_group), "senders_per_step": senders,
                           "root_ingress_GBps": round(per_group / mean_us / 1e3, 2),
                           "whole_slot_bytes_per_group": int(sum(n_ * slot for pl, (_, slot) in zip(plans, slot_views) for (_, _, n_) in pl["moves"]) /
                                                             max(1, sum(1 for pl in plans if pl["moves"]))),
                           "note": "events on rank 0's launch stream around its side of the exchange (one ncclGroup per stitcher); "
                                   "includes waiting for the senders' warps"})
            if per_rank_warp:
                launch_bytes = (src_b + dst_b) / NC * min(per_rank, NC)   # the cameras of ONE K1 launch of a rank (one stitcher's share)
                mg["per_rank_warp_GBps"] = [round(launch_bytes / (m / n * 1e-3) / 1e9, 1) if n else None for m, n in per_rank_warp]
            result["multi_gpu"] = mg
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
            nstream = max(300, args.steps)
            ts = time.perf_counter()
            for k in range(nstream + 1):
                s = k & 1
                if k < nstream:          # panorama k goes into slot s (panorama k - 2, its last user, was waited for at k - 1) ...
                    for grp in range(NG):
                        ctxs[grp].stream_submit(s)
                if k >= 1:               # ... and only then panorama k - 1 is waited for: two panoramas in flight
                    for grp in range(NG):
                        ctxs[grp].stream_wait(1 - s)
            rate = round(nstream / (time.perf_counter() - ts), 1)
            # "pano_compose incl. H2D of inputs ... >= 300 frames": the streaming entry (page-locked slots, H2D,
            # compose and D2H of consecutive panoramas overlapped); PCIe-inclusive, so never `value`
            result["h2d_inclusive_panoramas_per_s"] = rate
            result["host_streaming_panoramas_per_s"] = rate
            # ... and its own roofline: this path is bound by the host link, not by HBM.  Bytes per panorama pair that cross it: up =
            # the live source rectangles of the 8 frames (pano_get_source_rect), down = the two panoramas.  Ceiling = plain page-locked
            # hipMemcpyAsync of 64 MiB blocks, both directions at once on two streams (what the link gives a copy loop on this box);
            # frac = the time the link needs for one step's bytes at those rates (full duplex: the longer direction) / the step time
            try:
                up_b = sum(r[2] * r[3] for c in ctxs for r in (c.source_rect(i) for i in range(NC)))
                down_b = NG * ow * oh * 3
                nb = 64 << 20
                hp_up, hp_dn = torch.empty(nb, dtype=torch.uint8).pin_memory(), torch.empty(nb, dtype=torch.uint8).pin_memory()
                d_up, d_dn = torch.empty(nb, dtype=torch.uint8, device="cuda"), torch.empty(nb, dtype=torch.uint8, device="cuda")
                s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
                torch.cuda.synchronize()

                def both(reps):
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    with torch.cuda.stream(s_up):
                        ev[0].record()
                        for _ in range(reps):
                            d_up.copy_(hp_up, non_blocking=True)
                        ev[1].record()
                    with torch.cuda.stream(s_dn):
                        ev[2].record()
                        for _ in range(reps):
                            hp_dn.copy_(d_dn, non_blocking=True)
                        ev[3].record()
                    torch.cuda.synchronize()
                    return nb * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, nb * reps / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9
                both(2)
                c_up = c_dn = 0.0
                for _ in range(3):   # the best of three: a ceiling, and one hiccup of the host would put the fraction above 1
                    u_, d_ = both(12)
                    c_up, c_dn = max(c_up, u_), max(c_dn, d_)
                t_link = max(up_b / (c_up * 1e9), down_b / (c_dn * 1e9))
                result["h2d_inclusive"] = {
                    "panoramas_per_s": rate, "bound": "host link (PCIe), full duplex", "up_bytes_per_step": int(up_b), "down_bytes_per_step": int(down_b),
                    "up_GBps": round(rate * up_b / 1e9, 2), "down_GBps": round(rate * down_b / 1e9, 2),
                    "pinned_copy_ceiling_GBps": {"up": round(c_up, 2), "down": round(c_dn, 2),
                                                 "measured": "hipMemcpyAsync of page-locked 64 MiB blocks, 12 each way, both directions at once, best of 3"},
                    "frac": round(t_link * rate, 4),
                    "note": "frames of %d steps land in the library's page-locked slots; H2D, compose and D2H of consecutive steps overlap (pano_stream_*)" % nstream}
                del hp_up, hp_dn, d_up, d_dn
            except Exception as exc:  # noqa: BLE001 - a side measurement never breaks the line
                result["h2d_inclusive"] = {"panoramas_per_s": rate, "error": repr(exc)[:200]}
        if world == 1 and not args.no_c4:
            try:
                result["c4"] = config4_leg(pano, torch, max(60, min(args.steps, 200)))
            except Exception as exc:  # noqa: BLE001 - a side measurement never breaks the contract line
                result["c4"] = {"error": repr(exc)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(ctxs, g, args.bands)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def config4_leg(pano, torch, steps):
    """BASELINE.json configs[3] on the same GPU, through the same machinery: 4 x 3840x2160 BGR8 frames resident in HBM, cylindrical
    warp, exposure gain maps applied in K1 (the compensator apply of src/stitching_detailed.cpp:841), 7-band blend.  Panoramas/s
    with four frames in flight, then `steps` frames one at a time with the dispatch events on: K1 with the gain maps and the
    level-0 blend against the HBM roofline (the kernels run three to four times longer here than on config 2)."""
    from helpers import c4_gain_map, c4_rig, synth_frame
    g = c4_rig()
    W, H, NC, F = g["w"], g["h"], 4, 4
    ctx = pano.Context(NC, W, H, scale=g["scale"], projector=pano.CYLINDRICAL, num_bands=7, device=torch.cuda.current_device())
    for i in range(NC):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare()
    ctx.build_masks_voronoi()
    for i in range(NC):
        r = ctx.roi(i)
        ctx.set_gain_map(i, c4_gain_map((r[2] + 31) // 32, (r[3] + 31) // 32, 7 + i))
    ctx.set_frame_slots(F)
    frames = [torch.from_numpy(synth_frame(W, H, 900 + i)).cuda() for i in range(NC)]
    fp, strides = [t.data_ptr() for t in frames], [W * 3] * NC
    ow, oh = ctx.output_size()
    outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
    streams, _ = ctx.frame_streams(F)   # probed onto distinct hardware queues, like the timed region's

    def step(k):
        f = k % F
        ctx.select_frame_slot(f)
        ctx.compose(fp, strides, outs[f].data_ptr(), ow * 3, streams[f])
    for k in range(12):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.set_profiling(True)
    ctx.select_frame_slot(0)
    torch.cuda.synchronize()
    ctx.stage_stats(True)
    for k in range(steps):
        ctx.compose(fp, strides, outs[0].data_ptr(), ow * 3, streams[0])
    torch.cuda.synchronize()
    ms, n = ctx.stage_stats(True)
    sb, db = ctx.warp_bytes()

    def roof(kernel, nbytes, ms_total, launches):
        us = ms_total / max(launches, 1) * 1e3
        gbs = nbytes / (us * 1e-6) / 1e9
        return {"kernel": kernel, "bound": "hbm", "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_us": round(us, 2),
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    out = {"workload": "C4: 4x3840x2160 BGR8, cylindrical warp + exposure gain maps (32x32-pixel blocks, smooth field in [0.8, 1.25]) + "
                       "7-band blend, Voronoi seams, pano %dx%d" % (ow, oh),
           "panoramas_per_s": round(1.0 / dt, 1), "frames_in_flight": F, "steps": steps,
           "one_frame_at_a_time_stage_us": {k: round(ms[i] / max(n[i], 1) * 1e3, 2) for i, k in enumerate(("warp", "pyramid", "blend", "blend_level0"))},
           "roofline": {"warp_with_gains": roof("warp_tiles_lut_kernel<true>", sb + db, ms[0], n[0]),
                        "blend_level0": roof("blend_level_ordered_kernel<true,3>", 8.5 * ow * oh, ms[3], n[3]),
                        "measured": "%d frames one at a time, dispatch events of the kernels" % steps}}
    del ctx
    return out


def cpu_baseline(ctxs, g, bands):
    """the CPU oracle (checker infrastructure, used here only as the timed CPU leg) on bounded samples of the same
    workload: one thread, 16 threads (the CPU share of one GPU on this pool) and every core of the box"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pano_oracle as po
    from helpers import synth_frame
    W, H = g["w"], g["h"]
    masks = [ctxs[0].get_mask(i) for i in range(4)]
    frames = [synth_frame(W, H, 42 + i) for i in range(4)]
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # "all cores" = what this job may use: a cgroup CPU quota (16 for a one-GPU share of this pool) counts, 256 runnable threads
    # on a 16-CPU quota only measure the scheduler (0.06 panoramas/s when it was tried)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:  # noqa: BLE001
        quota = None
    usable = min(ncpu, quota) if quota else ncpu

    def run(threads, budget_s, max_reps):
        po.set_threads(threads)
        t0 = time.perf_counter()
        reps, stage = 0, [0.0, 0.0, 0.0]
        while True:
            for _ in range(2):   # one panorama = two groups
                _, ms = po.compose(frames, g["K"], g["R"], g["scale"], masks, bands)
                stage = [a + b for a, b in zip(stage, ms)]
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= max_reps:
                break
        return {"value": round(reps / el, 3), "cores": threads, "panoramas": reps,
                "ms_per_panorama": round(el / reps * 1e3, 1),
                "stage_ms_per_panorama": {"warp": round(stage[0] / reps, 1), "feed": round(stage[1] / reps, 1), "blend": round(stage[2] / reps, 1)}}

    # the timed leg runs the build BASELINE.md states (-O3 -march=native, same source, same -ffp-contract=off): a second library,
    # compiled on this host, and only after it has reproduced the checker build's panorama byte for byte
    build_flags = "-O2 (checker build)"
    try:
        po.set_threads(min(usable, 16))
        want, _ = po.compose(frames, g["K"], g["R"], g["scale"], masks, bands)
        po.select_build("timed")
        po.set_threads(min(usable, 16))
        got, _ = po.compose(frames, g["K"], g["R"], g["scale"], masks, bands)
        if np.array_equal(got, want):
            build_flags = "-O3 -march=native -ffp-contract=off -fopenmp (reproduces the checker build's bytes)"
        else:
            po.select_build("check")
            build_flags = "-O2 (checker build; the -O3 -march=native build did NOT reproduce its bytes and was not timed)"
    except Exception as exc:  # noqa: BLE001 - no compiler on the box: time the checker build
        po.select_build("check")
        build_flags = "-O2 (checker build; the timed build failed: %s)" % repr(exc)[:80]
    one = run(1, 8.0, 3)
    share = run(min(usable, 16), 7.0, 20)
    allc = run(min(usable, 64), 7.0, 20) if usable > 16 else share   # beyond 64 threads the row-parallel loops run out of rows
    po.set_threads(1)
    po.select_build("check")
    best = max((share, allc), key=lambda r: r["value"])
    return {"value": best["value"], "unit": "panoramas/s", "cores": best["cores"], "kind": "port", "host_cpus": ncpu, "cpu_quota": quota,
            "build": build_flags,
            "sample": "%d panoramas of the same C2 workload (8x1080p, 2 groups, %d bands), OpenMP over rows; the best of the "
                      "16-thread and all-core runs is `value`" % (best["panoramas"], bands),
            "one_thread": one, "threads_16": share, "all_cores": allc}


if __name__ == "__main__":
    main()

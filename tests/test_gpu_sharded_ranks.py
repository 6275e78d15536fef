"""The camera-sharded compose path with real kernels and several processes: world_size 2 and 4, gloo
backend, every rank on cuda:0 (slot ranges staged through host memory - RCCL needs one GPU per rank, the 8-GPU run is the
driver's).  Each rank warps + builds the pyramids of its cameras (pano_feed_cameras), the slot ranges land on rank 0
(pano_get_pyramid_slots), rank 0 blends (pano_blend): the panoramas must be the single-process ones, bit for bit."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _DevView:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from helpers import c2_group, synth_frame
    pano = importlib.import_module("img-stitching_amd")
    sh = importlib.import_module("img-stitching_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    d = c2_group(w=960, h=540, f=501.2)
    NG, NC = 2, 4
    ctxs, frames, outs = [], [], []
    for g in range(NG):
        ctx = pano.Context(NC, d["w"], d["h"], scale=d["scale"], num_bands=4, device=0)
        for i in range(NC):
            ctx.set_camera(i, d["K"][i], d["R"][i])
        ctx.prepare(); ctx.build_masks_voronoi()
        ctxs.append(ctx)
        frames.append([torch.from_numpy(synth_frame(d["w"], d["h"], 700 + NC * g + i)).cuda() for i in range(NC)])
        ow, oh = ctx.output_size()
        outs.append(torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda"))
    st = torch.cuda.current_stream().cuda_stream
    strides = [d["w"] * 3] * NC
    plans = sh.group_plan(NG * NC, NC, world, rank)
    ok = True
    for rep in range(2):
        for g, plan in enumerate(plans):
            ctx = ctxs[g]
            fp = [t.data_ptr() for t in frames[g]]
            if plan["bits"]:
                ctx.feed_cameras(plan["bits"], fp, strides, st)
            torch.cuda.synchronize()
            base, slot = ctx.pyramid_slots()
            buf = torch.as_tensor(_DevView(base, slot * NC), device="cuda")
            sh.exchange_slots(dist, rank, buf, slot, plan["moves"], via_host=True)
            if plan["blend_here"]:
                ctx.blend(outs[g].data_ptr(), outs[g].shape[1] * 3, st)
                torch.cuda.synchronize()
            if plan["pano_from"] != 0:
                if rank == plan["pano_from"]:
                    dist.send(outs[g].cpu(), dst=0)
                elif rank == 0:
                    host = outs[g].cpu()
                    dist.recv(host, src=plan["pano_from"])
                    outs[g].copy_(host)
        if rank == 0:
            for g in range(NG):  # the same frames through one process
                ref_ctx = pano.Context(NC, d["w"], d["h"], scale=d["scale"], num_bands=4, device=0)
                for i in range(NC):
                    ref_ctx.set_camera(i, d["K"][i], d["R"][i])
                ref_ctx.prepare(); ref_ctx.build_masks_voronoi()
                ref = torch.zeros_like(outs[g])
                ref_ctx.compose([t.data_ptr() for t in frames[g]], strides, ref.data_ptr(), ref.shape[1] * 3, st)
                torch.cuda.synchronize()
                ok &= bool(np.array_equal(ref.cpu().numpy(), outs[g].cpu().numpy()))
                ok &= int(outs[g].max().item()) > 0
        dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_compose_across_processes(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == list(range(world))
    assert all(ok for _, ok in res)


@pytest.mark.gpu
def test_rccl_communicator_of_the_library_on_one_rank():
    """what CAN be checked about the RCCL exchange on a one-GPU box: librccl loads, pano_rccl_unique_id + pano_rccl_comm_create
    make a world-size-1 communicator on the real device, pano_gather_slots runs its ncclGroupStart / End around zero transfers
    (every slot is owned by the root) and the blend that follows still gives the single-GPU panorama"""
    import importlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import c2_group, synth_frame
    pano = importlib.import_module("img-stitching_amd")
    d = c2_group(w=640, h=360, f=334.0)
    ctx = pano.Context(4, 640, 360, scale=d["scale"], num_bands=3, device=0)
    for i in range(4):
        ctx.set_camera(i, d["K"][i], d["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi()
    frames = [synth_frame(640, 360, 90 + i) for i in range(4)]
    want = ctx.compose_host(frames)
    uid = pano.Context.rccl_unique_id()
    try:
        comm = ctx.rccl_comm_create(uid, 1, 0)
    except pano.PanoError as e:
        # ncclCommInitRank itself can fail on a box ("unhandled cuda error" inside librccl, seen once in round 5 on a box where the
        # same tree passed minutes earlier on another): that is the box's RCCL, not this library's exchange - which the RCCL double
        # (tests/test_gpu_rccl_double.py) exercises between real peers either way.  Any OTHER failure is ours and fails the test
        if "unhandled cuda error" in str(e) or "unhandled system error" in str(e):
            pytest.skip("librccl's own ncclCommInitRank failed on this box: " + str(e))
        raise
    try:
        ctx.feed_cameras_host(0b1111, frames)
        ctx.gather_slots(comm, 0, 0, [0, 0, 0, 0])          # stream 0 = the ctx's own stream
        assert np.array_equal(ctx.blend_host(), want)
        with pytest.raises(pano.PanoError):
            ctx.gather_slots(None, 0, 0, [0, 0, 0, 0])      # no communicator
    finally:
        ctx.rccl_comm_destroy(comm)

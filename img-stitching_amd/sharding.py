"""Camera sharding of one panorama over the ranks of a torch.distributed job.

The compose path splits per camera up to the accumulate of MultiBandBlender::feed: warp + Gaussian pyramid
(`pano_feed_cameras`) are per camera, only `pano_blend` needs every camera.  All pyramid slots of a context are
one allocation of equal-sized slots (`pano_get_pyramid_slots`), so a rank's cameras are one contiguous byte
range and the root receives every range in place - the "single gather of the pyramid tiles onto rank 0" of the
north star, expressed as point-to-point RCCL sends because the ranges of different ranks differ in position,
not in size.  With `per_rank >= cams_per_group` a rank owns whole groups and only finished panoramas move.

Pure host logic over torch tensors: the same code runs on gloo/CPU tensors in tests/test_sharding_gloo.py.
"""


def camera_shards(total_cams, world):
    """contiguous camera ranges: rank r owns [r*total/world, (r+1)*total/world)"""
    if world < 1 or total_cams % world:
        raise ValueError("world size must divide the camera count")
    per = total_cams // world
    return [list(range(r * per, (r + 1) * per)) for r in range(world)]


def group_plan(total_cams, cams_per_group, world, rank):
    """what this rank does for each group: (camera bit mask to feed, [(peer, first_slot, n_slots)] to send to
    rank 0 or receive on rank 0, blend_here, panorama_src_rank)"""
    shards = camera_shards(total_cams, world)
    per = len(shards[0])
    plans = []
    for g in range(total_cams // cams_per_group):
        lo, hi = g * cams_per_group, (g + 1) * cams_per_group
        mine = [c - lo for c in shards[rank] if lo <= c < hi]
        bits = sum(1 << c for c in mine)
        owners = [r for r in range(world) if any(lo <= c < hi for c in shards[r])]
        if per >= cams_per_group:            # one rank owns the whole group: blend there, ship the panorama
            owner = owners[0]
            plans.append({"bits": bits, "moves": [], "blend_here": rank == owner, "pano_from": owner})
        else:                                # several ranks per group: their slot ranges go to rank 0
            moves = [(r, shards[r][0] - lo, per) for r in owners if r != 0]
            plans.append({"bits": bits, "moves": moves, "blend_here": rank == 0, "pano_from": 0})
    return plans


def owner_ranks(total_cams, cams_per_group, world, group):
    """for every camera of `group`, the rank that feeds it - the `owner_rank` array of pano_gather_slots"""
    shards = camera_shards(total_cams, world)
    lo = group * cams_per_group
    return [next(r for r in range(world) if (lo + c) in shards[r]) for c in range(cams_per_group)]


def exchange_slots(dist, rank, slot_buffer, slot_bytes, moves, via_host=False):
    """slot_buffer: 1-D uint8 tensor over all slots of the group's context.  Senders push their range to rank 0,
    rank 0 receives each range where it belongs - all transfers of the group posted together (batch_isend_irecv: one
    ncclGroupStart / End on RCCL), not one blocking launch after the other.  The torch.distributed twin of the C-ABI's
    pano_gather_slots.  via_host stages device tensors through host memory, for rehearsing the path with the gloo backend
    (several ranks on one GPU); RCCL sends device memory directly."""
    ops, landings = [], []
    for peer, first, count in moves:
        view = slot_buffer[first * slot_bytes:(first + count) * slot_bytes]
        if rank == peer:
            ops.append(dist.P2POp(dist.isend, view.cpu() if via_host else view, 0))
        elif rank == 0:
            if via_host:
                host = view.cpu()
                ops.append(dist.P2POp(dist.irecv, host, peer))
                landings.append((view, host))
            else:
                ops.append(dist.P2POp(dist.irecv, view, peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for view, host in landings:
        view.copy_(host)

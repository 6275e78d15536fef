// pano_sharded.cpp - the camera-sharded exchange of the C-ABI (include/pano.h) over RCCL: one process per GPU, the pyramid slots of
// every rank's cameras land in place on the root with ONE ncclGroup (replaces the reference's UDP + JPEG link, src/slave.cpp:88-145).

#include "pano_ctx.hpp"

extern "C" {

/* ---- the camera-sharded exchange over RCCL (one process per GPU) ------------------------------------------- */
#define RCCL_TRY(ctx, expr)                                                                      \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            if (ctx) (ctx)->err = std::string(#expr) + ": " + Rccl::get().GetErrorString(r_);   \
            return PANO_EHIP;                                                                    \
        }                                                                                        \
    } while (0)


pano_status pano_rccl_unique_id(char id[PANO_RCCL_ID_BYTES]) {
    static_assert(PANO_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return PANO_ENODEVICE;
    ncclUniqueId u;
    if (R.GetUniqueId(&u) != ncclSuccess) return PANO_EHIP;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return PANO_OK;
}

pano_status pano_rccl_comm_create(pano_ctx* c, const char id[PANO_RCCL_ID_BYTES], int world, int rank, void** comm) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return fail(c, PANO_ENODEVICE, R.error.c_str());
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t nc = nullptr;
    RCCL_TRY(c, R.CommInitRank(&nc, world, u, rank));
    *comm = nc;
    return PANO_OK;
}

pano_status pano_rccl_comm_destroy(void* comm) {
    if (!comm) return PANO_OK;
    Rccl& R = Rccl::get();
    if (!R.ok) return PANO_ENODEVICE;
    return R.CommDestroy((ncclComm_t)comm) == ncclSuccess ? PANO_OK : PANO_EHIP;
}

pano_status pano_rccl_comm_count(void* comm, int* ranks) {
    if (!comm || !ranks) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return PANO_ENODEVICE;
    return R.CommCount((ncclComm_t)comm, ranks) == ncclSuccess ? PANO_OK : PANO_EHIP;
}

const char* pano_rccl_library(void) {
    Rccl& R = Rccl::get();
    return R.ok ? R.path.c_str() : "";
}

// The packed form of the exchange: per camera the live rectangles of its pyramid levels (pano_get_live_rect: what the blend on the
// root reads; the rest of a slot - 30 % on config 2 - is never produced and never read), columns widened to 16-byte boundaries, as
// copy segments {plane of a level} -> one contiguous message per camera.  Every rank holds the same masks, so every rank derives the
// same segments and message sizes.  Rebuilt when the live rects change (a mask change).
static pano_status build_exchange_segments(pano_ctx* c) {
    const int n = c->plan.n;
    c->xch_segs.clear();
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        c->xch_first[i] = (int)c->xch_segs.size();
        c->xch_off[i] = off;
        c->xch_rows[i] = 0;
        for (int l = 0; l < c->levels; l++) {
            const int* r = c->live[i][l];
            if (r[2] < r[0] || r[3] < r[1]) continue;   // nothing live (an empty mask)
            const int pitch = c->lvl_pitch[i][l];
            const int x0 = r[0] & ~15, x1 = std::min(r[2] | 15, pitch - 1);   // rows are padded to 16 bytes
            const int w16 = (x1 - x0 + 1 + 15) / 16, rows = r[3] - r[1] + 1;
            for (int pl = 0; pl < 3; pl++) {
                XchSeg g{};
                g.slot_off = (size_t)i * c->slot_bytes + c->lvl_off[i][l] + (size_t)pl * c->lvl_plane[i][l] + (size_t)r[1] * pitch + x0;
                g.stage_off = off;
                g.pitch = pitch; g.width16 = w16; g.rows = rows;
                c->xch_segs.push_back(g);
                off += (size_t)w16 * 16 * rows;
            }
            c->xch_rows[i] = std::max(c->xch_rows[i], rows);
        }
    }
    c->xch_first[n] = (int)c->xch_segs.size();
    c->xch_off[n] = off;
    HIP_TRY(c, hipDeviceSynchronize());   // a frame in flight may still read the old table
    dfree(c->d_xch_segs);
    if (!c->xch_segs.empty()) {
        HIP_TRY(c, hipMalloc((void**)&c->d_xch_segs, c->xch_segs.size() * sizeof(XchSeg)));
        HIP_TRY(c, hipMemcpy(c->d_xch_segs, c->xch_segs.data(), c->xch_segs.size() * sizeof(XchSeg), hipMemcpyHostToDevice));
    }
    // one staging area per frame slot: frames in flight (pano_set_frame_slots) gather on streams of their own
    if (off > c->xch_stage_bytes) {
        dfree(c->xch_stage);
        HIP_TRY(c, hipMalloc((void**)&c->xch_stage, off * PANO_MAX_FRAME_SLOTS));
        c->xch_stage_bytes = off;
    }
    c->xch_dirty = false;
    return PANO_OK;
}

pano_status pano_get_exchange_stats(pano_ctx* c, uint64_t* packed_bytes_per_camera, uint64_t* slot_bytes, uint64_t* bytes_moved) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if ((st = ensure_weights(c, c->own_stream)) != PANO_OK) return st;   // the live rects follow the masks
    if (c->xch_dirty && (st = build_exchange_segments(c)) != PANO_OK) return st;
    if (packed_bytes_per_camera)
        for (int i = 0; i < c->plan.n; i++) packed_bytes_per_camera[i] = c->xch_whole_slots ? c->slot_bytes : c->xch_off[i + 1] - c->xch_off[i];
    if (slot_bytes) *slot_bytes = c->slot_bytes;
    if (bytes_moved) *bytes_moved = c->xch_bytes_moved;
    return PANO_OK;
}

pano_status pano_gather_slots(pano_ctx* c, void* comm, int rank, int root, const int* owner_rank, void* stream) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!comm || !owner_rank || rank < 0 || root < 0) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return fail(c, PANO_ENODEVICE, R.error.c_str());
    const int n = c->plan.n;
    if (!stream) stream = c->own_stream;  // callers without HIP types: the stream pano_feed_cameras_host / pano_blend_host use
    hipStream_t s = (hipStream_t)stream;
    // what travels: by default the live rectangles of every level, packed (every rank derives the same sizes from the same masks -
    // the root too, which may have fed none of this context's cameras: its weights, and with them the live rects, are made current
    // here); PANO_GATHER_WHOLE_SLOTS=1: whole slots, in place
    const bool packed = !c->xch_whole_slots;
    if (packed) {
        if ((st = ensure_weights(c, s)) != PANO_OK) return st;
        if (c->xch_dirty && (st = build_exchange_segments(c)) != PANO_OK) return st;
    }
    char* const slots = c->pyr_base;                                             // the frame slot in force (pano_select_frame_slot)
    uint8_t* const stage = c->xch_stage + (size_t)c->cur_slot * c->xch_stage_bytes;
    // consecutive slots with the same peer travel as one message: a rank's cameras are a contiguous range of slots and of the staging buffer
    if (packed && rank != root)
        for (int i = 0; i < n; i++)
            if (owner_rank[i] == rank && owner_rank[i] != root)
                launch_copy_segments(c->d_xch_segs, c->xch_first[i], c->xch_first[i + 1] - c->xch_first[i], c->xch_rows[i], (uint8_t*)slots, stage, false, s);
    RCCL_TRY(c, R.GroupStart());
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && owner_rank[j] == owner_rank[i]) j++;
        const int owner = owner_rank[i];
        char* base = packed ? (char*)stage + c->xch_off[i] : slots + (size_t)i * c->slot_bytes;
        const size_t bytes = packed ? c->xch_off[j] - c->xch_off[i] : (size_t)(j - i) * c->slot_bytes;
        ncclResult_t r = ncclSuccess;
        if (owner != root && bytes) {
            if (rank == root) r = R.Recv(base, bytes, ncclUint8, owner, (ncclComm_t)comm, s);
            else if (rank == owner) r = R.Send(base, bytes, ncclUint8, root, (ncclComm_t)comm, s);
            if (rank == root || rank == owner) c->xch_bytes_moved += bytes;
        }
        if (r != ncclSuccess) {
            (void)R.GroupEnd();
            c->err = std::string("ncclSend / ncclRecv: ") + R.GetErrorString(r);
            return PANO_EHIP;
        }
        i = j;
    }
    RCCL_TRY(c, R.GroupEnd());
    if (packed && rank == root)
        for (int i = 0; i < n; i++)
            if (owner_rank[i] != root)
                launch_copy_segments(c->d_xch_segs, c->xch_first[i], c->xch_first[i + 1] - c->xch_first[i], c->xch_rows[i], (uint8_t*)slots, stage, true, s);
    return PANO_OK;
}

}  // extern "C"

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

pano_status pano_gather_slots(pano_ctx* c, void* comm, int rank, int root, const int* owner_rank, void* stream) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!comm || !owner_rank || rank < 0 || root < 0) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return fail(c, PANO_ENODEVICE, R.error.c_str());
    const int n = c->plan.n;
    if (!stream) stream = c->own_stream;  // callers without HIP types: the stream pano_feed_cameras_host / pano_blend_host use
    // consecutive slots with the same peer travel as one message: a rank's cameras are a contiguous byte range
    RCCL_TRY(c, R.GroupStart());
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && owner_rank[j] == owner_rank[i]) j++;
        const int owner = owner_rank[i];
        char* base = c->pyr_base + (size_t)i * c->slot_bytes;
        const size_t bytes = (size_t)(j - i) * c->slot_bytes;
        ncclResult_t r = ncclSuccess;
        if (owner != root) {
            if (rank == root) r = R.Recv(base, bytes, ncclUint8, owner, (ncclComm_t)comm, (hipStream_t)stream);
            else if (rank == owner) r = R.Send(base, bytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream);
        }
        if (r != ncclSuccess) {
            (void)R.GroupEnd();
            c->err = std::string("ncclSend / ncclRecv: ") + R.GetErrorString(r);
            return PANO_EHIP;
        }
        i = j;
    }
    RCCL_TRY(c, R.GroupEnd());
    return PANO_OK;
}

}  // extern "C"

// fake_rccl.cpp - TEST DOUBLE for librccl.so (test infrastructure; never part of the product).
//
// libpano_hip.so opens its RCCL library by name (csrc/pano_rccl.hpp); with PANO_RCCL_LIB pointing here the C-ABI's one exchange,
// pano_gather_slots (include/pano.h), runs between real peers on a box with ONE GPU - RCCL itself refuses two ranks on one
// device.  The double exports the nine symbols the loader binds and
//   * MOVES THE BYTES between ranks - processes, or threads of one process - that share a device: a send copies the device
//     range to a POSIX shared-memory object of its own (D2H), the matching receive copies it from there into the receiver's
//     device range (H2D).  Host-synchronous: both sides drain `stream` first and the data is in place when ncclGroupEnd returns,
//     a stricter ordering than RCCL's (which enqueues on the stream), so anything correct against RCCL's is correct here;
//   * CHECKS what real RCCL would only punish with a hang or silent corruption: every send meets a receive of the same
//     count, datatype and peer, in per-pair order; sends and receives only inside ncclGroupStart / ncclGroupEnd (the library
//     promises ONE group per gather: an ungrouped call is an error here), group nesting balanced, no operation left pending
//     when the communicator is destroyed, ranks and peers inside the communicator, no send to self;
//   * COUNTS per communicator and rank: groups, sends, receives and bytes (fake_rccl_stats), so a test can prove that the
//     exchange it asserts on really happened.
// FAKE_RCCL_HOST_BUFFERS=1: the buffers are host memory and no HIP call is made - for the double's OWN unit tests on a box
// without a GPU (tests/test_fake_rccl.py).
// What it does not stand in for: the transport (xGMI, IPC, rings).  Rendezvous and matching time out (FAKE_RCCL_TIMEOUT_S,
// default 60 s) with ncclSystemError instead of hanging the box.
//
// Build: hipcc -O1 -shared -fPIC tests/src/fake_rccl.cpp -o <dir>/libfake_rccl.so -lrt -lpthread
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 64;
constexpr uint32_t kMagic = 0x46524343u;  // "FRCC"

// one per communicator, in shared memory, named after the unique id
struct Control {
    std::atomic<uint32_t> magic;
    std::atomic<int> world;
    std::atomic<int> joined[kMaxRanks];
    std::atomic<int> left;                          // ranks that have destroyed their communicator
    std::atomic<uint64_t> seq[kMaxRanks][kMaxRanks]; // [src][dst]: messages the SENDER has posted
    std::atomic<uint64_t> groups[kMaxRanks], sends[kMaxRanks], recvs[kMaxRanks], bytes_out[kMaxRanks], bytes_in[kMaxRanks];
};
// one per message, in a shared-memory object of its own: header + payload
struct MsgHeader {
    std::atomic<uint32_t> ready;
    uint32_t datatype;
    uint64_t count, bytes;
};

struct Op {
    bool send;
    void* buf;
    size_t count;
    ncclDataType_t dt;
    int peer;
    hipStream_t stream;
};

thread_local std::string g_err;

}  // namespace

struct ncclComm {  // the opaque type of <rccl/rccl.h>
    uint32_t magic = kMagic;
    std::string id;  // hex of the unique id: the prefix of every shared-memory name
    int world = 0, rank = -1, device = 0;
    Control* ctl = nullptr;
    uint64_t recv_seq[kMaxRanks] = {};  // per source: messages this rank has consumed
    std::vector<Op> pending;
};

namespace {

// group state is per thread, like RCCL's
thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm*, Op>> g_ops;

bool host_buffers() {
    const char* e = getenv("FAKE_RCCL_HOST_BUFFERS");
    return e && atoi(e) != 0;
}
hipError_t drain(hipStream_t s) { return host_buffers() ? hipSuccess : hipStreamSynchronize(s); }
hipError_t copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (host_buffers()) {
        std::memcpy(dst, src, bytes);
        return hipSuccess;
    }
    return hipMemcpy(dst, src, bytes, kind);
}
double timeout_s() {
    const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
    const double v = e ? atof(e) : 60.0;
    return v > 0 ? v : 60.0;
}
size_t dt_size(ncclDataType_t dt) {
    switch (dt) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
ncclResult_t bad(ncclResult_t r, const std::string& why) {
    g_err = why;
    fprintf(stderr, "[fake_rccl] %s\n", why.c_str());
    return r;
}
std::string msg_name(const ncclComm* c, int src, int dst, uint64_t seq) {
    return "/frccl_" + c->id + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq);
}
void* map_shm(const std::string& name, size_t bytes, bool create, double wait_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int fd = shm_open(name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
        if (fd >= 0) {
            if (create && ftruncate(fd, (off_t)bytes) != 0) {
                close(fd);
                shm_unlink(name.c_str());
                return nullptr;
            }
            if (!create) {  // the creator sizes it before it fills it: wait until the size is there
                struct stat st;
                while (fstat(fd, &st) == 0 && (size_t)st.st_size < bytes) {
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) {
                        close(fd);
                        return nullptr;
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            }
            void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            return p == MAP_FAILED ? nullptr : p;
        }
        if (create) return nullptr;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) return nullptr;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

ncclResult_t do_send(ncclComm* c, const Op& op) {
    const size_t bytes = op.count * dt_size(op.dt);
    const uint64_t seq = c->ctl->seq[c->rank][op.peer].load();
    const std::string name = msg_name(c, c->rank, op.peer, seq);
    char* m = (char*)map_shm(name, sizeof(MsgHeader) + bytes, true, 0);
    if (!m) return bad(ncclSystemError, "send: cannot create " + name + " (a stale object of an earlier run, or /dev/shm full)");
    MsgHeader* h = new (m) MsgHeader;
    h->ready.store(0);
    h->datatype = (uint32_t)op.dt;
    h->count = op.count;
    h->bytes = bytes;
    hipError_t e = drain(op.stream);  // what earlier launches on the stream produce is what travels
    if (e == hipSuccess && bytes) e = copy(m + sizeof(MsgHeader), op.buf, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        munmap(m, sizeof(MsgHeader) + bytes);
        shm_unlink(name.c_str());
        return bad(ncclUnhandledCudaError, std::string("send: ") + hipGetErrorString(e));
    }
    h->ready.store(1, std::memory_order_release);
    c->ctl->seq[c->rank][op.peer].fetch_add(1);
    munmap(m, sizeof(MsgHeader) + bytes);
    c->ctl->sends[c->rank]++;
    c->ctl->bytes_out[c->rank] += bytes;
    return ncclSuccess;
}
ncclResult_t do_recv(ncclComm* c, const Op& op) {
    const size_t bytes = op.count * dt_size(op.dt);
    const uint64_t seq = c->recv_seq[op.peer];
    const std::string name = msg_name(c, op.peer, c->rank, seq);
    const double T = timeout_s();
    // the header first: the sender may have posted another size (that is the mismatch this double exists to catch)
    MsgHeader* h = (MsgHeader*)map_shm(name, sizeof(MsgHeader), false, T);
    if (!h) return bad(ncclSystemError, "recv on rank " + std::to_string(c->rank) + ": no message " + std::to_string(seq) + " from rank " +
                                            std::to_string(op.peer) + " within " + std::to_string((int)T) + " s (unmatched ncclRecv)");
    const auto t0 = std::chrono::steady_clock::now();
    while (!h->ready.load(std::memory_order_acquire)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > T) {
            munmap(h, sizeof(MsgHeader));
            return bad(ncclSystemError, "recv: message from rank " + std::to_string(op.peer) + " never became ready");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    const uint64_t scount = h->count, sbytes = h->bytes;
    const uint32_t sdt = h->datatype;
    munmap(h, sizeof(MsgHeader));
    if (scount != op.count || sdt != (uint32_t)op.dt) {
        shm_unlink(name.c_str());
        return bad(ncclInvalidArgument, "send / recv mismatch between rank " + std::to_string(op.peer) + " and rank " + std::to_string(c->rank) +
                                            ": sent " + std::to_string(scount) + " x type " + std::to_string(sdt) + ", expected " +
                                            std::to_string(op.count) + " x type " + std::to_string((int)op.dt));
    }
    char* m = (char*)map_shm(name, sizeof(MsgHeader) + sbytes, false, T);
    if (!m) return bad(ncclSystemError, "recv: cannot map " + name);
    hipError_t e = drain(op.stream);  // nothing earlier on the stream still reads the landing range
    if (e == hipSuccess && bytes) e = copy(op.buf, m + sizeof(MsgHeader), bytes, hipMemcpyHostToDevice);
    munmap(m, sizeof(MsgHeader) + sbytes);
    shm_unlink(name.c_str());
    if (e != hipSuccess) return bad(ncclUnhandledCudaError, std::string("recv: ") + hipGetErrorString(e));
    c->recv_seq[op.peer]++;
    c->ctl->recvs[c->rank]++;
    c->ctl->bytes_in[c->rank] += bytes;
    return ncclSuccess;
}

ncclResult_t check_comm(ncclComm* c, const char* what) {
    if (!c || c->magic != kMagic || !c->ctl) return bad(ncclInvalidArgument, std::string(what) + ": not a communicator of this library");
    return ncclSuccess;
}
ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm* c, hipStream_t s) {
    const char* what = send ? "ncclSend" : "ncclRecv";
    ncclResult_t r = check_comm(c, what);
    if (r != ncclSuccess) return r;
    if (g_depth == 0)
        return bad(ncclInvalidUsage, std::string(what) + " outside ncclGroupStart / ncclGroupEnd: pano_gather_slots promises one group per gather");
    if (peer < 0 || peer >= c->world) return bad(ncclInvalidArgument, std::string(what) + ": peer " + std::to_string(peer) + " outside the communicator");
    if (peer == c->rank) return bad(ncclInvalidArgument, std::string(what) + " to self");
    if (!dt_size(dt)) return bad(ncclInvalidArgument, std::string(what) + ": datatype");
    if (count && !buf) return bad(ncclInvalidArgument, std::string(what) + ": null buffer");
    g_ops.push_back({c, Op{send, buf, count, dt, peer, s}});
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return bad(ncclInvalidArgument, "ncclGetUniqueId: null");
    std::random_device rd;
    std::memset(id->internal, 0, sizeof(id->internal));
    const uint64_t v[2] = {((uint64_t)rd() << 32) ^ rd() ^ ((uint64_t)getpid() << 17),
                           (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count()};
    std::memcpy(id->internal, "FRCC", 4);
    std::memcpy(id->internal + 8, v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return bad(ncclInvalidArgument, "ncclCommInitRank: arguments");
    if (std::memcmp(id.internal, "FRCC", 4) != 0) return bad(ncclInvalidArgument, "ncclCommInitRank: not a unique id of this library");
    ncclComm* c = new ncclComm;
    char hex[40];
    uint64_t v[2];
    std::memcpy(v, id.internal + 8, sizeof(v));
    snprintf(hex, sizeof(hex), "%016llx%016llx", (unsigned long long)v[0], (unsigned long long)v[1]);
    c->id = hex;
    c->world = nranks;
    c->rank = rank;
    if (!host_buffers()) (void)hipGetDevice(&c->device);
    const std::string name = "/frccl_" + c->id + "_ctl";
    // whoever comes first creates the control block; everybody maps it
    int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return bad(ncclSystemError, "ncclCommInitRank: shm_open " + name);
    }
    void* p = mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return bad(ncclSystemError, "ncclCommInitRank: mmap");
    }
    c->ctl = (Control*)p;  // a fresh object is zero-filled: every atomic starts at 0
    int w0 = 0;
    if (!c->ctl->world.compare_exchange_strong(w0, nranks) && w0 != nranks) {
        munmap(p, sizeof(Control));
        delete c;
        return bad(ncclInvalidArgument, "ncclCommInitRank: ranks disagree about the world size");
    }
    if (c->ctl->joined[rank].exchange(1) != 0) {
        munmap(p, sizeof(Control));
        delete c;
        return bad(ncclInvalidArgument, "ncclCommInitRank: rank " + std::to_string(rank) + " joined twice");
    }
    // rendezvous, like the real call: nobody returns before everybody is there
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int n = 0;
        for (int r = 0; r < nranks; r++) n += c->ctl->joined[r].load() ? 1 : 0;
        if (n == nranks) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
            munmap(p, sizeof(Control));
            delete c;
            return bad(ncclSystemError, "ncclCommInitRank: only " + std::to_string(n) + " of " + std::to_string(nranks) + " ranks arrived");
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    ncclResult_t r = check_comm(comm, "ncclCommCount");
    if (r != ncclSuccess) return r;
    if (!count) return bad(ncclInvalidArgument, "ncclCommCount: null");
    *count = comm->world;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    ncclResult_t r = check_comm(comm, "ncclCommDestroy");
    if (r != ncclSuccess) return r;
    for (auto& po : g_ops)
        if (po.first == comm) return bad(ncclInvalidUsage, "ncclCommDestroy inside an open group with operations of this communicator pending");
    // a message this rank was sent and never received is a send without its receive
    ncclResult_t res = ncclSuccess;
    for (int src = 0; src < comm->world; src++)
        if (src != comm->rank && comm->ctl->seq[src][comm->rank].load() != comm->recv_seq[src])
            res = bad(ncclInvalidUsage, "ncclCommDestroy on rank " + std::to_string(comm->rank) + ": " +
                                            std::to_string(comm->ctl->seq[src][comm->rank].load() - comm->recv_seq[src]) +
                                            " message(s) from rank " + std::to_string(src) + " were never received");
    const int left = comm->ctl->left.fetch_add(1) + 1;
    const std::string name = "/frccl_" + comm->id + "_ctl";
    const bool last = left == comm->world;
    munmap(comm->ctl, sizeof(Control));
    if (last) shm_unlink(name.c_str());
    comm->magic = 0;
    delete comm;
    return res;
}

ncclResult_t ncclGroupStart() {
    g_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth == 0) return bad(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return ncclSuccess;
    std::vector<std::pair<ncclComm*, Op>> ops;
    ops.swap(g_ops);
    std::vector<ncclComm*> seen;
    for (auto& po : ops) {
        bool dup = false;
        for (ncclComm* s : seen) dup |= s == po.first;
        if (!dup) {
            seen.push_back(po.first);
            po.first->ctl->groups[po.first->rank]++;
        }
    }
    // every send first (a send never waits: its message is an object of its own), then the receives - no order of
    // ranks can deadlock
    ncclResult_t res = ncclSuccess;
    for (auto& po : ops)
        if (po.second.send && res == ncclSuccess) res = do_send(po.first, po.second);
    for (auto& po : ops)
        if (!po.second.send && res == ncclSuccess) res = do_recv(po.first, po.second);
    return res;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(false, recvbuff, count, datatype, peer, comm, stream);
}

const char* ncclGetErrorString(ncclResult_t result) {
    static thread_local std::string s;
    s = "fake_rccl result " + std::to_string((int)result) + (g_err.empty() ? "" : ": " + g_err);
    return s.c_str();
}

// ---- not part of RCCL: what the tests read ----
// out[0..4] = groups, sends, receives, bytes sent, bytes received of `rank` (any rank of the communicator: the counters live in
// the shared control block) since the communicator was created
int fake_rccl_stats(ncclComm_t comm, int rank, unsigned long long out[5]) {
    if (check_comm(comm, "fake_rccl_stats") != ncclSuccess || rank < 0 || rank >= comm->world || !out) return -1;
    out[0] = comm->ctl->groups[rank].load();
    out[1] = comm->ctl->sends[rank].load();
    out[2] = comm->ctl->recvs[rank].load();
    out[3] = comm->ctl->bytes_out[rank].load();
    out[4] = comm->ctl->bytes_in[rank].load();
    return 0;
}
const char* fake_rccl_identity(void) { return "fake_rccl test double (tests/src/fake_rccl.cpp)"; }

}  // extern "C"

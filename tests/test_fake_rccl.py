"""the RCCL test double (tests/src/fake_rccl.cpp) checked on its own, without a GPU (FAKE_RCCL_HOST_BUFFERS=1: host buffers,
no HIP call): it moves the bytes between ranks, matches sends and receives per pair and in order, and refuses what real RCCL
would answer with a hang - a count mismatch, an ungrouped call, an unmatched receive, a send nobody received.  The GPU tests
(tests/test_gpu_rccl_double.py) put libpano_hip.so's pano_gather_slots on top of it."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

NCCL_UINT8 = 1  # ncclUint8 of <rccl/rccl.h>
NCCL_FLOAT32 = 7


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.fixture(scope="module")
def fk(fake_rccl_lib):
    os.environ["FAKE_RCCL_HOST_BUFFERS"] = "1"
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "1.5"
    lib = C.CDLL(fake_rccl_lib)
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.fake_rccl_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    lib.ncclGetErrorString.restype = C.c_char_p
    yield lib
    os.environ.pop("FAKE_RCCL_HOST_BUFFERS", None)
    os.environ.pop("FAKE_RCCL_TIMEOUT_S", None)


def run_ranks(fk, world, body):
    """body(lib, comm, rank) on one thread per rank (ctypes releases the GIL in every call); returns the per-rank results"""
    uid = UniqueId()
    assert fk.ncclGetUniqueId(C.byref(uid)) == 0
    res = [None] * world

    def rank_main(r):
        comm = C.c_void_p()
        st = fk.ncclCommInitRank(C.byref(comm), world, uid, r)
        if st != 0:
            res[r] = ("init", st)
            return
        try:
            res[r] = body(fk, comm, r)
        finally:
            d = fk.ncclCommDestroy(comm)
            res[r] = (res[r], d)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th)
    return res


def test_gather_moves_the_bytes_and_counts_them(fk):
    """the shape of pano_gather_slots at world 4: ranks 1..3 send a range each to rank 0 inside one group; rank 0 receives every
    range in place"""
    n = 100_000

    def body(lib, comm, r):
        cnt = C.c_int(0)
        assert lib.ncclCommCount(comm, C.byref(cnt)) == 0 and cnt.value == 4
        buf = np.zeros(4 * n, np.uint8)
        buf[r * n:(r + 1) * n] = np.arange(n, dtype=np.uint32).astype(np.uint8) + r
        assert lib.ncclGroupStart() == 0
        for peer in range(1, 4):
            view = buf[peer * n:(peer + 1) * n]
            if r == 0:
                assert lib.ncclRecv(view.ctypes.data, n, NCCL_UINT8, peer, comm, None) == 0
            elif r == peer:
                assert lib.ncclSend(view.ctypes.data, n, NCCL_UINT8, 0, comm, None) == 0
        st = lib.ncclGroupEnd()
        stats = (C.c_ulonglong * 5)()
        assert lib.fake_rccl_stats(comm, r, stats) == 0
        ok = True
        if r == 0:
            for peer in range(4):
                ok &= bool((buf[peer * n:(peer + 1) * n] == np.arange(n, dtype=np.uint32).astype(np.uint8) + peer).all())
        return st, ok, list(stats)

    res = run_ranks(fk, 4, body)
    for r, ((st, ok, stats), destroyed) in enumerate(res):
        assert st == 0 and ok and destroyed == 0
        assert stats[0] == 1                                   # one group
        assert stats[1:] == ([0, 3, 0, 3 * n] if r == 0 else [1, 0, n, 0])


def test_messages_of_a_pair_match_in_order(fk):
    def body(lib, comm, r):
        a, b = np.full(16, 1 + 10 * r, np.float32), np.full(8, 2 + 10 * r, np.float32)
        assert lib.ncclGroupStart() == 0 and lib.ncclGroupStart() == 0       # nested groups: one exchange at the outer end
        if r == 1:
            lib.ncclSend(a.ctypes.data, 16, NCCL_FLOAT32, 0, comm, None)
            lib.ncclSend(b.ctypes.data, 8, NCCL_FLOAT32, 0, comm, None)
        else:
            lib.ncclRecv(a.ctypes.data, 16, NCCL_FLOAT32, 1, comm, None)
            lib.ncclRecv(b.ctypes.data, 8, NCCL_FLOAT32, 1, comm, None)
        assert lib.ncclGroupEnd() == 0
        st = lib.ncclGroupEnd()
        return st, float(a[0]), float(b[0])

    res = run_ranks(fk, 2, body)
    assert res[0] == ((0, 11.0, 12.0), 0) and res[1] == ((0, 11.0, 12.0), 0)


def test_what_real_rccl_would_hang_on_is_an_error_here(fk):
    # a count mismatch between the two ends
    def mismatch(lib, comm, r):
        buf = np.zeros(64, np.uint8)
        lib.ncclGroupStart()
        if r == 1:
            lib.ncclSend(buf.ctypes.data, 64, NCCL_UINT8, 0, comm, None)
        else:
            lib.ncclRecv(buf.ctypes.data, 32, NCCL_UINT8, 1, comm, None)
        return lib.ncclGroupEnd()

    res = run_ranks(fk, 2, mismatch)
    assert res[0][0] != 0 and res[1][0] == 0

    # send / recv outside a group, to self, to a rank outside the communicator; ncclGroupEnd without a start
    def usage(lib, comm, r):
        buf = np.zeros(8, np.uint8)
        out = [lib.ncclSend(buf.ctypes.data, 8, NCCL_UINT8, 1 - r, comm, None)]
        lib.ncclGroupStart()
        out.append(lib.ncclSend(buf.ctypes.data, 8, NCCL_UINT8, r, comm, None))
        out.append(lib.ncclRecv(buf.ctypes.data, 8, NCCL_UINT8, 5, comm, None))
        out.append(lib.ncclGroupEnd())   # nothing was queued: fine
        out.append(lib.ncclGroupEnd())   # unbalanced
        return out

    res = run_ranks(fk, 2, usage)
    for (out, destroyed) in res:
        assert out[0] != 0 and out[1] != 0 and out[2] != 0 and out[3] == 0 and out[4] != 0 and destroyed == 0

    # a receive nobody sends to times out instead of hanging; a send nobody receives is reported when the receiver's
    # communicator is destroyed
    def unmatched(lib, comm, r):
        buf = np.zeros(8, np.uint8)
        lib.ncclGroupStart()
        if r == 0:
            lib.ncclRecv(buf.ctypes.data, 8, NCCL_UINT8, 1, comm, None)
        return lib.ncclGroupEnd()

    res = run_ranks(fk, 2, unmatched)
    assert res[0][0] != 0 and res[1][0] == 0

    def orphan(lib, comm, r):
        buf = np.zeros(8, np.uint8)
        lib.ncclGroupStart()
        if r == 1:
            lib.ncclSend(buf.ctypes.data, 8, NCCL_UINT8, 0, comm, None)
        st = lib.ncclGroupEnd()
        import time
        time.sleep(0.2)   # rank 0 destroys its communicator after the send is posted
        return st

    res = run_ranks(fk, 2, orphan)
    assert res[0] == (0, res[0][1]) and res[0][1] != 0 and res[1] == (0, 0)
    # the orphan's shared-memory object would outlive the test: remove what this test left behind
    for f in os.listdir("/dev/shm"):
        if f.startswith("frccl_"):
            os.unlink(os.path.join("/dev/shm", f))

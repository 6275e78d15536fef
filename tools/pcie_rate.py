#!/usr/bin/env python3
"""what the host link of this box gives: page-locked H2D / D2H rate (one stream, two streams), pageable H2D, and the
host memcpy rate of N threads - the bounds of pano_compose_host and pano_stream_* """
import threading
import time

import numpy as np
import torch


def rate(fn, nbytes, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t) / 1e9


def main():
    n = 48 << 20
    hp = torch.empty(n, dtype=torch.uint8).pin_memory()
    hp2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    hq = torch.empty(n, dtype=torch.uint8)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out = {}
    out["h2d_pinned_GBps"] = rate(lambda: d.copy_(hp, non_blocking=True), n)
    out["d2h_pinned_GBps"] = rate(lambda: hp.copy_(d, non_blocking=True), n)
    out["h2d_pageable_GBps"] = rate(lambda: d.copy_(hq), n, 3)

    def two():
        with torch.cuda.stream(s1):
            d.copy_(hp, non_blocking=True)
        with torch.cuda.stream(s2):
            d2.copy_(hp2, non_blocking=True)
    out["h2d_pinned_two_streams_GBps"] = rate(two, 2 * n)

    def duplex():
        with torch.cuda.stream(s1):
            d.copy_(hp, non_blocking=True)
        with torch.cuda.stream(s2):
            hp2.copy_(d2, non_blocking=True)
    out["h2d_plus_d2h_GBps"] = rate(duplex, 2 * n)
    a = np.empty(n, np.uint8); b = np.empty(n, np.uint8)
    a[:] = 1; b[:] = 2
    for nt in (1, 2, 4, 8, 16):
        parts = np.array_split(np.arange(n), nt)
        sl = [(int(p[0]), int(p[-1]) + 1) for p in parts]

        def work(lo, hi):
            for _ in range(10):
                np.copyto(b[lo:hi], a[lo:hi])
        th = [threading.Thread(target=work, args=x) for x in sl]
        t = time.perf_counter()
        [x.start() for x in th]; [x.join() for x in th]
        out["host_memcpy_%d_threads_GBps" % nt] = n * 10 / (time.perf_counter() - t) / 1e9
    print({k: round(v, 1) for k, v in out.items()})


if __name__ == "__main__":
    main()

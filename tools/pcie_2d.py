#!/usr/bin/env python3
"""rate of RECTANGULAR page-locked host -> device copies (hipMemcpy2DAsync) against linear ones: what uploading only the live
columns of a frame (pano_compose_host / pano_stream_submit) can expect from the link.  8 frames of 1920 x 1080 x 3 per round."""
import ctypes as C
import json
import time

import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
H2D = 1
W, H, N = 1920, 1080, 8
pitch = W * 3
host = [torch.empty(pitch * H, dtype=torch.uint8).pin_memory() for _ in range(N)]
dev = [torch.empty(pitch * H, dtype=torch.uint8, device="cuda") for _ in range(N)]
st = torch.cuda.Stream()
s = C.c_void_p(st.cuda_stream)


def run(fn, nbytes, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return {"GBps": round(nbytes / dt / 1e9, 2), "ms_per_8_frames": round(dt * 1e3, 3)}


out = {}
out["linear_full"] = run(lambda: [hip.hipMemcpyAsync(d.data_ptr(), h.data_ptr(), pitch * H, H2D, s) for d, h in zip(dev, host)], N * pitch * H)
for name, x0, wbytes in (("2d_70pct_aligned64", 832, 4032), ("2d_70pct_odd", 835, 4031), ("2d_50pct_aligned64", 1408, 2880), ("2d_full_width", 0, pitch)):
    out[name] = run(lambda: [hip.hipMemcpy2DAsync(d.data_ptr() + x0, pitch, h.data_ptr() + x0, pitch, wbytes, H, H2D, s) for d, h in zip(dev, host)],
                    N * wbytes * H)
    out[name]["row_bytes"] = wbytes
out["linear_70pct_rows"] = run(lambda: [hip.hipMemcpyAsync(d.data_ptr(), h.data_ptr(), pitch * 756, H2D, s) for d, h in zip(dev, host)], N * pitch * 756)
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Diagnostic: rocprof-free timing of the level-0 blend kernel and its ablations (diag library)."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child():
    import torch
    from helpers import c2_group, synth_frame
    pano = importlib.import_module("img-stitching_amd")
    g = c2_group()
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi()
    frames = [torch.from_numpy(synth_frame(g["w"], g["h"], 42 + i)).cuda() for i in range(4)]
    ow, oh = ctx.output_size()
    out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ctx.compose([t.data_ptr() for t in frames], [g["w"] * 3] * 4, out.data_ptr(), ow * 3, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        ctx.blend(out.data_ptr(), ow * 3, st)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        ctx.blend(out.data_ptr(), ow * 3, st)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"abl": os.environ.get("PANO_K3_ABL", "0"), "blend_all_levels_us": round(e0.elapsed_time(e1) * 10, 2)}))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        names = {0: "full", 1: "no canvas L1 loads", 2: "no coarse tile loads", 3: "no level-0 tile loads", 4: "no stores", 5: "fast path only", 6: "level 0 skipped"}
        for a in (0, 5, 6):
            env = dict(os.environ, PANO_K3_ABL=str(a), PANO_LIB=os.path.join(ROOT, "img-stitching_amd", "libpano_hip_diag.so"))
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(names[a], r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)

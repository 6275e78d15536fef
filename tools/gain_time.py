import importlib, json, os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c2_group()
ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
for i in range(4): ctx.set_camera(i, g["K"][i], g["R"][i])
ctx.prepare(); ctx.build_masks_voronoi()
rng = np.random.default_rng(3)
for i in range(4):
    r = ctx.roi(i)
    ctx.set_gain_map(i, (0.9 + 0.2 * rng.random(((r[3] + 31) // 32, (r[2] + 31) // 32))).astype(np.float32))
frames = [torch.from_numpy(synth_frame(g["w"], g["h"], 42 + i)).cuda() for i in range(4)]
ow, oh = ctx.output_size(); out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ctx.set_profiling(True)
for _ in range(10): ctx.compose([t.data_ptr() for t in frames], [g["w"] * 3] * 4, out.data_ptr(), ow * 3, st)
ctx.stage_stats(True)
for _ in range(100): ctx.compose([t.data_ptr() for t in frames], [g["w"] * 3] * 4, out.data_ptr(), ow * 3, st)
torch.cuda.synchronize()
ms, n = ctx.stage_stats(True)
print(json.dumps({"c2_group_with_gains_warp_us": round(ms[0] / n[0] * 1e3, 2)}))

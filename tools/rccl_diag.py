import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pano = importlib.import_module("img-stitching_amd")
ctx = pano.Context(1, 64, 64, scale=50.0, num_bands=0, device=0)
ctx.set_camera(0, [50.0, 0, 32, 0, 50.0, 32, 0, 0, 1], [1.0, 0, 0, 0, 1, 0, 0, 0, 1]); ctx.prepare()
uid = pano.Context.rccl_unique_id()
try:
    comm = ctx.rccl_comm_create(uid, 1, 0); print("comm ok", ctx.rccl_comm_count(comm)); ctx.rccl_comm_destroy(comm)
except Exception as e:
    print("FAILED", e)

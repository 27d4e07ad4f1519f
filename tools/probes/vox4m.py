"""voxel build of the 4 M-point map (40 tiles of the 100 k bench cloud): wall time per build, leaves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.cuda.init()
import lvx, synth  # noqa: E402
ctx = lvx.Context(0)
cloud = synth.make_voxel_cloud(seed=2, n=100_000)
for tiles in (4, 40):
    big = synth.tile_voxel_cloud(cloud, tiles)
    t = lvx.upstream_bench(ctx, "voxel_build", (big, 0.5), reps=10)
    info = ctx.voxel_info()
    print("voxel_build %8d points: %.1f us, %d leaves" % (len(big), 1e6 * t, info["n_leaves"] if isinstance(info, dict) else getattr(info, "n_leaves", -1)))
ctx.close()

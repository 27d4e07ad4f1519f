"""Probe: one sweep of lvx_scan_register_batch_d (config 1: 16 rings x 1800 points), repeated; for rocprofv3 --kernel-trace timelines."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "lvi-exc_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
torch.cuda.init()
import lvx, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sw = synth.make_vlp16_sweep(seed=1)
pts = np.ascontiguousarray(sw, dtype=lvx.RS_POINT)
allp = np.concatenate([pts] * S)
off = np.arange(S + 1, dtype=np.int32) * len(pts)
pd = torch.from_numpy(allp.view(np.uint8)).cuda()
g = lvx.Context(0)
for _ in range(5):
    lvx.scan_register_batch_d(g, pd.data_ptr(), off, 16, 0.3)
t0 = time.perf_counter()
n = 50
for _ in range(n):
    lvx.scan_register_batch_d(g, pd.data_ptr(), off, 16, 0.3)
print("scan_register_batch_d %d sweeps: %.1f us per call" % (S, 1e6 * (time.perf_counter() - t0) / n))

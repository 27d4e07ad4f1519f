"""Surfel association alone, 64 scans x 2 000 surfels resident on the device, surfel map prepared once (as bench.py's assoc_metric): ms per call and Gpts/s.
Quick iteration on k_assoc_hits / k_assoc_select: run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lvi-exc_amd"))
import lvx  # noqa: E402
import synth  # noqa: E402

if __name__ == "__main__":
    torch.cuda.init()
    ctx = lvx.Context(0)
    scan, p4, bmin, bmax = synth.make_assoc_problem(seed=5, H=16, W=1800, n_planes=2000)
    H, W, P = scan.shape[0], scan.shape[1], len(p4)
    dev = torch.device("cuda", 0)
    pl = torch.from_numpy(np.concatenate([p4.ravel(), bmin.ravel(), bmax.ravel()])).to(dev)
    ctx._ck(ctx._l.lvx_surfel_map_prepare_d(ctx._h, C.c_int(P), C.c_void_p(pl.data_ptr())))
    for S in (4, 16, 64):
        local = torch.from_numpy(np.ascontiguousarray(scan, np.float32)).to(dev).unsqueeze(0).repeat(S, 1, 1, 1).contiguous()
        local[:, :, :, 0] += 1e-3 * torch.arange(0, S, device=dev, dtype=torch.float32).view(-1, 1, 1)
        flags = torch.empty((S, H * W), dtype=torch.int32, device=dev)
        step = lambda: ctx._ck(ctx._l.lvx_surfel_assoc_batch_d(ctx._h, C.c_int(S), C.c_int(H), C.c_int(W), C.c_void_p(local.data_ptr()), C.c_int(P), C.c_void_p(pl.data_ptr()), C.c_double(0.05), C.c_int(2),
                                                               C.c_void_p(flags.data_ptr())))
        for _ in range(3):
            step()
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        ctx.synchronize(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
        print("surfel_assoc %2d scans: %.3f ms per call, %.2f Gpts/s, %d associated" % (S, 1e3 * t, S * H * W / t / 1e9, int((flags >= 0).sum().item())))
    ctx._ck(ctx._l.lvx_surfel_map_release(ctx._h))

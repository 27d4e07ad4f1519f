import sys, time
sys.path.insert(0,'/root/repo/lvi-exc_amd'); sys.path.insert(0,'/root/repo')
import numpy as np, synth, lvx
P=synth.make_bench_problem(seed=4)
g=lvx.Context(0); lvx.load_problem(g,P,lvx.LOCK_LIDAR_TAU|lvx.LOCK_CAM_TAU)
print(g.layout())
g.set_profiling(True)
t=time.time(); x,s=g.lm_solve(P['state0'],max_iterations=4); dt=time.time()-t
ms,n=g.kernel_ms()
print('lm 4 iters wall',dt, s['termination'], s['cost_history'], s['accepted'])
names=["gyro","accel","prior","surfel","reproj","camsurf","fold","solve"]
print({names[i]:(ms[i]/max(1,n[i]),int(n[i])) for i in range(8)})

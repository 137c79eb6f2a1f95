import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
sys.argv = ['bench.py']
import bench
from slam_amd import pipeline, dist as sdist
from slam_amd.synth import SynthConfig, SynthSequence
F = 3450
scfg = SynthConfig(n_frames=F, n_features=2000, seed=1234)
desc, xy, count = SynthSequence(scfg).frames(0, F)
n_max = scfg.n_max
d = (torch.from_numpy(desc.reshape(2*F,n_max,64)).cuda(), torch.from_numpy(xy.reshape(2*F,n_max,2)).cuda(), torch.from_numpy(count.reshape(2*F)).cuda())
f0, f1 = pipeline.consecutive_pairs(F)
seeds = 1000 + np.arange(F-1)
trk = pipeline.ShardTracker(F, F-1, n_max)
rel_buf = np.empty((F, 3, 4)); rel_buf[0] = np.eye(3,4)
def step():
    res = trk.run(*d, f0, f1, seeds, unpack_mask=False, copy=False)
    rel_all, _ = sdist.gather_shard_results(res.pose, res.stats, F-1)
    rel_buf[1:] = rel_all[:F-1]
    bad = ~np.isfinite(rel_buf.reshape(F, 12)).all(axis=1)
    if bad.any(): rel_buf[bad] = np.eye(3,4)
    return res, pipeline.compose_global(rel_buf)
for _ in range(5): step()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): res,_ = step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/20*1e3, "call_wall", res.timing_ms[5])
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)

"""Tracked-set sizes, iteration and draw counts of BASELINE configs[4] (loop-closure candidates), revisits against
unrelated frames: what the later RANSAC rounds of that configuration are made of."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import pipeline
from slam_amd.config import HotPathConfig
from slam_amd.synth import SynthConfig, SynthSequence, loop_candidate_pairs
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3450
scfg = SynthConfig(n_frames=F, n_features=2000, seed=1234)
desc, xy, count = SynthSequence(scfg).frames(0, F)
n_max = scfg.n_max
g0, g1 = loop_candidate_pairs(F)
P = len(g0)
trk = pipeline.ShardTracker(F, P, n_max, HotPathConfig(), n_chunks=1)
res = trk.run(torch.from_numpy(desc.reshape(2*F,n_max,64)).cuda(), torch.from_numpy(xy.reshape(2*F,n_max,2)).cuda(),
              torch.from_numpy(count.reshape(2*F)).cuda(), g0.astype(np.int32), g1.astype(np.int32), 9000+np.arange(P), unpack_mask=False)
near = np.arange(P) % 5 < 2
st = res.stats
for name, m in (("revisits", near), ("unrelated", ~near)):
    T, it, dr, sup = st[m,0], st[m,2], st[m,3], st[m,1]
    print(f"{name:10s} pairs {m.sum():6d}  T mean {T.mean():7.1f} p10/50/90 {np.percentile(T,[10,50,90])}  supporters mean {sup.mean():7.1f}  iterations mean {it.mean():7.1f}  draws mean {dr.mean():7.1f} max {dr.max()}")
print("timing", res.timing_ms)

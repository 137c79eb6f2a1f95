import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
from slam_amd import pipeline
from slam_amd.config import HotPathConfig
from slam_amd.synth import SynthConfig, SynthSequence
F=3450
scfg = SynthConfig(n_frames=F, n_features=2000, seed=1234)
desc, xy, count = SynthSequence(scfg).frames(0, F)
n_max = scfg.n_max
f0, f1 = pipeline.consecutive_pairs(F)
trk = pipeline.ShardTracker(F, F-1, n_max, HotPathConfig())
res = trk.run(torch.from_numpy(desc.reshape(2*F,n_max,64)).cuda(), torch.from_numpy(xy.reshape(2*F,n_max,2)).cuda(), torch.from_numpy(count.reshape(2*F)).cuda(), f0, f1, 1000+np.arange(F-1))
d = res.stats[:,3]
print("draws: mean", d.mean(), "percentiles 50/90/95/99/max", np.percentile(d,[50,90,95,99,100]))
for k in (8,12,16,20,24,32,48,64): print(k, (d<=k).mean())
print("T mean", res.stats[:,0].mean(), "supp mean", res.stats[:,1].mean())

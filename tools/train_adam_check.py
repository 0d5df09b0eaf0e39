import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from alphazero_rs_amd import engine as E
from net_ref import layout
from train_ref import adam_reference
import test_train_gpu as T
C = 128; T.C = C
e = E.Engine(device=0, max_batch=1024, net_channels=C)
e.set_option("train_dropout_e6", 0)
p = T.perturbed_params(e, 3, seed=11)
batches = [T.make_batch(32, seed=200 + i) for i in range(6)]
e.train_begin(3)
losses = [e.train_step(*bt, apply=True)[0] for bt in batches]
e.train_end(4)
got = e.net_get_params(4).astype(np.float64)
ref, rl = adam_reference(p, C, batches)
for a, b in zip(losses, rl): print("loss", a, b)
for k, (o, shp) in layout(C)[0].items():
    n = int(np.prod(shp))
    d = np.abs(got[o:o + n] - ref[o:o + n]); mv = np.abs(ref[o:o + n] - p[o:o + n])
    print(f"{k:10s} moved max {mv.max():.2e} diff median {np.median(d):.2e} q99 {np.quantile(d, .99):.2e} q999 {np.quantile(d, .999):.2e} max {d.max():.2e} frac>1e-4 {np.mean(d > 1e-4):.4f}")

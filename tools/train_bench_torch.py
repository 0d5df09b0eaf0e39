"""Time the PyTorch-ROCm autograd trainer (alphazero-rs_amd/trainer.py) on the same workload as tools/train_bench.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphazero_rs_amd import trainer as T
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = steps * batch
rng = np.random.default_rng(0)
boards = (rng.random((n, 2, 6, 7)) < 0.2).astype(np.float32)
pis = rng.dirichlet(np.ones(7), n).astype(np.float32)
vs = rng.choice([-1.0, 1.0], n).astype(np.float32)
_, total = T.layout(C)
p = (rng.normal(size=total) * 0.02).astype(np.float32)
for k, (o, shp) in T.layout(C)[0].items():
    if k.endswith("_bn"):
        p[o:o + shp[1]] = 1; p[o + 3 * shp[1]:o + 4 * shp[1]] = 1
tr = T.Trainer(channels=C, batch_size=batch, epochs=1)
tr.train(p, boards[:4 * batch], pis[:4 * batch], vs[:4 * batch])
torch.cuda.synchronize(); t0 = time.time()
tr.train(p, boards, pis, vs)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"torch trainer C={C} batch={batch} steps={steps}: {dt / steps * 1e3:.3f} ms/step, {n / dt:.0f} samples/s")

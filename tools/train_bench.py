"""Time az_net_train (NNet::train on the device): C=512, batch 64, `steps` optimisation steps in one epoch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_rs_amd import engine as E

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = steps * batch
rng = np.random.default_rng(0)
boards = (rng.random((n, 2, 6, 7)) < 0.2).astype(np.float32)
pis = rng.dirichlet(np.ones(7), n).astype(np.float32)
vs = rng.choice([-1.0, 1.0], n).astype(np.float32)
e = E.Engine(device=0, max_batch=1024, net_channels=C)
e.net_init_random(0, seed=1)
e.set_option("train_epochs", 1); e.set_option("train_batch", batch)
if os.environ.get("TRAIN_GRAPH"):
    e.set_option("train_graph", int(os.environ["TRAIN_GRAPH"]))
if os.environ.get("TRAIN_FWD_DMA"):
    e.set_option("train_fwd_dma", int(os.environ["TRAIN_FWD_DMA"]))
if os.environ.get("TRAIN_FORK"):
    e.set_option("train_fork", int(os.environ["TRAIN_FORK"]))
if os.environ.get("TRAIN_G3RING"):
    e.set_option("train_gemm3_ring", int(os.environ["TRAIN_G3RING"]))
if os.environ.get("TRAIN_FWD_X3"):
    e.set_option("train_fwd_x3", int(os.environ["TRAIN_FWD_X3"]))
if os.environ.get("TRAIN_WGRAD_TR"):
    e.set_option("train_wgrad_tr", int(os.environ["TRAIN_WGRAD_TR"]))
if os.environ.get("TRAIN_IMPLICIT"):
    e.set_option("train_implicit", int(os.environ["TRAIN_IMPLICIT"]))
if os.environ.get("TRAIN_GEMM"):
    e.set_option("train_gemm", int(os.environ["TRAIN_GEMM"]))
e.train(0, 1, boards[: 4 * batch], pis[: 4 * batch], vs[: 4 * batch])     # warm-up (allocations, code load)
t0 = time.time()
hist = e.train(0, 1, boards, pis, vs)
dt = time.time() - t0
flop = 3 * 2 * 164_493_312 * (C / 512) ** 2 * batch    # ~3x the forward MACs (forward + dgrad + wgrad), C^2 scaling approx.
import hashlib
print("params sha", hashlib.sha256(e.net_get_params(1).tobytes()).hexdigest()[:16])
print(f"C={C} batch={batch} steps={steps}: {dt / steps * 1e3:.3f} ms/step, {n / dt:.0f} samples/s, ~{flop / (dt / steps) / 1e12:.1f} TFLOP/s f32, loss {hist}")

"""Per-tensor gradient error of one az_net_train_step against the float64 autograd reference (diagnostic)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from alphazero_rs_amd import engine as E
from net_ref import layout
from train_ref import step_reference
import test_train_gpu as T

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = int(sys.argv[2]) if len(sys.argv) > 2 else 37
T.C = C
e = E.Engine(device=0, max_batch=1024, net_channels=C)
e.set_option("train_dropout_e6", 0)
if os.environ.get("TRAIN_FWD_X3"):
    e.set_option("train_fwd_x3", int(os.environ["TRAIN_FWD_X3"]))
if os.environ.get("TRAIN_WGRAD_TR"):
    e.set_option("train_wgrad_tr", int(os.environ["TRAIN_WGRAD_TR"]))
if os.environ.get("TRAIN_IMPLICIT"):
    e.set_option("train_implicit", int(os.environ["TRAIN_IMPLICIT"]))
if os.environ.get("TRAIN_GEMM"):
    e.set_option("train_gemm", int(os.environ["TRAIN_GEMM"]))
p = T.perturbed_params(e, 1, seed=b)
boards, pis, vs = T.make_batch(b, seed=100 + b)
e.train_begin(1)
(lp, lv), g = e.train_step(boards, pis, vs, apply=False, want_grads=True)
rlp, rlv, rg, _ = step_reference(p, C, boards, pis, vs)
print("loss", lp, rlp, lv, rlv)
for k, (o, shp) in layout(C)[0].items():
    n = int(np.prod(shp))
    a, r = g[o:o + n].astype(np.float64), rg[o:o + n]
    print(f"{k:10s} |ref| {np.linalg.norm(r):.3e} rel err {np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-30):.3e} max abs {np.abs(a - r).max():.3e}")
# step timing
e.train_begin(1)
ts = []
for i in range(30):
    t0 = time.time()
    e.train_step(boards, pis, vs, apply=True)
    ts.append((time.time() - t0) * 1e3)
print("ms per az_net_train_step call (incl. host copies + sync):", " ".join(f"{x:.2f}" for x in ts))

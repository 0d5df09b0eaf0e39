"""A few full-batch net forwards with the default kernels (target of rocprofv3 --pmc passes)."""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from alphazero_rs_amd import engine as azeng
from _states import random_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
e = azeng.Engine(device=0, max_batch=B, diag=True)
e.net_init_random(0, 1)
if len(sys.argv) > 3:
    e.set_option("gemm_variant", int(sys.argv[3]))
for kv in filter(None, os.environ.get("OPT", "").split(",")):      # OPT=key=value,...: e.g. conv3_pp=1 (diagnostic library)
    k, v = kv.split("=")
    e.set_option(k, int(v))
uniq = random_states(512, 3)
states = uniq[np.random.default_rng(0).integers(0, 512, B)]
for _ in range(reps):
    e.predict_states(states, 0)
print("done")

import sys, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R,'tests'))
from alphazero_rs_amd import engine as azeng
from oracle import oracle_py as orc
from net_ref import *
from test_net_gpu import random_states
e = azeng.Engine(device=0, max_batch=2048)
for seed, B in ((1, 64), (2, 200)):
    params = random_params(512, seed)
    e.net_set_params(2, params)
    st = random_states(orc, B, seed)
    boards = np.stack([orc.c4_features(int(m), int(t)) for m, t in st])
    pi, v = e.predict_states(st, 2)
    rpi, rv = forward_ref(params, boards, 512, True)
    fpi, fv = forward_ref(params, boards, 512, False)
    print('B', B, 'emul: dpi', np.abs(pi-rpi).max(), 'dv', np.abs(v-rv).max(), '| f32: dpi', np.abs(pi-fpi).max(), 'dv', np.abs(v-fv).max(),
          '| ref emul-vs-f32 dpi', np.abs(rpi-fpi).max(), 'dv', np.abs(rv-fv).max(), '| pi range', pi.min(), pi.max(), 'v range', v.min(), v.max())
    print(' mean abs: emul', np.abs(pi-rpi).mean(), np.abs(v-rv).mean())
e.net_init_random(3, 5)
st = random_states(orc, 64, 9)
boards = np.stack([orc.c4_features(int(m), int(t)) for m, t in st])
p3 = e.net_get_params(3)
pi, v = e.predict_states(st, 3)
rpi, rv = forward_ref(p3, boards, 512, True)
fpi, fv = forward_ref(p3, boards, 512, False)
print('glorot init: emul dpi', np.abs(pi-rpi).max(), 'dv', np.abs(v-rv).max(), 'f32 dpi', np.abs(pi-fpi).max(), 'dv', np.abs(v-fv).max(), 'pi', pi[0], 'v', v[:4])

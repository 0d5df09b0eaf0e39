import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from alphazero_rs_amd import engine as azeng
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prof = int(sys.argv[3]) if len(sys.argv) > 3 else 1
e = azeng.Engine(device=0, max_batch=max(G, 128), profile=bool(prof))
e.net_init_random(0, 1)
e.net_set_kind(1, azeng.NET_STUB)
for model, name in ((1, 'stub'), (0, 'conv')):
    e.reset_stats()
    t = time.time()
    r = e.selfplay(n_games=G, num_sims=sims, model_id=model, seed=1, want_boards=False)
    dt = time.time() - t
    st = e.stats()
    print(f"{name}: G={G} sims={sims} time {dt:.3f}s games/s {G/dt:.1f} plies {r['game_len'].mean():.1f} sims/s {st['simulations']/dt:.3e} "
          f"exp/s {st['expansions']/dt:.3e} leaf_evals {st['leaf_evals']} dbar {st['depth_sum']/max(1,st['simulations']):.2f}")
    if st['net_launches']:
        print(f"   conv2: {st['net_conv2_flops']/st['net_conv2_ms']/1e9:.1f} TFLOP/s over {st['net_launches']} launches, avg {st['net_conv2_ms']/st['net_launches']:.3f} ms; "
              f"net total {st['net_total_flops']/st['net_total_ms']/1e9:.1f} TFLOP/s, net ms {st['net_total_ms']:.1f}, tree ms {st['tree_ms']:.1f}, wall ms {dt*1e3:.1f}")
    else:
        print(f"   tree ms {st['tree_ms']:.1f}")

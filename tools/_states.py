"""Random legal Connect Four positions for the tools/ scripts (product-side helpers only; the CPU checker is not used here)."""
import numpy as np
from alphazero_rs_amd.engine import c4_play


def _four(b):
    for d in (1, 7, 6, 8):
        m = b & (b >> d)
        if m & (m >> (2 * d)):
            return True
    return False


def random_states(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        s = (0, 0)
        for _ in range(int(rng.integers(0, 30))):
            mask = s[0] | s[1]
            legal = [c for c in range(7) if not (mask >> (c * 7 + 5)) & 1]
            if not legal:
                break
            nxt = c4_play(s[0], s[1], int(rng.choice(legal)))
            if _four(nxt[1]) or (nxt[0] | nxt[1]) == 0x0FDFBF7EFDFBF:
                break
            s = nxt
        out.append(s)
    return np.array(out, dtype=np.uint64)

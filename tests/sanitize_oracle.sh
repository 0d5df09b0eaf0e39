#!/bin/bash
# CPU sanitizers over the oracle (test infrastructure): the known-answer tests, and the C-API paths of the lock-step schedule, the arena
# and the second game through ctypes, built with -fsanitize=address,undefined into a scratch directory (nothing in the tree is touched).
# GPU sanitizers are not available on this pool.  Usage: bash tests/sanitize_oracle.sh   (about a minute)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
D="$(mktemp -d /tmp/az_san.XXXXXX)"
mkdir -p "$D/oracle"
cp "$R"/oracle/*.py "$D/oracle/"
FLAGS="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -pthread -I $R/oracle"
g++ $FLAGS "$R/oracle/test_oracle.cpp" -o "$D/test_oracle"
"$D/test_oracle" | tail -1
g++ $FLAGS -fPIC -shared "$R/oracle/az_oracle_capi.cpp" -o "$D/oracle/libaz_oracle.so"
cat > "$D/drive.py" <<PY
import sys, numpy as np
sys.path.insert(0, "$D")
from oracle import oracle_py as orc
orc.build = lambda force=False: orc.LIB_PATH          # the sanitized library built above, not a rebuild
r = orc.selfplay(24, 48, net_kind=orc.NET_HASH, salt=5, seed=3, threads=4, sim_threads=4)
w = orc.arena_ex(12, 60, net_kind=orc.NET_HASH, salt=9, seed=2, new_model_id=0, old_model_id=1, threads=4, sim_threads=3)
t = orc.Tree(96, net_kind=orc.NET_HASH, salt=1, threads=8)
s = (0, 0)
for mv in range(20):
    pi, cnt, q = t.get_action_prob(s[0], s[1], 0.0, seed=1, game_id=2)
    s = orc.c4_play(s[0], s[1], int(np.argmax(cnt)))
    if orc.c4_ended(*s) != 0.0:
        break
r3 = orc.selfplay(16, 24, net_kind=orc.NET_HASH, salt=5, seed=6, threads=4, game_kind=orc.GAME_CONNECT3, sim_threads=4)
print("sanitized C-API paths ran:", int(r["game_len"].sum()), w[0].tolist(), t.stats()["abandoned"], int(r3["game_len"].sum()))
PY
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" python3 "$D/drive.py"
rm -rf "$D"

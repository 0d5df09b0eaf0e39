"""Per-layer GEMM time for forced LDS-DMA ring tiles (az_set_option "ring_tile") as a function of the row count.
  rocprofv3 --kernel-trace --output-format csv -d D -o rt -- python3 tools/ring_tiles.py run
  python3 tools/ring_tiles.py fold D/rt_kernel_trace.csv"""
import sys, os, csv, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
ROWS = [1024, 1536, 2048, 2560, 2816, 3072, 3328, 3584, 4096, 4608, 5632, 6656, 8192]
TILES = [0, 642, 644, 962, 964, 1282, 1284, 1602, 1922]
if sys.argv[1] == "run":
    from alphazero_rs_amd import engine as azeng
    from _states import random_states
    e = azeng.Engine(device=0, max_batch=8192, diag=True)
    e.net_init_random(0, 1)
    uniq = random_states(8192, 3)
    for L in ROWS:
        for t in TILES:
            for layer in (3, 4, 5):
                e.set_option("ring_tile", layer * 10000 + t)
            for _ in range(3):
                e.predict_states(uniq[:L], 0)
else:
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    fw, cur = [], []
    for r in rows:
        name = r["Kernel_Name"]
        if "az::" not in name: continue
        short = re.sub(r"\(.*", "", name.replace("void ", "").replace("az::", ""))
        cur.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        if short.startswith("k_heads"):
            fw.append(cur); cur = []
    fw = fw[-len(ROWS) * len(TILES) * 3:]
    i = 0
    for L in ROWS:
        line = {3: [], 4: [], 5: []}
        for t in TILES:
            f = fw[i + 2]; i += 3
            ks = [x for x in f if x[0].startswith("k_gemm")]
            for layer, (n, us) in zip((3, 4, 5), ks):
                line[layer].append(f"{t}:{us:.0f}")
        for layer in (3, 4, 5):
            print(f"rows {L:5d} layer {layer}: " + "  ".join(line[layer]))

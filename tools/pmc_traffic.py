"""Fold two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of the bench command into per-kernel HBM bytes.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <bench_line.json> <out.json>
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are reported in
KB; FETCH_SIZE counts half the bytes of 16-B-per-lane streaming reads, so hbm_bytes = (2*FETCH + WRITE) * 1024.
"""
import csv, json, re, sys
from collections import defaultdict


def fold(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
            tot[name] += float(r["Counter_Value"])
            n[name] += 1
    return tot, n


def main():
    fetch, nf = fold(sys.argv[1], "FETCH_SIZE")
    write, nw = fold(sys.argv[2], "WRITE_SIZE")
    line = json.loads([l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1])
    # rows the net really ran (leaf de-duplication): every per-leaf figure below is per EXECUTED row
    per_sec = line["leaf_rows"]["executed_per_sec"] if "leaf_rows" in line else line["leaf_evals_per_sec"]
    leaf_evals = per_sec * line["ms_per_step"] * line["steps"] / 1e3
    conv = [k for k in fetch if "k_heads" in k]      # one k_heads launch per forward, whatever kernels the layers use
    launches = nf[conv[0]] if conv else max(nf.values())
    per_launch_leaves = leaf_evals / launches
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --episodes 16384 "
                      "--no-cpu-baseline --no-profile --no-train-probe --no-aux (two separate passes; tools/collect_profiles.sh)",
           "units": "FETCH_SIZE / WRITE_SIZE in KB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports "
                    "half the bytes of 16-B-per-lane streaming reads, MI355X_MICROARCH.md HBM section; WRITE_SIZE of the 8-B-per-lane "
                    "epilogue stores is uncalibrated)",
           "avg_leaves_per_launch": per_launch_leaves, "kernels": {},
           # what the fold describes: bench.py flags it as stale when the kernel sources have changed since
           "csrc_sha": line.get("config", {}).get("csrc_sha"), "git_sha": (sys.argv[5] if len(sys.argv) > 5 else None)}
    for k in fetch:
        if nf[k] == 0:
            continue
        f_kb, w_kb = fetch[k] / nf[k], (write[k] / nw[k] if nw.get(k) else 0.0)
        b = (2 * f_kb + w_kb) * 1024
        out["kernels"][k] = {"launches": nf[k], "fetch_kb_per_launch": f_kb, "write_kb_per_launch": w_kb, "hbm_bytes_per_launch": b,
                             "hbm_bytes_per_leaf": b / per_launch_leaves}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, v in out["kernels"].items():
        if "gemm" in k or "conv" in k or "backup_select" in k:
            print(f"{k:45s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch {v['hbm_bytes_per_leaf'] / 1e3:8.1f} KB/leaf")


if __name__ == "__main__":
    main()

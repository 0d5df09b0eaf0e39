"""The C-ABI library builds for gfx950, loads on a machine without a GPU, and exports exactly the symbols
include/az_engine.h declares.  No compute calls here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(engine_mod):
    lib = ctypes.CDLL(engine_mod.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/az_engine.h but not exported"
    assert sorted(engine_mod.EXPORTS) == decl


def test_struct_layouts_match_header(engine_mod):
    assert ctypes.sizeof(engine_mod.az_config) == 16
    assert ctypes.sizeof(engine_mod.az_selfplay_params) == 64
    assert ctypes.sizeof(engine_mod.az_samples) == 64
    assert ctypes.sizeof(engine_mod.az_arena_params) == 48
    assert ctypes.sizeof(engine_mod.az_stats) == 17 * 8


def test_no_gpu_fails_loudly(engine_mod):
    """Without a usable HIP device az_create reports AZ_ERR_HIP; nothing falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    try:
        engine_mod.Engine(device=0)
    except engine_mod.AzError as ex:
        assert ex.status == 3
    else:
        raise AssertionError("az_create succeeded without a GPU")


def test_product_code_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "alphazero-rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                # comments may NAME the oracle twin of a function; code may not include, import, link or load it
                assert not re.search(r'#\s*include\s*["<][^">]*oracle', text), f
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", text, flags=re.M), f
                assert not re.search(r"(CDLL|dlopen|LoadLibrary)\([^)]*oracle", text), f
    assert "oracle" not in open(os.path.join(ROOT, "include", "az_engine.h")).read().lower()


def test_host_helpers_match_oracle(engine_mod, oracle):
    import numpy as np
    rng = np.random.default_rng(0)
    for _ in range(200):
        s = (0, 0)
        for _ in range(int(rng.integers(0, 20))):
            vm = oracle.c4_valid_mask(*s)
            a = int(rng.choice([c for c in range(7) if (vm >> c) & 1]))
            n1, n2 = engine_mod.c4_play(s[0], s[1], a), oracle.c4_play(s[0], s[1], a)
            assert n1 == n2
            if oracle.c4_ended(*n2) != 0.0:
                break
            s = n2
        assert np.array_equal(engine_mod.c4_features(*s), oracle.c4_features(*s))

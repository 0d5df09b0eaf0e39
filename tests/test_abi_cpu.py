"""The C-ABI library builds for gfx950, loads on a machine without a GPU, and exports exactly the symbols
include/az_engine.h declares.  No compute calls here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(az_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_are_exported(engine_mod):
    lib = ctypes.CDLL(engine_mod.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/az_engine.h but not exported"
    assert sorted(engine_mod.EXPORTS) == decl


def test_struct_layouts_match_header(engine_mod):
    assert ctypes.sizeof(engine_mod.az_config) == 20
    assert ctypes.sizeof(engine_mod.az_selfplay_params) == 64
    assert ctypes.sizeof(engine_mod.az_samples) == 64
    assert ctypes.sizeof(engine_mod.az_arena_params) == 80
    assert ctypes.sizeof(engine_mod.az_stats) == 36 * 8


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


def test_cpp_host_mirror_rules_and_shuffle(oracle, tmp_path):
    """include/az_host.hpp without an engine: the Connect Four rules on bitboards agree with the oracle on the
    reference's diagonal game (connect_four_game.rs:244-264), and the counter-RNG shuffle is the Python host's."""
    import json
    import subprocess
    from alphazero_rs_amd import coach
    exe = os.path.join(tmp_path, "test_host_cpu")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_host_cpu.cpp"), "-o", exe])
    for n, seed, it in ((10, 3, 2), (1000, 123456789, 17), (1, 0, 0)):
        got = json.loads(subprocess.run([exe, str(n), str(seed), str(it)], check=True, stdout=subprocess.PIPE, text=True).stdout)
        assert got["perm"] == coach.shuffle_permutation(n, seed, it).tolist()
    # replay the diagonal game on the oracle's canonical bitboards: player +1 moves on even plies
    s, ended = (0, 0), []
    for ply, a in enumerate([0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3]):
        s = oracle.c4_play(s[0], s[1], a)               # canonical: (side to move, other) after the move
        e = oracle.c4_ended(*s)                         # from the side to move's view: -1 = the mover just won
        ended.append(0 if e == 0 else (1 if (e < 0) == (ply % 2 == 0) else -1))
    assert got["ended"] == ended and ended[-1] == 1 and sum(ended[:-1]) == 0
    # after 11 plies player -1 is to move: canonical (mine, theirs) = (minus, plus)
    assert (got["canon_plus"], got["canon_minus"]) == s == (got["minus"], got["plus"])
    assert got["valid"] == [(oracle.c4_valid_mask(*s) >> c) & 1 for c in range(7)] and got["features_sum"] == 11


# ---- the Rust shim crate (rust/az-engine-sys/src/lib.rs) against the header: it cannot be compiled here, so it is parsed ----
_C2RUST = {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8",
           "int8_t": "i8", "float": "f32", "double": "f64", "char": "c_char", "az_status": "c_int", "az_net_kind": "c_int",
           "az_engine": "az_engine", "az_tree": "az_tree", "az_config": "az_config", "az_stats": "az_stats",
           "az_selfplay_params": "az_selfplay_params", "az_samples": "az_samples", "az_arena_params": "az_arena_params"}


def _rust_type(ctype):
    """'const float*' -> '*const f32', 'az_tree**' -> '*mut *mut az_tree', 'uint64_t out_wld[3]' handled by the caller."""
    c = ctype.strip()
    const = c.startswith("const ")
    c = c[6:].strip() if const else c
    stars = c.count("*")
    base = _C2RUST[c.replace("*", "").strip()]
    if stars == 0:
        return base
    out = base
    for i in range(stars):
        out = ("*const " if (const and i == 0) else "*mut ") + out
    return out


def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    fns = {}
    for ret, name, args in re.findall(r"\b([A-Za-z_0-9]+\s*\**)\s*\b(az_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", hdr):
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            m = re.match(r"(.*?)([A-Za-z_0-9]+)(\[[A-Za-z_0-9]*\])?$", a)
            ctype = m.group(1).strip() + ("*" if m.group(3) else "")
            params.append(_rust_type(ctype))
        r = ret.replace(" ", "")
        fns[name] = (None if r == "void" else ("*const c_char" if r in ("char*", "constchar*") else _rust_type(ret)), params)
    # `const char* az_last_error(...)`: the regex above sees `char*` after `const`
    return fns


def _header_structs():
    hdr = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef struct [a-z_]+ \{(.*?)\}\s*([a-z_]+);", hdr, flags=re.S):
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            m = re.match(r"(.*?)([A-Za-z_0-9]+)(\[(\d+)\])?$", decl)
            rt = _rust_type(m.group(1))
            fields.append((m.group(2), "[%s; %s]" % (rt, m.group(4)) if m.group(3) else rt))
        out[name] = fields
    return out


def test_rust_shim_matches_the_header():
    """rust/az-engine-sys/src/lib.rs binds every export of include/az_engine.h with the same argument count and types, and
    its #[repr(C)] structs have the header's fields in the header's order."""
    src = open(os.path.join(ROOT, "rust", "az-engine-sys", "src", "lib.rs")).read()
    ext = src[src.index('extern "C" {'):]
    ext = ext[:ext.index("\n}\n")]
    ext = re.sub(r"//[^\n]*", "", ext)
    rust = {}
    for name, args, ret in re.findall(r"pub fn (az_[a-z_0-9]+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        rust[name] = (ret.strip() if ret else None, params)
    hdr = _header_functions()
    assert sorted(rust) == sorted(hdr) == declared_symbols()
    for name, (ret, params) in hdr.items():
        rret, rparams = rust[name]
        assert rparams == params, (name, rparams, params)
        assert rret == ret, (name, rret, ret)
    structs = _header_structs()
    assert set(structs) == {"az_config", "az_stats", "az_selfplay_params", "az_samples", "az_arena_params"}
    for sname, fields in structs.items():
        m = re.search(r"pub struct %s \{(.*?)\}" % sname, src, flags=re.S)
        rfields = [(n, t.strip()) for n, t in re.findall(r"pub ([a-z_0-9]+):\s*(\[[^\]]+\]|[^,}]+)", m.group(1))]
        assert rfields == fields, (sname, rfields, fields)
    # the NNet / AsyncMcts surfaces the reference's callers use
    for needle in ("pub fn new<P: AsRef<Path>>", "pub fn predict(", "pub fn train(", "pub fn from_state(", "pub fn get_action_prob("):
        assert needle in src, needle


def test_integration_md_matches_the_header():
    """INTEGRATION.md is what a maintainer copies from: every `pub fn az_*` and `pub struct az_*` in its rust blocks matches
    include/az_engine.h (names, argument counts and types, field order), and every struct literal names exactly the struct's fields."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = "\n".join(re.findall(r"```rust\n(.*?)```", md, flags=re.S))
    code = re.sub(r"//[^\n]*", "", blocks)
    hdr, structs = _header_functions(), _header_structs()
    seen_fn = 0
    for name, args, ret in re.findall(r"pub fn (az_[a-z_0-9]+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", code, flags=re.S):
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        assert name in hdr, name
        assert params == hdr[name][1], (name, params, hdr[name][1])
        assert (ret.strip() if ret else None) == hdr[name][0], name
        seen_fn += 1
    assert seen_fn >= 15
    seen_st = 0
    for sname, body in re.findall(r"pub struct (az_[a-z_]+) \{\n(.*?)\n\}", code + "\n", flags=re.S) + \
            re.findall(r"pub struct (az_[a-z_]+) \{([^\n]*?)\}", code):
        if sname in ("az_engine", "az_tree"):
            continue
        rfields = [(n, t.strip()) for n, t in re.findall(r"pub ([a-z_0-9]+):\s*(\[[^\]]+\]|[^,}]+)", body)]
        assert rfields == structs[sname], (sname, rfields, structs[sname])
        seen_st += 1
    assert seen_st >= 4
    # struct literals (`az_selfplay_params { field: .., }` without `..base`): all fields, in any order
    for sname, body in re.findall(r"= (az_[a-z_]+) \{(.*?)\};", code, flags=re.S):
        if ".." in re.sub(r"\[[^\]]*\]", "", body).replace("...", ""):
            continue
        names = set(re.findall(r"(?:^|[,{\s])([a-z_0-9]+)\s*(?::|,|$)", re.sub(r"\([^()]*\)|\[[^\]]*\]", "", body)))
        want = {n for n, _ in structs[sname]}
        assert want <= names, (sname, sorted(want - names))


def test_tools_and_examples_compile():
    """Every measurement helper and example parses (they only run on the GPU box, so a syntax error would otherwise surface there)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "examples", "*.py")) +
                   [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")])
    assert len(files) > 20
    for f in files:
        compile(open(f).read(), f, "exec")

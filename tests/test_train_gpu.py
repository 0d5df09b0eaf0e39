"""NNet::train on the device (src/nnet.rs:38; csrc/az_train.hip) against a float64 autograd reference.

The trainer is a floating-point kernel set, so the checker is PyTorch (tests/train_ref.py, double precision on the
CPU) and the bars are stated per test.  Everything goes through the C ABI (az_net_train_begin / _step / _end,
az_net_train).  The reference project has no runnable training code to pin against (its TF1 script is broken,
SURVEY.md B11): parity is with the recipe, restated in train_ref.py.
"""
import numpy as np
import pytest

from net_ref import layout
from train_ref import adam_reference, mix64, step_reference

pytestmark = pytest.mark.gpu

C = 128


def make_batch(b, seed):
    rng = np.random.default_rng(seed)
    boards = np.zeros((b, 2, 6, 7), np.float32)
    for j in range(b):                                   # random legal-looking fillings: gravity respected per column
        for col in range(7):
            h = rng.integers(0, 7)
            owner = rng.integers(0, 2, size=h)
            for r in range(h):
                boards[j, owner[r], 5 - r, col] = 1.0
    logits = rng.normal(size=(b, 7))
    pis = np.exp(logits) / np.exp(logits).sum(axis=1, keepdims=True)
    vs = rng.choice([-1.0, 1.0, 1e-4], size=b)
    return boards, pis.astype(np.float32), vs.astype(np.float32)


def perturbed_params(engine, model_id, seed):
    """Glorot init from the engine, then non-trivial BatchNorm parameters, moving averages and biases."""
    engine.net_init_random(model_id, seed=seed)
    p = engine.net_get_params(model_id)
    rng = np.random.default_rng(seed)
    for k, (o, shp) in layout(C)[0].items():
        n = int(np.prod(shp))
        if k.endswith("_bn"):
            c = shp[1]
            p[o:o + c] = rng.uniform(0.5, 1.5, c)              # gamma
            p[o + c:o + 2 * c] = rng.normal(0, 0.2, c)         # beta
            p[o + 2 * c:o + 3 * c] = rng.normal(0, 0.1, c)     # moving mean
            p[o + 3 * c:o + 4 * c] = rng.uniform(0.5, 1.5, c)  # moving variance
        elif k.endswith("_b"):
            p[o:o + n] = rng.normal(0, 0.1, n)
    engine.net_set_params(model_id, p)
    return p


@pytest.fixture(scope="module", params=[(1, 1, 1), (1, 1, 0), (0, 1, 0), (0, 0, 0)],
                ids=["default_f16x3_forward_bf16x3_backward", "f32_forward_bf16x3_backward", "f32_gemms", "f32_register_staged"])
def tengine(engine_mod, request):
    """The GEMM sets of the trainer: the default (conv2..conv4 forward as f16 x 3 on the f16 matrix cores, backward GEMMs as bf16 x 3 on the
    bf16 matrix cores), the forward on v_mfma_f32_16x16x4_f32 fed by LDS-DMA instead ("train_fwd_x3" = 0), every GEMM on the f32
    matrix cores ("train_gemm" = 0), and that with round 2's register-staged forward kernel ("train_fwd_dma" = 0).  Same bars for all."""
    e = engine_mod.Engine(device=0, max_batch=1024, net_channels=C)
    e.set_option("train_gemm", request.param[0])
    e.set_option("train_fwd_dma", request.param[1])
    e.set_option("train_fwd_x3", request.param[2])
    yield e
    e.close()


def compare_grads(g, ref, tol_rel, what):
    off, _ = layout(C)
    scale = np.abs(ref).max()
    for k, (o, shp) in off.items():
        n = int(np.prod(shp))
        a, r = g[o:o + n].astype(np.float64), ref[o:o + n]
        if k.endswith("_bn"):
            c = shp[1]
            assert np.all(a[2 * c:] == 0), k                   # the moving averages have no gradient
            a, r = a[:2 * c], r[:2 * c]
        if k.endswith("_b") and not k.startswith(("pi", "v")):
            # a bias in front of a BatchNorm has gradient identically 0 (the batch mean absorbs it)
            assert np.abs(r).max() <= 1e-9 * max(scale, 1.0)       # float64 autograd: rounding residue only
            assert np.all(a == 0), (what, k)                        # the kernels write the exact value
            continue
        err = np.linalg.norm(a - r) / max(np.linalg.norm(r), 1e-12)
        assert err <= tol_rel, (what, k, err)


@pytest.mark.parametrize("b", [64, 37])
def test_gradients_match_autograd(tengine, b):
    """Loss and every gradient of one step, dropout off.  Bar: relative L2 error per tensor <= 1e-3 (f32 kernels vs
    float64 autograd).  Typical errors are 3e-6; the bar leaves room for ReLU mask flips: an activation within f32
    rounding of zero is cut on one side only, and ONE flip among FC1's 38k activations moves every upstream
    gradient tensor by ~3e-4 (measured with tools/train_check.py: b = 37 has exactly one such element)."""
    tengine.set_option("train_dropout_e6", 0)
    p = perturbed_params(tengine, 1, seed=b)
    boards, pis, vs = make_batch(b, seed=100 + b)
    tengine.train_begin(1)
    (lp, lv), g = tengine.train_step(boards, pis, vs, apply=False, want_grads=True)
    rlp, rlv, rg, rstats = step_reference(p, C, boards, pis, vs)
    assert abs(lp - rlp) <= 1e-5 * max(1, abs(rlp)) and abs(lv - rlv) <= 1e-5 * max(1, abs(rlv)), (lp, rlp, lv, rlv)
    compare_grads(g, rg, 1e-3, f"b={b}")
    # apply = False leaves the weights alone but BatchNorm's moving averages advanced (training-mode forward)
    tengine.train_end(2)
    p2 = tengine.net_get_params(2)
    for k, (o, shp) in layout(C)[0].items():
        n = int(np.prod(shp))
        if k.endswith("_bn"):
            c = shp[1]
            assert np.array_equal(p2[o:o + 2 * c], p[o:o + 2 * c])
            assert np.abs(p2[o + 2 * c:o + 3 * c] - rstats[k][0]).max() <= 1e-6, k
            assert np.abs(p2[o + 3 * c:o + 4 * c] - rstats[k][1]).max() <= 1e-5, k
        else:
            assert np.array_equal(p2[o:o + n], p[o:o + n]), k


def test_gradients_match_autograd_at_the_bench_width(engine_mod):
    """The same check for the default GEMM set at C = 512, batch 64 -- the shapes bench.py's nnet_train probe and the example's
    Coach run, where the big GEMMs take the 256 x 128 ring kernels (at C = 128 most of them fall back to the 128 x 128 ones)
    and a forward GEMM sums 4608 products: typical error 1e-5 per tensor (tools/train_check.py), bar 1e-3."""
    global C
    saved, C = C, 512
    e = engine_mod.Engine(device=0, max_batch=1024, net_channels=512)
    try:
        e.set_option("train_dropout_e6", 0)
        p = perturbed_params(e, 1, seed=64)
        boards, pis, vs = make_batch(64, seed=164)
        e.train_begin(1)
        (lp, lv), g = e.train_step(boards, pis, vs, apply=False, want_grads=True)
        rlp, rlv, rg, _ = step_reference(p, 512, boards, pis, vs)
        assert abs(lp - rlp) <= 1e-5 * max(1, abs(rlp)) and abs(lv - rlv) <= 1e-5 * max(1, abs(rlv)), (lp, rlp, lv, rlv)
        compare_grads(g, rg, 1e-3, "C=512 b=64")
    finally:
        C = saved
        e.close()


def test_dropout_masks_match_the_counter_rng(tengine):
    """With dropout 0.3 the masks are the build's counter RNG (az_common.h dropout_keep): the reference applies the
    same masks explicitly, so the gradients still agree to 1e-3; another mask_seed changes them."""
    tengine.set_option("train_dropout_e6", 300000)
    try:
        p = perturbed_params(tengine, 1, seed=5)
        boards, pis, vs = make_batch(48, seed=9)
        tengine.train_begin(1)
        (lp, lv), g = tengine.train_step(boards, pis, vs, mask_seed=0xABCDEF0123, apply=False, want_grads=True)
        rlp, rlv, rg, _ = step_reference(p, C, boards, pis, vs, mask_seed=0xABCDEF0123, dropout=0.3)
        assert abs(lp - rlp) <= 1e-5 * max(1, abs(rlp)) and abs(lv - rlv) <= 1e-5
        compare_grads(g, rg, 1e-3, "dropout")
        (lp2, _), g2 = tengine.train_step(boards, pis, vs, mask_seed=77, apply=False, want_grads=True)
        assert lp2 != lp and not np.array_equal(g, g2)
        # the same step again is bit-identical (fixed-order reductions, no atomics) -- moving averages aside
        tengine.train_begin(1)
        (lp3, lv3), g3 = tengine.train_step(boards, pis, vs, mask_seed=0xABCDEF0123, apply=False, want_grads=True)
        assert (lp3, lv3) == (lp, lv) and np.array_equal(g3, g)
    finally:
        tengine.set_option("train_dropout_e6", 300000)


def test_adam_update_replays_from_the_device_gradients(tengine):
    """The optimiser in isolation.  Adam divides by sqrt(v), so over several steps an f32 and an f64 run drift apart
    chaotically (a gradient that is pure rounding noise still moves its weight by ~lr; measured with
    tools/train_adam_check.py: losses agree to 1e-5 after one step and to 3e-3 after six).  The bar that IS exact:
    feeding the gradients the device reports for each step into torch.optim.Adam's formula in float64 reproduces
    the device's final weights to 1e-6 (five f32 roundings of parameters up to 1.5 in size: ulp 1.2e-7 each; measured
    2.5e-7)."""
    tengine.set_option("train_dropout_e6", 0)
    try:
        p = perturbed_params(tengine, 3, seed=11)
        batches = [make_batch(32, seed=200 + i) for i in range(5)]
        tengine.train_begin(3)
        out = [tengine.train_step(*bt, apply=True, want_grads=True) for bt in batches]
        tengine.train_end(4)
        got = tengine.net_get_params(4).astype(np.float64)
        q = p.astype(np.float64).copy()
        m, v = np.zeros_like(q), np.zeros_like(q)
        for t, (_, g) in enumerate(out, start=1):
            g = g.astype(np.float64)
            m = 0.9 * m + 0.1 * g
            v = 0.999 * v + 0.001 * g * g
            q -= (1e-3 / (1 - 0.9 ** t)) * m / (np.sqrt(v) / np.sqrt(1 - 0.999 ** t) + 1e-8)
        stat = np.zeros(q.size, bool)
        for k, (o, shp) in layout(C)[0].items():
            if k.endswith("_bn"):
                stat[o + 2 * shp[1]:o + 4 * shp[1]] = True          # moving averages: not Adam's (checked above)
        assert np.abs(got - q)[~stat].max() <= 1e-6, np.abs(got - q)[~stat].max()
        assert np.abs(got - p)[~stat].max() > 1e-3                   # and the weights did move
        # the first two steps of the float64 trajectory are still close: losses within 1e-4
        _, rlosses = adam_reference(p, C, batches[:2])
        for ((lp, lv), _), (rlp, rlv) in zip(out[:2], rlosses):
            assert abs(lp - rlp) <= 1e-4 * max(1, abs(rlp)) and abs(lv - rlv) <= 1e-4 * max(1, abs(rlv))
    finally:
        tengine.set_option("train_dropout_e6", 300000)


def test_az_net_train_fits_a_target(tengine, engine_mod):
    """NNet::train end to end through the one-call entry: epochs x (n / batch) steps on batches drawn by the counter
    RNG.  The epoch losses fall, the stored model's INFERENCE path (BatchNorm folded, bf16 MFMA kernels) reproduces
    the training targets better than the starting weights, and the same call twice gives identical weights."""
    rng = np.random.default_rng(3)
    n = 1024
    boards, _, _ = make_batch(n, seed=31)
    # a learnable target: the policy prefers the emptiest column, the value is the sign of the stone difference
    fill = boards.sum(axis=(1, 2))                                     # [n, 7] stones per column
    pis = np.exp(-fill) / np.exp(-fill).sum(axis=1, keepdims=True)
    vs = np.sign(boards[:, 0].sum(axis=(1, 2)) - boards[:, 1].sum(axis=(1, 2)) + 0.5)
    tengine.net_init_random(5, seed=1)
    for key, val in (("train_epochs", 4), ("train_batch", 64), ("train_seed", 9)):
        tengine.set_option(key, val)
    try:
        hist = tengine.train(5, 6, boards, pis.astype(np.float32), vs.astype(np.float32))
        assert len(hist) == 4
        assert hist[-1][0] < hist[0][0] and hist[-1][1] < hist[0][1], hist
        pi0, v0 = tengine.predict(boards, 5)
        pi1, v1 = tengine.predict(boards, 6)
        ce = lambda q: float(-(pis * np.log(np.maximum(q, 1e-9))).sum(axis=1).mean())
        assert ce(pi1) < ce(pi0) - 0.05, (ce(pi0), ce(pi1))
        assert np.mean((v1 - vs) ** 2) < np.mean((v0 - vs) ** 2) - 0.05
        p6 = tengine.net_get_params(6)
        tengine.train(5, 7, boards, pis.astype(np.float32), vs.astype(np.float32))
        assert np.array_equal(p6, tengine.net_get_params(7))
        tengine.set_option("train_seed", 10)                       # another seed: other batches, other weights
        tengine.train(5, 7, boards, pis.astype(np.float32), vs.astype(np.float32))
        assert not np.array_equal(p6, tengine.net_get_params(7))
    finally:
        for key, val in (("train_epochs", 10), ("train_batch", 64), ("train_seed", 0)):
            tengine.set_option(key, val)


def test_az_net_train_is_the_documented_sequence_of_steps(tengine):
    """az_net_train = epochs x (n / batch) az_net_train_step calls on batches drawn by the counter RNG: row j of global
    step t is sample floor(rng_draw(seed, t, j, RNG_BATCH) * n / 2^64), its dropout masks are keyed by
    mix64(mix64(seed ^ 0xD6E8FEB86659FD93) ^ t).  Replaying that by hand through the step entry gives bit-identical
    weights, with the captured-graph replay (default) and with direct launches ("train_graph" 0)."""
    n, batch, epochs, seed = 256, 32, 2, 1234
    boards, pis, vs = make_batch(n, seed=77)
    tengine.net_init_random(8, seed=3)
    for key, val in (("train_epochs", epochs), ("train_batch", batch), ("train_seed", seed)):
        tengine.set_option(key, val)
    try:
        hist = tengine.train(8, 9, boards, pis, vs)
        got = tengine.net_get_params(9)
        tengine.set_option("train_graph", 0)
        tengine.train(8, 10, boards, pis, vs)
        assert np.array_equal(got, tengine.net_get_params(10))
        tengine.set_option("train_graph", 1)
        for graph in (1, 0):                                      # the wgrad chains on a second stream branch: the same kernels, the same bits
            tengine.set_option("train_fork", 1)
            tengine.set_option("train_graph", graph)
            tengine.train(8, 10, boards, pis, vs)
            assert np.array_equal(got, tengine.net_get_params(10)), graph
        tengine.set_option("train_fork", 0)
        tengine.set_option("train_graph", 1)

        def draw(t, j):
            r = int(mix64(np.uint64(seed)))
            for x in (t, j, 4):                                   # rng_draw(seed, game_id = t, ply = j, purpose = RNG_BATCH)
                r = int(mix64(np.uint64(r ^ x)))
            return (r * n) >> 64
        key = int(mix64(np.uint64(seed ^ 0xD6E8FEB86659FD93)))
        tengine.train_begin(8)
        steps = n // batch
        losses = []
        for t in range(epochs * steps):
            idx = np.array([draw(t, j) for j in range(batch)])
            (lp, lv), _ = tengine.train_step(boards[idx], pis[idx], vs[idx], mask_seed=int(mix64(np.uint64(key ^ t))), apply=True)
            losses.append((lp, lv))
        tengine.train_end(11)
        assert np.array_equal(got, tengine.net_get_params(11))
        for ep in range(epochs):                                  # the history is the per-epoch mean of the step losses
            m = np.mean(losses[ep * steps:(ep + 1) * steps], axis=0)
            assert np.allclose(hist[ep], m, rtol=1e-6), (hist[ep], m)
    finally:
        tengine.set_option("train_graph", 1)
        tengine.set_option("train_fork", 0)
        for key, val in (("train_epochs", 10), ("train_batch", 64), ("train_seed", 0)):
            tengine.set_option(key, val)


def test_train_on_very_few_samples(tengine):
    """len(examples) < batch_size: one step per epoch on a batch drawn with replacement (connect_four_net.py:127-130
    would compute 0 batches; the build keeps at least one), all-identical rows included (BatchNorm variance 0)."""
    boards, pis, vs = make_batch(3, seed=2)
    tengine.net_init_random(12, seed=4)
    for key, val in (("train_epochs", 2), ("train_batch", 64)):
        tengine.set_option(key, val)
    try:
        for n in (3, 1):
            hist = tengine.train(12, 13, boards[:n], pis[:n], vs[:n])
            assert len(hist) == 2 and all(np.isfinite(h).all() for h in hist)
            p = tengine.net_get_params(13)
            assert np.isfinite(p).all() and not np.array_equal(p, tengine.net_get_params(12))
            pi, v = tengine.predict(boards[:n], 13)
            assert np.isfinite(pi).all() and np.isfinite(v).all()
    finally:
        for key, val in (("train_epochs", 10), ("train_batch", 64)):
            tengine.set_option(key, val)


def test_train_argument_checks(tengine, engine_mod):
    boards, pis, vs = make_batch(4, seed=1)
    e2 = engine_mod.Engine(device=0, max_batch=256, net_channels=C)
    try:
        with pytest.raises(Exception):
            e2.train_step(boards, pis, vs)                        # no begin
        with pytest.raises(Exception):
            e2.train_begin(42)                                    # unknown model
        e2.net_set_kind(0, engine_mod.NET_STUB, 0)
        with pytest.raises(Exception):
            e2.train_begin(0)                                     # the stub net has no parameters
        e2.net_init_random(1, seed=0)
        e2.train_begin(1)
        with pytest.raises(Exception):
            e2.train_step(boards[:1], pis[:1], vs[:1])            # BatchNorm needs at least two rows
        with pytest.raises(Exception):
            e2.set_option("train_batch", 1000)
    finally:
        e2.close()


def test_trained_weights_of_the_gemm_sets_drift_alike(engine_mod):
    """The GEMM sets of NNet::train are numerics classes: per-step gradients agree with float64 autograd to 1e-5 / 1e-6
    (test_gradients_match_autograd), the TRAINED weights are not bit-identical between any two of them, and nothing in the
    reference pins them (its TF1 script cannot run, SURVEY.md B11: parity unpinned).  The recipe itself is chaotic (Adam's
    first steps are sign-like, dropout + training-mode BatchNorm): two all-f32 sets that differ only in SUMMATION ORDER
    ("train_fwd_dma" 0 / 1) end 32 steps 0.28 of the update's own norm apart (measured).  What this test pins is that the
    default set (f16 x 3 forward, bf16 x 3 backward) is not a worse class than that: its drift from the f32 set is within
    3 x the f32-vs-f32 drift, every set's epoch losses fall, and the sets' losses stay within 20 % of each other."""
    n = 512
    boards, pis, vs = make_batch(n, seed=41)
    got = {}
    e = engine_mod.Engine(device=0, max_batch=1024, net_channels=C)
    try:
        e.net_init_random(1, seed=5)
        start = e.net_get_params(1).astype(np.float64)
        for key, val in (("train_epochs", 4), ("train_batch", 64), ("train_seed", 3)):
            e.set_option(key, val)
        for i, (name, g, d, x3) in enumerate((("f32", 0, 1, 0), ("default", 1, 1, 1), ("f32_other_order", 0, 0, 0))):
            e.set_option("train_gemm", g)
            e.set_option("train_fwd_dma", d)
            e.set_option("train_fwd_x3", x3)
            hist = e.train(1, 2 + i, boards, pis, vs)
            got[name] = (np.array(hist, np.float64), e.net_get_params(2 + i).astype(np.float64))
    finally:
        e.close()
    h_ref, p_ref = got["f32"]
    moved = np.linalg.norm(p_ref - start)
    drift = {k: np.linalg.norm(p - p_ref) / moved for k, (_, p) in got.items()}
    print("drift from the f32 LDS-DMA set, in units of its update norm:", drift)
    assert 0 < drift["f32_other_order"] and 0 < drift["default"] <= 3 * drift["f32_other_order"], drift
    for k, (h, _) in got.items():
        assert h[-1, 0] < h[0, 0] and h[-1, 1] < h[0, 1], (k, h)
        assert np.allclose(h, h_ref, rtol=0.2), (k, h, h_ref)

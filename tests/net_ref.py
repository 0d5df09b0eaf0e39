"""Plain PyTorch fp32 reference of the policy+value net (connect_four_net.py:20-95, repaired: 7 actions,
[B,2,6,7] input) built from the engine's flat parameter vector (layout: DESIGN.md "weights file").

Used by the GPU net tests only.  Two modes:
  emulate_bf16=True  -- same arithmetic contract as the HIP kernels (BN folded in f32, weights rounded to
                        bf16, activations rounded to bf16 after every fused ReLU, f32 accumulation): isolates
                        kernel correctness (tile indexing, tap offsets, fragment layouts) at tight tolerance.
  emulate_bf16=False -- the textbook f32 net with explicit BatchNorm: shows the bf16 error budget.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def layout(C):
    off, o = {}, 0
    for l in range(4):
        cin = 2 if l == 0 else C
        off[f"conv{l+1}_w"] = (o, (3, 3, cin, C)); o += 9 * cin * C
        off[f"conv{l+1}_b"] = (o, (C,)); o += C
        off[f"conv{l+1}_bn"] = (o, (4, C)); o += 4 * C
    for l, (fi, fo) in enumerate(((6 * C, 1024), (1024, 512))):
        off[f"fc{l+1}_w"] = (o, (fi, fo)); o += fi * fo
        off[f"fc{l+1}_b"] = (o, (fo,)); o += fo
        off[f"fc{l+1}_bn"] = (o, (4, fo)); o += 4 * fo
    off["pi_w"] = (o, (512, 7)); o += 512 * 7
    off["pi_b"] = (o, (7,)); o += 7
    off["v_w"] = (o, (512, 1)); o += 512
    off["v_b"] = (o, (1,)); o += 1
    return off, o


def unpack(params, C):
    off, total = layout(C)
    assert params.size == total, (params.size, total)
    return {k: torch.from_numpy(params[o:o + int(np.prod(shp))].reshape(shp).copy()) for k, (o, shp) in off.items()}


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def random_params(C, seed, bn_random=True):
    """Glorot-ish weights with NON-trivial biases and BatchNorm statistics so the fold is exercised."""
    g = np.random.default_rng(seed)
    off, total = layout(C)
    p = np.zeros(total, np.float32)
    for k, (o, shp) in off.items():
        n = int(np.prod(shp))
        if k.endswith("_w"):
            fan_in = int(np.prod(shp[:-1]))
            lim = np.sqrt(6.0 / (fan_in + shp[-1] * (9 if k.startswith("conv") else 1)))
            # He-like gain keeps the ReLU stack alive; small head weights keep the logits O(1) so that an
            # absolute tolerance on pi / v is meaningful (saturated outputs would hide or amplify errors)
            gain = 0.3 if k in ("pi_w", "v_w") else 1.4
            p[o:o + n] = g.uniform(-lim, lim, n) * (gain if bn_random else 1.0)
        elif k.endswith("_b"):
            p[o:o + n] = g.uniform(-0.1, 0.1, n) if bn_random else 0.0
        else:  # bn: gamma, beta, mean, var
            c = shp[1]
            if bn_random:
                p[o:o + c] = g.uniform(0.5, 1.5, c)
                p[o + c:o + 2 * c] = g.uniform(-0.2, 0.2, c)
                p[o + 2 * c:o + 3 * c] = g.uniform(-0.2, 0.2, c)
                p[o + 3 * c:o + 4 * c] = g.uniform(0.5, 1.5, c)
            else:
                p[o:o + c] = 1.0
                p[o + 3 * c:o + 4 * c] = 1.0
    return p


def forward_ref(params, boards, C, emulate_bf16=True):
    """boards [B,2,6,7] f32 -> (pi [B,7], v [B]) in f32 on the CPU."""
    P = unpack(np.asarray(params, np.float32), C)
    x = torch.from_numpy(np.asarray(boards, np.float32).reshape(-1, 2, 6, 7))

    def fold(w, b, bn):
        gamma, beta, mean, var = bn[0], bn[1], bn[2], bn[3]
        s = gamma / torch.sqrt(var + np.float32(BN_EPS))
        return w * s, b * s + (beta - mean * s)

    with torch.no_grad():
        for l in range(4):
            w, b, bn = P[f"conv{l+1}_w"], P[f"conv{l+1}_b"], P[f"conv{l+1}_bn"]
            pad = 1 if l < 2 else 0
            if emulate_bf16:
                wf, bf = fold(w, b, bn)
                if l > 0:
                    wf = bf16_round(wf)           # conv1 stays f32 in the engine (VALU kernel)
                x = F.conv2d(x, wf.permute(3, 2, 0, 1).contiguous(), bf, padding=pad)
                x = bf16_round(torch.relu(x))
            else:
                x = F.conv2d(x, w.permute(3, 2, 0, 1).contiguous(), b, padding=pad)
                x = F.batch_norm(x, bn[2], bn[3], bn[0], bn[1], training=False, eps=BN_EPS)
                x = torch.relu(x)
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)     # NHWC flatten: index (y*3+x)*C + c
        for l in range(2):
            w, b, bn = P[f"fc{l+1}_w"], P[f"fc{l+1}_b"], P[f"fc{l+1}_bn"]
            if emulate_bf16:
                wf, bf = fold(w, b, bn)
                x = bf16_round(torch.relu(x @ bf16_round(wf) + bf))
            else:
                x = x @ w + b
                x = F.batch_norm(x, bn[2], bn[3], bn[0], bn[1], training=False, eps=BN_EPS)
                x = torch.relu(x)
        pi = torch.softmax(x @ P["pi_w"] + P["pi_b"], dim=1)
        v = torch.tanh(x @ P["v_w"] + P["v_b"]).reshape(-1)
    return pi.numpy(), v.numpy()

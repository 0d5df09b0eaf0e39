"""float64 PyTorch autograd reference of ONE training step of the policy+value net (test infrastructure).

Same recipe as csrc/az_train.hip / alphazero-rs_amd/trainer.py (connect_four_net.py:102-151): BatchNorm in training
mode on every conv / FC, ReLU, dropout on the two FC layers with EXPLICIT masks (the build's counter RNG, restated
here in numpy), loss = softmax cross-entropy(pi) + mean squared error(v).  Everything runs in double precision on
the CPU, so the difference to the f32 kernels is the kernels' own rounding.
"""
import numpy as np
import torch
import torch.nn.functional as F

from net_ref import layout

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(x):
    """az_common.h mix64 on numpy uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (np.asarray(x, np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return z ^ (z >> np.uint64(31))


def dropout_mask(mask_seed, layer, rows, cols, p):
    """az_common.h dropout_keep for element r*cols + c of `layer` -> float64 mask already scaled by 1/(1-p)."""
    if p <= 0:
        return np.ones((rows, cols))
    keep = np.float32(1.0) - np.float32(p)
    thresh = np.uint64(int(np.float32(keep * np.float32(16777216.0))))
    scale = np.float32(1.0) / keep
    with np.errstate(over="ignore"):
        key = mix64(np.uint64(mask_seed) ^ np.uint64(((layer + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF))
        idx = np.arange(rows * cols, dtype=np.uint64)
        r = mix64(key ^ idx)
    return ((r >> np.uint64(40)) < thresh).reshape(rows, cols).astype(np.float64) * float(scale)


def step_reference(params, C, boards, pis, vs, mask_seed=0, dropout=0.0):
    """-> (loss_pi, loss_v, grads flat [total] float64, new running stats dict name -> (mean, var))."""
    off, total = layout(C)
    leaf, stats = {}, {}
    for k, (o, shp) in off.items():
        t = torch.from_numpy(np.asarray(params[o:o + int(np.prod(shp))], np.float64).reshape(shp).copy())
        if k.endswith("_bn"):
            leaf[k + "_gamma"] = t[0].clone().requires_grad_(True)
            leaf[k + "_beta"] = t[1].clone().requires_grad_(True)
            stats[k] = (t[2].clone(), t[3].clone())
        else:
            leaf[k] = t.requires_grad_(True)

    def bn(x, name):
        rm, rv = stats[name]
        return F.batch_norm(x, rm, rv, leaf[name + "_gamma"], leaf[name + "_beta"], True, 1.0 - BN_MOMENTUM, BN_EPS)

    b = boards.shape[0]
    x = torch.from_numpy(np.asarray(boards, np.float64).reshape(b, 2, 6, 7))
    for l in range(4):
        w = leaf[f"conv{l+1}_w"].permute(3, 2, 0, 1)
        x = torch.relu(bn(F.conv2d(x, w, leaf[f"conv{l+1}_b"], padding=1 if l < 2 else 0), f"conv{l+1}_bn"))
    x = x.permute(0, 2, 3, 1).reshape(b, -1)
    for l in range(2):
        x = torch.relu(bn(x @ leaf[f"fc{l+1}_w"] + leaf[f"fc{l+1}_b"], f"fc{l+1}_bn"))
        x = x * torch.from_numpy(dropout_mask(mask_seed, 4 + l, b, x.shape[1], dropout))
    logits = x @ leaf["pi_w"] + leaf["pi_b"]
    v = torch.tanh(x @ leaf["v_w"] + leaf["v_b"]).reshape(-1)
    tp = torch.from_numpy(np.asarray(pis, np.float64))
    tv = torch.from_numpy(np.asarray(vs, np.float64))
    loss_pi = -(tp * F.log_softmax(logits, dim=1)).sum(dim=1).mean()
    loss_v = F.mse_loss(v, tv)
    (loss_pi + loss_v).backward()
    grads = np.zeros(total, np.float64)
    for k, (o, shp) in off.items():
        n = int(np.prod(shp))
        if k.endswith("_bn"):
            c = shp[1]
            grads[o:o + c] = leaf[k + "_gamma"].grad.numpy()
            grads[o + c:o + 2 * c] = leaf[k + "_beta"].grad.numpy()
        else:
            grads[o:o + n] = leaf[k].grad.reshape(-1).numpy()
    return float(loss_pi.detach()), float(loss_v.detach()), grads, {k: (m.numpy(), v_.numpy()) for k, (m, v_) in stats.items()}


def adam_reference(params, C, batches, lr=1e-3, mask_seeds=None, dropout=0.0):
    """Several steps of torch.optim.Adam on explicit batches, in float64 -> (params, [(loss_pi, loss_v)])."""
    off, total = layout(C)
    p = np.asarray(params, np.float64).copy()
    m, v = np.zeros(total), np.zeros(total)
    losses = []
    for t, (boards, pis, vs) in enumerate(batches, start=1):
        lp, lv, g, stats = step_reference(p, C, boards, pis, vs, mask_seed=(mask_seeds[t - 1] if mask_seeds else 0), dropout=dropout)
        losses.append((lp, lv))
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
        p = p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + 1e-8)
        for k, (o, shp) in off.items():
            if k.endswith("_bn"):
                c = shp[1]
                p[o + 2 * c:o + 3 * c], p[o + 3 * c:o + 4 * c] = stats[k]
    return p, losses

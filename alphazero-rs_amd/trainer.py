"""PyTorch autograd RESTATEMENT of NNet::train (src/nnet.rs:38) -- a checker and a stand-in, not the product trainer.

The product's NNet::train is `az_net_train` (csrc/az_train.hip: f32 MFMA forward/backward/Adam kernels behind the C ABI).
This module states the same recipe in plain PyTorch on the engine's parameter layout; it is what the CPU tests' fake
engine trains with (tests/test_coach_cpu.py: there is no GPU, hence no az_net_train), the yardstick of
tools/train_bench_torch.py, and an optional `trainer=` for Coach.setup.  The recipe is the reference's (TensorFlow 1.x,
examples/connect_four_lib/connect_four_net.py:102-151, internally broken: B11): loss = softmax cross-entropy(pi) + mean
squared error(v) (:104-108), Adam with lr 1e-3 (:21, :112), dropout 0.3 on the two FC layers (:15, :72-89), BatchNorm in
training mode (:39-77), batch 64 (:14), 10 epochs (:13).

Several ranks: REPLICAS.  Every rank holds the same all-gathered samples and draws the same batches and dropout masks from
the same seeds, so every rank computes the same step and ends with the same weights -- no gradient exchange (splitting the
recipe's batch of 64 over ranks would make each step's GEMMs smaller and add a 43 MB all-reduce per ~1.5 ms step).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # tf.layers.batch_normalization default epsilon
BN_MOMENTUM = 0.99     # tf.layers.batch_normalization default momentum (moving = 0.99*moving + 0.01*batch)


def layout(C):
    """Offsets of the flat f32 parameter vector (== include/az_engine.h weights file)."""
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


class PolicyValueNet(torch.nn.Module):
    """connect_four_net.py:20-95 (repaired: 7 actions, [B,2,6,7] input) on the engine's parameter layout."""

    def __init__(self, channels, flat_params, device):
        super().__init__()
        self.C = channels
        self.off, self.total = layout(channels)
        flat = torch.as_tensor(np.asarray(flat_params, np.float32))
        assert flat.numel() == self.total
        self.bn_names = []
        for k, (o, shp) in self.off.items():
            t = flat[o:o + int(np.prod(shp))].reshape(shp).clone().to(device)
            if k.endswith("_bn"):
                self.register_parameter(k + "_gamma", torch.nn.Parameter(t[0].clone()))
                self.register_parameter(k + "_beta", torch.nn.Parameter(t[1].clone()))
                self.register_buffer(k + "_mean", t[2].clone())
                self.register_buffer(k + "_var", t[3].clone())
                self.bn_names.append(k)
            else:
                self.register_parameter(k, torch.nn.Parameter(t))

    def _bn(self, x, name):
        return F.batch_norm(x, getattr(self, name + "_mean"), getattr(self, name + "_var"), getattr(self, name + "_gamma"),
                            getattr(self, name + "_beta"), self.training, 1.0 - BN_MOMENTUM, BN_EPS)

    def forward(self, boards, dropout=0.0):
        x = boards.reshape(-1, 2, 6, 7)
        for l in range(4):
            w = getattr(self, f"conv{l+1}_w").permute(3, 2, 0, 1)
            x = F.conv2d(x, w, getattr(self, f"conv{l+1}_b"), padding=1 if l < 2 else 0)
            x = torch.relu(self._bn(x, f"conv{l+1}_bn"))
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)        # NHWC flatten, index (y*3+x)*C + c
        for l in range(2):
            x = x @ getattr(self, f"fc{l+1}_w") + getattr(self, f"fc{l+1}_b")
            x = torch.relu(self._bn(x, f"fc{l+1}_bn"))
            x = F.dropout(x, dropout, self.training)
        logits = x @ self.pi_w + self.pi_b
        v = torch.tanh(x @ self.v_w + self.v_b).reshape(-1)
        return logits, v

    def flat_params(self):
        out = np.zeros(self.total, np.float32)
        for k, (o, shp) in self.off.items():
            n = int(np.prod(shp))
            if k.endswith("_bn"):
                t = torch.stack([getattr(self, k + s).detach() for s in ("_gamma", "_beta", "_mean", "_var")])
            else:
                t = getattr(self, k).detach()
            out[o:o + n] = t.reshape(-1).float().cpu().numpy()
        return out


def loss_fn(logits, v, target_pi, target_v):
    """connect_four_net.py:104-108: softmax_cross_entropy(target_pis, pi) + mean_squared_error(target_vs, v)."""
    loss_pi = -(target_pi * F.log_softmax(logits, dim=1)).sum(dim=1).mean()
    loss_v = F.mse_loss(v, target_v)
    return loss_pi, loss_v


class Trainer:
    """NNet::train(examples, previous_model_id, model_id): starts from the previous model's weights, returns the new
    model's flat parameters (the caller uploads them under `model_id` with az_net_set_params)."""

    def __init__(self, channels=512, lr=1e-3, batch_size=64, epochs=10, dropout=0.3, device=None):
        self.C, self.lr, self.batch_size, self.epochs, self.dropout = channels, lr, batch_size, epochs, dropout
        self.device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        self.history = []

    def train(self, prev_params, boards, pis, vs, seed=0):
        torch.manual_seed(seed)                  # dropout masks (batches use their own generator below)
        net = PolicyValueNet(self.C, prev_params, self.device)
        net.train()
        opt = torch.optim.Adam(net.parameters(), lr=self.lr)
        X = torch.as_tensor(np.asarray(boards, np.float32).reshape(-1, 2, 6, 7), device=self.device)
        P = torch.as_tensor(np.asarray(pis, np.float32), device=self.device)
        V = torch.as_tensor(np.asarray(vs, np.float32), device=self.device)
        n = X.shape[0]
        gen = torch.Generator(device="cpu").manual_seed(seed)
        steps = max(1, n // self.batch_size)
        self.history = []
        for epoch in range(self.epochs):
            tot = [0.0, 0.0]
            for _ in range(steps):
                idx = torch.randint(0, n, (self.batch_size,), generator=gen).to(self.device)   # :130 randint batches
                logits, v = net(X[idx], self.dropout)
                lp, lv = loss_fn(logits, v, P[idx], V[idx])
                opt.zero_grad(set_to_none=True)
                (lp + lv).backward()
                opt.step()
                tot[0] += lp.item()
                tot[1] += lv.item()
            self.history.append((tot[0] / steps, tot[1] / steps))
        net.eval()
        return net.flat_params()

    def evaluate(self, params, boards, pis, vs):
        net = PolicyValueNet(self.C, params, self.device)
        net.eval()
        with torch.no_grad():
            X = torch.as_tensor(np.asarray(boards, np.float32).reshape(-1, 2, 6, 7), device=self.device)
            logits, v = net(X)
            lp, lv = loss_fn(logits, v, torch.as_tensor(np.asarray(pis, np.float32), device=self.device),
                             torch.as_tensor(np.asarray(vs, np.float32), device=self.device))
        return float(lp), float(lv)

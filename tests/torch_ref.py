"""Independent second formulation of the v3 / v3-slim graph in torch CPU ops.

Written separately from oracle/cv_oracle.c (F.pad + conv2d on NCHW with HWIO->OIHW
weights, max_pool2d, matmul) and run in float64 to bound the oracle's fp32 error.
Follows /root/reference/clairvoyante/clairvoyante_v3.py:54-151 and selu.py:21-25.
Used by tests and by tests/golden/make_golden.py.
"""
import torch
import torch.nn.functional as F

ALPHA = 1.6732632423543772848170429916717
SCALE = 1.0507009873554804934193349852946

CFG = {
    "full": dict(kh=(1, 2, 3), cout=(16, 32, 48), pool=(5, 4, 3), fc4=336, fc5=168),
    "slim": dict(kh=(1, 3, 5), cout=(8, 16, 32), pool=(1, 1, 1), fc4=36, fc5=18),
}


def selu(x):
    return SCALE * torch.where(x >= 0, x, ALPHA * (torch.exp(x) - 1))


def same_pad(k):
    total = k - 1
    before = total // 2
    return before, total - before


def forward(arch, params, x, dtype=torch.float64, mask4=None, rate4=0.0, want_logits=False, device=None):
    """x [n,33,4,4] (NHWC) -> dict with out16 and intermediates (NHWC).  device: where the torch ops run (default: CPU;
    bench.py runs the float64 formulation over all timed candidates with stock torch ops on the GPU)."""
    cfg = CFG[arch]
    p = {k: torch.as_tensor(v).to(device=device, dtype=dtype) for k, v in params.items()}
    t = torch.as_tensor(x).to(device=device, dtype=dtype).permute(0, 3, 1, 2)  # NCHW: C=matrix, H=position, W=base
    inter = {}
    for l in range(3):
        w = p["conv%d/kernel" % (l + 1)].permute(3, 2, 0, 1)  # HWIO -> OIHW
        pt, pb = same_pad(cfg["kh"][l])
        pl, pr = same_pad(4)
        t = F.conv2d(F.pad(t, (pl, pr, pt, pb)), w, p["conv%d/bias" % (l + 1)])
        inter["pre%d" % (l + 1)] = t.permute(0, 2, 3, 1)
        t = selu(t)
        inter["act%d" % (l + 1)] = t.permute(0, 2, 3, 1)
        if cfg["pool"][l] > 1:
            t = F.max_pool2d(t, kernel_size=(cfg["pool"][l], 1), stride=1)
        inter["pool%d" % (l + 1)] = t.permute(0, 2, 3, 1)
    flat = t.permute(0, 2, 3, 1).reshape(t.shape[0], -1)
    fc4 = selu(flat @ p["fc4/kernel"] + p["fc4/bias"])
    inter["fc4"] = fc4
    d4 = fc4
    if mask4 is not None:
        ap = -1.7580993408473766
        q = 1.0 - rate4
        a = (1.0 / (q * ((1 - q) * ap * ap + 1.0))) ** 0.5
        b = -a * ((1 - q) * ap)
        m = torch.as_tensor(mask4).to(device=device, dtype=dtype)
        d4 = a * (fc4 * m + ap * (1 - m)) + b
    inter["d4"] = d4
    fc5 = selu(d4 @ p["fc5/kernel"] + p["fc5/bias"])
    inter["fc5"] = fc5
    base = torch.sigmoid(d4 @ p["YBaseChangeSigmoid/kernel"] + p["YBaseChangeSigmoid/bias"])
    lz = selu(fc5 @ p["YZygosityFC/kernel"] + p["YZygosityFC/bias"]) + 1e-10
    lt = selu(fc5 @ p["YVarTypeFC/kernel"] + p["YVarTypeFC/bias"]) + 1e-10
    ll = selu(fc5 @ p["YIndelLengthFC/kernel"] + p["YIndelLengthFC/bias"]) + 1e-10
    out = torch.cat([base, torch.softmax(lz, 1), torch.softmax(lt, 1), torch.softmax(ll, 1)], 1)
    inter["out"] = out
    if want_logits:
        inter["logits"] = (lz, lt, ll)
        inter["base"] = base
    return inter


def loss(arch, params, x, y, lam, dtype=torch.float64, mask4=None, rate4=0.0):
    """Scalar loss of clairvoyante_v3.py:140-151 (sums over the batch)."""
    r = forward(arch, params, x, dtype, mask4, rate4, want_logits=True)
    y = torch.as_tensor(y).to(dtype)
    lz, lt, ll = r["logits"]
    l1 = ((r["base"] - y[:, 0:4]) ** 2).sum()
    l2 = (-y[:, 4:6] * torch.log_softmax(lz, 1)).sum()
    l3 = (-y[:, 6:10] * torch.log_softmax(lt, 1)).sum()
    l4 = (-y[:, 10:16] * torch.log_softmax(ll, 1)).sum()
    reg = sum((torch.as_tensor(v).to(dtype) ** 2).sum() / 2 for k, v in params.items() if "bias" not in k)
    return l1 + l2 + l3 + l4 + lam * reg, (l1, l2, l3, l4, lam * reg)

"""Deterministic synthetic pileup tensors (the bench / parity workload).

The value semantics follow the producer of the reference's text tensors
(/root/reference/dataPrepScripts/CreateTensor.py:32-48): per position and base,
matrix 0 counts reference bases of aligned reads, matrix 1 query bases plus
inserted bases, matrix 2 reference bases plus deleted bases, matrix 3 query
bases; the reader then subtracts matrix 0 from matrices 1..3
(/root/reference/clairvoyante/utils_v2.py:46).  All values are small signed
integers held in fp32 (CreateTensor.py:24 prints "%0.1f"; --dcov 250).

Generator spec: seed = 20260927 + rank; reference bases uniform;
depth ~ clip(Poisson(40), 4, 250); class ~ {ref .70, het-SNP .12, hom-SNP .08,
INS .05, DEL .05}.  Written with torch ops so the same code fills HBM directly
on an MI355X (bench) or runs on CPU (tests, fixtures).
"""
import numpy as np
import torch

H, W, C = 33, 4, 4
CENTER = 16
BASE_SEED = 20260927
CLASS_P = (0.70, 0.12, 0.08, 0.05, 0.05)  # ref, het SNP, hom SNP, INS, DEL


def _binomial(n, p, gen):
    # n: float tensor of counts, p: float or tensor -> Binomial(n, p) via torch.binomial
    pt = p if torch.is_tensor(p) else torch.full_like(n, float(p))
    return torch.binomial(n, pt, generator=gen)


def make_candidates(n, seed=BASE_SEED, device="cpu", return_class=False):
    """-> X [n,33,4,4] fp32 (matrices 1..3 already minus matrix 0), optionally class ids."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    f32 = torch.float32
    ref = torch.randint(0, 4, (n, H), generator=gen, device=dev)
    depth = torch.poisson(torch.full((n,), 40.0, device=dev), generator=gen).clamp_(4, 250)
    cls = torch.multinomial(torch.tensor(CLASS_P, device=dev), n, replacement=True, generator=gen)
    onehot_ref = torch.nn.functional.one_hot(ref, 4).to(f32)            # [n,H,4]
    dp = _binomial(depth[:, None].expand(n, H).contiguous(), 0.95, gen)  # per-position depth
    m0 = onehot_ref * dp[..., None]
    # matrix 3: query bases = ref bases minus sequencing errors moved to another base
    err = _binomial(dp, 0.01, gen)
    other = (ref + torch.randint(1, 4, (n, H), generator=gen, device=dev)) % 4
    onehot_other = torch.nn.functional.one_hot(other, 4).to(f32)
    m3 = onehot_ref * (dp - err)[..., None] + onehot_other * err[..., None]
    # SNP at the centre: move an alt fraction (0.5 het / 1.0 hom) to the alt base
    frac = torch.where(cls == 1, 0.5, torch.where(cls == 2, 1.0, 0.0)).to(f32)
    alt = (ref[:, CENTER] + torch.randint(1, 4, (n,), generator=gen, device=dev)) % 4
    altc = _binomial(dp[:, CENTER], frac, gen)
    m3[:, CENTER] = m3[:, CENTER] - onehot_ref[:, CENTER] * torch.minimum(altc, m3[:, CENTER].gather(
        1, ref[:, CENTER:CENTER + 1]).squeeze(1))[:, None] + torch.nn.functional.one_hot(alt, 4).to(f32) * altc[:, None]
    m3.clamp_(min=0)
    # indels: length 1..6 at offsets centre+1.., carried by ~half of the reads
    ilen = torch.randint(1, 7, (n,), generator=gen, device=dev)
    pos = torch.arange(H, device=dev)[None, :]
    span = (pos > CENTER) & (pos <= CENTER + ilen[:, None])
    carr = _binomial(depth, 0.5, gen)[:, None] * span.to(f32)            # [n,H]
    ins_base = torch.randint(0, 4, (n, H), generator=gen, device=dev)
    m1 = m3 + torch.nn.functional.one_hot(ins_base, 4).to(f32) * (carr * (cls == 3)[:, None])[..., None]
    m2 = m0 + onehot_ref * (carr * (cls == 4)[:, None])[..., None]
    # sparse alignment noise on the insertion matrix
    noise = _binomial(torch.ones((n, H, 4), device=dev), 0.02, gen)
    m1 = m1 + noise
    x = torch.stack([m0, m1 - m0, m2 - m0, m3 - m0], dim=-1).contiguous()  # [n,H,4,4]
    if return_class:
        return x, cls, ref, alt, ilen
    return x


def make_stress(n, seed=BASE_SEED, device="cpu"):
    """Uniform integers in [-250, 250] (worst-case magnitudes after subtraction)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) + 7)
    return torch.randint(-250, 251, (n, H, W, C), generator=gen, device=dev).to(torch.float32)


def make_labels(cls, ref, alt, ilen):
    """16-vector labels with the reference's encoding
    (/root/reference/clairvoyante/utils_v2.py:90-117,142-147)."""
    n = cls.shape[0]
    y = torch.zeros((n, 16), dtype=torch.float32, device=cls.device)
    r = ref[:, CENTER]
    idx = torch.arange(n, device=cls.device)
    isref = cls == 0
    het = cls == 1
    hom = cls == 2
    ins = cls == 3
    dele = cls == 4
    # base change
    y[idx[isref], r[isref]] = 1.0
    y[idx[het], r[het]] = 0.5
    y[idx[het], alt[het]] = 0.5
    y[idx[hom], alt[hom]] = 1.0
    indel = ins | dele
    y[idx[indel], r[indel]] = 0.5          # het indel: utils_v2.py:95-96
    # zygosity: HET(4) for het SNP and (het) indels, HOM(5) for ref and hom SNP
    y[:, 4] = (het | indel).float()
    y[:, 5] = (isref | hom).float()
    # type REF SNP INS DEL
    y[:, 6] = isref.float()
    y[:, 7] = (het | hom).float()
    y[:, 8] = ins.float()
    y[:, 9] = dele.float()
    # length 0,1,2,3,4,>4
    ln = torch.where(indel, ilen.clamp(max=5), torch.zeros_like(ilen))
    y[idx, 10 + ln] = 1.0
    return y


# ---- seeded weights (bench / smoke / parity workload) ---------------------------------------------------------
_TOPOLOGY = {   # kh, cout, pool, fc4, fc5: clairvoyante_v3.py:9-12 / clairvoyante_v3_slim.py:9-11
    "full": ((1, 2, 3), (16, 32, 48), (5, 4, 3), 336, 168),
    "slim": ((1, 3, 5), (8, 16, 32), (1, 1, 1), 36, 18),
}


def param_shapes(arch):
    """TF variable name -> shape in TF layout (HWIO kernels, [in, out] dense), in checkpoint order."""
    kh, cout, pool, fc4, fc5 = _TOPOLOGY[arch]
    cin = (C, cout[0], cout[1])
    h = H - sum(p - 1 for p in pool)
    shapes = {}
    for l in range(3):
        shapes["conv%d/kernel" % (l + 1)] = (kh[l], W, cin[l], cout[l])
        shapes["conv%d/bias" % (l + 1)] = (cout[l],)
    shapes.update({"fc4/kernel": (h * W * cout[2], fc4), "fc4/bias": (fc4,), "fc5/kernel": (fc4, fc5), "fc5/bias": (fc5,),
                   "YBaseChangeSigmoid/kernel": (fc4, 4), "YBaseChangeSigmoid/bias": (4,),
                   "YZygosityFC/kernel": (fc5, 2), "YZygosityFC/bias": (2,),
                   "YVarTypeFC/kernel": (fc5, 4), "YVarTypeFC/bias": (4,),
                   "YIndelLengthFC/kernel": (fc5, 6), "YIndelLengthFC/bias": (6,)})
    return shapes


def seeded_params(arch, seed=0, bias_scale=0.0):
    """Weights drawn with the reference's initialisers from a seeded numpy stream: truncated normal with sigma
    sqrt(1.3 * 2 / fan_in) cut at 2 sigma for conv / fc4 / fc5 (variance_scaling_initializer, clairvoyante_v3.py:57),
    glorot-uniform heads (tf.layers.dense default, v3.py:125-135); biases zero, or N(0, bias_scale^2) so that the
    bias path carries values."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(arch).items():
        if name.endswith("bias"):
            out[name] = (bias_scale * rng.standard_normal(shp)).astype(np.float32)
            continue
        fan_in, fan_out = int(np.prod(shp[:-1])), shp[-1]
        if name.startswith("Y"):
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            out[name] = rng.uniform(-lim, lim, shp).astype(np.float32)
            continue
        v = rng.standard_normal(shp)
        bad = np.abs(v) > 2.0
        while bad.any():
            v[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(v) > 2.0
        out[name] = (v * np.sqrt(1.3 * 2.0 / fan_in)).astype(np.float32)
    return out


def bench_params(arch, seed=1):
    """The weight set of bench.py, smoke() and the parity tests: seeded_params with conv1 scaled by 1/32 so that
    count-valued inputs give O(1) logits (un-scaled He-initialised weights saturate every softmax),
    and non-zero biases."""
    P = seeded_params(arch, seed=seed, bias_scale=0.05)
    P["conv1/kernel"] = (P["conv1/kernel"] * np.float32(1.0 / 32.0)).astype(np.float32)
    return P

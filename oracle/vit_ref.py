"""CPU ORACLE (test infrastructure, NOT product code) for the frame-embedding half of the hot path.

The reference has NO local vision encoder: the forward pass sits behind an HTTP endpoint
(src/pipeline/vlm_extractor.py:130-185, config/base_config.yaml:6-9).  So there is no reference arithmetic to
restate and no reference test to pin against -> "PARITY UNPINNED by the reference" for the encoder.  What pins
this oracle instead (tests/golden/make_vit_golden.py, run in the authoring container):

  * the forward below is checked against the locally installed ``transformers`` model classes
    (``ViTModel`` / ``CLIPVisionModel``) constructed FROM A CONFIG OBJECT (never from_pretrained) with the same
    build-generated weights;
  * its outputs on seeded inputs are committed as golden vectors under tests/golden/.

Modes:
  quant=None        plain fp32 forward (the "what the model means" answer)
  quant="f16"/"bf16" same forward with a round-to-dtype at exactly the points where the HIP path stores a
                    16-bit value (POINTS below) - fp32 everywhere else, like the kernels' accumulators and the
                    residual stream.  With quant="bf16" the three points that are NOT matrix operands (patch rows,
                    the two residual-branch outputs) are rounded to fp16, as the device stores them (DEVICE_POINTS).
  quant={point: dtype} rounds only the named points, each to its own dtype: how tests/golden/bf16_floor.py prices
                    every storage point on its own.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

VIT_B16_224 = dict(
    arch="vit_b16_224", image=224, patch=16, hidden=768, layers=12, heads=12, mlp=3072, act="gelu",
    ln_eps=1e-12, pre_ln=False, patch_bias=True, proj_dim=0,
    mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5),
)
CLIP_L14_336 = dict(
    arch="clip_l14_336", image=336, patch=14, hidden=1024, layers=24, heads=16, mlp=4096, act="quick_gelu",
    ln_eps=1e-5, pre_ln=True, patch_bias=False, proj_dim=0,
    mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711),
)
SPECS = {"vit_b16_224": VIT_B16_224, "clip_l14_336": CLIP_L14_336}


def tiny_spec(image=32, patch=8, hidden=64, layers=2, heads=2, mlp=128, act="gelu", pre_ln=False,
              patch_bias=True, ln_eps=1e-12, proj_dim=0):
    return dict(arch="tiny", image=image, patch=patch, hidden=hidden, layers=layers, heads=heads, mlp=mlp,
                act=act, ln_eps=ln_eps, pre_ln=pre_ln, patch_bias=patch_bias, proj_dim=proj_dim,
                mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))


def _torch_dtype(quant: Optional[str]):
    return {None: None, "f16": torch.float16, "bf16": torch.bfloat16}[quant]


# every place the HIP path stores a 16-bit value
POINTS = ("weights", "pixels", "patch", "ln", "qkv", "probs", "ctx", "proj_out", "act", "fc2_out", "head_in", "out")
# matrix-unit operands must be in the encoder's dtype; the other three are only ever added to the fp32 residual
# stream by a vector kernel and are stored as fp16 whatever the encoder's dtype (csrc/gemm.hip EPI_DELTA16)
OPERAND_POINTS = ("weights", "pixels", "ln", "qkv", "probs", "ctx", "act", "head_in")
DELTA_POINTS = ("patch", "proj_out", "fc2_out")


def device_points(dtype: Optional[str]):
    """{point: dtype} of the HIP path for an encoder of `dtype` (None -> no rounding anywhere)."""
    if dtype is None:
        return {}
    d = {p: dtype for p in POINTS}
    for p in DELTA_POINTS:
        d[p] = "f16"
    return d


def _rounder(quant):
    table = device_points(quant) if (quant is None or isinstance(quant, str)) else dict(quant)

    def q(x: torch.Tensor, point: str) -> torch.Tensor:
        dt = _torch_dtype(table.get(point))
        return x if dt is None else x.to(dt).to(torch.float32)
    return q


def _layernorm(x, g, b, eps):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g + b


def _act(x, kind):
    if kind == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(kind)


def patchify(pixels_chw: torch.Tensor, patch: int) -> torch.Tensor:
    """[B,3,S,S] -> [B, (S/p)^2, 3*p*p] with the conv-weight flattening order (c, py, px)."""
    B, C, S, _ = pixels_chw.shape
    g = S // patch
    x = pixels_chw.reshape(B, C, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, g * g, C * patch * patch)


@torch.no_grad()
def vit_forward_ref(spec: Dict, w: Dict[str, np.ndarray], pixels_chw, quant: Optional[str] = None,
                    l2_normalise: bool = True, return_tokens: bool = False):
    """pixels_chw: [B,3,S,S] float (already normalised).  Returns [B, D] fp32 numpy (D = proj_dim or hidden)."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
    q_ = _rounder(quant)
    qw = lambda a: q_(t(a), "weights")  # GEMM weights are stored 16-bit on the device
    H, heads, L = spec["hidden"], spec["heads"], spec["layers"]
    hd = H // heads
    x = q_(t(pixels_chw), "pixels")
    B = x.shape[0]
    patches = patchify(x, spec["patch"])  # [B,P,K]
    tok = patches @ qw(w["patch_w"]).T
    if spec["patch_bias"]:
        tok = tok + t(w["patch_b"])
    tok = q_(tok, "patch")  # the patch GEMM stores 16-bit rows; pos / cls are added in fp32 afterwards
    cls = t(w["cls"]).reshape(1, 1, H).expand(B, 1, H)
    x = torch.cat([cls, tok], dim=1) + t(w["pos"]).unsqueeze(0)  # fp32 residual stream
    if spec["pre_ln"]:
        x = _layernorm(x, t(w["pre_ln_g"]), t(w["pre_ln_b"]), spec["ln_eps"])
    N = x.shape[1]
    scale = 1.0 / math.sqrt(hd)
    for l in range(L):
        p = lambda n: w[f"l{l}.{n}"]
        h = q_(_layernorm(x, t(p("ln1_g")), t(p("ln1_b")), spec["ln_eps"]), "ln")
        qkv = q_(h @ qw(p("qkv_w")).T + t(p("qkv_b")), "qkv")  # [B,N,3H]
        q, k, v = qkv.split(H, dim=-1)
        q = q.reshape(B, N, heads, hd).transpose(1, 2)
        k = k.reshape(B, N, heads, hd).transpose(1, 2)
        v = v.reshape(B, N, heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale
        s = s - s.max(dim=-1, keepdim=True).values
        e = torch.exp(s)
        den = e.sum(dim=-1, keepdim=True)  # fp32 sum of the unquantised numerators
        ctx = (q_(e, "probs") @ v) / den
        ctx = q_(ctx.transpose(1, 2).reshape(B, N, H), "ctx")
        x = x + q_(ctx @ qw(p("proj_w")).T + t(p("proj_b")), "proj_out")  # 16-bit branch output, fp32 residual
        h = q_(_layernorm(x, t(p("ln2_g")), t(p("ln2_b")), spec["ln_eps"]), "ln")
        a = q_(_act(h @ qw(p("fc1_w")).T + t(p("fc1_b")), spec["act"]), "act")
        x = x + q_(a @ qw(p("fc2_w")).T + t(p("fc2_b")), "fc2_out")
    if return_tokens:
        return _layernorm(x, t(w["ln_g"]), t(w["ln_b"]), spec["ln_eps"]).numpy()
    pooled = _layernorm(x[:, 0], t(w["ln_g"]), t(w["ln_b"]), spec["ln_eps"])
    if spec.get("proj_dim", 0):
        pooled = q_(pooled, "head_in") @ qw(w["proj_w"]).T
    if l2_normalise:
        n = torch.sqrt((pooled * pooled).sum(dim=-1, keepdim=True))
        pooled = pooled / torch.clamp(n, min=1e-12)
    return q_(pooled, "out").numpy()


@torch.no_grad()
def vit_forward_fast(spec: Dict, w: Dict[str, np.ndarray], pixels_chw, l2_normalise: bool = True,
                     tensors: Optional[Dict[str, torch.Tensor]] = None) -> np.ndarray:
    """The SAME fp32 forward as ``vit_forward_ref(quant=None)`` written the way a CPU deployment would run it: weights
    converted once (``tensors`` = ``fast_weights(w)``), ``torch.nn.functional`` linear / layer_norm / gelu / fused
    scaled-dot-product attention on the whole batch.  bench.py times this as the best-effort CPU line (cpu_baseline
    R2); tests/test_vit_oracle.py holds it to the restatement above at fp32 rounding."""
    import torch.nn.functional as Fn
    W = tensors if tensors is not None else fast_weights(w)
    H, heads, L = spec["hidden"], spec["heads"], spec["layers"]
    x = torch.as_tensor(np.asarray(pixels_chw), dtype=torch.float32)
    B = x.shape[0]
    tok = Fn.linear(patchify(x, spec["patch"]), W["patch_w"], W["patch_b"] if spec["patch_bias"] else None)
    x = torch.cat([W["cls"].reshape(1, 1, H).expand(B, 1, H), tok], dim=1) + W["pos"].unsqueeze(0)
    if spec["pre_ln"]:
        x = Fn.layer_norm(x, (H,), W["pre_ln_g"], W["pre_ln_b"], spec["ln_eps"])
    N = x.shape[1]
    for l in range(L):
        p = lambda n: W[f"l{l}.{n}"]
        h = Fn.layer_norm(x, (H,), p("ln1_g"), p("ln1_b"), spec["ln_eps"])
        qkv = Fn.linear(h, p("qkv_w"), p("qkv_b")).reshape(B, N, 3, heads, H // heads).permute(2, 0, 3, 1, 4)
        ctx = Fn.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        x = x + Fn.linear(ctx.transpose(1, 2).reshape(B, N, H), p("proj_w"), p("proj_b"))
        h = Fn.layer_norm(x, (H,), p("ln2_g"), p("ln2_b"), spec["ln_eps"])
        a = Fn.linear(h, p("fc1_w"), p("fc1_b"))
        a = Fn.gelu(a) if spec["act"] == "gelu" else a * torch.sigmoid(1.702 * a)
        x = x + Fn.linear(a, p("fc2_w"), p("fc2_b"))
    pooled = Fn.layer_norm(x[:, 0], (H,), W["ln_g"], W["ln_b"], spec["ln_eps"])
    if spec.get("proj_dim", 0):
        pooled = Fn.linear(pooled, W["proj_w"])
    if l2_normalise:
        pooled = pooled / torch.clamp(torch.sqrt((pooled * pooled).sum(dim=-1, keepdim=True)), min=1e-12)
    return pooled.numpy()


def fast_weights(w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(np.asarray(v), dtype=torch.float32).contiguous() for k, v in w.items()}

"""Encoder family descriptions (BASELINE.json configs: ViT-B/16-224 for C1/C2/C4/C5, CLIP-ViT-L/14-336 for C3).

The reference names no vision model at all (config/base_config.yaml:9 says only ``qwen-vlm``, served remotely);
these two are the ones BASELINE.json's configs quote.  Preprocessing constants are build-defined:
ViT mean = std = 0.5; CLIP's published mean / std.
"""
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

# algorithmic FLOPs per frame (2 * MACs of the GEMMs and of QK^T / PV).
# executed=True: what vm_encode actually runs - the LAST layer's projection and MLP only on the CLS row, the one row the
# pooled embedding is read from (csrc/encoder.hip): 32.71 instead of 35.13 GFLOP for ViT-B/16-224.
def flops_per_frame(spec, executed: bool = False) -> float:
    H, L, M, P = spec["hidden"], spec["layers"], spec["mlp"], spec["patch"]
    n = (spec["image"] // P) ** 2 + 1
    qkv_attn = 2 * n * H * 3 * H + 2 * 2 * n * n * H
    branch = 2 * n * H * H + 2 * 2 * n * H * M          # attention projection + FC1 + FC2 over n rows
    patch = 2 * (n - 1) * (3 * P * P) * H
    total = L * (qkv_attn + branch) + patch
    if executed:
        total -= branch - branch // n                    # last layer: one row instead of n ...
        q = 2 * n * H * H                                # ... and only that row's query; QK^T / PV for one query TILE (16)
        total -= q - q // n
        total -= (2 * 2 * n * n * H) * (n - 16) // n
    return float(total)

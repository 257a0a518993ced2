"""Pins oracle/vit_ref.py and writes tests/golden/vit_*.npz.  Run in the authoring container:

    python tests/golden/make_vit_golden.py

1. Cross-checks ``vit_forward_ref`` (fp32) against the locally installed ``transformers`` classes built FROM A
   CONFIG OBJECT (ViTModel / CLIPVisionModel; never from_pretrained, nothing is downloaded) loaded with the same
   seeded weights.  The reference repo has no encoder code at all (it POSTs frames to a remote VLM,
   src/pipeline/vlm_extractor.py:130-185), so this is the only independent implementation available offline.
2. Writes golden vectors: seeded pixel inputs are regenerated from the seed by the tests; the file stores only
   the expected embeddings (fp32 + quantisation-aware variants) for tiny and full-size specs.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vidmem  # noqa: E402
from vidmem import synthetic as syn  # noqa: E402
from oracle import vit_ref as V  # noqa: E402


def _hf_vit(spec, w):
    from transformers import ViTConfig, ViTModel
    cfg = ViTConfig(hidden_size=spec["hidden"], num_hidden_layers=spec["layers"],
                    num_attention_heads=spec["heads"], intermediate_size=spec["mlp"], image_size=spec["image"],
                    patch_size=spec["patch"], hidden_act="gelu", layer_norm_eps=spec["ln_eps"], qkv_bias=True)
    m = ViTModel(cfg, add_pooling_layer=False).eval()
    T, H, P = torch.tensor, spec["hidden"], spec["patch"]
    new = {
        "embeddings.cls_token": T(w["cls"]).reshape(1, 1, H),
        "embeddings.position_embeddings": T(w["pos"]).unsqueeze(0),
        "embeddings.patch_embeddings.projection.weight": T(w["patch_w"]).reshape(H, 3, P, P),
        "embeddings.patch_embeddings.projection.bias": T(w["patch_b"]),
        "layernorm.weight": T(w["ln_g"]), "layernorm.bias": T(w["ln_b"]),
    }
    for l in range(spec["layers"]):
        pre = f"layers.{l}."
        for i, nm in enumerate(["q_proj", "k_proj", "v_proj"]):
            new[pre + f"attention.{nm}.weight"] = T(w[f"l{l}.qkv_w"][i * H:(i + 1) * H])
            new[pre + f"attention.{nm}.bias"] = T(w[f"l{l}.qkv_b"][i * H:(i + 1) * H])
        new[pre + "attention.o_proj.weight"] = T(w[f"l{l}.proj_w"])
        new[pre + "attention.o_proj.bias"] = T(w[f"l{l}.proj_b"])
        new[pre + "layernorm_before.weight"] = T(w[f"l{l}.ln1_g"])
        new[pre + "layernorm_before.bias"] = T(w[f"l{l}.ln1_b"])
        new[pre + "layernorm_after.weight"] = T(w[f"l{l}.ln2_g"])
        new[pre + "layernorm_after.bias"] = T(w[f"l{l}.ln2_b"])
        new[pre + "mlp.fc1.weight"] = T(w[f"l{l}.fc1_w"])
        new[pre + "mlp.fc1.bias"] = T(w[f"l{l}.fc1_b"])
        new[pre + "mlp.fc2.weight"] = T(w[f"l{l}.fc2_w"])
        new[pre + "mlp.fc2.bias"] = T(w[f"l{l}.fc2_b"])
    assert set(new) == set(m.state_dict()), set(new) ^ set(m.state_dict())
    m.load_state_dict(new)
    return lambda px: m(pixel_values=torch.tensor(px)).last_hidden_state[:, 0].numpy()


def _hf_clip(spec, w):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(hidden_size=spec["hidden"], intermediate_size=spec["mlp"],
                           num_hidden_layers=spec["layers"], num_attention_heads=spec["heads"],
                           image_size=spec["image"], patch_size=spec["patch"], hidden_act="quick_gelu",
                           layer_norm_eps=spec["ln_eps"])
    m = CLIPVisionModel(cfg).eval()
    T, H, P = torch.tensor, spec["hidden"], spec["patch"]
    new = {
        "embeddings.class_embedding": T(w["cls"]),
        "embeddings.position_embedding.weight": T(w["pos"]),
        "embeddings.patch_embedding.weight": T(w["patch_w"]).reshape(H, 3, P, P),
        "pre_layrnorm.weight": T(w["pre_ln_g"]), "pre_layrnorm.bias": T(w["pre_ln_b"]),
        "post_layernorm.weight": T(w["ln_g"]), "post_layernorm.bias": T(w["ln_b"]),
    }
    for l in range(spec["layers"]):
        pre = f"encoder.layers.{l}."
        for i, nm in enumerate(["q_proj", "k_proj", "v_proj"]):
            new[pre + f"self_attn.{nm}.weight"] = T(w[f"l{l}.qkv_w"][i * H:(i + 1) * H])
            new[pre + f"self_attn.{nm}.bias"] = T(w[f"l{l}.qkv_b"][i * H:(i + 1) * H])
        new[pre + "self_attn.out_proj.weight"] = T(w[f"l{l}.proj_w"])
        new[pre + "self_attn.out_proj.bias"] = T(w[f"l{l}.proj_b"])
        new[pre + "layer_norm1.weight"] = T(w[f"l{l}.ln1_g"])
        new[pre + "layer_norm1.bias"] = T(w[f"l{l}.ln1_b"])
        new[pre + "layer_norm2.weight"] = T(w[f"l{l}.ln2_g"])
        new[pre + "layer_norm2.bias"] = T(w[f"l{l}.ln2_b"])
        new[pre + "mlp.fc1.weight"] = T(w[f"l{l}.fc1_w"])
        new[pre + "mlp.fc1.bias"] = T(w[f"l{l}.fc1_b"])
        new[pre + "mlp.fc2.weight"] = T(w[f"l{l}.fc2_w"])
        new[pre + "mlp.fc2.bias"] = T(w[f"l{l}.fc2_b"])
    sd = {k.replace("vision_model.", ""): v for k, v in m.state_dict().items() if "position_ids" not in k}
    assert set(new) == set(sd), set(new) ^ set(sd)
    prefix = "vision_model." if any(k.startswith("vision_model.") for k in m.state_dict()) else ""
    m.load_state_dict({prefix + k: v for k, v in new.items()}, strict=False)
    return lambda px: m(pixel_values=torch.tensor(px)).pooler_output.numpy()


CASES = [
    # name, spec, weight seed, weight std, n frames, weight distribution (synthetic.encoder_weights tail=)
    ("tiny_vit", V.tiny_spec(), 42, 0.2, 3, "normal"),
    ("tiny_clip", V.tiny_spec(act="quick_gelu", pre_ln=True, patch_bias=False, ln_eps=1e-5, image=28, patch=7),
     43, 0.2, 3, "normal"),
    ("vit_b16_224", V.VIT_B16_224, 42, 0.02, 8, "normal"),
    ("vit_b16_224_heavy", V.VIT_B16_224, 1789, 0.02, 8, "heavy"),
    ("clip_l14_336", V.CLIP_L14_336, 42, 0.02, 4, "normal"),
    ("clip_l14_336_heavy", V.CLIP_L14_336, 1848, 0.02, 4, "heavy"),
]
SPEC_OF = {name: spec for name, spec, *_ in CASES}


def main(only=None):
    path = os.path.join(ROOT, "tests", "golden", "vit_golden.npz")
    out = dict(np.load(path)) if (only and os.path.exists(path)) else {}
    with torch.no_grad():
        for name, spec, seed, std, n, tail in CASES:
            if only and name not in only:
                continue
            w = syn.encoder_weights(spec, seed=seed, std=std, tail=tail)
            px = syn.normal(1000 + seed, "pixels_" + name.replace("_heavy", ""), (n, 3, spec["image"], spec["image"]))
            hf = (_hf_clip if spec["pre_ln"] else _hf_vit)(spec, w)
            want = hf(px)
            got = V.vit_forward_ref(spec, w, px, quant=None, l2_normalise=False)
            err = np.abs(want - got).max() / np.abs(want).max()
            print(f"{name}: oracle vs transformers max rel err {err:.2e}", flush=True)
            assert err < 2e-5, err
            fast = V.vit_forward_fast(spec, w, px, l2_normalise=False)
            errf = np.abs(want - fast).max() / np.abs(want).max()
            print(f"{name}: functional (bench cpu_baseline) forward vs transformers max rel err {errf:.2e}", flush=True)
            assert errf < 2e-5, errf
            out[name + "/fp32"] = V.vit_forward_ref(spec, w, px, quant=None)
            out[name + "/f16"] = V.vit_forward_ref(spec, w, px, quant="f16")
            out[name + "/bf16"] = V.vit_forward_ref(spec, w, px, quant="bf16")
            # what NO implementation with 16-bit matrix operands can beat: only the operands rounded, per frame
            for dt in ("f16", "bf16"):
                fl = V.vit_forward_ref(spec, w, px, quant={p: dt for p in V.OPERAND_POINTS})
                d = fl.astype(np.float64) - out[name + "/fp32"].astype(np.float64)
                out[name + f"/floor_{dt}"] = np.concatenate([
                    [np.linalg.norm(d) / np.linalg.norm(out[name + "/fp32"].astype(np.float64))],
                    np.linalg.norm(d, axis=1) / np.linalg.norm(out[name + "/fp32"].astype(np.float64), axis=1)])
                print(f"{name}: operand-only {dt} floor, batch {out[name + f'/floor_{dt}'][0]:.2e}, worst frame "
                      f"{out[name + f'/floor_{dt}'][1:].max():.2e}", flush=True)
            out[name + "/seed_std_n"] = np.array([seed, std, n], dtype=np.float64)
            out[name + "/tail"] = np.array([1.0 if tail == "heavy" else 0.0])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(set(sys.argv[1:]) or None)

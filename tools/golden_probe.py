"""Developer probe: per-frame relative error of the full ViT-B/16 fp16 encoder against the committed goldens (the numbers
tests/test_encoder_gpu.py::test_full_models_match_golden asserts on), printed, twice (determinism).
   [TOOLS_DEV=1 VIDMEM_...=..] python tools/golden_probe.py [name=vit_b16_224_heavy] [dtype=f16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vidmem
from _dev import maybe_dev; maybe_dev()
from vidmem import synthetic as syn
from vidmem.encoder import FrameEncoder
from oracle import vit_ref as V
name = sys.argv[1] if len(sys.argv) > 1 else "vit_b16_224_heavy"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
golden = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vit_golden.npz"))
spec = V.SPECS[name.replace("_heavy", "")]
seed, std, n = golden[name + "/seed_std_n"]
tail = "heavy" if float(golden[name + "/tail"][0]) else "normal"
w = syn.encoder_weights(spec, seed=int(seed), std=float(std), tail=tail)
px = syn.normal(1000 + int(seed), "pixels_" + name.replace("_heavy", ""), (int(n), 3, spec["image"], spec["image"]))
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
floor = golden[name + "/floor_" + dtype]
prev = None
for rep in range(2):
    enc = FrameEncoder(spec, w, dtype, device=0)
    got = enc.encode_patches(enc.patches_from_pixels(torch.from_numpy(px))).float().cpu().numpy()
    g64, f64 = got.astype(np.float64), golden[name + "/fp32"].astype(np.float64)
    per32 = np.array([rel(g64[i], f64[i]) for i in range(g64.shape[0])])
    print(f"{name} {dtype} rep {rep}: batch {rel(got, golden[name + '/fp32']):.4e} (vs quant-aware {rel(got, golden[name + '/' + dtype]):.4e}, floor {floor[0]:.3e}); floors per frame {np.array2string(floor[1:], precision=6)}; per frame {np.array2string(per32, precision=6)}; "
          f"bars {np.array2string(np.maximum(1e-3, 1.65 * floor[1:]), precision=6)}; same bits as rep 0: {prev is None or np.array_equal(prev, got)}")
    prev = got if prev is None else prev

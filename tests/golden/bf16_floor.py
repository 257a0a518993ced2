"""Prices every 16-bit storage point of the HIP encoder path on its own (CPU only, oracle/vit_ref.py).

    python tests/golden/bf16_floor.py [--model clip_l14_336|vit_b16_224] [--dtype bf16|f16] [--frames 2]

For a full-size model with the golden weights / pixels it prints the relative L2 distance between the fp32 oracle and
the oracle with
  * each storage point rounded ALONE (what that point costs),
  * only the matrix-unit operands rounded (the FLOOR of any implementation whose GEMMs take 16-bit operands),
  * the floor + the 16-bit output embedding,
  * every point in the encoder's dtype (the round-1 device path),
  * the device path as built (operands in the encoder's dtype, the three non-operand points in fp16),
and writes the table to tests/golden/bf16_floor_<model>_<dtype>.json.  tests/test_encoder_gpu.py takes its bars for
the bf16 path from that file (north_star says 1e-3 relative; a bf16 OUTPUT alone costs more than that, which this
script measures instead of asserting).  The reference has no encoder arithmetic to compare with
(src/pipeline/vlm_extractor.py:130-185 is an HTTP call): parity unpinned by the reference, pinned by this oracle.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vidmem  # noqa: E402,F401
from vidmem import synthetic as syn  # noqa: E402
from oracle import vit_ref as V  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="clip_l14_336")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    spec = V.SPECS[args.model]
    dt = args.dtype
    w = syn.encoder_weights(spec, seed=args.seed, std=0.02)
    px = syn.normal(1000 + args.seed, "pixels_" + args.model, (args.frames, 3, spec["image"], spec["image"]))
    t0 = time.time()
    ref = V.vit_forward_ref(spec, w, px, quant=None)
    print(f"fp32 oracle: {time.time() - t0:.1f} s for {args.frames} frames", flush=True)
    rows = {}

    def run(name, table):
        rows[name] = rel(V.vit_forward_ref(spec, w, px, quant=table), ref)
        print(f"  {name:46s} {rows[name]:.3e}", flush=True)

    for p in V.POINTS:
        if p == "head_in" and not spec.get("proj_dim", 0):
            continue
        run(f"only {p} -> {dt}", {p: dt})
    floor = {p: dt for p in V.OPERAND_POINTS}
    run(f"FLOOR: matrix operands only -> {dt}", floor)
    run(f"floor + out -> {dt}", dict(floor, out=dt))
    run(f"floor + deltas -> f16", dict(floor, **{p: "f16" for p in V.DELTA_POINTS}))
    run(f"every point -> {dt} (round-1 device path)", {p: dt for p in V.POINTS})
    run(f"device path as built (deltas f16, out {dt})", V.device_points(dt))
    dev_no_out = V.device_points(dt)
    dev_no_out.pop("out")
    run("device path, fp32 output", dev_no_out)
    out = {"model": args.model, "dtype": dt, "frames": args.frames, "seed": args.seed,
           "metric": "||oracle(points) - oracle(fp32)|| / ||oracle(fp32)|| over the L2-normalised embeddings",
           "rows": rows}
    path = os.path.join(ROOT, "tests", "golden", f"bf16_floor_{args.model}_{dt}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    main()

"""GPU parity of the frame-embedding path (preprocess + encoder) through the C ABI.

The reference has no encoder arithmetic (remote service) -> parity is against the build's own oracle
(oracle/vit_ref.py, pinned to the ``transformers`` classes by tests/golden/make_vit_golden.py) and its committed
golden vectors.  Tolerance from BASELINE.json's north_star: embeddings within 1e-3 relative.  It is applied as
||gpu - oracle|| / ||oracle|| against the oracle evaluated at the SAME 16-bit storage points (quant-aware mode);
the distance to the pure-fp32 oracle is checked against the precision of the storage type itself.
"""
import os

import numpy as np
import pytest
import torch

from oracle import frames_ref as F
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu

TD = {"f16": torch.float16, "bf16": torch.bfloat16}
# Bars.  north_star: embeddings within 1e-3 relative of the CPU path.
#   fp16: 1e-3, against BOTH oracles (same storage points; plain fp32).
#   bf16: 1e-3 is below what ANY implementation with bf16 matrix operands can reach - tests/golden/bf16_floor.py prices
#         every 16-bit storage point of CLIP-L/14-336 on the CPU: rounding ONLY the operands of the matrix units costs
#         4.9e-3 (the weights alone 3.6e-3), the bf16 output alone 1.6e-3.  The bar is therefore 1.5 x that measured
#         floor (committed table tests/golden/bf16_floor_clip_l14_336_bf16.json; for the short stacks below the floor of
#         the SAME stack is evaluated in the test), and the device stores everything that is not a matrix operand in
#         fp16 (csrc/vm_kernels.h EPI_DELTA16).  Against the quant-aware oracle the bar is the same: one flipped bf16
#         rounding is 2^-8 of an element, so that comparison is as noisy as the fp32 one.
#         That is a STATED DEVIATION from BASELINE.json's 1e-3: bf16 embedding parity is unpinned by the
#         reference (it holds no encoder) and is NOT "within 1e-3".  Against the quant-aware golden - the same rounding
#         points, so only accumulation order and flipped roundings remain - the full models are held to the tighter
#         QUANT_SLACK x floor (measured 0.93 x batch, 0.97-1.01 x worst frame).
CONTRACT = 1e-3
FLOOR_SLACK = 1.5
QUANT_SLACK = 1.15
# Per-FRAME bars of the full models: max(contract, FRAME_SLACK x that frame's operand-only floor).  A single frame's distance
# to the fp32 oracle after 12-24 layers moves by up to 9 % under changes that are not errors: round 4 measured the same
# kernels with the attention context rounded once (v_fma_mixlo_f16) or twice (fp32 product, then fp16 - what is now
# pinned in csrc/attention.hip and what the quant-aware oracle does): frame 5 of the heavy-tailed ViT-B/16 golden
# 1.025e-3 <-> 1.117e-3 (floor 7.40e-4, i.e. 1.39 x <-> 1.51 x), frame 7 1.091e-3 <-> 0.993e-3, the batch 9.80e-4 <-> 9.59e-4
# (tools/golden_probe.py).  The batch bars above are unchanged; the per-frame slack leaves that much room.
FRAME_SLACK = 1.65
_FLOOR_JSON = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_floor_clip_l14_336_bf16.json")


def floor_bar_full_clip():
    import json
    rows = json.load(open(_FLOOR_JSON))["rows"]
    return FLOOR_SLACK * rows["FLOOR: matrix operands only -> bf16"]


def bar_for(dtype, spec, w, px, want_32):
    """The contract, unless 1.5 x the operand-only floor of this very stack (oracle, CPU) is above it: what no
    implementation with `dtype` matrix operands can beat.  That is every bf16 stack, and the fp16 short stacks with
    the wide (std 0.05) test weights, whose floor alone is 1.1e-3; the full fp16 models (floor 7.6e-4) are held to the
    contract itself in test_full_models_match_golden."""
    floor = rel(V.vit_forward_ref(spec, w, px, quant={p: dtype for p in V.OPERAND_POINTS}), want_32)
    return max(CONTRACT, FLOOR_SLACK * floor)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _encoder(spec, w, dtype, **options):
    from vidmem.encoder import FrameEncoder
    return FrameEncoder(spec, w, dtype=dtype, **options)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "vit_golden.npz"))


def test_specs_match_oracle():
    from vidmem import specs
    assert specs.VIT_B16_224 == V.VIT_B16_224 and specs.CLIP_L14_336 == V.CLIP_L14_336


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("hw", [(224, 224), (360, 640), (1080, 1920), (97, 131)])
def test_preprocess_matches_oracle(hw, dtype):
    from vidmem import synthetic as syn
    from vidmem import specs
    spec = specs.VIT_B16_224 if dtype == "f16" else specs.CLIP_L14_336
    w = syn.encoder_weights(dict(spec, layers=1), seed=1)
    enc = _encoder(dict(spec, layers=1), w, dtype)
    frames = syn.frames_u8(1234, 2, hw[0], hw[1])
    ft = torch.from_numpy(frames).cuda()
    for layout in ("chw", "patches"):
        got = enc.preprocess(ft, layout=layout).float().cpu().numpy()
        want = F.preprocess_ref(frames, spec["image"], spec["mean"], spec["std"], layout=layout,
                                patch=spec["patch"], k_pad=enc.patch_k)
        want16 = torch.from_numpy(want).to(TD[dtype]).float().numpy()
        # same fp32 expression on both sides; allow one 16-bit rounding flip
        ulp = 2.0 ** (-10 if dtype == "f16" else -7)
        assert np.abs(got - want16).max() <= ulp * max(1.0, np.abs(want16).max()), (hw, layout)
        assert (got == want16).mean() > 0.999
        if layout == "patches":
            assert (got[..., 3 * spec["patch"] ** 2:] == 0).all()


def test_preprocess_identity_is_exact():
    """224x224 source: bilinear taps land on pixel centres -> exact (x/255 - .5)/.5 in fp16."""
    from vidmem import synthetic as syn, specs
    spec = dict(specs.VIT_B16_224, layers=1)
    enc = _encoder(spec, syn.encoder_weights(spec, seed=1), "f16")
    frames = syn.frames_u8(5, 1, 224, 224)
    got = enc.preprocess(torch.from_numpy(frames).cuda(), layout="chw").cpu()
    rgb = torch.from_numpy(frames[..., ::-1].copy()).permute(0, 3, 1, 2).float()
    a = np.float32(1.0) / (np.float32(255.0) * np.float32(0.5))
    want = (rgb * float(a) + (-1.0)).to(torch.float16)
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype,layers,act,pre_ln,hidden,heads,mlp,image,patch", [
    ("f16", 1, "gelu", False, 768, 12, 3072, 224, 16),
    ("f16", 2, "gelu", False, 768, 12, 3072, 224, 16),
    ("bf16", 2, "quick_gelu", True, 1024, 16, 4096, 336, 14),
    ("f16", 2, "quick_gelu", True, 256, 4, 512, 64, 16),       # 17 tokens: ragged tiles everywhere
    ("bf16", 1, "gelu", False, 512, 8, 1024, 96, 16),          # 37 tokens
    ("f16", 1, "gelu", False, 256, 4, 512, 160, 16),           # 101 tokens: 7 of 13 key tiles (masked-tile path)
    ("bf16", 1, "quick_gelu", True, 256, 4, 512, 224, 14),     # 257 tokens: two-pass kernel, 17 of 37 key tiles
    ("f16", 1, "gelu", False, 256, 4, 512, 240, 16),           # 226 tokens: two-pass kernel just above 13 tiles
])
def test_short_stacks_match_oracle(dtype, layers, act, pre_ln, hidden, heads, mlp, image, patch):
    from vidmem import synthetic as syn
    spec = dict(arch="t", image=image, patch=patch, hidden=hidden, layers=layers, heads=heads, mlp=mlp, act=act,
                ln_eps=1e-5 if pre_ln else 1e-12, pre_ln=pre_ln, patch_bias=not pre_ln, proj_dim=0,
                mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
    w = syn.encoder_weights(spec, seed=3, std=0.05)
    B = 3
    px = syn.normal(77, "px", (B, 3, image, image))
    enc = _encoder(spec, w, dtype)
    got = enc.encode_patches(enc.patches_from_pixels(torch.from_numpy(px))).float().cpu().numpy()
    want_q = V.vit_forward_ref(spec, w, px, quant=dtype)
    want_32 = V.vit_forward_ref(spec, w, px, quant=None)
    bar = bar_for(dtype, spec, w, px, want_32)
    print(f"{dtype} L={layers} H={hidden}: vs quant-aware {rel(got, want_q):.2e}, vs fp32 {rel(got, want_32):.2e}, "
          f"bar {bar:.2e}")
    assert rel(got, want_q) < bar, (rel(got, want_q), bar)
    assert rel(got, want_32) < bar, (rel(got, want_32), bar)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=4e-3)


def test_projection_head_and_no_l2():
    from vidmem import synthetic as syn
    spec = dict(arch="t", image=64, patch=16, hidden=256, layers=1, heads=4, mlp=512, act="gelu", ln_eps=1e-6,
                pre_ln=False, patch_bias=True, proj_dim=128, mean=(0.5,) * 3, std=(0.5,) * 3)
    w = syn.encoder_weights(spec, seed=4, std=0.05)
    px = syn.normal(78, "px", (2, 3, 64, 64))
    enc = _encoder(spec, w, "f16")
    got = enc.encode_patches(enc.patches_from_pixels(torch.from_numpy(px)), l2_normalise=False)
    assert got.shape == (2, 128)
    want = V.vit_forward_ref(spec, w, px, quant="f16", l2_normalise=False)
    assert rel(got.float().cpu().numpy(), want) < CONTRACT


@pytest.mark.parametrize("name,dtype", [("vit_b16_224", "f16"), ("vit_b16_224_heavy", "f16"),
                                        ("clip_l14_336", "bf16"), ("clip_l14_336_heavy", "bf16")])
def test_full_models_match_golden(name, dtype, golden):
    """Full-size models against the committed goldens: 8 ViT-B/16 frames and 4 CLIP-L/14-336 frames for each of two
    weight sets - N(0, 0.02^2), and a heavy-tailed one (Student-t weights, six 6x LayerNorm outlier channels:
    synthetic.encoder_weights tail="heavy").  Bars: the batch as a whole within the contract (fp16: 1e-3; bf16: 1.5 x
    the operand-only floor, which no bf16-operand implementation can beat); every single frame within
    max(contract, FRAME_SLACK x THAT frame's operand-only floor) - the floors are part of the golden file
    (tests/golden/make_vit_golden.py), so the test evaluates nothing but the device output."""
    from vidmem import synthetic as syn
    spec = V.SPECS[name.replace("_heavy", "")]
    seed, std, n = golden[name + "/seed_std_n"]
    tail = "heavy" if float(golden[name + "/tail"][0]) else "normal"
    w = syn.encoder_weights(spec, seed=int(seed), std=float(std), tail=tail)
    px = syn.normal(1000 + int(seed), "pixels_" + name.replace("_heavy", ""), (int(n), 3, spec["image"], spec["image"]))
    enc = _encoder(spec, w, dtype)
    got = enc.encode_patches(enc.patches_from_pixels(torch.from_numpy(px))).float().cpu().numpy()
    assert np.isfinite(got).all()
    floor = golden[name + "/floor_" + dtype]                     # [batch, frame 0, frame 1, ...]
    e_q, e_32 = rel(got, golden[name + "/" + dtype]), rel(got, golden[name + "/fp32"])
    bar = CONTRACT if dtype == "f16" else max(CONTRACT, FLOOR_SLACK * float(floor[0]))
    g64, f64 = got.astype(np.float64), golden[name + "/fp32"].astype(np.float64)
    q64 = golden[name + "/" + dtype].astype(np.float64)
    per32 = np.array([rel(g64[i], f64[i]) for i in range(g64.shape[0])])
    perq = np.array([rel(g64[i], q64[i]) for i in range(g64.shape[0])])
    frame_bar = np.maximum(CONTRACT, FRAME_SLACK * floor[1:])
    print(f"{name} {dtype}: batch rel err vs quant-aware golden {e_q:.2e}, vs fp32 golden {e_32:.2e}, bar {bar:.2e} "
          f"(operand-only floor {floor[0]:.2e}); per frame vs fp32: worst {per32.max():.2e} (its floor "
          f"{floor[1:][per32.argmax()]:.2e}, bar {frame_bar[per32.argmax()]:.2e}), vs quant-aware worst {perq.max():.2e}")
    assert e_q < bar and e_32 < bar
    assert (per32 < frame_bar).all(), (per32, frame_bar)
    if dtype == "f16":
        assert (perq < frame_bar).all(), (perq, frame_bar)
    else:   # same rounding points on both sides: the tighter bar (a real kernel error moves this first)
        assert e_q < max(CONTRACT, QUANT_SLACK * float(floor[0])), (e_q, floor[0])
        assert (perq < np.maximum(CONTRACT, QUANT_SLACK * floor[1:])).all(), (perq, floor[1:])
    # the same per frame as an angle: for unit vectors 1 - cos = e^2 / 2 (directions compared in fp64; the 16-bit
    # output's own norm is 1 only to an ulp)
    cos = (g64 * f64).sum(1) / (np.linalg.norm(g64, axis=1) * np.linalg.norm(f64, axis=1))
    assert (1.0 - cos < 0.5 * frame_bar * frame_bar).all(), (1.0 - cos).max()


def test_bf16_encoder_survives_activations_beyond_fp16_range():
    """Residual-branch outputs (patch rows, attention projection, FC2) are stored as fp16 whatever the encoder's dtype
    (csrc/vm_kernels.h EPI_DELTA16).  A bf16 model may have been chosen for its RANGE: an FC2 bias of 1e5 puts those
    outputs past fp16's 65504.  The store must saturate instead of writing inf (which LayerNorm turns into NaN rows):
    the embedding stays finite and - LayerNorm being scale-free, the huge bias common to every token - close to the
    oracle's."""
    from vidmem import synthetic as syn
    spec = dict(arch="t", image=64, patch=16, hidden=256, layers=2, heads=4, mlp=512, act="quick_gelu", ln_eps=1e-5,
                pre_ln=True, patch_bias=False, proj_dim=0, mean=(0.5,) * 3, std=(0.5,) * 3)
    w = syn.encoder_weights(spec, seed=6, std=0.05)
    w["l0.fc2_b"] = w["l0.fc2_b"].copy()
    w["l0.fc2_b"][:8] = np.float32(1.0e5)
    px = syn.normal(79, "px", (3, 3, 64, 64))
    enc = _encoder(spec, w, "bf16")
    got = enc.encode_patches(enc.patches_from_pixels(torch.from_numpy(px))).float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=8e-3)
    # ... and a NaN must stay visible: the saturating store keeps it (csrc/gemm.hip sat_f16), so a broken activation
    # cannot hide behind a finite, wrong embedding
    w["l0.fc2_b"][8] = np.float32("nan")
    bad = _encoder(spec, w, "bf16")
    got = bad.encode_patches(bad.patches_from_pixels(torch.from_numpy(px))).float().cpu().numpy()
    assert np.isnan(got).all()


def test_batching_is_invisible():
    """Frame i's embedding must not depend on what else is in the launch or where micro-batches split."""
    from vidmem import synthetic as syn, specs
    spec = dict(specs.VIT_B16_224, layers=2)
    w = syn.encoder_weights(spec, seed=9)
    enc = _encoder(spec, w, "f16", micro_batch=64)     # 150 frames = passes of 64, 64, 22 (two streams: the default)
    frames = torch.from_numpy(syn.frames_u8(99, 150, 224, 224)).cuda()
    all_emb = enc.embed_frames(frames)
    head = enc.embed_frames(frames[:3])
    tail = enc.embed_frames(frames[147:])
    assert torch.equal(all_emb[:3], head) and torch.equal(all_emb[147:], tail)
    assert torch.isfinite(all_emb.float()).all()


@pytest.mark.parametrize("dtype,image,patch,hidden,heads,n", [("f16", 224, 16, 768, 12, 130), ("bf16", 336, 14, 1024, 16, 40)])
def test_last_layer_on_cls_rows_is_invisible(dtype, image, patch, hidden, heads, n):
    """vm_encode runs the LAST layer's query projection, attention query tile, attention projection, LN2 and MLP on the
    CLS rows only (the one row the embedding is pooled from).  Against an encoder with VM_ENC_OPT_LAST_LAYER = 0
    (everything on every row) the embeddings must be identical BIT FOR BIT, for both attention kernels (197 / 577
    tokens) and for batches that take the persistent 256 x 256 GEMM (n x tokens rows) as well as 3-frame ones."""
    from vidmem import synthetic as syn, specs
    base = specs.VIT_B16_224 if image == 224 else specs.CLIP_L14_336
    spec = dict(base, layers=3)
    assert (spec["image"], spec["patch"], spec["hidden"], spec["heads"]) == (image, patch, hidden, heads)
    w = syn.encoder_weights(spec, seed=31)
    frames = torch.from_numpy(syn.frames_u8(77, n, image, image)).cuda()
    full = _encoder(spec, w, dtype, last_layer=0)
    half = _encoder(spec, w, dtype, last_layer=1)        # projection + MLP pruned, attention on every row
    pruned = _encoder(spec, w, dtype)
    want = full.embed_frames(frames)
    assert torch.isfinite(want.float()).all()
    assert torch.equal(half.embed_frames(frames), want)
    assert torch.equal(pruned.embed_frames(frames), want)
    assert torch.equal(pruned.embed_frames(frames[:3]), want[:3])
    one = dict(spec, layers=1)                           # the last layer is also the first: no pending branch outputs
    w1 = syn.encoder_weights(one, seed=32)
    full1 = _encoder(one, w1, dtype, last_layer=0)
    assert torch.equal(_encoder(one, w1, dtype).embed_frames(frames[:5]), full1.embed_frames(frames[:5]))


@pytest.mark.parametrize("name,dtype,mbatch,n", [("vit_b16_224", "f16", 64, 200), ("clip_l14_336", "bf16", 28, 70)])
def test_two_stream_mode_is_invisible(name, dtype, mbatch, n):
    """The DEFAULT schedule (VM_SCHED_AUTO): consecutive micro-batch passes of a call alternate between two internal
    streams with a workspace each, and the LayerNorms run their low-register build.  Embeddings must equal the
    one-stream encoder's bit for bit - an even and an odd number of passes, a ragged last pass, and calls of one
    pass (where the schedule does nothing) in between; also with the schedule forced, and with per-kernel timing on
    (where AUTO falls back to one stream)."""
    from vidmem import synthetic as syn
    spec = dict(V.SPECS[name], layers=3)
    w = syn.encoder_weights(spec, seed=17)
    plain = _encoder(spec, w, dtype, micro_batch=mbatch, schedule="one_stream")
    dual = _encoder(spec, w, dtype, micro_batch=mbatch)       # the default path
    forced = _encoder(spec, w, dtype, micro_batch=mbatch, schedule="two_streams")
    assert dual.schedule == "auto" and plain.schedule == "one_stream" and forced.schedule == "two_streams"
    assert dual.workspace_bytes(n) == 2 * plain.workspace_bytes(n) and dual.workspace_bytes(3) == plain.workspace_bytes(3)
    S = spec["image"]
    frames = torch.from_numpy(syn.frames_u8(5, n, S, S)).cuda()
    for count in (n, 2 * mbatch, 3, 3 * mbatch - 5):          # 4 passes (ragged) / 2 / 1 / 3 (ragged)
        want = plain.embed_frames(frames[:count])
        assert torch.isfinite(want.float()).all()
        for label, enc in (("auto", dual), ("two_streams", forced)):
            got = enc.embed_frames(frames[:count])
            if not torch.equal(got, want):    # say where: which rows (frames) differ tells which pass / stream
                rows = ((got.float() - want.float()).abs().amax(dim=1) > 0).nonzero().flatten().tolist()
                raise AssertionError(f"{label}, {count} frames: {len(rows)} embeddings differ, first rows {rows[:10]}")
    # per-kernel timing on: AUTO runs one stream (clean event durations), TWO_STREAMS stays on two; same bits
    dual.ctx.profile_enable(4096)
    try:
        assert torch.equal(dual.embed_frames(frames), plain.embed_frames(frames))
        assert torch.equal(forced.embed_frames(frames), plain.embed_frames(frames))
        assert dual.ctx.profile_read()["layernorm"][1] > 0
    finally:
        dual.ctx.profile_enable(0)


@pytest.mark.parametrize("name,dtype,n", [("vit_b16_224", "f16", 1000), ("clip_l14_336", "bf16", 260)])
def test_bench_size_batches_take_the_big_kernels_and_agree(name, dtype, n):
    """BASELINE-size micro-batches (883 ViT-B frames / 224 CLIP-L frames per pass) run the persistent 256x256 GEMM
    (ragged last row panel, head-major QKV stores) and the streaming / two-pass attention kernels, which the few-frame
    tests above never reach.  Size-independent checks: (1) every frame's embedding equals what the same frame gets
    in a 3-frame launch (128x128 GEMM, one attention item per workgroup) BIT FOR BIT - both tilings accumulate each
    output in the same MFMA order, so any difference is a misplaced tile or a stale LDS image; (2) a sample of frames against the quant-aware oracle at the parity bar."""
    from vidmem import synthetic as syn
    spec = V.SPECS[name]
    w = syn.encoder_weights(spec, seed=21)
    enc = _encoder(spec, w, dtype)
    S = spec["image"]
    # vm_encode_micro_batch: frames per pass.  Never more than the call brings; for 577-token sequences (one attention
    # workgroup per (frame, head) and CU) a whole number of rounds of the device's CUs
    mb = enc.micro_batch(10 ** 6)
    assert enc.micro_batch(3) == 3 and 3 < mb < n and enc.micro_batch(mb + 7) == mb
    if enc.tokens > 208:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        assert (mb * spec["heads"]) % cus == 0, (mb, spec["heads"], cus)
    frames = torch.from_numpy(syn.frames_u8(500, n, S, S)).cuda()
    big = enc.embed_frames(frames)
    assert torch.isfinite(big.float()).all()
    picks = [0, 1, 2, n // 2, n // 2 + 1, n // 2 + 2, n - 3, n - 2, n - 1]   # first / middle / ragged-tail panels
    worst = 0.0
    for lo in (0, n // 2, n - 3):
        small = enc.embed_frames(frames[lo:lo + 3])
        worst = max(worst, rel(big[lo:lo + 3].float().cpu().numpy(), small.float().cpu().numpy()))
    print(f"{name}: big-batch vs 3-frame launches, worst rel diff {worst:.2e}")
    assert worst == 0.0
    px = F.preprocess_ref(frames[picks[:3]].cpu().numpy(), S, spec["mean"], spec["std"], layout="chw")
    want = V.vit_forward_ref(spec, w, px, quant=dtype)
    bar = CONTRACT if dtype == "f16" else floor_bar_full_clip()
    assert rel(big[:3].float().cpu().numpy(), want) < bar


def test_long_attention_variants_agree(tmp_path):
    """The 577-token attention has four builds (csrc/attention.hip): two query tiles per walk with a lazily raised
    softmax reference, as a PERSISTENT workgroup per CU whose idle waves fetch the next (frame, head)'s K / V rows into
    registers during the walk (what the release library runs); the same walk with one workgroup per item and an LDS-DMA
    fill (VIDMEM_ATTN_PERSIST=0) - identical bits required, on a batch that gives some CUs two items and in both the
    all-rows and the CLS-rows-only layer; one tile per walk (VIDMEM_ATTN_PAIR=0); and the two-pass softmax
    against the exact row maximum (VIDMEM_ATTN_ONLINE=0).  Same softmax in all of them: the embeddings may differ by the
    rounding of the 16-bit probabilities only.  The other builds are reachable only in the DEVELOPER library
    (make -C csrc dev: -DVM_DEV_SWITCHES; the release library reads no environment variable), where the switches are
    read once per process, hence one child process per variant; skipped when that library has not been built."""
    import subprocess
    import sys
    from vidmem import _lib
    if not os.path.exists(_lib.DEV_LIB_PATH):
        pytest.skip("libvidmem_dev.so not built (make -C real-time-brain-inspired-video-memory_amd/csrc dev)")
    prog = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import vidmem
from vidmem import _lib
_lib.use_dev_library()
from vidmem import synthetic as syn
from vidmem.encoder import FrameEncoder
spec = dict(arch="t", image=336, patch=14, hidden=256, layers=2, heads=4, mlp=512, act="quick_gelu", ln_eps=1e-5,
            pre_ln=True, patch_bias=False, proj_dim=0, mean=(0.5,) * 3, std=(0.5,) * 3)
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=3, std=0.05), "f16")
assert enc.tokens == 577
px = syn.normal(77, "px", (70, 3, 336, 336))     # 70 frames x 4 heads = 280 items > the CUs of the device
np.save(sys.argv[2], enc.encode_patches(enc.patches_from_pixels(torch.from_numpy(px))).float().cpu().numpy())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, env in (("pair", {}), ("one_item", {"VIDMEM_ATTN_PERSIST": "0"}), ("single", {"VIDMEM_ATTN_PAIR": "0"}),
                      ("two_pass", {"VIDMEM_ATTN_ONLINE": "0"})):
        path = str(tmp_path / f"{name}.npy")
        subprocess.check_call([sys.executable, "-c", prog, root, path], env=dict(os.environ, **env))
        outs[name] = np.load(path)
    assert np.isfinite(outs["pair"]).all()
    assert np.array_equal(outs["one_item"], outs["pair"]), rel(outs["one_item"], outs["pair"])
    for name in ("single", "two_pass"):
        d = rel(outs[name], outs["pair"])
        print(f"{name} vs pair: {d:.2e}")
        # one tile per walk is the same arithmetic (measured: bit-identical); the two-pass build exponentiates against a
        # different reference, so every 16-bit probability rounds independently (fp16: 2^-11 = 4.9e-4 per value;
        # measured 3.5e-4 on the embedding of this two-layer stack)
        assert d < (1e-6 if name == "single" else 6e-4), (name, d)

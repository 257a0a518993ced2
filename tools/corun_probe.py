"""Developer probe: can an HBM-bound, low-register kernel make progress on a second stream WHILE the encoder's GEMMs
run, and what does it cost them?  (The persistent GEMM holds 2 waves x 224 VGPRs per SIMD and 156 KiB of LDS per CU:
only a kernel of <= 64 VGPRs and no LDS fits beside it.)  Stream A: encoder passes.  Stream B: torch elementwise
kernels (few registers) streaming a 1 GiB buffer."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
spec = specs.VIT_B16_224
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), "f16")
fr = torch.randint(0, 256, (441, 224, 224, 3), device="cuda", dtype=torch.uint8)
pat = enc.preprocess(fr)
x = torch.empty(256 << 20, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
N = 6
def enc_loop():
    with torch.cuda.stream(sa):
        for _ in range(N): enc.encode_patches(pat)
def copy_loop(n):
    with torch.cuda.stream(sb):
        for _ in range(n): torch.add(x, 1.0, out=y)     # 2 GiB of traffic per call
for _ in range(2): enc.encode_patches(pat)
copy_loop(2); torch.cuda.synchronize()
t0 = time.perf_counter(); enc_loop(); torch.cuda.synchronize(); t_enc = (time.perf_counter() - t0) / N
t0 = time.perf_counter(); copy_loop(40); torch.cuda.synchronize(); t_copy = (time.perf_counter() - t0) / 40
print(f"alone: encoder pass {t_enc*1e3:.2f} ms, copy {t_copy*1e3:.3f} ms = {2*x.numel()*4/t_copy/1e12:.2f} TB/s")
ncopy = int(N * t_enc / t_copy * 0.5)   # about half the encoder's time worth of copies, if they ran alone
ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(sa): ea.record()
with torch.cuda.stream(sb): ca.record()
enc_loop(); copy_loop(ncopy)
with torch.cuda.stream(sa): eb.record()
with torch.cuda.stream(sb): cb.record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"together: {N} encoder passes took {ea.elapsed_time(eb):.1f} ms on their stream (alone {N*t_enc*1e3:.1f}), "
      f"{ncopy} copies took {ca.elapsed_time(cb):.1f} ms (alone {ncopy*t_copy*1e3:.1f}); wall {wall*1e3:.1f} ms vs "
      f"serial {(N*t_enc + ncopy*t_copy)*1e3:.1f} ms")

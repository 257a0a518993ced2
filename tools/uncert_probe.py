"""Developer probe (DESIGN 9.6 of round 3): which per-chunk searches go to the exhaustive redo when a clip is processed
twice, and why.  Rebuilds the extractor leg's second pass: a 100 k-row random memory + the clip's 4,096 frame
embeddings (first pass), then every 16-frame chunk again: top-10 of its 16 queries (list scan, KL = 16), flags read back
per launch (vm_topk_flag: 1 = gap below the fp32 bound, 3 = candidate overflow), then the append.  For every flagged
query: the exact scores around rank k .. KL from the all-pairs fp64 kernel, i.e. the gap the certification needed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from _dev import maybe_dev; maybe_dev()
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
from vidmem.memory import EmbeddingMemory
R, D, k, NF = 100_000, 768, 10, 4096
QC = int(sys.argv[1]) if len(sys.argv) > 1 else 16        # queries per search (16 = one chunk)
g = torch.Generator(device="cuda").manual_seed(7)
spec = specs.VIT_B16_224
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), "f16")
fr = torch.randint(0, 256, (NF, 224, 224, 3), generator=torch.Generator(device="cuda").manual_seed(11), device="cuda",
                   dtype=torch.uint8)
emb = torch.cat([enc.embed_frames(fr[i:i + 512]) for i in range(0, NF, 512)])
del fr
mem = EmbeddingMemory(R + 2 * NF, D, "f16", ring=False)
x = torch.randn((R, D), generator=g, device="cuda")
mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
mem.append(emb)                                           # first pass
KL = 16
eps = 2.0 * (D + 8) * 2.0 ** -24
print(f"fp32 certification bound eps = {eps:.3e}; second pass, {QC} queries per search")
flagged = []
for c0 in range(0, NF, QC):
    q = emb[c0:c0 + QC]
    s, r = mem.topk(q, k)
    fl = mem._scratch.flags[:q.shape[0]].clone()
    if int(fl.sum()):
        rows_now = len(mem)
        allrows = mem.rows_tensor()[:rows_now]
        for qi in fl.nonzero().flatten().tolist():
            ex = mem.cosine_exact(q[qi:qi + 1], allrows)[0]
            top = torch.sort(ex, descending=True).values[:KL + 2].tolist()
            flagged.append((c0 + qi, int(fl[qi]), rows_now, top))
    mem.append(q)
print(f"flagged {len(flagged)} of {NF} queries; redone {mem.uncertified_count}")
for (qi, why, rows_now, top) in flagged[:40]:
    gap = top[k - 1] - top[KL]
    print(f"  query {qi}: flag {why} ({'gap' if why == 1 else 'overflow'}), rows {rows_now}, exact rank {k} = {top[k-1]:.7f}, "
          f"rank {KL + 1} (best row a KL = {KL} list can reject) = {top[KL]:.7f}, gap {gap:.3e} = {gap / eps:.2f} eps; "
          f"ranks 1..3 {[round(v, 6) for v in top[:3]]}")

"""Developer probe: emit scan vs list scan on the same inputs (flag counts, result equality)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch, vidmem
    from vidmem.memory import EmbeddingMemory
    for D, M, Q, k in ((128, 70_000, 200, 10), (768, 70_000, 200, 10), (128, 70_000, 64, 10), (256, 200_000, 130, 10)):
        rng = np.random.default_rng(33)
        m = torch.tensor(rng.standard_normal((M, D)), dtype=torch.float32).to(torch.float16)
        q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
        mem = EmbeddingMemory(M, D, "f16"); mem.append(m)
        mem.reset_uncertified()
        s, r = mem.topk(q, k, redo=False)
        flagged = mem._scratch.flags[:Q].cpu().numpy()
        s2, r2 = mem.topk(q, k, exact=True)
        bad = (r != r2).any(dim=1).cpu().numpy()
        print(f"EMIT={os.environ.get('VIDMEM_TOPK_EMIT','1')} D={D} M={M} Q={Q}: flagged {int(flagged.sum())}, wrong-without-redo {int(bad.sum())}, wrong&unflagged {int((bad & (flagged == 0)).sum())}", flush=True)
else:
    for e in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, VIDMEM_TOPK_EMIT=e))

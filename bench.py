#!/usr/bin/env python3
"""bench.py - the hot path of BASELINE.json on N MI355X GPUs of one node.

One "step" = one pass of the hot path over one batch of synthetic frames, per GPU:
    uint8 BGR frames (resident in HBM) -> preprocess -> ViT-B/16 encoder (fp16) -> L2-normalised embeddings
    -> [N>1: RCCL all-gather of the step's query embeddings] -> cosine top-10 of every query over the local
    100k x 768 memory shard -> [N>1: RCCL all-gather of the per-shard candidates + merge] -> append the new
    embeddings to the (ring) memory shard.
Workload = BASELINE.json configs[1]: "ViT-B/16 encoder, 4k frames (chunk_size=16), fp16; top-10 over 100k x 768".
Frames are sharded by chunk, memory rows by rank; per-GPU work is fixed as N grows ("weak").

Prints ONE JSON line (rank 0).  `value` = frame embeddings per second over all GPUs, inputs resident in HBM.
Extra objects: `roofline` (dominant kernel, HIP-event timed inside the timed region) and, at N=1 only, `cpu_baseline`
(the oracle on the host cores, bounded sample), `knn` (the kNN half of the metric on a 1M x 768 index), `streaming`
(BASELINE configs[4]: 16 x 1080p chunks through one hipGraph) and `c3` (BASELINE configs[2]: CLIP-ViT-L/14-336 bf16).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between the ranks' processes); exported on the boxes already
os.environ.setdefault("VIDGRAPH_LOG_LEVEL", "WARNING")   # the drop-in classes log to stdout like the reference's; this
                                                          # program's stdout carries exactly one JSON line

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chunks-per-step", type=int, default=110,
                    help="chunks of 16 frames encoded per step per GPU (110 chunks = 1,760 frames = two encoder passes of "
                         "880, which the default two-stream schedule runs side by side; 55 = one pass, one stream)")
    ap.add_argument("--memory-rows", type=int, default=None,
                    help="rows of the memory shard per GPU (default: 100,000 = BASELINE configs[1] on one GPU; "
                         "1,048,576 = BASELINE configs[3], 8 M rows over 8 GPUs, when --gpus > 1)")
    ap.add_argument("--no-extractor", action="store_true", help="skip the plugin-path leg (process_video on a host clip)")
    ap.add_argument("--extractor-frames", type=int, default=4096)
    ap.add_argument("--extractor-long-frames", type=int, default=16384,
                    help="frames of the second, longer clip of the plugin-path leg (a 9-minute video at 30 fps; the "
                         "leg's fixed head and tail weigh less on it); 0 = skip")
    ap.add_argument("--look-ahead-chunks", type=int, default=0,
                    help="chunks per encoder call in the plugin-path leg (0 = auto, the config default: two encoder passes)")
    ap.add_argument("--no-two-stream", action="store_true", help="skip the one-stream A/B beside the main leg")
    ap.add_argument("--schedule", default="auto", choices=["auto", "one_stream", "two_streams"],
                    help="encoder schedule of the main leg (developer A/B; auto is the product default)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the mfma_ceiling leg (vm_probe_mfma)")
    ap.add_argument("--ceiling-seconds", type=float, default=1.5, help="seconds per vm_probe_mfma variant")
    ap.add_argument("--no-rccl-world1", action="store_true",
                    help="skip the one-rank RCCL self-test (tools/rccl_world1.py as a child process)")
    ap.add_argument("--no-c4", action="store_true", help="skip the one-GPU rank-share leg of BASELINE configs[3]")
    ap.add_argument("--c4-world", type=int, default=8, help="ranks of the job whose per-rank step the c4 leg runs")
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--knn-rows", type=int, default=1_000_000)
    ap.add_argument("--no-streaming", action="store_true")
    ap.add_argument("--no-c3", action="store_true", help="skip the CLIP-ViT-L/14-336 bf16 leg (BASELINE configs[2])")
    ap.add_argument("--c3-frames", type=int, default=2048 + 224, help="frames per timing of the C3 leg (rounded down "
                    "to whole encoder passes of 224 frames; BASELINE configs[2] names 32k frames)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to rehearse N>1 on one GPU")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--dry-run", action="store_true", help="launcher self-test: rendezvous + barrier on the CPU (gloo), "
                    "rank 0 prints a stub line; no GPU work, nothing measured")
    ap.add_argument("--stream-rows", type=int, default=2_097_152, help="rolling memory rows of the C5 latency leg")
    ap.add_argument("--stream-replays", type=int, default=2020, help="chunks the host producer feeds in the C5 leg")
    ap.add_argument("--stream-period-ms", type=float, default=1000.0 / 30.0,
                    help="period of the feed: one chunk per 30 fps frame time by default (16x real time)")
    return ap.parse_args()


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores() -> int:
    """Physical cores this process may actually run on: /proc/cpuinfo core ids, cut by the affinity mask and by a
    cgroup CPU quota (a GPU box hands a one-GPU job a share of its host: 128 threads on a 16-CPU share is what made
    round 2's CPU line 98 GFLOP/s)."""
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip() and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    n = len(phys) or (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(spec, weights, mem_cpu_f16, k, enc_frames=256, r1_rows=10_000, r1_queries=4):
    """the oracle timed on the host cores, two stated lines.
      R1 "reference-faithful": oracle.similarity_ref.calculate_batch_similarities_ref - the pure-Python loop of
         src/components/pre_llm_injector.py:346-388, ONE thread (the reference is single-threaded) - on a
         `r1_rows`-row subset for `r1_queries` queries, extrapolated linearly to the full shard and chunk.
      R2 "best-effort CPU": the same step on the cores this process may use: preprocess (oracle.frames_ref) + the fp32
         encoder as one batched torch.nn.functional forward (oracle.vit_ref.vit_forward_fast: weights converted once,
         fused attention; held to the restatement and to `transformers` by tests/golden/make_vit_golden.py) on
         `enc_frames` frames, + the same cosine semantics as one fp32 torch matmul + top-k over the full shard.
    `value` = R2's frame-embeddings/s for the whole step (encode + top-k of every frame over the shard)."""
    from oracle import frames_ref, similarity_ref, vit_ref
    from vidmem import synthetic as syn
    nthreads = usable_cores()
    torch.set_num_threads(nthreads)
    D = mem_cpu_f16.shape[1]
    R = mem_cpu_f16.shape[0]
    # ---- R2: all usable cores
    frames = syn.frames_u8(1234, enc_frames, spec["image"], spec["image"])
    wt = vit_ref.fast_weights(weights)
    vit_ref.vit_forward_fast(spec, weights, frames_ref.preprocess_ref(frames[:8], spec["image"], spec["mean"], spec["std"],
                                                                      layout="chw"), tensors=wt)   # warm-up
    t0 = time.perf_counter()
    px = frames_ref.preprocess_ref(frames, spec["image"], spec["mean"], spec["std"], layout="chw")
    emb = vit_ref.vit_forward_fast(spec, weights, px, tensors=wt)
    t1 = time.perf_counter()
    mem32 = torch.from_numpy(mem_cpu_f16).float()
    q32 = torch.from_numpy(emb).float()
    t2 = time.perf_counter()
    sc = (q32 @ mem32.T) / (q32.norm(dim=1, keepdim=True) * mem32.norm(dim=1).unsqueeze(0))
    torch.topk(sc, k, dim=1)
    t3 = time.perf_counter()
    r2_enc_fps = enc_frames / (t1 - t0)
    r2_knn_qps = enc_frames / (t3 - t2)
    r2_fps = enc_frames / ((t1 - t0) + (t3 - t2))
    # ---- R1: one thread, pure Python
    rows_sub = min(r1_rows, R)
    existing = {f"c{i}": mem_cpu_f16[i].astype(np.float64).tolist() for i in range(rows_sub)}
    queries = [emb[i].astype(np.float16).astype(np.float64).tolist() for i in range(r1_queries)]
    t4 = time.perf_counter()
    similarity_ref.calculate_batch_similarities_ref(queries, existing, k)
    t5 = time.perf_counter()
    pairs_per_s = rows_sub * r1_queries / (t5 - t4)
    r1_s_per_query_full = R / pairs_per_s
    return {
        "value": r2_fps, "unit": "frame-embeddings/s", "cores": nthreads, "kind": "port",
        "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
        "sample": f"R2: {enc_frames} frames {spec['image']}x{spec['image']} through oracle/frames_ref + "
                  f"oracle/vit_ref.vit_forward_fast (fp32, one batched torch.nn.functional forward, {nthreads} threads = "
                  f"usable physical cores: {t1 - t0:.2f} s) + fp32 matmul cosine + top-{k} of those {enc_frames} "
                  f"queries over all {R} x {D} rows ({t3 - t2:.2f} s); R1: see r1_reference_faithful",
        "r2_best_effort_cpu": {"threads": nthreads, "encoder_frames_per_s": r2_enc_fps,
                               "encoder_gflops": r2_enc_fps * 35.13,
                               "knn_queries_per_s": r2_knn_qps, "step_frames_per_s": r2_fps},
        "r1_reference_faithful": {
            "what": "oracle.similarity_ref.calculate_batch_similarities_ref (pure-Python restatement of "
                    "src/components/pre_llm_injector.py:346-388), 1 thread",
            "sample": f"{r1_queries} queries x {rows_sub} of {R} rows x {D} in {t5 - t4:.2f} s, extrapolated "
                      f"linearly in rows",
            "pairs_per_s": pairs_per_s, "seconds_per_query_full_shard": r1_s_per_query_full,
            "queries_per_s_full_shard": 1.0 / r1_s_per_query_full, "cores": 1},
    }


def pmc_traffic(kernel: str, shape: str, leg: str = "main"):
    """HBM/fabric bytes per launch of `kernel` from the newest committed PMC summary (tools/pmc_traffic.py; the --pmc
    passes are separate rocprofv3 runs of this bench, as MI355X_MICROARCH.md prescribes).  The number is only returned
    when the summary was taken on the launch shape this run uses; the source is stamped either way."""
    import glob
    import hashlib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        raw = open(path, "rb").read()
        d = json.loads(raw)
        ent = d.get("kernels", {}).get(kernel)
        if ent is None:
            continue
        blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()
        prof_shape = d.get("shapes", {}).get(leg)
        src = {"file": os.path.relpath(path, ROOT), "git_blob": blob, "kernel": kernel,
               "profile_shape": prof_shape, "run_shape": shape}
        ok = prof_shape == shape
        if not ok:
            src["note"] = "launch shape of the profile differs from this run: traffic withheld"
        return (ent.get("traffic_bytes") if ok else None), src
    return None, {"file": None, "kernel": kernel, "run_shape": shape, "note": "no committed PMC summary holds this kernel"}


def note(msg: str) -> None:
    """Progress line on stderr (rank 0's stdout carries exactly one JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
        sys.stderr.flush()


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (torch.distributed.run) before
    this process has touched the GPU, relay rank 0's JSON line, return the child's exit code.  Never re-exec."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    elif proc.stdout:
        sys.stderr.write(proc.stdout[-4000:])
    return proc.returncode if (proc.returncode != 0 or line is not None) else 1


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))       # parent: no GPU call before or after this point
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            dist.barrier()
            dist.destroy_process_group()
            assert int(t.item()) == world
        if rank == 0:
            print(json.dumps({"metric": "dry-run (launcher self-test, nothing measured)", "value": None,
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup}))
        return
    # ---- RCCL on this box, before this process touches the GPU: a child process runs dist.ShardedRetriever on a
    # one-rank "nccl" group with both all-gathers forced (tools/rccl_world1.py); a hang there is a timeout here
    rccl_world1 = None
    if world == 1 and rank == 0 and not args.no_rccl_world1:
        import subprocess
        note("rccl world-1 self-test (child process)")
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py"), "--rows", "100000",
                                 "--queries", str(args.chunks_per_step * 16), "--topk", str(args.topk)],
                                capture_output=True, text=True, timeout=420)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            rccl_world1 = json.loads(lines[-1]) if lines else {"ok": False, "error": cp.stderr[-600:]}
            rccl_world1["exit_code"] = cp.returncode
        except Exception as exc:   # a timeout or a missing librccl must not cost the run its headline
            rccl_world1 = {"ok": False, "error": f"{type(exc).__name__}: {exc}"[:600]}
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import vidmem  # noqa: F401
    from vidmem import specs, synthetic as syn, _lib
    from vidmem.encoder import FrameEncoder
    from vidmem.memory import EmbeddingMemory, topk_merge
    from vidmem.dist import ShardedRetriever

    spec = specs.VIT_B16_224
    D, k = spec["hidden"], args.topk
    F = args.chunks_per_step * 16
    weights = syn.encoder_weights(spec, seed=42)
    enc = FrameEncoder(spec, weights, dtype="f16", device=local_rank, schedule=args.schedule)
    ctx = enc.ctx

    # memory shard: R L2-normalised rows (seed 7 + rank), ring so the size stays R while steps append
    R = args.memory_rows if args.memory_rows else (1_048_576 if world > 1 else 100_000)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    mem_rows = torch.randn((R, D), generator=g, device=dev, dtype=torch.float32)
    mem_rows = (mem_rows / mem_rows.norm(dim=1, keepdim=True)).to(torch.float16)
    memory = EmbeddingMemory(R, D, "f16", ring=True, device=local_rank)
    memory.append(mem_rows)
    retriever = ShardedRetriever(memory, rank, world)

    # synthetic frames resident in HBM: a pool of distinct uint8 frames, cycled through the steps
    # (distinct frames for every step up to 40 steps - the timed ones and the event-timed ones behind them: re-embedding a frame plants exact duplicates in the memory, and
    # enough exact ties make a query uncertifiable on the fast path)
    pool_steps = min(40, args.warmup + 2 * args.steps + 2)
    gf = torch.Generator(device=dev).manual_seed(1234 + rank)
    frame_pool = torch.randint(0, 256, (pool_steps, F, 224, 224, 3), generator=gf, device=dev, dtype=torch.uint8)

    def step(i):
        frames = frame_pool[i % pool_steps]
        emb = enc.embed_frames(frames)                      # [F, D] fp16, L2-normalised
        scores, rows = retriever.search(emb, k)             # global top-k for this rank's F queries
        memory.append(emb)
        return scores, rows

    note(f"main leg: {F} frames per step, {R}-row shard, world {world}")
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    # per-kernel breakdown: ONE extra untimed step with every launch bracketed by HIP events (two event records cost
    # ~7 us per launch here, 12 % of the step).  While per-kernel timing is on, the encoder's default schedule
    # (VM_SCHED_AUTO, include/vidmem.h) runs ONE stream: with two, a kernel's event duration includes its wait for the
    # other stream's kernels.
    mb_frames = enc.micro_batch(F)          # frames per encoder pass (csrc/encoder.hip, micro_batch_of)
    passes = -(-F // mb_frames)
    two_streams = passes >= 2 and args.schedule != "one_stream"   # what VM_SCHED_AUTO does with this step when timing is off
    per_step_events = passes * (7 * spec["layers"] + 8) + 16
    ctx.profile_enable(per_step_events + 64)
    step(args.warmup)
    breakdown = ctx.profile_read()
    # the two GEMM kernel instantiations, as rocprofv3 names them: the 16-bit-store epilogue (patch embed, QKV,
    # attention projection, FC2) and the GELU epilogue (FC1)
    kern_cats = {"gemm256p_kernel<f16, STORE16>": ("gemm_patch", "gemm_qkv", "gemm_resid"),
                 "gemm256p_kernel<f16, GELU16>": ("gemm_act",)}
    dom = max(kern_cats, key=lambda kname: sum(breakdown[c][0] for c in kern_cats[kname]))
    ctx.profile_enable(0)

    # ---- THE TIMED REGION: exactly args.steps steps of the product's default path (no per-kernel events) -------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + 1 + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    uncert = retriever.uncertified_total()   # queries the fp32 scan could not certify; redone exhaustively in-step

    # ---- the dominant kernel's launch durations: the SAME steps again with its HIP events on (recorded on the launch
    # stream; the one-stream schedule, see above).  Also the A/B of the two schedules: same embeddings, bit for bit.
    one_stream = None
    prof = breakdown
    if not args.no_profile:
        ctx.profile_enable(args.steps * per_step_events + 64)
        ctx.profile_mask(list(kern_cats[dom]))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + 1 + args.steps + i)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / args.steps
        prof = ctx.profile_read()
        ctx.profile_enable(0)
        ctx.profile_mask(None)
        if two_streams and not args.no_two_stream:
            emb_two = enc.embed_frames(frame_pool[0])
            enc.set_schedule("one_stream")
            emb_one = enc.embed_frames(frame_pool[0])
            enc.set_schedule(args.schedule)
            one_stream = {
                "what": "the same steps on ONE stream (VM_SCHED_ONE_STREAM is what VM_SCHED_AUTO falls back to while "
                        "per-kernel timing is on; these are the steps roofline.avg_launch_ms was taken from, the "
                        "dominant kernel's events recorded: ~0.25 ms per step of event overhead)",
                "frames_per_s": F / dt1, "ms_per_step": dt1 * 1e3,
                "embeddings_bit_identical": bool(torch.equal(emb_two, emb_one)),
            }

    t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    frames_total = F * args.steps * world
    value = frames_total / elapsed

    out = {
        "metric": "frame-embeddings/sec (+ kNN queries/sec, see knn)", "value": value, "unit": "frame-embeddings/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: ViT-B/16-224 encoder fp16, chunks of 16 frames, cosine top-10 "
                                f"over a {R}-row x 768 fp16 memory shard per GPU") if world == 1 else
                               (f"BASELINE configs[3]: {world} GPUs, frames sharded by chunk ({F} per step per GPU, "
                                f"ViT-B/16-224 fp16), memory sharded by row ({R} x 768 fp16 rows per GPU = "
                                f"{R * world} rows), every shard scores all {F * world} queries of a step, all-gather "
                                f"of queries and candidates, global top-{k}"),
                   "frames_per_step_per_gpu": F, "chunk_size": 16, "memory_rows_per_gpu": R, "top_k": k,
                   "parallelism": (f"dp{world}: frames by chunk, memory by row, "
                                   + ("RCCL all-gather over xGMI" if args.backend == "nccl" else
                                      f"{args.backend} all-gather (rehearsal backend, host-staged)")
                                   + " of queries and candidates") if world > 1 else "single GPU"},
        "queries_per_s": value, "uncertified_queries_redone": uncert,
        "encoder_schedule": ("two streams (VM_SCHED_AUTO, the default: the step's %d encoder passes of <= %d frames "
                             "alternate between two internal streams)" % (passes, mb_frames)) if two_streams else
                            "one stream (a step of one encoder pass)",
    }
    if one_stream is not None:
        out["one_stream_encoder"] = one_stream
    if rccl_world1 is not None:
        out["rccl_world1"] = dict(rccl_world1, what=(
            "dist.ShardedRetriever on a ONE-rank nccl (= RCCL) process group with both all-gathers forced, in a child "
            "process before this one touched the GPU (tools/rccl_world1.py): fp16 queries, fp64 scores and int64 rows "
            "all-gathered on device tensors, result compared bit for bit with the local search.  One rank moves nothing "
            "over xGMI: the times are RCCL's per-collective overhead, not a link measurement"))

    if rank == 0:
        # ---- roofline of the dominant kernel (by total time inside the timed region) ------------------------
        T = enc.tokens
        H, M = spec["hidden"], spec["mlp"]
        mbs = []  # micro-batch sizes vm_encode used for F frames (csrc/encoder.hip micro_batch_of)
        mb = mb_frames
        left = F
        while left > 0:
            mbs.append(min(mb, left))
            left -= mbs[-1]
        rows = [b * T for b in mbs]
        # the last layer's projection / FC1 / FC2 run on the CLS rows only (one per frame): rows -> frames there
        Lf = spec["layers"] - 1
        flops = {  # algorithmic FLOPs of one step's launches of each GEMM instantiation (2*M*N*K)
            "gemm_qkv": sum(2.0 * r * 3 * H * H for r in rows) * Lf      # last layer: K, V of every row ...
                        + sum(2.0 * r * 2 * H * H for r in rows),
            "gemm_act": sum(2.0 * r * M * H for r in rows) * Lf,
            "gemm_resid": sum(2.0 * r * H * H + 2.0 * r * H * M for r in rows) * Lf,
            # ... and Q, projection, FC1, FC2 of the CLS rows (category gemm_cls: the 128 x 128 kernel)
            "gemm_cls": sum(2.0 * b * H * H * 2 + 2.0 * b * H * M * 2 for b in mbs),
            "gemm_patch": sum(2.0 * b * (T - 1) * (3 * 16 * 16) * H for b in mbs),
        }
        abytes = {  # operands read once + output written once, per step
            "gemm_qkv": sum(2.0 * (r * H + 3 * H * H + r * 3 * H) for r in rows) * Lf
                        + sum(2.0 * (r * H + 2 * H * H + r * 2 * H) for r in rows),
            "gemm_act": sum(2.0 * (r * H + M * H + r * M) for r in rows) * Lf,
            "gemm_resid": sum(2.0 * (r * H + H * H + r * H) + 2.0 * (r * M + H * M + r * H) for r in rows) * Lf,
            "gemm_patch": sum(2.0 * (b * (T - 1) * 768 + 768 * H + b * (T - 1) * H) for b in mbs),
        }
        ms = sum(prof[c][0] for c in kern_cats[dom])
        launches = sum(prof[c][1] for c in kern_cats[dom])
        nsteps_prof = args.steps if not args.no_profile else 1
        dom_flops = sum(flops[c] for c in kern_cats[dom])          # algorithmic FLOPs of those launches per step
        achieved = dom_flops * nsteps_prof / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM/fabric bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
        shape = f"F{F},mb{mb_frames},R{R},k{k}"
        rocname = "void (anonymous namespace)::gemm256p_kernel<0, %d, 0>" % (0 if "STORE16" in dom else 1)
        traffic, traffic_source = pmc_traffic(rocname, shape)
        out["roofline"] = {
            "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
            "traffic_source": traffic_source,
            "timing_source": ("HIP events on the launch stream over %d steps run right after the timed region on the "
                              "one-stream schedule (one_stream_encoder); the timed region itself runs two streams, where "
                              "an event pair also spans the wait for the other stream's kernels" % args.steps)
                             if (two_streams and not args.no_profile) else
                             ("HIP events on the launch stream over %d steps of the same schedule, run right after the "
                              "timed region" % args.steps if not args.no_profile else "the one event-bracketed step"),
            "traffic_note": "HBM/fabric bytes per launch = 2*FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc "
                            "passes of this bench (traffic_source); algorithmic_bytes = operands + output once per "
                            "launch",
            "algorithmic_bytes": sum(abytes[c] for c in kern_cats[dom]) / max(launches / nsteps_prof, 1),
            "avg_launch_ms": ms / max(launches, 1), "launches": launches,
            "flops_per_launch": dom_flops * nsteps_prof / max(launches, 1),
        }
        out["kernel_time_ms_per_step"] = {c: round(v[0], 4) for c, v in breakdown.items() if v[1]}
        out["kernel_time_note"] = "one untimed step with every launch event-bracketed (adds ~7 us per launch)"
        enc_ms = sum(breakdown[c][0] for c in ("gemm_patch", "gemm_qkv", "gemm_act", "gemm_resid", "gemm_cls", "attention",
                                                "layernorm", "pool"))
        # executed FLOPs (last layer's MLP on the CLS rows only: 33.05 GFLOP per frame, not the 35.13 of every row)
        out["encoder_tflops"] = (specs.flops_per_frame(spec, executed=True) * F / (enc_ms * 1e-3) / 1e12
                                 if enc_ms > 0 else None)
        out["encoder_gflop_per_frame"] = {"executed": specs.flops_per_frame(spec, executed=True) / 1e9,
                                          "every_row": specs.flops_per_frame(spec) / 1e9}

    # ---- what this GPU sustains on 16-bit MFMA work, here and now (vm_probe_mfma) ---------------------
    # The chip lowers its clock under matrix load: the data sheet's 2.5 PFLOP/s is not what ANY kernel reaches on random
    # operands.  Three synthetic loops - registers only; + the GEMM tile's fragment reads; + its L2 -> LDS staging -
    # put a measured ceiling beside the fraction of the data-sheet peak.
    if rank == 0 and world == 1 and not args.no_ceiling:
        note("mfma ceiling leg")
        secs = args.ceiling_seconds
        names = ("registers_only", "with_fragment_reads", "with_fragment_reads_and_staging")
        ceil = {n: ctx.probe_mfma(v, secs) for v, n in enumerate(names)}
        zero = ctx.probe_mfma(2, min(secs, 1.0), zero_operands=True)
        out["mfma_ceiling"] = {
            "what": "vm_probe_mfma (csrc/probe.hip): v_mfma_f32_16x16x32_f16 loops with the operand pattern of the GEMM's "
                    "128 x 64 wave tile, two waves per SIMD on every CU, RANDOM operands, %.1f s each: (0) registers only, "
                    "(1) + 12 conflict-free ds_read_b128 per 32 MFMAs, (2) + 4 KiB of LDS-DMA per wave and 32 MFMAs out "
                    "of L2 = the K loop of gemm256p_kernel with no epilogue, barrier, tile boundary, miss or store" % secs,
            "unit": "TFLOP/s", **ceil,
            "with_fragment_reads_and_staging_zero_operands": zero,
            "data_sheet_peak": MFMA_PEAK_TFLOPS,
        }
        sustained = ceil["with_fragment_reads_and_staging"]
        out["roofline"]["sustained_ceiling"] = sustained
        out["roofline"]["frac_of_sustained"] = out["roofline"]["achieved"] / sustained if sustained > 0 else None
        out["roofline"]["sustained_note"] = ("sustained_ceiling = mfma_ceiling.with_fragment_reads_and_staging, measured "
                                             "in this run on this GPU; frac stays achieved / the 2.5 PFLOP/s data sheet")

    # ---- kNN half of the metric: Q=16 queries/launch over a 1M x 768 index (single GPU part of every rank 0) ----
    if rank == 0 and world == 1 and not args.no_knn:
        note("knn leg")
        del frame_pool
        Mk = args.knn_rows
        big = EmbeddingMemory(Mk, D, "f16", device=local_rank)
        gk = torch.Generator(device=dev).manual_seed(7)
        for lo in range(0, Mk, 250_000):
            n = min(250_000, Mk - lo)
            x = torch.randn((n, D), generator=gk, device=dev, dtype=torch.float32)
            big.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
        out["knn"] = {"index": f"{Mk} x {D} f16", "k": k, "batches": {}}
        bytes_scan = Mk * D * 2
        for Qk in (16, 64, 256):
            q = torch.randn((Qk, D), generator=gk, device=dev, dtype=torch.float32).to(torch.float16)
            for _ in range(3):
                big.topk(q, k)
            torch.cuda.synchronize()
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                big.topk(q, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ctx.profile_enable(8 * reps)          # kernel split from a separate, event-bracketed pass
            for _ in range(reps):
                big.topk(q, k)
            p2 = ctx.profile_read()
            ctx.profile_enable(0)
            scan_ms = p2["topk_scan"][0] / reps
            out["knn"]["batches"][f"Q{Qk}"] = {
                "queries_per_s": Qk * reps / dt, "ms_per_launch": 1e3 * dt / reps, "scan_kernels_ms": scan_ms,
                "finalize_kernels_ms": p2["topk_finalize"][0] / reps, "scan_GBps": bytes_scan / (scan_ms * 1e-3) / 1e9,
            }
        best = max(out["knn"]["batches"].values(), key=lambda v: v["queries_per_s"])
        q16 = out["knn"]["batches"]["Q16"]
        out["knn"]["queries_per_s"] = best["queries_per_s"]
        out["knn"]["uncertified_queries_redone"] = big.uncertified_count
        out["knn"]["roofline"] = {"bound": "hbm", "kernel": "topk_scan_kernel<f16, KL=16, QT=1> (Q=16)",
                                  "achieved": q16["scan_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": q16["scan_GBps"] / HBM_PEAK_GBPS, "traffic": None,
                                  "algorithmic_bytes": bytes_scan}
        tk, tk_src = pmc_traffic("void (anonymous namespace)::topk_scan_kernel<0, 16, 1>", f"F{F},mb{mb_frames},R{R},k{k}")
        out["knn"]["roofline"]["traffic"] = tk if Mk == 1_000_000 else None
        out["knn"]["roofline"]["traffic_source"] = tk_src
        big.close()

    # ---- BASELINE configs[3] on ONE GPU: one rank's share of a step of the 8-GPU job, without the collectives --------
    # Every rank of that job encodes its own 880 frames and then scores ALL ranks' 8 x 880 = 7,040 queries against its
    # 1,048,576-row shard (dist.ShardedRetriever: all-gather queries -> local exhaustive top-k -> all-gather candidates
    # -> 8-part merge of its own 880).  Here the other ranks' queries are embeddings of other synthetic frames, the
    # other ranks' candidate lists are this shard's lists of their queries; nothing crosses a link.
    if rank == 0 and world == 1 and not args.no_c4:
        note("c4 rank-share leg")
        W4 = args.c4_world
        R4 = 1_048_576
        shard = EmbeddingMemory(R4, D, "f16", ring=True, device=local_rank)
        g4 = torch.Generator(device=dev).manual_seed(4004)
        for lo in range(0, R4, 262_144):
            x = torch.randn((262_144, D), generator=g4, device=dev, dtype=torch.float32)
            shard.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
        others = torch.cat([enc.embed_frames(torch.randint(0, 256, (F, 224, 224, 3), generator=g4, device=dev,
                                                           dtype=torch.uint8)) for _ in range(W4 - 1)])
        # distinct frames for every step of this leg (re-embedding a frame plants exact duplicates of a query)
        fr4 = torch.randint(0, 256, (17, F, 224, 224, 3), generator=g4, device=dev, dtype=torch.uint8)

        def share_step(i, all_ranks=True):
            emb = enc.embed_frames(fr4[i])
            if all_ranks:
                q_all = torch.cat([emb, others])                    # stands for the query all-gather's output
                s_l, r_l = shard.topk(q_all, k, row_stride=W4, row_offset=0)      # scan + device-side flagged redo
                s_m, r_m = topk_merge(ctx, s_l.view(W4, F, k), r_l.view(W4, F, k))  # W4-part merge of the own F
            else:
                s_m, r_m = shard.topk(emb, k, row_stride=W4, row_offset=0)
            shard.append(emb)
            return s_m, r_m

        def timed(all_ranks, first, n=6):
            for i in range(2):
                share_step(first + i, all_ranks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                share_step(first + 2 + i, all_ranks)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        shard.reset_uncertified()
        ms_share = timed(True, 0)
        redone4 = shard.uncertified_count
        ms_alone = timed(False, 8)
        shard.reset_uncertified()
        ctx.profile_enable(4096)
        share_step(16, True)
        bd4 = ctx.profile_read()
        ctx.profile_enable(0)
        redone4_prof = shard.uncertified_count
        Q4 = W4 * F
        scan_ms = bd4["topk_scan"][0]
        topk_ms = sum(bd4[c][0] for c in ("topk_scan", "topk_finalize", "topk_exact", "topk_merge"))
        eff = ms_alone / ms_share
        out["c4_rank_share"] = {
            "workload": f"BASELINE configs[3], one rank's step of the {W4}-GPU job on ONE GPU, no collectives: {F} frames "
                        f"encoded (ViT-B/16-224 fp16) + {Q4} queries x {R4} x {D} fp16 rows top-{k} (scan + flagged "
                        f"redo) + {W4}-part merge of the own {F} + append",
            "ms_per_step": ms_share, "topk_ms": topk_ms, "scan_ms": scan_ms,
            "scan_tflops": 2.0 * Q4 * R4 * D / (scan_ms * 1e-3) / 1e12,
            "scan_frac_of_mfma_peak": 2.0 * Q4 * R4 * D / (scan_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
            "uncertified_queries_redone": redone4, "uncertified_queries_redone_in_the_profiled_step": redone4_prof,
            "same_shard_own_queries_only_ms_per_step": ms_alone,
            "projected_weak_scaling_efficiency": eff, "projected_speedup_at_world": eff * W4,
            # the two all-gathers of the real job, as fields: what one rank ends up holding per step, what crosses each
            # of its links (one peer's share), and what that costs at the per-link rate of the 8-GPU xGMI mesh
            "all_gather_bytes_per_rank_step": {"queries_f16": Q4 * D * 2, "candidates_f64_i64": W4 * Q4 * k * 16},
            "all_gather_bytes_per_link_step": {"queries_f16": F * D * 2, "candidates_f64_i64": Q4 * k * 16},
            "all_gather_expected_ms_per_step": {
                "assumption": "fully connected xGMI, ~153 GB/s per link and direction (SURVEY.md 5), every peer's share "
                              "over its own link in parallel, + ~30 us of RCCL launch latency per collective (three "
                              "collectives per step: queries, scores, rows); NOT measured",
                "queries_f16": (F * D * 2) / 153e9 * 1e3 + 0.03,
                "candidates_f64_i64": (Q4 * k * 16) / 153e9 * 1e3 + 0.06,
            },
            "projection_note": f"PROJECTED, NOT MEASURED: (step of one GPU alone on the same {R4}-row shard, {F} queries) / "
                               f"(this rank-share step); excludes the two all-gathers of the real job "
                               f"({Q4 * D * 2 / 1e6:.1f} MB of queries + {Q4 * k * 16 / 1e6:.1f} MB of candidates per rank "
                               f"and step over xGMI)",
            "kernel_time_ms_per_step": {c: round(v[0], 4) for c, v in bd4.items() if v[1]},
        }
        del fr4, others
        shard.close()

    # ---- the plugin path itself: FrameEmbeddingExtractor.process_video over a 4,096-frame clip, chunk_size = 16 --------
    # (BASELINE configs[1] literally; reference loop src/pipeline/vlm_extractor.py:44-74).  `value` above batches 55
    # chunks per encoder call from frames resident in HBM; this leg drives the drop-in class from a file on the host:
    # frame reads, pinned staging, H2D, per-chunk top-k + append, JSON and metrics files included.
    if rank == 0 and world == 1 and not args.no_extractor:
        note("extractor (plugin path) leg")
        import asyncio
        import tempfile
        from vidmem import config as vcfg
        from vidmem.extractor import FrameEmbeddingExtractor
        with tempfile.TemporaryDirectory() as td:
            nfr = args.extractor_frames
            clip = os.path.join(td, "clip.npy")
            np.save(clip, np.random.default_rng(11).integers(0, 256, size=(nfr, 224, 224, 3), dtype=np.uint8))
            cwd = os.getcwd()
            os.chdir(td)                                   # metrics/ and logs/ of the runs land in the temp dir
            ext = {}
            try:
                for la in (1, args.look_ahead_chunks):
                    cfg = vcfg.from_dict({
                        # the .npy source plays at 30 fps (extractor.open_source): 16-frame chunks, all 16 frames picked
                        "video": {"chunk_size_seconds": 16.0 / 30.0 + 1e-9, "frames_per_chunk": 16},
                        "encoder": {"arch": "vit_b16_224", "dtype": "f16", "top_k": k, "look_ahead_chunks": la},
                        "memory": {"capacity": R + 2 * nfr, "ring": False}})
                    # the reference's store only grows (Neo4j): R rows to start with, the clip's frames appended twice
                    exm = EmbeddingMemory(R + 2 * nfr, D, "f16", ring=False, device=local_rank)
                    exm.append(mem_rows)
                    ex = FrameEmbeddingExtractor(cfg, encoder=enc, memory=exm)
                    asyncio.run(ex.process_video(clip, os.path.join(td, "warm.json")))       # warm-up run
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    asyncio.run(ex.process_video(clip, os.path.join(td, "out.json")))
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    ext[la] = nfr / dt
                    exm.close()
                if args.extractor_long_frames > nfr:
                    # the same class on a longer clip (distinct random frames, kernels warm from the runs above)
                    nlong = args.extractor_long_frames
                    clip2 = os.path.join(td, "clip_long.npy")
                    np.save(clip2, np.random.default_rng(12).integers(0, 256, size=(nlong, 224, 224, 3), dtype=np.uint8))
                    exm = EmbeddingMemory(R + nlong, D, "f16", ring=False, device=local_rank)
                    exm.append(mem_rows)
                    # (cfg of the last run above: the configured look-ahead; its memory section is unused - a memory is passed)
                    ex = FrameEmbeddingExtractor(cfg, encoder=enc, memory=exm)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    asyncio.run(ex.process_video(clip2, os.path.join(td, "out_long.json")))
                    torch.cuda.synchronize()
                    ext["long"] = nlong / (time.perf_counter() - t0)
                    exm.close()
            finally:
                os.chdir(cwd)
        out["extractor"] = {
            "workload": f"FrameEmbeddingExtractor.process_video on a {nfr}-frame 224x224 .npy clip (host file), chunks of "
                        f"16 frames, top-{k} + append per chunk against a memory of {R} rows that grows with the clip",
            "frames_per_s_look_ahead_1": ext[1],
            "look_ahead_chunks": args.look_ahead_chunks if args.look_ahead_chunks else "0 = auto (config default)",
            "frames_per_s": ext[args.look_ahead_chunks],
            "fraction_of_value": ext[args.look_ahead_chunks] / value,
        }
        if "long" in ext:
            out["extractor"]["long_clip"] = {
                "frames": args.extractor_long_frames, "frames_per_s": ext["long"], "fraction_of_value": ext["long"] / value,
                "what": "one process_video run of the same extractor on a longer clip of distinct random frames: the head "
                        "(first read + staging, ramp of small groups) and the tail (last read-back, JSON + metrics files) "
                        "are per clip, not per frame"}

    # ---- streaming leg (BASELINE configs[4]): 16 x 1080p frames per chunk, rolling 2M-row memory, one hipGraph ------
    if rank == 0 and world == 1 and not args.no_streaming:
        note("streaming leg")
        from vidmem.streaming import StreamingSession
        Ms = args.stream_rows
        ring = EmbeddingMemory(Ms, D, "f16", ring=True, device=local_rank)
        gs = torch.Generator(device=dev).manual_seed(99)
        for lo in range(0, Ms, 262_144):
            n = min(262_144, Ms - lo)
            x = torch.randn((n, D), generator=gs, device=dev, dtype=torch.float32)
            ring.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
        sess = StreamingSession(enc, ring, 16, 1080, 1920, top_k=k, warmup=2)
        chunk = torch.empty((16, 1080, 1920, 3), device=dev, dtype=torch.uint8)
        lat = []
        for i in range(60):
            chunk.random_(0, 256, generator=gs)   # a new chunk every push (repeats would plant exact ties)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sess.stream):
                e0.record(sess.stream)
                sess.push(chunk)
                e1.record(sess.stream)
            e1.synchronize()
            lat.append(e0.elapsed_time(e1))
        lat = sorted(lat[10:])
        # The same session as a FEED (BASELINE configs[4]; reference loop: src/pipeline/vlm_extractor.py:44-74, one
        # chunk in flight): a host producer thread writes a new 16 x 1080p chunk into the stager's pinned slot every
        # `period` (where a decoder would write it) while the previous chunk runs; the consumer commits it (H2D on the
        # copy stream) and replays the graph.  Latency = chunk complete in host memory -> results of its replay
        # available, host clock, H2D included; the only waits are on the replay's own event (no device-wide sync).
        # Default period = one 30 fps FRAME time per 16-frame chunk, i.e. the feed runs 16x faster than real time
        # (2,000 chunks at the real 533 ms chunk period would take 18 minutes); --stream-period-ms 533.3 is real time.
        note(f"streaming feed: {args.stream_replays} chunks, one every {args.stream_period_ms:.1f} ms")
        import queue
        import threading
        n_feed, period = args.stream_replays, args.stream_period_ms * 1e-3
        pool_n = 8          # x 256 per-push constants = 2,048 distinct chunks before one repeats (a repeat plants exact ties)
        pool = np.random.default_rng(5).integers(0, 256, size=(pool_n, 16, 1080, 1920, 3), dtype=np.uint8)
        stager = sess.stager
        ready: "queue.Queue" = queue.Queue()
        # FrameStager hands out slot (commits so far) % depth: the producer may run ONE chunk ahead of the commits (it
        # writes slot s+1 while the H2D of slot s and its replay run); next_slot itself waits for the slot's last copy
        slot_free = threading.Semaphore(1)

        def producer():
            try:
                t_next = time.perf_counter()
                for i in range(n_feed):
                    slot_free.acquire()
                    view = stager.next_slot()
                    # a new, distinct chunk per push without a 100 MB RNG call: pool chunk + a per-push constant (mod 256)
                    np.add(pool[i % pool_n], np.uint8(((i // pool_n) * 37 + 1) % 256), out=view)
                    t_next += period
                    delay = t_next - time.perf_counter()
                    if delay > 0:
                        time.sleep(delay)
                    ready.put(time.perf_counter())
            finally:
                ready.put(None)      # also on an exception: the consumer must never wait for a dead producer

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        lat_f, parts = [], []
        # The consumer loop of a latency-bound feed: two event pairs created ONCE and reused (a chunk is synchronised
        # before the next is taken), and Python's cyclic garbage collector off for the duration - round 4's breakdown
        # traced the feed's rare 10-70 ms outliers to the host's submit step (an event created per chunk + a
        # collection pause), never to the device (2.50 ms per replay, every chunk)
        import gc
        ev_ring = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
        gc.collect()
        gc.disable()
        while True:
            t_arr = ready.get(timeout=120.0)
            if t_arr is None:
                break
            t_pop = time.perf_counter()              # the consumer thread is awake and holds the chunk
            ticket = stager.commit(16)
            slot_free.release()
            ev_a, ev = ev_ring[len(lat_f) & 1]
            ev_a.record(sess.stream)
            sess.push_staged(ticket)
            ev.record(sess.stream)
            t_sub = time.perf_counter()              # H2D and replay are queued
            ev.synchronize()
            t_done = time.perf_counter()
            lat_f.append((t_done - t_arr) * 1e3)
            # where a chunk's latency went: waking the consumer, queueing the work, waiting for the device - and, on
            # the device's own clock, the replay stream from "previous work done" to "results ready" (H2D wait included)
            parts.append(((t_pop - t_arr) * 1e3, (t_sub - t_pop) * 1e3, (t_done - t_sub) * 1e3, ev_a.elapsed_time(ev)))
        th.join()
        gc.enable()
        feed_redone = sess.uncertified_last_push
        lf = sorted(lat_f[20:])
        order = sorted(range(20, len(lat_f)), key=lambda i: lat_f[i])
        names = ("consumer_wakeup_ms", "host_submit_ms", "host_wait_for_device_ms", "device_stream_ms")

        def part_row(i):
            return dict({"latency_ms": round(lat_f[i], 3), "chunk": i}, **{n: round(v, 3) for n, v in zip(names, parts[i])})

        def pct(v, p):
            return v[min(len(v) - 1, int(len(v) * p))]
        out["streaming"] = {
            "workload": f"chunk of 16 x 1080p uint8 frames -> preprocess + ViT-B/16 fp16 + top-{k} over a rolling "
                        f"{Ms}-row x {D} ring + append, one hipGraph replay per chunk",
            "p50_ms": lat[len(lat) // 2], "p99_ms": lat[min(len(lat) - 1, int(len(lat) * 0.99))], "max_ms": lat[-1],
            "device_resident_replays": len(lat),
            "feed": {
                "what": f"{len(lf)} chunks (after 20 warm-up) from a host producer thread, one every "
                        f"{args.stream_period_ms:.2f} ms, pinned slot -> H2D -> replay; latency = chunk ready on the "
                        f"host -> results ready, host clock",
                "period_ms": args.stream_period_ms, "chunks": len(lf),
                "p50_ms": pct(lf, 0.50), "p99_ms": pct(lf, 0.99), "p99_9_ms": pct(lf, 0.999), "max_ms": lf[-1],
                "deadline_ms": 33.0, "missed_deadlines": sum(1 for v in lf if v > 33.0),
                "uncertified_queries_redone_last_push": feed_redone,
                "breakdown": {
                    "what": "host clock: chunk ready -> consumer thread awake (queue + GIL) -> H2D and replay queued -> "
                            "replay's event done; device_stream_ms = the same replay between two events on its stream",
                    "median_chunk": part_row(order[len(order) // 2]),
                    "slowest_chunks": [part_row(i) for i in order[-3:][::-1]],
                },
            },
            "budget_ms": 33.0,
        }
        del pool
        del sess, chunk
        ring.close()

    # ---- BASELINE configs[2] leg: CLIP-ViT-L/14-336 bf16 encoder + top-20 over a 1M x 1024 bf16 index --------------
    if rank == 0 and world == 1 and not args.no_c3:
        note("c3 leg (CLIP-L/14-336 bf16)")
        del enc
        torch.cuda.empty_cache()
        spec3 = specs.CLIP_L14_336
        enc3 = FrameEncoder(spec3, syn.encoder_weights(spec3, seed=42), dtype="bf16", device=local_rank)
        g3 = torch.Generator(device=dev).manual_seed(4321)
        mb3 = enc3.micro_batch(args.c3_frames)                                # 224 frames per encoder pass
        F3 = args.c3_frames // mb3 * mb3                                      # >= 2048 frames per timing
        fr3 = torch.randint(0, 256, (F3, 336, 336, 3), generator=g3, device=dev, dtype=torch.uint8)
        M3, D3, k3 = 1_000_000, 1024, 20
        mem3 = EmbeddingMemory(M3, D3, "bf16", device=local_rank)
        for lo in range(0, M3, 250_000):
            x = torch.randn((250_000, D3), generator=g3, device=dev, dtype=torch.float32)
            mem3.append((x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16))
        e3 = enc3.embed_frames(fr3[:2 * mb3])
        mem3.topk(e3[:16], k3)
        torch.cuda.synchronize()
        # the timed pass: the product's default schedule (two streams: the 10 passes alternate), no per-kernel events
        t0 = time.perf_counter()
        e3 = enc3.embed_frames(fr3)
        torch.cuda.synchronize()
        dt_enc = time.perf_counter() - t0
        # the dominant GEMM instantiation's launch durations: the same frames again with its HIP events on (one stream
        # while timing is on, as in the main leg)
        ctx.profile_enable(F3 // mb3 * (7 * spec3["layers"] + 8) + 64)
        ctx.profile_mask(["gemm_patch", "gemm_qkv", "gemm_resid"])
        t0 = time.perf_counter()
        e3b = enc3.embed_frames(fr3)
        torch.cuda.synchronize()
        dt_enc_one = time.perf_counter() - t0
        p3 = ctx.profile_read()
        ctx.profile_enable(0)
        ctx.profile_mask(None)
        c3_same = bool(torch.equal(e3, e3b))
        del e3b
        t0 = time.perf_counter()
        for i in range(20):
            mem3.topk(e3[16 * i:16 * i + 16], k3)
        torch.cuda.synchronize()
        dt_knn = (time.perf_counter() - t0) / 20
        ctx.profile_enable(4096)
        enc3.embed_frames(fr3[:2 * mb3])
        bd3 = ctx.profile_read()
        ctx.profile_enable(0)
        T3, H3, M3m = enc3.tokens, spec3["hidden"], spec3["mlp"]
        rows3 = mb3 * T3
        passes3 = F3 // mb3
        # projection + FC2 (and Q) on every row of all layers but the last, on the CLS rows there
        fl3 = passes3 * ((spec3["layers"] - 1) * 2.0 * rows3 * 3 * H3 * H3
                         + 2.0 * rows3 * 2 * H3 * H3          # last layer: K, V of every row (its CLS-row GEMMs: gemm_cls)
                         + (spec3["layers"] - 1) * (2.0 * rows3 * H3 * H3 + 2.0 * rows3 * H3 * M3m)
                         + 2.0 * mb3 * (T3 - 1) * enc3.patch_k * H3)
        ms3 = sum(p3[c][0] for c in ("gemm_patch", "gemm_qkv", "gemm_resid"))
        n3 = sum(p3[c][1] for c in ("gemm_patch", "gemm_qkv", "gemm_resid"))
        ach3 = fl3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else 0.0
        # (a launch's traffic depends on the frames per encoder pass, not on how many passes a timing holds)
        c3_traffic, c3_src = pmc_traffic("void (anonymous namespace)::gemm256p_kernel<1, 0, 0>", f"mb{mb3}", "c3")
        out["c3"] = {
            "workload": f"BASELINE configs[2]: CLIP-ViT-L/14-336 bf16, {F3} of configs[2]'s 32,768 frames per timing "
                        f"({passes3} encoder passes of {mb3}; a RATE, not the whole job); top-{k3} of 16 queries over "
                        f"{M3} x {D3} bf16",
            "frames_per_s": F3 / dt_enc, "encoder_tflops": F3 / dt_enc * specs.flops_per_frame(spec3, executed=True) / 1e12,
            "encoder_schedule": "two streams (VM_SCHED_AUTO); one_stream_frames_per_s = the event-timed repeat the roofline "
                                "was taken from",
            "one_stream_frames_per_s": F3 / dt_enc_one, "embeddings_bit_identical": c3_same,
            "knn_queries_per_s": 16 / dt_knn, "knn_scan_GBps": M3 * D3 * 2 / dt_knn / 1e9,
            "uncertified_queries_redone": mem3.uncertified_count,
            "roofline": {"bound": "mfma", "kernel": "gemm256p_kernel<bf16, STORE16> (patch, QKV, proj, FC2)",
                         "achieved": ach3, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach3 / MFMA_PEAK_TFLOPS, "traffic": c3_traffic, "traffic_source": c3_src,
                         "avg_launch_ms": ms3 / max(n3, 1), "launches": n3, "flops_per_launch": fl3 / max(n3, 1)},
            "kernel_time_ms_per_2_passes": {c: round(v[0], 3) for c, v in bd3.items() if v[1]},
        }
        mem3.close()
        del enc3, fr3

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline")
        out["cpu_baseline"] = cpu_baseline(spec, weights, mem_rows[:R].cpu().numpy(), k)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

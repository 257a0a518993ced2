"""FrameEmbeddingExtractor: drop-in for the reference's VLMExtractor on the frame-embedding path.

Mirrors ``VLMExtractor`` (src/pipeline/vlm_extractor.py):
  * ``FrameEmbeddingExtractor(config)`` - constructible from the pipeline config ALONE, like ``VLMExtractor(config)``
    (:19-23; caller src/cli/main.py:28).  It reads ``config.video.chunk_size_seconds`` / ``frames_per_chunk`` (:38,:101)
    and the two new sections ``config.encoder`` / ``config.memory`` (config.py: which encoder stands where the remote
    VLM was, how big the HBM-resident memory is); a reference ``PipelineConfig`` without those sections gets their
    defaults.  ``FrameEmbeddingExtractor(config, encoder, memory)`` still accepts ready-made objects (shared memory,
    tests).
  * ``async process_video(video_path, output_path) -> str`` runs the same serial chunk loop (:44-74) with the same
    integers: chunk_size_frames = int(chunk_size_seconds * fps) (:38), total_chunks = max(1, int(total/chunk)) (:39),
    frame picks np.linspace(start, end-1, n, dtype=int) (:107), label "MM:SS-MM:SS" (:57-59), and writes the same JSON
    (:77-88) - ``content`` carries a short placeholder and the new fields ``embedding_rows`` / ``similar`` carry what
    the hot path produced for the chunk - and the same timing file ``metrics/vlm_<run_id>.json`` (:73,:91; metrics.py).
  * where the reference POSTs JPEGs to a remote VLM (:121-185), this encodes the frames on the GPU, retrieves the top-k
    most similar stored frames and appends the new embeddings to the memory.

ONE deliberate deviation, decided here and nowhere else: the reference shrinks frames taller than 720 rows with
``cv2.resize`` before JPEG-encoding them (:114-116) - a payload optimisation for the HTTP call.  This path has no
payload: the preprocess kernel resamples the ORIGINAL frame to the encoder's input size in one bilinear pass
(csrc/context.hip), which is both cheaper and closer to the source pixels than resize -> resize.  For sources of more
than 720 rows the encoder input therefore differs from what a 720-row intermediate would give; the reference defines
no encoder input at all, so nothing is pinned either way
(tests/test_frames_oracle.py::test_tall_frames_are_not_downscaled_on_the_host).

Frame sources: an OpenCV-readable video when ``cv2`` is importable (it is NOT installed in the build image), or an
``.npy`` / ``.npz`` stack of uint8 BGR frames ``[N,H,W,3]`` (``.npz``: arrays ``frames`` and optional ``fps``).
"""
from __future__ import annotations

import json
import time
import uuid
from typing import Any, List, Optional, Tuple

import numpy as np
import torch

from . import config as cfgmod
from .encoder import FrameEncoder
from .memory import EmbeddingMemory
from .metrics import MetricsTracker, get_logger

logger = get_logger("vidmem.extractor")


class _ArraySource:
    def __init__(self, frames: np.ndarray, fps: float):
        self.frames, self.fps, self.total = frames, float(fps), int(frames.shape[0])
        self.frame_hw = (int(frames.shape[1]), int(frames.shape[2]))

    def read(self, idx: int) -> Optional[np.ndarray]:
        return self.frames[idx] if 0 <= idx < self.total else None

    def read_many(self, picks: List[int]) -> np.ndarray:
        """The frames at ``picks`` (all readable) as ONE [n,H,W,3] array: a view of the clip when the picks are
        consecutive (every frame of consecutive chunks: the staging copy is then one slab), a gathered copy otherwise."""
        if picks and picks[-1] - picks[0] == len(picks) - 1 and all(b - a == 1 for a, b in zip(picks, picks[1:])):
            return self.frames[picks[0]:picks[-1] + 1]
        return self.frames[np.asarray(picks, dtype=np.int64)]

    def release(self):
        pass


class _Cv2Source:
    def __init__(self, path: str):
        import cv2  # noqa: F401  (absent in the build image; present on a deployment box)
        self.cv2 = cv2
        self.cap = cv2.VideoCapture(path)
        self.fps = self.cap.get(cv2.CAP_PROP_FPS)
        self.total = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        h, w = self.cap.get(cv2.CAP_PROP_FRAME_HEIGHT), self.cap.get(cv2.CAP_PROP_FRAME_WIDTH)
        self.frame_hw = (int(h), int(w)) if h and w else None

    def read(self, idx: int) -> Optional[np.ndarray]:
        self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, idx)
        ret, frame = self.cap.read()
        return frame if ret else None

    def release(self):
        self.cap.release()


def open_source(video_path: str, default_fps: float = 30.0):
    if video_path.endswith(".npy"):
        return _ArraySource(np.load(video_path, mmap_mode="r"), default_fps)
    if video_path.endswith(".npz"):
        z = np.load(video_path)
        return _ArraySource(z["frames"], float(z["fps"]) if "fps" in z else default_fps)
    return _Cv2Source(video_path)


def chunk_plan(fps: float, total_frames: int, chunk_size_seconds: float, frames_per_chunk: int):
    """The reference's chunking integers (src/pipeline/vlm_extractor.py:38-46, :101-107, :57-59)."""
    chunk_size_frames = int(chunk_size_seconds * fps)
    total_chunks = max(1, int(total_frames / chunk_size_frames))
    plan = []
    for chunk_idx in range(total_chunks):
        start_frame = chunk_idx * chunk_size_frames
        end_frame = min((chunk_idx + 1) * chunk_size_frames, total_frames)
        n = min(frames_per_chunk, end_frame - start_frame)
        if n <= 0:
            continue
        idx = np.linspace(start_frame, end_frame - 1, n, dtype=int)
        start_time, end_time = start_frame / fps, end_frame / fps
        time_str = (f"{int(start_time // 60):02d}:{int(start_time % 60):02d}-"
                    f"{int(end_time // 60):02d}:{int(end_time % 60):02d}")
        plan.append((chunk_idx, start_frame, end_frame, [int(i) for i in idx], time_str))
    return total_chunks, plan


def build_encoder(enc_cfg) -> FrameEncoder:
    """``config.encoder`` -> FrameEncoder.  No checkpoint can be fetched offline and the reference names no vision
    model (config/base_config.yaml:9 says only ``qwen-vlm``): ``weights`` is an .npz of named fp32 arrays
    (synthetic.encoder_weight_shapes), otherwise seeded synthetic weights of the architecture."""
    from . import specs, synthetic
    spec = specs.SPECS[enc_cfg.arch]
    if enc_cfg.weights:
        z = np.load(enc_cfg.weights, allow_pickle=False)
        weights = {k: z[k] for k in z.files}
        missing = set(synthetic.encoder_weight_shapes(spec)) - set(weights)
        if missing:
            raise ValueError(f"{enc_cfg.weights}: missing weight arrays {sorted(missing)[:4]}...")
    else:
        weights = synthetic.encoder_weights(spec, seed=int(enc_cfg.seed))
    return FrameEncoder(spec, weights, dtype=enc_cfg.dtype, device=int(enc_cfg.device))


def build_memory(mem_cfg, encoder: FrameEncoder) -> EmbeddingMemory:
    import os
    dtype = mem_cfg.dtype or encoder.dtype_name
    if mem_cfg.snapshot and os.path.exists(mem_cfg.snapshot):
        return EmbeddingMemory.restore(mem_cfg.snapshot, capacity=int(mem_cfg.capacity), ring=bool(mem_cfg.ring),
                                       device=encoder.device.index or 0)
    return EmbeddingMemory(int(mem_cfg.capacity), encoder.out_dim, dtype, ring=bool(mem_cfg.ring),
                           device=encoder.device.index or 0)


class FrameEmbeddingExtractor:
    def __init__(self, config: Any, encoder: Optional[FrameEncoder] = None, memory: Optional[EmbeddingMemory] = None,
                 top_k: Optional[int] = None, stager_factory=None):
        self.config = config
        enc_cfg = cfgmod.section(config, "encoder", cfgmod.ENCODER_DEFAULTS)
        mem_cfg = cfgmod.section(config, "memory", cfgmod.MEMORY_DEFAULTS)
        self.encoder = encoder if encoder is not None else build_encoder(enc_cfg)
        self.memory = memory if memory is not None else build_memory(mem_cfg, self.encoder)
        self.top_k = int(enc_cfg.top_k if top_k is None else top_k)
        self.metrics = MetricsTracker()              # src/pipeline/vlm_extractor.py:21
        self.metrics_dir = "metrics"                 # :91 writes metrics/vlm_<run_id>.json relative to the cwd
        self.last_metrics_path: Optional[str] = None
        self.timings: List[Tuple[str, float]] = []
        self._stager = None
        self._stager_factory = stager_factory  # (frames, H, W, device) -> ingest.FrameStager-like; tests inject a host one

    def _read_chunk(self, src, indices) -> List[np.ndarray]:
        """The reference's per-chunk frame pick (:107-111): unreadable frames are dropped.  No 720-row downscale
        (module docstring: the one deliberate deviation)."""
        return [f for f in (src.read(i) for i in indices) if f is not None]

    def _stage(self, frames: List[np.ndarray], group_chunks: int = 1):
        """Start the H2D copy of a group of chunks (ingest.FrameStager: pinned slot + copy stream); the stager is sized
        on the first group and rebuilt if the frame size changes.  Returns (stager, ticket): a ticket is redeemed at the
        stager that issued it, which the consumer therefore gets with it - ``self._stager`` may already be a newer one
        (this runs on the staging thread, ahead of the consumer)."""
        h, w = frames[0].shape[:2]
        fpc = int(cfgmod.section(self.config, "video", cfgmod.VIDEO_DEFAULTS).frames_per_chunk)
        cap = max(len(frames), fpc * max(1, group_chunks))
        if self._stager is None or self._stager.shape[1:3] != (h, w) or self._stager.shape[0] < len(frames):
            factory = self._stager_factory
            if factory is None:
                from .ingest import FrameStager as factory
            old = self._stager
            self._stager = factory(cap, h, w, self.encoder.device.index or 0)
            if old is not None and hasattr(old, "close_pool"):
                old.close_pool()   # its copy threads; its slots live as long as a ticket in flight still refers to it
        return self._stager, self._stager.stage(frames)

    def close(self) -> None:
        """Release the staging slots (pinned host + device) and their copy threads."""
        if self._stager is not None and hasattr(self._stager, "close"):
            self._stager.close()
        self._stager = None

    def _auto_look_ahead(self, src, frames_per_chunk: int, slot_bytes: int = 320 << 20) -> int:
        """``look_ahead_chunks: 0`` (the default): as many chunks per group as fill the encoder's passes
        (FrameEncoder.micro_batch: 883 frames of ViT-B/16, 224 of CLIP-L/14-336) - TWO passes when the encoder runs its
        two-stream schedule (two passes of one call side by side, include/vidmem.h VM_ENC_OPT_SCHEDULE), one otherwise -
        as long as one staging slot of the group's source frames stays under ``slot_bytes`` of pinned host memory (two
        such slots exist, and two on the device: 1080p frames are 6.2 MB each).  A source that does not say its frame
        size, or an encoder stand-in without ``micro_batch``, gets 1."""
        hw = getattr(src, "frame_hw", None)
        mb = getattr(self.encoder, "micro_batch", None)
        if hw is None or mb is None or frames_per_chunk <= 0:
            return 1
        passes = 2 if getattr(self.encoder, "schedule", "one_stream") != "one_stream" else 1
        by_pass = passes * int(mb(10 ** 6)) // frames_per_chunk
        by_bytes = slot_bytes // max(1, frames_per_chunk * int(hw[0]) * int(hw[1]) * 3)
        return max(1, min(by_pass, by_bytes))

    def _group_search(self, emb_all: torch.Tensor, counts: List[int]):
        """Neighbours of every frame of a look-ahead group in ONE pass, equal to the chunk-by-chunk loop of
        src/pipeline/vlm_extractor.py:44-74 (search chunk i against everything before it, then append it):
          (1) top-k of all the group's frames over the memory as it stands before the group (one launch);
          (2) the group's own earlier chunks: all-pairs exact reference cosines (``vm_cosine_exact``: the same fp64
              arithmetic on the same 16-bit values the memory's re-scoring uses), masked to "chunks before mine", k best by
              (score descending, row ascending) - a stable sort over append order, row id = the id the row is about to get;
          (3) ``vm_topk_merge`` of the two lists (the kernel that merges per-shard lists: same order relation).
        Only called when no stored row can be overwritten while the group is appended (see the caller).
        The (score descending, row ascending) rule lives in the library alone: ``vm_topk_select`` ranks every frame
        against the columns before its own chunk, ``vm_topk_merge`` folds the two lists."""
        from .memory import topk_merge, topk_select
        F, k = emb_all.shape[0], self.top_k
        dev = emb_all.device
        base = len(self.memory)                                   # id of the group's first row
        if self.memory.searchable:
            s_mem, r_mem = self.memory.topk(emb_all, k)
        else:
            s_mem = torch.zeros((F, k), dtype=torch.float64, device=dev)
            r_mem = torch.full((F, k), -1, dtype=torch.int64, device=dev)
        if all(c == counts[0] for c in counts):
            # equal chunks (every group but a clip's ragged last one): the first row of each frame's chunk is computed
            # on the device.  A host list would have to be COPIED in stream order, i.e. the host would sit behind the
            # encoder launches it has just queued, and the GPU would idle afterwards while the host catches up
            starts = torch.arange(F, device=dev) // counts[0] * counts[0]
        else:
            first, starts_h = 0, []
            for c in counts:
                starts_h += [first] * c
                first += c
            starts = torch.tensor(starts_h, dtype=torch.int64, device=dev)
        g = self.memory.cosine_exact(emb_all, emb_all)            # [F, F] fp64, the reference's arithmetic
        s_grp, r_grp = topk_select(self.memory.ctx, g, k, col_limit=starts, row_base=base)
        return topk_merge(self.memory.ctx, torch.stack([s_mem, s_grp]), torch.stack([r_mem, r_grp]))

    async def process_video(self, video_path: str, output_path: str) -> str:
        run_id = str(uuid.uuid4())
        logger.info(f"Starting frame-embedding extraction with run ID: {run_id}")
        video_cfg = cfgmod.section(self.config, "video", cfgmod.VIDEO_DEFAULTS)
        src = open_source(video_path)
        try:
            fps, total_frames = src.fps, src.total
            total_chunks, plan = chunk_plan(fps, total_frames, video_cfg.chunk_size_seconds,
                                            video_cfg.frames_per_chunk)
            results = []
            result_lines: List[str] = []      # results[i] as JSON text (finish)
            # Look-ahead groups (config.encoder.look_ahead_chunks, default 0 = auto, 1 = the reference's one chunk at a time,
            # :44-74): the frames of N consecutive chunks go through ONE encoder call - the encoder fills the chip only
            # from a few hundred frames up - and then every chunk of the group, in chunk order, gets its own top-k
            # against the memory as it stands (chunks < i, the group's earlier chunks included) followed by its own
            # append: the same neighbours and the same rows as N = 1, bit for bit (an embedding does not depend on the
            # batch it was computed in: tests/test_encoder_gpu.py::test_bench_size_batches...).  Results are read
            # back once per group, AFTER the next group's launches are queued, so the GPU never waits for the host.
            L = int(cfgmod.section(self.config, "encoder", cfgmod.ENCODER_DEFAULTS).look_ahead_chunks)
            if L <= 0:
                L = self._auto_look_ahead(src, int(video_cfg.frames_per_chunk))
            # the first groups ramp up (1/8, 1/4, 1/2 of L): the GPU starts on a small group while the host is still
            # reading the first full one (grouping never changes a result, only when work is launched)
            groups, i, size = [], 0, max(1, L // 8)
            while i < len(plan):
                groups.append(plan[i:i + size])
                i += size
                size = min(L, size * 2)
            # ... and the last group is a small one (a quarter of L): a group's results are read back and serialised while
            # the NEXT group runs, so only the last group's share of that work is exposed at the end of a clip
            tail = max(1, L // 4)
            if len(groups) >= 2 and len(groups[-1]) > tail + max(1, L // 8):
                last = groups.pop()
                groups += [last[:-tail], last[-tail:]]
            dev = self.encoder.device

            def read_and_stage(gi):
                """Host side of group gi: frame reads + pinned staging + start of the H2D copy."""
                if gi >= len(groups):
                    return None
                counts, frames = [], []
                read_many = getattr(src, "read_many", None)
                if read_many is not None:     # array sources: the group's picks in one gather (no per-frame Python)
                    picks: List[int] = []
                    for (_ci, _s, _e, indices, _t) in groups[gi]:
                        ok = [i for i in indices if 0 <= i < src.total]      # an unreadable frame is dropped (:111)
                        counts.append(len(ok))
                        picks += ok
                    return counts, (self._stage(read_many(picks), L) if picks else (None, None))
                for (_ci, _s, _e, indices, _t) in groups[gi]:
                    fr = self._read_chunk(src, indices)
                    counts.append(len(fr))
                    frames += fr
                return counts, (self._stage(frames, L) if frames else (None, None))

            def finish(pend):
                """Read one group's neighbours back (its event has fired long before the host gets here).
                ``processing_time`` (:62-71 times ONE chunk around its own work): the group's own DEVICE time - from
                the first kernel of its encoder call to its last append, a pair of events on the launch stream, so
                neither the host work queued for the next group nor the frame reads on the staging thread are in it -
                divided by the chunks of the group; ``group_time`` / ``group_chunks`` carry the undivided figure.
                With one chunk per group (the config default of the reference's loop) it is that chunk's device time.
                A host stand-in for the encoder (CPU tests) has no events: wall time since the group was taken up."""
                chunks, host_s, host_r, ev0, ev, t_start = pend
                if ev is not None:
                    ev.synchronize()
                    group_time = ev0.elapsed_time(ev) * 1e-3
                else:
                    group_time = time.perf_counter() - t_start
                live = [c for c in chunks if c["nframes"]]
                off = 0
                # row -> id as EmbeddingMemory.id_of does it, without 10 method calls per frame (the last group's share
                # of this loop is the tail of every clip: nothing is left to overlap it with)
                ids_tab, tab_base = getattr(self.memory, "ids", None), getattr(self.memory, "table_base", 0)
                tab_len = len(ids_tab) if ids_tab is not None else 0
                id_of = self.memory.id_of       # a memory stand-in without the table (tests): its own lookup
                all_s = host_s.tolist() if host_s is not None else []
                all_r = host_r.tolist() if host_r is not None else []
                for c in live:
                    similar = []
                    if c["searched"]:
                        n = c["nframes"]
                        for s_row, r_row in zip(all_s[off:off + n], all_r[off:off + n]):
                            if ids_tab is None:
                                similar.append([(id_of(r), float(s)) for s, r in zip(s_row, r_row) if r >= 0])
                            else:
                                similar.append([(ids_tab[r - tab_base] if 0 <= r - tab_base < tab_len else None, s)
                                                for s, r in zip(s_row, r_row) if r >= 0])
                        off += n
                    chunk_time = group_time / len(live)
                    results.append({
                        "time": c["time"],
                        "content": f"[{c['nframes']} frame embeddings]",
                        "chunk_idx": c["chunk_idx"],
                        "processing_time": chunk_time,
                        "group_time": group_time,
                        "group_chunks": len(live),
                        "embedding_rows": list(range(c["first"], c["first"] + c["nframes"])),
                        "similar": similar,
                    })
                    # serialised here, while the GPU works on the next group, not in one piece after the last one
                    result_lines.append(json.dumps(results[-1], default=str))
                    self.timings.append((f"chunk_{c['chunk_idx']}", chunk_time))
                    self.metrics.record_timing(f"chunk_{c['chunk_idx']}", "vlm_inference", chunk_time)   # key of :73

            # Host side on its own thread: frame reads, the copy into the pinned slot and the start of the H2D copy run
            # AHEAD of the launches (numpy copies release the GIL), bounded by the stager's device slots: group n may be
            # staged once the launches that read group n - depth have been issued (``done`` records the event the copy
            # stream waits for).  The reference reads the frames of a chunk inline (:98-119); so does look_ahead 1 on a
            # source that must stay on one thread - the staging thread is the only one that touches ``src``.
            import queue
            import threading
            depth = 2
            credits = threading.Semaphore(depth)
            staged_q: "queue.Queue" = queue.Queue()
            stop = threading.Event()

            def stage_all():
                try:
                    for gi in range(len(groups)):
                        credits.acquire()
                        if stop.is_set():      # the consumer gave up (an exception in the main loop): touch nothing more
                            return
                        staged_q.put(read_and_stage(gi))
                except BaseException as exc:   # surfaces in the main loop
                    staged_q.put(exc)

            stager_thread = threading.Thread(target=stage_all, name="vidmem-stage", daemon=True)
            stager_thread.start()
            try:
                pending = None
                for gi, grp in enumerate(groups):
                    t_start = time.perf_counter()
                    staged = staged_q.get()
                    if isinstance(staged, BaseException):
                        raise staged
                    counts, (stager, ticket) = staged
                    emb_all = ev0 = None
                    if ticket is not None:
                        frames_dev = stager.get(ticket)
                        if getattr(frames_dev, "is_cuda", False):
                            ev0 = torch.cuda.Event(enable_timing=True)
                            ev0.record(torch.cuda.current_stream(dev))
                        emb_all = self.encoder.embed_frames(frames_dev)   # asynchronous launches
                        stager.done(ticket)
                    credits.release()
                    chunks, dev_s, dev_r, off = [], [], [], 0
                    total = sum(counts)
                    # One search for the whole group when that is provably the same thing: every chunk of the group still
                    # sees exactly "the memory before the group + the group's earlier chunks", i.e. no row of the memory is
                    # overwritten while the group is appended (not a ring, or a ring that does not wrap inside the group).
                    batched = (len(grp) > 1 and emb_all is not None and self.top_k > 0 and emb_all.is_cuda and
                               (not self.memory.ring or len(self.memory) + total <= self.memory.capacity))
                    if batched:
                        grp_s, grp_r = self._group_search(emb_all, counts)
                    for (chunk_idx, start, end, indices, time_str), nframes in zip(grp, counts):
                        c = {"chunk_idx": chunk_idx, "time": time_str, "nframes": nframes, "searched": False, "first": 0}
                        chunks.append(c)
                        if not nframes:
                            continue
                        emb = emb_all[off:off + nframes]
                        if batched:
                            if self.memory.searchable + off > 0:       # what `memory.searchable` will be when chunk i's turn comes
                                dev_s.append(grp_s[off:off + nframes])
                                dev_r.append(grp_r[off:off + nframes])
                                c["searched"] = True
                        elif self.memory.searchable and self.top_k > 0:
                            scores, rows = self.memory.topk(emb, self.top_k)
                            dev_s.append(scores)
                            dev_r.append(rows)
                            c["searched"] = True
                        off += nframes
                        ids = [f"{run_id}_{chunk_idx}_{i}" for i in range(nframes)]  # pre_llm_injector.py:91 id scheme
                        created = time.strftime("%Y-%m-%dT%H:%M:%S+00:00", time.gmtime())   # Chunk.created_at of the export
                        c["ids"] = ids
                        c["meta"] = [{"time": time_str, "content": None, "batch_id": chunk_idx, "created_at": created}] * nframes
                        if not batched:
                            c["first"] = self.memory.append(emb, ids=ids, meta=c["meta"])
                    if batched:      # one append for the group: rows, ids and meta in chunk order
                        live = [c for c in chunks if c["nframes"]]
                        first = self.memory.append(emb_all, ids=[i for c in live for i in c["ids"]],
                                                   meta=[m for c in live for m in c["meta"]])
                        for c in live:
                            c["first"] = first
                            first += c["nframes"]
                    host_s = host_r = ev = None
                    if dev_s:
                        cat_s, cat_r = torch.cat(dev_s), torch.cat(dev_r)
                        host_s = torch.empty(cat_s.shape, dtype=cat_s.dtype).pin_memory()
                        host_r = torch.empty(cat_r.shape, dtype=cat_r.dtype).pin_memory()
                        host_s.copy_(cat_s, non_blocking=True)
                        host_r.copy_(cat_r, non_blocking=True)
                    if ev0 is not None:   # (a host stand-in for the encoder in the CPU tests: no event)
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(torch.cuda.current_stream(dev))
                    if pending is not None:
                        finish(pending)
                    pending = (chunks, host_s, host_r, ev0, ev, t_start)
                if pending is not None:
                    finish(pending)
            finally:
                # the staging thread is the only one that touches ``src`` and the pinned slots: it is stopped and JOINED
                # before ``src.release()`` below runs, whatever ended the loop (a full non-ring memory, an encoder or
                # top-k error, a failing consumer).  It is either inside read_and_stage - which returns - or waiting for
                # a credit, which it now gets and then sees ``stop``.
                stop.set()
                for _ in range(depth):
                    credits.release()
                stager_thread.join()
                if self._stager is not None and hasattr(self._stager, "close_pool"):
                    self._stager.close_pool()     # the slab-copy workers; the slots stay for the next clip
            output_data = {
                "metadata": {"run_id": run_id, "video_path": video_path, "total_chunks": total_chunks,
                             "config": cfgmod.config_dict(self.config)},
                "results": results,
            }
            # same document as json.dump(output_data, f, indent=2) would give a parser (:86-88); the per-chunk entries go
            # one per line through the C encoder (the indenting encoder is pure Python: 0.1 s for 4,096 frames' neighbours)
            with open(output_path, "w") as f:
                f.write('{\n  "metadata": ')
                f.write(json.dumps(output_data["metadata"], indent=2, default=str).replace("\n", "\n  "))
                f.write(',\n  "results": [')
                f.write(",".join("\n    " + line for line in result_lines))
                f.write("\n  ]\n}\n" if results else "]\n}\n")
            logger.info(f"Frame-embedding extraction completed. Output saved to: {output_path}")
            import os
            self.last_metrics_path = os.path.join(self.metrics_dir, f"vlm_{run_id}.json")
            self.metrics.save_metrics(self.last_metrics_path)
            mem_cfg = cfgmod.section(self.config, "memory", cfgmod.MEMORY_DEFAULTS)
            if mem_cfg.snapshot:
                self.memory.snapshot(mem_cfg.snapshot)
            return output_path
        finally:
            src.release()

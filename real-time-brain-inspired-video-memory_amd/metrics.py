"""Timing JSON and logger in the reference's shapes, so downstream tooling reads either producer.

``MetricsTracker`` restates src/core/metrics.py:9-66 (keys ``start_time / timings / counts / batch_metrics``, then
``end_time`` and ``summary{total_runtime, timing_averages, counts, batch_count}`` on save; timings keyed
``"<category>.<operation>"``; json.dump(indent=2, default=str)).  The extractor records ``chunk_<i>.vlm_inference`` -
the key src/pipeline/vlm_extractor.py:73 uses - so a chunk's time lands where the remote-VLM latency used to - and
saves to ``metrics/vlm_<run_id>.json`` (:91).  ``get_logger`` honours ``VIDGRAPH_LOG_LEVEL`` like
src/core/logger.py:7-51 (stdout handler + ``logs/<name>.log``; the directory can be moved with VIDGRAPH_LOG_DIR, and
an unwritable one degrades to stdout only instead of failing the import)."""
from __future__ import annotations

import json
import logging
import os
import sys
import time
from pathlib import Path
from typing import Any, Dict


class MetricsTracker:
    def __init__(self):
        self.metrics: Dict[str, Any] = {"start_time": time.time(), "timings": {}, "counts": {}, "batch_metrics": []}

    def record_timing(self, category: str, operation: str, duration: float) -> None:
        self.metrics["timings"].setdefault(f"{category}.{operation}", []).append(duration)

    def record_count(self, category: str, operation: str, count: int) -> None:
        key = f"{category}.{operation}"
        self.metrics["counts"][key] = self.metrics["counts"].get(key, 0) + count

    def add_batch_metrics(self, batch_metrics: Dict[str, Any]) -> None:
        self.metrics["batch_metrics"].append(batch_metrics)

    def get_summary(self) -> Dict[str, Any]:
        return {
            "total_runtime": time.time() - self.metrics["start_time"],
            "timing_averages": {k: sum(v) / len(v) for k, v in self.metrics["timings"].items() if v},
            "counts": self.metrics["counts"],
            "batch_count": len(self.metrics["batch_metrics"]),
        }

    def save_metrics(self, path: str) -> None:
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        self.metrics["end_time"] = time.time()
        self.metrics["summary"] = self.get_summary()
        with open(path, "w") as f:
            json.dump(self.metrics, f, indent=2, default=str)


def get_logger(name: str, level: int = logging.INFO) -> logging.Logger:
    logger = logging.getLogger(name)
    env_level = os.getenv("VIDGRAPH_LOG_LEVEL")
    if env_level:
        level = getattr(logging, env_level.upper(), level)
    logger.setLevel(level)
    if logger.handlers:
        return logger
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    console = logging.StreamHandler(sys.stdout)
    console.setLevel(level)
    console.setFormatter(fmt)
    logger.addHandler(console)
    try:
        logs_dir = Path(os.getenv("VIDGRAPH_LOG_DIR", "logs"))
        logs_dir.mkdir(exist_ok=True)
        fh = logging.FileHandler(logs_dir / f"{name}.log")
        fh.setLevel(level)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    except OSError:
        pass
    return logger

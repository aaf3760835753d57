"""Local SFT datasets for ``aria_amd.train`` -- the on-disk format and the mixing rule of the reference (aria/data.py:123-233):

    <dataset>/train.jsonl [+ test.jsonl]   one JSON object per line:
        {"messages": [{"role": "user" | "assistant", "content": [{"type": "text" | "image" | "video", "text": str | null}, ...]}, ...],
         "images": ["image_folder/0001.jpg", ...] | null,          (paths relative to the dataset directory)
         "video": {"path": ..., "num_frames": ...} | null}

``dataset_mixer: {path: frac}`` of the recipes (recipes/config_full.yaml:5-8): ``frac <= 1`` keeps the first ``int(frac * n)`` rows,
``frac > 1`` repeats the whole set ``int(frac)`` times; the training rows of all sets are concatenated and shuffled with seed 42 (the
permutation HF ``datasets.shuffle(seed=42)`` draws: ``numpy.random.default_rng(42).permutation(n)``); test rows are concatenated unshuffled.
Plain ``json`` + lists (no ``datasets`` dependency, no worker processes): rows are small dicts, images stay paths until ``collate_fn``.
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Dict, Iterator, List, Optional

COLUMNS = ("images", "messages", "video")


def _read_jsonl(path: str) -> List[dict]:
    with open(path) as f:
        return [json.loads(line) for line in f if line.strip()]


def _absolute(item: dict, root: str) -> dict:
    item = {k: item.get(k) for k in COLUMNS}
    if item["images"] and item["video"]:
        raise ValueError("Simultaneous input of images and video is not supported.")
    if item["images"] is not None:
        item["images"] = [f"{root}/{p}" for p in item["images"]]
    if item["video"] is not None:
        video = dict(item["video"])
        if video.get("num_frames") is None or video["num_frames"] <= 0:
            warnings.warn("`num_frames` is set to 8 by default because of a negative value or `None`.")
            video["num_frames"] = 8
        video["path"] = f"{root}/{video['path']}"
        item["video"] = video
    return item


def load_local_dataset(path: str) -> Dict[str, List[dict]]:
    """{"train": rows[, "test": rows]} with image / video paths made absolute (aria/data.py:123-199)."""
    if not os.path.exists(f"{path}/train.jsonl"):
        raise FileNotFoundError(f"train.jsonl not found in {path}")
    out = {"train": [_absolute(r, path) for r in _read_jsonl(f"{path}/train.jsonl")]}
    if os.path.exists(f"{path}/test.jsonl"):
        out["test"] = [_absolute(r, path) for r in _read_jsonl(f"{path}/test.jsonl")]
    return out


def mix_datasets(dataset_config: Dict[str, float], seed: int = 42) -> Dict[str, Optional[List[dict]]]:
    """aria/data.py:202-233."""
    import numpy as np

    train, test = [], []
    for path, frac in dataset_config.items():
        frac = float(frac)
        ds = load_local_dataset(path)
        rows = ds["train"]
        train.extend(rows[: int(frac * len(rows))] if frac <= 1 else rows * int(frac))
        test.extend(ds.get("test", []))
    order = np.random.default_rng(seed).permutation(len(train))
    return {"train": [train[int(i)] for i in order], "test": test or None}


def batches(rows: List[dict], batch_size: int, rank: int = 0, world: int = 1, epochs: float = 1.0, drop_last: bool = True) -> Iterator[List[dict]]:
    """Per-rank batches of row dicts: rank r takes rows r, r + world, ... of every epoch (every rank gets the same number of batches, the
    tail that does not fill one batch on every rank is dropped like the HF Trainer's ``dataloader_drop_last``)."""
    per_rank = len(rows) // world if drop_last else -(-len(rows) // world)
    n_batches = per_rank // batch_size if drop_last else -(-per_rank // batch_size)
    whole, part = int(epochs), epochs - int(epochs)
    for epoch in range(whole + (1 if part > 0 else 0)):
        limit = n_batches if epoch < whole else int(part * n_batches)
        mine = rows[rank::world][:per_rank]
        for b in range(limit):
            yield list(mine[b * batch_size:(b + 1) * batch_size])

"""aria_amd.data (aria/data.py:123-233: dataset directory format, dataset_mixer fractions, seed-42 shuffle) against the live reference
where it is importable, and a short fine-tune of aria_amd.train on such a dataset (real collate path, emulator kernels)."""
import json

import numpy as np
import pytest
import torch

from aria_amd import data as D
from oracle.ref_processing import StubTokenizer
from oracle.ref_shims import reference_available


def make_dataset(root, n, tag, with_test=False, extra=False):
    from PIL import Image

    (root / "image_folder").mkdir(parents=True)
    rng = np.random.default_rng(len(tag) + n)
    rows = []
    for i in range(n):
        images = None
        content = [{"type": "text", "text": f"{tag} question {i}?"}]
        if i % 2 == 0:
            name = f"image_folder/{i:03d}.png"
            Image.fromarray(rng.integers(0, 255, (40 + i, 64, 3), dtype=np.uint8)).save(root / name)
            images = [name]
            content.insert(0, {"type": "image", "text": None})
        rows.append({"messages": [{"role": "user", "content": content},
                                  {"role": "assistant", "content": [{"type": "text", "text": f"answer {tag} {i}"}]}],
                     "images": images, "video": None, **({"extra_column": i} if extra else {})})
    with open(root / "train.jsonl", "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows) + "\n")
    if with_test:
        with open(root / "test.jsonl", "w") as f:
            f.write(json.dumps(rows[0]) + "\n")
    return rows


def test_mixing_rule_and_layout(tmp_path):
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    make_dataset(a, 6, "a", with_test=True, extra=True), make_dataset(b, 5, "b"), make_dataset(c, 3, "c")  # extra columns are dropped
    ds = D.load_local_dataset(str(a))
    assert len(ds["train"]) == 6 and len(ds["test"]) == 1 and set(ds["train"][0]) == {"images", "messages", "video"}
    assert ds["train"][0]["images"] == [f"{a}/image_folder/000.png"] and ds["train"][1]["images"] is None
    mixed = D.mix_datasets({str(a): 1, str(b): 0.5, str(c): 2})                    # recipes/config_full.yaml:5-8 semantics
    texts = [r["messages"][0]["content"][-1]["text"] for r in mixed["train"]]
    assert len(texts) == 6 + 2 + 6 and sum(t.startswith("b ") for t in texts) == 2 and sum(t.startswith("c ") for t in texts) == 6
    assert sorted(t for t in texts if t.startswith("b ")) == ["b question 0?", "b question 1?"]     # frac <= 1: the FIRST int(frac * n) rows
    unshuffled = ([f"a question {i}?" for i in range(6)] + ["b question 0?", "b question 1?"] + [f"c question {i}?" for i in range(3)] * 2)
    assert texts == [unshuffled[i] for i in np.random.default_rng(42).permutation(14)]
    assert len(mixed["test"]) == 1
    with pytest.raises(FileNotFoundError):
        D.load_local_dataset(str(tmp_path / "nope"))
    got = list(D.batches(list(range(23)), 2, rank=1, world=2, epochs=1.5))
    assert len(got) == 5 + 2 and got[0] == [1, 3] and got[5] == [1, 3]


@pytest.mark.skipif(not reference_available(), reason="reference checkout not present")
def test_mix_datasets_equals_the_reference(tmp_path):
    from oracle.ref_processing import load_reference_processing

    ns = load_reference_processing()
    a, b = tmp_path / "a", tmp_path / "b"
    make_dataset(a, 7, "a", with_test=True), make_dataset(b, 4, "b")
    cfg = {str(a): 0.6, str(b): 3}
    want = ns.data.mix_datasets(cfg)
    got = D.mix_datasets(cfg)
    assert len(want["train"]) == len(got["train"])
    for w, g in zip(want["train"], got["train"]):
        assert w["images"] == g["images"] and w["video"] == g["video"]
        assert [(m["role"], [(c["type"], c["text"]) for c in m["content"]]) for m in w["messages"]] == \
               [(m["role"], [(c["type"], c["text"]) for c in m["content"]]) for m in g["messages"]]
    assert len(want["test"]) == len(got["test"]) == 1


def test_finetune_on_a_local_dataset(tmp_path):
    from tests.emu import emu_lib

    from aria_amd.train import main

    root = tmp_path / "ds"
    make_dataset(root, 2, "toy")                            # one row with an image (7 s of emulated ViT), one text-only
    tok = StubTokenizer()
    emu_lib.install()
    try:
        hist = main(["--tiny", "per_device_train_batch_size=1", "gradient_accumulation_steps=1", "max_seq_length=400", "max_image_size=490",
                     "num_train_epochs=1", "max_steps=0", "learning_rate=1e-2", "weight_decay=0.0", "warmup_ratio=0.0", "logging_steps=100",
                     "tiny_image_size=490", "save_final=true", f"output_dir={tmp_path / 'out'}", f"image_token_index={StubTokenizer.SPECIAL.index('<|img|>')}", f'dataset_mixer={{"{root}": 1}}'], tokenizer=tok)
    finally:
        emu_lib.uninstall()
    assert (tmp_path / "out" / "config.json").exists() and json.load(open(tmp_path / "out" / "preprocessor_config.json"))["max_image_size"] == 490
    assert len(hist) == 2 and all(np.isfinite(hist))        # 2 rows / batch 1 = 2 optimizer steps in one epoch

"""Checkpoint / wire formats, CPU only: the product converters are pure layout, so every check is bit-exact.
The HF -> gptfast conversion is pinned against tests/golden/gptfast.pt, whose converted weights the reference's own gptfast model
accepted (its logits match the reference HF model on them, oracle/make_golden.py)."""
import json
import os

import torch

from aria_amd import checkpoint as ck

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gptfast.pt")


def _hf_weights():
    g = torch.load(GOLDEN)
    cfg = g["cfg"]
    hf = {"language_model." + k: v for k, v in g["weights"].items()}
    # a state dict of the full model also carries vision / projector tensors: they must pass through untouched
    hf["vision_tower.vision_model.embeddings.patch_embedding.weight"] = torch.randn(8, 3, 2, 2)
    hf["multi_modal_projector.query"] = torch.randn(4, 8)
    hf["language_model.model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.randn(4)  # dropped by the reference (:97)
    return g, cfg, hf


def test_hf_to_gptfast_matches_the_conversion_the_reference_accepted():
    g, cfg, hf = _hf_weights()
    hd = cfg["hidden_size"] // cfg["num_attention_heads"]
    got = ck.hf_to_gptfast(hf, cfg["num_attention_heads"], hd, cfg["num_key_value_heads"])
    want = g["gptfast_weights"]
    llm = {k[4:]: v for k, v in got.items() if k.startswith("llm.")}
    assert set(llm) == set(want)
    for k in want:
        assert torch.equal(llm[k], want[k]), k
    assert torch.equal(got["multi_modal_projector.query"], hf["multi_modal_projector.query"])
    assert not any("inv_freq" in k for k in got)


def test_round_trip_is_the_identity():
    g, cfg, hf = _hf_weights()
    hd = cfg["hidden_size"] // cfg["num_attention_heads"]
    back = ck.gptfast_to_hf(ck.hf_to_gptfast(hf, cfg["num_attention_heads"], hd), cfg["num_attention_heads"], hd)
    ref = {k: v for k, v in hf.items() if "inv_freq" not in k}
    assert set(back) == set(ref)
    for k in ref:
        assert torch.equal(back[k], ref[k]), k


def test_sharded_safetensors_directory_round_trip_and_model_pth(tmp_path):
    g, cfg, hf = _hf_weights()
    hf = {k: v for k, v in hf.items() if "inv_freq" not in k}
    d = str(tmp_path / "ckpt")
    ck.save_checkpoint_dir(hf, d, max_shard_bytes=64 << 10)
    index = json.load(open(os.path.join(d, "model.safetensors.index.json")))
    assert len(set(index["weight_map"].values())) > 1  # really sharded
    assert index["metadata"]["total_size"] == sum(v.numel() * v.element_size() for v in hf.values())
    loaded = ck.load_checkpoint_dir(d)
    assert set(loaded) == set(hf) and all(torch.equal(loaded[k], hf[k]) for k in hf)
    hd = cfg["hidden_size"] // cfg["num_attention_heads"]
    pth = ck.convert_hf_checkpoint(d, cfg["num_attention_heads"], hd)
    conv = torch.load(pth)
    for k, v in g["gptfast_weights"].items():
        assert torch.equal(conv["llm." + k], v), k


def test_load_hf_into_module_checks_shapes_and_names():
    lin = torch.nn.Linear(4, 3, bias=False)
    missing, unexpected = ck.load_hf_into(lin, {"weight": torch.ones(3, 4), "rotary_emb.inv_freq": torch.ones(2)})
    assert missing == [] and unexpected == [] and float(lin.weight.sum()) == 12
    try:
        ck.load_hf_into(lin, {"weight": torch.ones(4, 3)})
    except ValueError:
        pass
    else:
        raise AssertionError("shape mismatch must raise")


def test_save_and_from_pretrained_round_trip(tmp_path):
    """config.json + sharded safetensors written by save_pretrained come back through from_pretrained bit for bit (no kernels involved)."""
    from aria_amd.modeling_aria import AriaConfig, AriaForConditionalGeneration

    cfg = AriaConfig(vision_config=dict(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=48, image_size=28),
                     text_config=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, vocab_size=96, moe_intermediate_size=16,
                                      moe_num_experts=4, moe_topk=2, max_position_embeddings=128),
                     projector_patch_to_query_dict={4: 2}, image_token_index=7)
    model = AriaForConditionalGeneration(cfg)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_((torch.randn(p.shape) * 0.1).to(p.dtype))
    d = str(tmp_path / "aria_ckpt")
    model.save_pretrained(d, max_shard_bytes=32 << 10)
    again = AriaForConditionalGeneration.from_pretrained(d)
    a, b = model.state_dict(), again.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert again.config.image_token_index == 7 and again.config.projector_patch_to_query_dict == {4: 2}
    assert again.config.text_config.moe_num_experts == 4 and again.config.vision_config.image_size == 28
    assert again.config.text_config.max_position_embeddings == 128


def test_sharded_directory_streams_shard_by_shard(tmp_path):
    """load_hf_dir_into = load_hf_into over iter_checkpoint_shards: same result as loading the whole directory, shape / strictness checks
    included (from_pretrained of a ~50 GB checkpoint never holds more than one shard on the host)."""
    import torch
    from torch import nn

    from aria_amd import checkpoint as C

    torch.manual_seed(0)
    src = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4))
    sd = {k: v.detach().clone() for k, v in src.state_dict().items()}
    C.save_checkpoint_dir(sd, str(tmp_path), max_shard_bytes=300)          # forces several shards
    shards = list(C.iter_checkpoint_shards(str(tmp_path)))
    assert len(shards) >= 3 and sorted(k for s in shards for k in s) == sorted(sd)
    dst = nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 4))
    assert C.load_hf_dir_into(dst, str(tmp_path)) == ([], [])
    assert all(torch.equal(a, b) for a, b in zip(dst.state_dict().values(), src.state_dict().values()))
    import pytest

    with pytest.raises(KeyError):
        C.load_hf_dir_into(nn.Sequential(nn.Linear(8, 16)), str(tmp_path))             # unexpected keys, strict
    with pytest.raises(ValueError):
        C.load_hf_dir_into(nn.Sequential(nn.Linear(8, 16), nn.Linear(16, 5)), str(tmp_path))   # shape mismatch

"""SURVEY row a3, the integer part: position ids of the ViT for EVERY valid patch grid.  The reference inherits the computation from
transformers' Idefics2VisionEmbeddings (fp32 fractional coordinates bucketized against arange(1/n, 1, 1/n), modeling_idefics2.py:130-173);
here (1) the oracle is pinned against that very module -- run with zero patch weights and a position table that stores its own index --
and (2) the device kernel (through the emulator) against the oracle, exhaustively at the 490-px grid (35 x 35: all 1225 valid h x w
rectangles) and on a lattice of the 980-px grid (70 x 70).  (The full 70 x 70 sweep, 4900 grids, was run once offline: no mismatch.)"""
import pytest
import torch

from oracle import aria_oracle as O


def _masks(n_side, step=1):
    pairs = [(h, w) for h in range(1, n_side + 1, step) for w in range(1, n_side + 1, step)]
    pm = torch.zeros(len(pairs), n_side, n_side, dtype=torch.bool)
    for j, (h, w) in enumerate(pairs):
        pm[j, :h, :w] = True
    return pm


@pytest.mark.parametrize("n_side,step", [(35, 1), (70, 5)])
def test_oracle_equals_transformers_embeddings(n_side, step):
    from transformers.models.idefics2.configuration_idefics2 import Idefics2VisionConfig
    from transformers.models.idefics2.modeling_idefics2 import Idefics2VisionEmbeddings

    cfg = Idefics2VisionConfig(hidden_size=1, image_size=n_side * 14, patch_size=14, num_channels=3, num_attention_heads=1, intermediate_size=1,
                               num_hidden_layers=1)
    emb = Idefics2VisionEmbeddings(cfg)
    with torch.no_grad():
        emb.patch_embedding.weight.zero_()
        emb.patch_embedding.bias.zero_()
        emb.position_embedding.weight.copy_(torch.arange(n_side * n_side, dtype=torch.float32)[:, None])   # the table returns its index
    pm = _masks(n_side, step)
    for i in range(0, pm.shape[0], 128):
        chunk = pm[i:i + 128]
        with torch.no_grad():
            ids = emb(torch.zeros(chunk.shape[0], 3, n_side * 14, n_side * 14), chunk)[..., 0].round().long()
        want = O.vit_position_ids(chunk, n_side)
        valid = chunk.view(chunk.shape[0], -1)
        assert torch.equal(ids[valid], want[valid])


@pytest.mark.parametrize("n_side,step", [(35, 1), (70, 5)])
def test_kernel_equals_oracle(n_side, step):
    from tests.emu import emu_lib

    from aria_amd import ops
    from aria_amd.vision import AriaVisionConfig, AriaVisionModel

    emu_lib.install()
    try:
        vit = AriaVisionModel(AriaVisionConfig(hidden_size=16, num_hidden_layers=1, num_attention_heads=1, intermediate_size=16,
                                               image_size=n_side * 14, patch_size=14))
        boundaries = vit.vision_model.embeddings.boundaries("cpu")
        pm = _masks(n_side, step)
        for i in range(0, pm.shape[0], 256):
            chunk = pm[i:i + 256]
            pixels = chunk.repeat_interleave(14, 1).repeat_interleave(14, 2)
            got = ops.vit_pos_ids(ops.vit_patch_mask(pixels, 14), boundaries, n_side)
            assert torch.equal(got.long(), O.vit_position_ids(chunk, n_side))
    finally:
        emu_lib.uninstall()

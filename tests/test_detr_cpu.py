"""scene_graph_commonsense_amd/detr.py (the DETR-101 feature extractor the reference downloads through torch.hub).
Pinned against the reference: the state-dict contract (names and count of its own key list).  The rest checks the restated
architecture against torch.nn building blocks and literal formulas - the hub code itself is not available offline, so numerical
parity with it is unpinned (said so in the module header and DESIGN.md)."""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))


def _small(**kw):
    from scene_graph_commonsense_amd.detr import DETR
    torch.manual_seed(0)
    return DETR(blocks=(1, 1, 1, 1), **kw).eval()


def test_state_dict_names_are_the_reference_checkpoint_names():
    from scene_graph_commonsense_amd.detr import DETR
    want = [l.strip() for l in open(os.path.join(HERE, "golden", "detr101_keys.txt")) if l.strip()]
    with torch.device("meta"):
        model = DETR(num_outputs=151)
    have = list(model.state_dict().keys())
    assert len(want) == 713 and sorted(have) == sorted(want)
    sd = model.state_dict()
    assert tuple(sd["class_embed.weight"].shape) == (151, 256) and tuple(sd["query_embed.weight"].shape) == (100, 256)
    assert tuple(sd["input_proj.weight"].shape) == (256, 2048, 1, 1)
    assert tuple(sd["transformer.encoder.layers.5.self_attn.in_proj_weight"].shape) == (768, 256)
    assert tuple(sd["backbone.0.body.layer3.22.conv2.weight"].shape) == (256, 256, 3, 3)
    assert tuple(sd["bbox_embed.layers.2.weight"].shape) == (4, 256)


def test_frozen_batchnorm_and_bottleneck_stride():
    from scene_graph_commonsense_amd.detr import Bottleneck, FrozenBatchNorm2d
    torch.manual_seed(1)
    fb, bn = FrozenBatchNorm2d(6), nn.BatchNorm2d(6).eval()
    for name in ("weight", "bias", "running_mean"):
        v = torch.randn(6)
        getattr(fb, name).copy_(v)
        getattr(bn, name).data.copy_(v)
    v = torch.rand(6) + 0.5
    fb.running_var.copy_(v)
    bn.running_var.copy_(v)
    x = torch.randn(2, 6, 5, 7)
    assert torch.allclose(fb(x), bn(x), atol=1e-6)
    blk = Bottleneck(16, 8, stride=2, downsample=True)
    assert tuple(blk(torch.randn(1, 16, 12, 12)).shape) == (1, 32, 6, 6)


def test_encoder_layer_is_the_post_norm_layer_with_positions_on_queries_and_keys_only():
    from scene_graph_commonsense_amd.detr import TransformerEncoderLayer
    torch.manual_seed(2)
    mine = TransformerEncoderLayer(d=32, nhead=4, ff=64, dropout=0.0).eval()
    ref = nn.TransformerEncoderLayer(32, 4, 64, dropout=0.0, activation="relu", norm_first=False).eval()
    ref.load_state_dict(mine.state_dict())
    src = torch.randn(10, 3, 32)
    pad = torch.zeros(3, 10, dtype=torch.bool)
    pad[1, 7:] = True
    with torch.no_grad():
        assert torch.allclose(mine(src, src_key_padding_mask=pad), ref(src, src_key_padding_mask=pad), atol=1e-5)
        # literal attention with the position embedding added to q and k but not to v
        pos = torch.randn(10, 3, 32)
        W, b = mine.self_attn.in_proj_weight, mine.self_attn.in_proj_bias
        q = F.linear(src + pos, W[:32], b[:32]).view(10, 3, 4, 8)
        k = F.linear(src + pos, W[32:64], b[32:64]).view(10, 3, 4, 8)
        v = F.linear(src, W[64:], b[64:]).view(10, 3, 4, 8)
        att = torch.einsum("qbhd,kbhd->bhqk", q, k) / math.sqrt(8)
        att = att.masked_fill(pad[:, None, None, :], float("-inf")).softmax(-1)
        o = torch.einsum("bhqk,kbhd->qbhd", att, v).reshape(10, 3, 32)
        x = mine.norm1(src + mine.self_attn.out_proj(o))
        want = mine.norm2(x + mine.linear2(F.relu(mine.linear1(x))))
        assert torch.allclose(mine(src, src_key_padding_mask=pad, pos=pos), want, atol=1e-5)


def test_sine_position_embedding_against_the_formula():
    from scene_graph_commonsense_amd.detr import NestedTensor, PositionEmbeddingSine
    mask = torch.zeros(1, 5, 6, dtype=torch.bool)
    mask[0, 4:, :] = True
    mask[0, :, 5:] = True                                   # a 4 x 5 image padded to 5 x 6
    pe = PositionEmbeddingSine(4)(NestedTensor(torch.zeros(1, 3, 5, 6), mask))[0]
    assert tuple(pe.shape) == (8, 5, 6)
    for (yy, xx) in ((0, 0), (2, 3), (3, 4)):
        ye, xe = (yy + 1) / (4 + 1e-6) * 2 * math.pi, (xx + 1) / (5 + 1e-6) * 2 * math.pi
        dim = [10000.0 ** (2 * (k // 2) / 4) for k in range(4)]
        wy = [math.sin(ye / dim[0]), math.cos(ye / dim[1]), math.sin(ye / dim[2]), math.cos(ye / dim[3])]
        wx = [math.sin(xe / dim[0]), math.cos(xe / dim[1]), math.sin(xe / dim[2]), math.cos(xe / dim[3])]
        np.testing.assert_allclose(pe[:, yy, xx].numpy(), np.array(wy + wx), rtol=1e-5, atol=1e-6)


def test_feature_extraction_and_detection_through_the_reference_call_sequence():
    """``process_image_features`` (train_utils.py:9-18) and ``detr(nested_tensor)`` (evaluate.py:309) on a small trunk."""
    from scene_graph_commonsense_amd import train_utils as TU
    from scene_graph_commonsense_amd.detr import nested_tensor_from_tensor_list
    model = _small(num_outputs=151)
    images = [torch.randn(3, 128, 128) for _ in range(2)]
    args = {"models": {"num_img_feature": 256, "feature_size": 4}}
    with torch.no_grad():
        feats = TU.process_image_features(args, images, model, "cpu")
        assert tuple(feats.shape) == (2, 256, 4, 4) and torch.isfinite(feats).all()
        assert torch.allclose(feats, model.encode(torch.stack(images)), atol=1e-5)
        out = model(nested_tensor_from_tensor_list([images[0], images[1][:, :96, :64]]))      # ragged sizes -> padding mask
    assert tuple(out["pred_logits"].shape) == (2, 100, 151) and tuple(out["pred_boxes"].shape) == (2, 100, 4)
    assert float(out["pred_boxes"].min()) > 0 and float(out["pred_boxes"].max()) < 1
    # padding must not leak: the un-padded first image alone gives the same queries
    with torch.no_grad():
        alone = model(nested_tensor_from_tensor_list([images[0]]))
    assert torch.allclose(alone["pred_logits"][0], out["pred_logits"][0], atol=1e-4)


def test_build_detr101_loads_a_checkpoint_with_the_reference_key_renaming(tmp_path):
    from scene_graph_commonsense_amd.detr import DETR, build_detr101
    torch.manual_seed(3)
    src = DETR(num_outputs=151, blocks=(1, 1, 1, 1))
    state = dict(src.state_dict())
    # two keys stored under other names, renamed back through the two lists exactly as utils.py:96-109 does
    state["backbone.stem.conv1.weight"] = state.pop("backbone.0.body.conv1.weight")
    state["criterion.empty_weight"] = torch.ones(152)
    ck = tmp_path / "ckpt.pth"
    torch.save({"model": state}, ck)
    (tmp_path / "before.txt").write_text("backbone.stem.conv1.weight\n")
    (tmp_path / "after.txt").write_text("backbone.0.body.conv1.weight\n")
    import scene_graph_commonsense_amd.detr as D
    orig = D.DETR
    D.DETR = lambda num_outputs: orig(num_outputs=num_outputs, blocks=(1, 1, 1, 1))
    try:
        args = {"dataset": {"dataset": "vg"}, "models": {"detr101_pretrained_vg": str(ck), "detr101_key_before": str(tmp_path / "before.txt"),
                                                           "detr101_key_after": str(tmp_path / "after.txt")}}
        got = build_detr101(args)
    finally:
        D.DETR = orig
    for k, v in src.state_dict().items():
        assert torch.equal(got.state_dict()[k], v), k

"""DETR-101 feature extraction on the MI355X (scene_graph_commonsense_amd/detr.py) feeding the fused relation head: pixels ->
encoder features [B,256,32,32] -> all ordered pairs, through the reference's call sequence (train_utils.py:9-18)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pixels_to_relation_scores_and_bf16_encoder_agrees_with_f32():
    from scene_graph_commonsense_amd import train_utils as TU
    from scene_graph_commonsense_amd.detr import build_detr101
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    torch.manual_seed(0)
    cfg = HeadConfig()
    args = cfg.args()
    args["models"].update(num_img_feature=256, feature_size=32)
    detr = build_detr101(args).cuda().eval()                       # no checkpoint configured: random weights
    images = [torch.randn(3, 1024, 1024, device="cuda") for _ in range(2)]
    with torch.no_grad():
        feats = TU.process_image_features(args, images, detr, "cuda:0")
        assert tuple(feats.shape) == (2, 256, 32, 32) and torch.isfinite(feats).all()
        same = detr.encode(torch.stack(images))
        assert torch.allclose(feats, same, atol=1e-4, rtol=1e-4)
        half = detr.encode(torch.stack(images), autocast=torch.bfloat16)
        rel = float((half - feats).norm() / feats.norm())
        assert rel < 5e-2, rel                                     # bf16 matrix cores for the frozen extractor: a per-cent-level change
        out = detr(images)
        assert tuple(out["pred_logits"].shape) == (2, 100, 151)
    model = BayesianRelationClassifier(args).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=1, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, [7, 5], seed=2)
    batch.image_feature = feats.cpu()
    sc = flatten_scene(cfg, batch, "cuda:0")
    res = model.forward_pairs(sc)
    assert res.relation.shape[0] == 7 * 6 + 5 * 4 and torch.isfinite(res.relation).all()

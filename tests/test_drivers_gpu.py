"""The same-named driver modules of SURVEY 8(b) - ``train_test.training`` / ``testing`` and ``evaluate.eval_pc`` with the
reference's signatures - end to end on one GPU: a tiny dataset of precomputed DETR features in the reference dataloader's
tuple layout (``dataloader.py:163-165``), RCCL process group of one rank, checkpoint in the reference's naming with the
``module.`` prefix, per-rank result files."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyVG(torch.utils.data.Dataset):
    """Items are the reference loader's 9-tuples with the image slot already holding the [256,32,32] encoder features."""

    def __init__(self, cfg, n_items, seed, none_at=()):
        from scene_graph_commonsense_amd.synthetic import make_scene_batch
        sizes = [3 + (i * 5) % 4 for i in range(n_items)]
        self.b = make_scene_batch(cfg, sizes, seed=seed, connect_frac=0.5)
        self.none_at = set(none_at)

    def __len__(self):
        return len(self.b.bbox)

    def __getitem__(self, i):
        if i in self.none_at:
            return None                                        # the reference's loader drops such images (collate_fn filters them)
        b = self.b
        return (b.image_feature[i], b.image_feature[i] * 0.9 + 0.05, b.image_depth[i], b.categories[i], b.super_categories[i], b.bbox[i],
                b.relationships[i], b.subj_or_obj[i], "img_%d_annotations.pkl" % i)


def _args(tmp_path, run_mode):
    from scene_graph_commonsense_amd.synthetic import HeadConfig
    from tests.golden_cases import GOLDEN
    cfg = HeadConfig()
    args = cfg.args(run_mode=run_mode, fixtures=os.path.join(GOLDEN, "ref_fixtures") + os.sep)
    args["models"]["feature_encoder"] = "precomputed"
    args["training"].update(batch_size=2, num_epoch=1, start_epoch=0, continue_train=False, scheduler_param1=5, scheduler_param2=8,
                            print_freq=1, print_freq_test=1, eval_freq=1, eval_freq_test=1, test_epoch=0, save_vis_results=False,
                            result_path=str(tmp_path) + os.sep, checkpoint_path=str(tmp_path) + os.sep, learning_rate=1e-5)
    return cfg, args


class TinyOIV6(torch.utils.data.Dataset):
    """OpenImages-style items: 601 classes, no super-categories (``None`` in that slot, ``dataloader.py`` OpenImageV6Dataset)."""

    def __init__(self, cfg, n_items, seed):
        from scene_graph_commonsense_amd.synthetic import make_scene_batch
        self.b = make_scene_batch(cfg, [3 + (i * 3) % 4 for i in range(n_items)], seed=seed, connect_frac=0.5)

    def __len__(self):
        return len(self.b.bbox)

    def __getitem__(self, i):
        b = self.b
        return (b.image_feature[i], b.image_feature[i] * 0.9 + 0.05, b.image_depth[i], b.categories[i], None, b.bbox[i],
                b.relationships[i], b.subj_or_obj[i], "img_%d_annotations.pkl" % i)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_training_then_eval_pc_drivers(tmp_path, monkeypatch):
    from scene_graph_commonsense_amd import evaluate, train_test
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", _free_port())
    monkeypatch.setenv("WORLD_SIZE", "1")
    cfg, args = _args(tmp_path, "train")
    train, test = TinyVG(cfg, 4, seed=41, none_at=(2,)), TinyVG(cfg, 4, seed=42)
    train_test.training(0, args, train, test)
    ckpt = os.path.join(str(tmp_path), "HierRelationModel_Baseline_motif0_0.pth")
    assert os.path.exists(ckpt)
    sd = torch.load(ckpt, map_location="cpu")
    assert all(k.startswith("module.") for k in sd) and sd["module.fc1.weight"].shape == (4096, 65536)
    from scene_graph_commonsense_amd.synthetic import make_state_dict
    rec = json.load(open(os.path.join(str(tmp_path), "train_results_0.json")))
    assert len(rec) == 2 and all(np.isfinite(r["total_losses"]) and r["total_losses"] > 0 for r in rec)
    assert rec[0]["num_connected"] + rec[0]["num_not_connected"] > 0
    trec = json.load(open(os.path.join(str(tmp_path), "test_results_0.json")))
    assert len(trec) == 2 and all(0.0 <= r <= 1.0 for r in trec[-1]["recall_relationship"])
    # evaluation driver on the checkpoint the training driver wrote (the reference loads "<...>_<epoch>_0.pth")
    monkeypatch.setenv("MASTER_PORT", _free_port())
    _, args_e = _args(tmp_path, "eval")
    recall, mean_recall = evaluate.eval_pc(0, args_e, test)
    assert len(recall) == 3 and all(0.0 <= float(r) <= 1.0 for r in recall)
    erec = json.load(open(os.path.join(str(tmp_path), "test_results_0.json")))
    assert len(erec) == 2 and erec[-1]["num_connected"] >= 0
    assert not torch.distributed.is_initialized()


class FakeDetr:
    """Stands in for DETR-101 in the driver tests: features pass through (the image slot holds precomputed encoder features) and
    ``detect`` returns the decoder outputs carried in the second image slot ([100, 151 + 4] = logits | cxcywh boxes)."""

    def eval(self):
        return self

    def __call__(self, images):
        return images

    def detect(self, images):
        x = torch.stack(list(images))
        return {"pred_logits": x[:, :, :151].contiguous(), "pred_boxes": x[:, :, 151:].contiguous()}


class TinySGDet(torch.utils.data.Dataset):
    def __init__(self, cfg, n_items, seed):
        from scene_graph_commonsense_amd.synthetic import make_scene_batch
        from tests import sgdet_case
        self.b = make_scene_batch(cfg, [4 + i % 3 for i in range(n_items)], seed=seed, connect_frac=0.6)
        inv = np.argsort(sgdet_case.alp2fre_table())
        g = torch.Generator().manual_seed(seed)
        self.det = []
        for i in range(n_items):
            lg = torch.randn(100, 151, generator=g) * 0.3
            lg[:, 150] += 10.0
            bx = torch.rand(100, 4, generator=g) * 0.2 + 0.4
            for o in range(self.b.bbox[i].shape[0]):
                c = int(self.b.categories[i][o])
                lg[2 * o + 1, 150] -= 10.0
                lg[2 * o + 1, int(inv[c])] += 14.0
                lg[2 * o + 1, int(inv[(c * 7 + 3) % 150])] += 9.0
                x0, x1, y0, y1 = [float(v) for v in self.b.bbox[i][o]]
                bx[2 * o + 1] = torch.tensor([(x0 + x1) / 64.0, (y0 + y1) / 64.0, (x1 - x0) / 32.0, (y1 - y0) / 32.0])
            self.det.append(torch.cat((lg, bx.clamp(0, 1)), dim=1))

    def __len__(self):
        return len(self.det)

    def __getitem__(self, i):
        b = self.b
        return (b.image_feature[i], self.det[i], b.image_depth[i], b.categories[i], b.super_categories[i], b.bbox[i],
                b.relationships[i], b.subj_or_obj[i], "img_%d_annotations.pkl" % i)


@pytest.mark.parametrize("mode", ["sgd", "sgc"])
def test_eval_sgd_and_sgc_drivers(tmp_path, monkeypatch, mode):
    """``evaluate.eval_sgd`` / ``eval_sgc`` (evaluate.py:230-461, 464-702) end to end: DETR decoder outputs -> HIP object
    front-end (-> label matching for SGCLS) -> fused pair path over the predicted objects -> Evaluator(predcls=False)."""
    from scene_graph_commonsense_amd import evaluate, train_test
    from scene_graph_commonsense_amd.synthetic import make_state_dict
    from tests import sgdet_case
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", _free_port())
    monkeypatch.setenv("WORLD_SIZE", "1")
    cfg, args = _args(tmp_path, "eval")
    args["models"].update(feature_encoder=FakeDetr(), topk_cat=2, nms=0.5)
    args["dataset"]["object_class_alp2fre"] = sgdet_case.alp2fre_table().tolist()
    args["training"]["eval_mode"] = mode
    model = train_test.build_classifier(args, 0)
    model.load_state_dict(make_state_dict(cfg, seed=9, head_gain=5.0))
    train_test.save_checkpoint(model, train_test.checkpoint_name(args, 0, False)[1])       # "<...>_0_0.pth", the spelling the reference loads
    data = TinySGDet(cfg, 4, seed=77)
    recall, mean_recall = (evaluate.eval_sgd if mode == "sgd" else evaluate.eval_sgc)(0, args, data)
    assert len(recall) == 3 and all(0.0 <= float(r) <= 1.0 for r in recall)
    rec = json.load(open(os.path.join(str(tmp_path), "test_results_0.json")))
    assert len(rec) == 2 and len(rec[-1]["recall_relationship"]) == 3
    assert not torch.distributed.is_initialized()


def test_training_driver_on_openimages_shaped_data(tmp_path, monkeypatch):
    """The OIV6 branch of ``training`` / ``testing`` (601 classes, (4,2,24) head, no super-categories, weighted mean AP through
    ``Evaluator.compute_precision`` instead of zero-shot recall / Top-3)."""
    from scene_graph_commonsense_amd import train_test
    from scene_graph_commonsense_amd.synthetic import HeadConfig
    from tests.golden_cases import GOLDEN
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", _free_port())
    monkeypatch.setenv("WORLD_SIZE", "1")
    cfg = HeadConfig(dataset="oiv6", num_classes=601, num_super_classes=0, num_geometric=4, num_possessive=2, num_semantic=24)
    args = cfg.args(run_mode="train", fixtures=os.path.join(GOLDEN, "ref_fixtures") + os.sep)
    args["models"]["feature_encoder"] = "precomputed"
    args["training"].update(batch_size=2, num_epoch=1, start_epoch=0, continue_train=False, scheduler_param1=5, scheduler_param2=8,
                            print_freq=1, print_freq_test=1, eval_freq=1, eval_freq_test=1, test_epoch=0, save_vis_results=False,
                            result_path=str(tmp_path) + os.sep, checkpoint_path=str(tmp_path) + os.sep, learning_rate=1e-5)
    train_test.training(0, args, TinyOIV6(cfg, 4, seed=51), TinyOIV6(cfg, 2, seed=52))
    rec = json.load(open(os.path.join(str(tmp_path), "train_results_0.json")))
    assert len(rec) == 2 and all(np.isfinite(r["total_losses"]) for r in rec) and "wmap_rel" in rec[-1]
    trec = json.load(open(os.path.join(str(tmp_path), "test_results_0.json")))
    assert len(trec) == 1 and "wmap_rel" in trec[-1]
    sd = torch.load(os.path.join(str(tmp_path), "HierRelationModel_Baseline_motif0_0.pth"), map_location="cpu")
    assert sd["module.fc2.weight"].shape == (512, 4096 + 2 * 601)

"""Drop-in for the reference's ``evaluate.py``: ``eval_pc`` (``evaluate.py:29-227``), ``eval_sgd`` (``:230-461``) and ``eval_sgc``
(``:464-702``) with the reference's signatures, so ``main.py``'s ``from evaluate import eval_pc, eval_sgc, eval_sgd`` +
``mp.spawn(eval_*, ...)`` (``main.py:15,112-123``) run on the MI355X path.

One process per GPU over RCCL; the pair loops are one fused call per minibatch (``pair_loop.evaluate_minibatch`` /
``evaluate_sgdet_minibatch``); the DETR decoder outputs go through the HIP object front-end (``object_frontend.DetrFrontEnd``:
soft-max / top-k / class re-indexing / boxes / per-class NMS, and the SGCLS label matching).  Evaluation needs no collective;
every rank writes its own ``test_results_<rank>.json`` as in the reference.  Out of scope here as in DESIGN.md: the LLM
``prepare_cs`` pipeline and ``save_vis_results`` (they raise).
"""
from __future__ import annotations

import json
import os

import torch
import torch.distributed as dist

from .evaluator import Evaluator, Evaluator_Top3
from .object_frontend import DetrFrontEnd
from .pair_loop import MinibatchLookahead, evaluate_minibatch, evaluate_sgdet_minibatch, freeze_setup_objects
from .pairs import flatten_scene
from .train_test import (_collate, _host, _record_test, _to_batch, build_classifier, build_feature_encoder, load_checkpoint, setup)
from .train_utils import process_image_features


def _loader(args, test_subset, rank, world_size):
    sampler = torch.utils.data.distributed.DistributedSampler(test_subset, num_replicas=world_size, rank=rank)
    return torch.utils.data.DataLoader(test_subset, batch_size=args["training"]["batch_size"], shuffle=False,
                                       collate_fn=_host("collate_fn", _collate), num_workers=0, drop_last=True, sampler=sampler)


def _start(gpu, args, test_subset):
    rank = gpu
    world_size = int(os.environ.get("WORLD_SIZE", 0)) or max(torch.cuda.device_count(), 1)
    setup(rank, world_size)
    print("rank", rank, "torch.distributed.is_initialized", dist.is_initialized())
    loader = _loader(args, test_subset, rank, world_size)
    print("Finished loading the datasets...")
    with open(args["training"]["result_path"] + "test_results_" + str(rank) + ".json", "w") as f:
        json.dump([], f)
    model = build_classifier(args, rank)
    detr = build_feature_encoder(args, rank)
    model.eval()
    load_checkpoint(model, args, args["training"]["test_epoch"], args["training"]["run_mode"] == "eval_cs", rank)
    return rank, loader, model, detr


def _finish(name):
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    print("FINISHED TESTING %s\n" % name)


def eval_pc(gpu, args, test_subset, curr_dataset=None, prepare_cs_step=-1):
    """Predicate classification (``evaluate.py:29-227``)."""
    T = args["training"]
    if T["run_mode"] == "prepare_cs" or prepare_cs_step != -1:
        raise NotImplementedError("the LLM commonsense collection (run_mode prepare_cs) is outside this package's scope (DESIGN.md)")
    if T.get("save_vis_results"):
        raise NotImplementedError("save_vis_results is outside this package's scope (DESIGN.md)")
    rank, loader, model, detr = _start(gpu, args, test_subset)
    hier, vg = args["models"]["hierarchical_pred"], args["dataset"]["dataset"] == "vg"
    Recall = Evaluator(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100])
    Recall_top3 = Evaluator_Top3(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100]) if vg else None
    record_test = _host("record_test_results", _record_test)
    test_record = []
    stats = torch.zeros(5, dtype=torch.int64, device=rank)
    recall = recall_top3 = mean_recall_top3 = mean_recall = recall_zs = mean_recall_zs = wmap_rel = wmap_phrase = None
    skip = bool(T.get("skip_filtered_pairs", False))
    print("Start Testing PC...")
    freeze_setup_objects()
    cfg, dev = model.head_config(), next(model.parameters()).device

    def prepare(batch_count, data):
        """minibatch k+1 is loaded, encoded and flattened while the device scores minibatch k (pair_loop.MinibatchLookahead)"""
        try:
            batch, _, annot_path = _to_batch(args, data, detr, rank, with_aug=False)
        except (ValueError, IndexError):
            return None
        return batch, annot_path, flatten_scene(cfg, batch, dev)
    with torch.no_grad():
        ahead = MinibatchLookahead(loader, prepare)
        for batch_count, (batch, annot_path, scene) in ahead:
            Recall.load_annotation_paths(annot_path)
            last = batch_count + 1 == len(loader)
            feed = batch_count % T["eval_freq_test"] == 0 or last
            evaluate_minibatch(model, batch, Recall if feed else None, Recall_top3 if (feed and hier) else None, skip_filtered=skip,
                               scene=scene, while_running=ahead.fetch_next)
            if model.last_connectivity_stats is not None:
                stats += model.last_connectivity_stats
            if feed:
                if vg:
                    recall, _, mean_recall, recall_zs, _, mean_recall_zs = Recall.compute(per_class=True)
                    if hier:
                        recall_top3, _, mean_recall_top3 = Recall_top3.compute(per_class=True)
                        Recall_top3.clear_data()
                else:
                    recall, _, mean_recall, _, _, _ = Recall.compute(per_class=True)
                    wmap_rel, wmap_phrase = Recall.compute_precision()
                if batch_count % T["print_freq_test"] == 0 or last:
                    s = stats.tolist()
                    record_test(args, test_record, rank, T["test_epoch"], recall_top3, recall, mean_recall_top3, mean_recall, recall_zs,
                                mean_recall_zs, torch.tensor(float(s[4])), s[1], s[0], torch.tensor(float(s[3])), s[2], wmap_rel, wmap_phrase)
                Recall.clear_data()
    _finish("PC")
    return recall, mean_recall


def _detr_outputs(detr, image2, rank):
    """``detr(nested_tensor_from_tensor_list(image2))`` -> (pred_logits [B,100,C+1], pred_boxes [B,100,4]) (``evaluate.py:309``);
    a feature encoder that is not the DETR module must offer ``detect(images)`` returning the same dict."""
    images = [im.to(rank) for im in image2]
    if hasattr(detr, "detect"):
        out = detr.detect(images)
    else:
        try:
            from utils import nested_tensor_from_tensor_list        # host repository helper
        except Exception:
            from .detr import nested_tensor_from_tensor_list
        out = detr(nested_tensor_from_tensor_list(images))
    return out["pred_logits"], out["pred_boxes"]


def _sg_common(gpu, args, test_subset, sgcls: bool):
    T = args["training"]
    rank, loader, model, detr = _start(gpu, args, test_subset)
    Recall = Evaluator(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100])
    sub2super = torch.load(args["dataset"]["sub2super_cat_dict"])
    try:
        from dataset_utils import object_class_alp2fre                # the host repository's own class-index table
        alp2fre = object_class_alp2fre()
    except Exception:
        alp2fre = args["dataset"].get("object_class_alp2fre")
        if alp2fre is None:
            raise RuntimeError("the DETR (alphabetical) -> dataset (frequency) class table is needed: make the host repository's "
                               "dataset_utils importable or pass args['dataset']['object_class_alp2fre']")
    fe = DetrFrontEnd(alp2fre, num_classes=args["models"]["num_classes"], topk_cat=args["models"]["topk_cat"],
                      feature_size=args["models"]["feature_size"], nms=args["models"]["nms"])
    record_test = _host("record_test_results", _record_test)
    test_record = []
    recall = mean_recall = recall_zs = mean_recall_zs = None
    name = "SGC" if sgcls else "SGD"
    print("Start Testing %s..." % name)
    freeze_setup_objects()
    with torch.no_grad():
        for batch_count, data in enumerate(loader):
            try:
                images, image2, image_depth, categories_target, _, bbox_target, relationships, subj_or_obj = data[:8]
            except ValueError:
                continue
            image_feature = process_image_features(args, images, detr, rank)
            depth = torch.stack([d.to(rank) for d in image_depth])
            logits, boxes = _detr_outputs(detr, image2, rank)
            categories_pred, cat_conf, bbox_pred, kept = fe.sgdet(logits, boxes)
            if len(kept) != int(image_feature.shape[0]):
                continue              # an image without any detected object: the reference's index bookkeeping breaks there too
            cats_t = [c.to(rank) for c in categories_target]
            box_t = [b.to(rank) for b in bbox_target]
            if sgcls:                 # ground-truth boxes, labels matched from the predictions (utils.py:376-425)
                matched = fe.match_object_categories(categories_pred, cat_conf, bbox_pred, box_t)
                if matched[0] is None or matched[1] is None:
                    continue
                categories_pred, cat_conf, bbox_pred = matched[0], matched[1], matched[2]
            last = batch_count + 1 == len(loader)
            if batch_count % T["eval_freq_test"] == 0 or last:
                evaluate_sgdet_minibatch(model, image_feature, depth, categories_pred, cat_conf, bbox_pred, Recall, sub2super=sub2super,
                                         targets=(relationships, subj_or_obj, cats_t, box_t),
                                         skip_filtered=bool(T.get("skip_filtered_pairs", False)))
                recall, _, mean_recall, recall_zs, _, mean_recall_zs = Recall.compute(per_class=True, predcls=False)
                Recall.clear_data()
                if batch_count % T["print_freq_test"] == 0 or last:
                    record_test(args, test_record, rank, T["test_epoch"], None, recall, None, mean_recall, recall_zs, mean_recall_zs,
                                torch.tensor(0.0), 0.0, 0.0, torch.tensor(0.0), 0.0, None, None)
    _finish(name)
    return recall, mean_recall


def eval_sgd(gpu, args, test_subset):
    """Scene graph detection (``evaluate.py:230-461``): predicted boxes and labels."""
    return _sg_common(gpu, args, test_subset, sgcls=False)


def eval_sgc(gpu, args, test_subset):
    """Scene graph classification (``evaluate.py:464-702``): ground-truth boxes, predicted labels."""
    return _sg_common(gpu, args, test_subset, sgcls=True)

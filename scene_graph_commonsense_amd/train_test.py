"""Drop-in for the reference's ``train_test.py``: ``setup``, ``training(gpu, args, train_subset, test_subset)`` and
``testing(args, detr, relation_classifier, test_loader, test_record, epoch, rank, writer)`` with the reference's signatures
(``train_test.py:25-28,31-330,333-465``), so ``main.py``'s ``from train_test import training`` + ``mp.spawn(training, ...)``
(``main.py:14,103``) run on the MI355X path.

What changed against the reference, and why:
* one process per GPU over **RCCL** (``backend="nccl"``) on 127.0.0.1 instead of gloo on ``localhost``; ``dist.barrier`` instead
  of the gloo-only ``monitored_barrier``;
* the ``graph_iter`` / ``edge_iter`` loops (hundreds of classifier calls per minibatch) are ONE fused call per minibatch
  (``pair_loop.train_minibatch`` / ``evaluate_minibatch``): same pairs, same loss (running-sum step weights included), same
  evaluator feed order, gradients mean-reduced across ranks by ``distributed.GradReducer`` instead of a DDP wrapper;
* everything that is NOT the hot path is taken from the host repository when it is importable (``utils.collate_fn``,
  ``utils.build_detr101``, ``utils.record_train_results`` / ``record_test_results``, ``utils.get_num_each_class_reordered``), with
  small local fall-backs, so this file also runs stand-alone on datasets of precomputed DETR features
  (``args['models']['feature_encoder'] = 'precomputed'`` or any callable).
Checkpoints keep the reference's names and its ``module.`` key prefix (``train_test.py:310-322``), and load with or without it.
"""
from __future__ import annotations

import json
import math
import os
import shutil

import numpy as np
import torch
import torch.distributed as dist

from .distributed import GradReducer, ShardedSGD
from .evaluator import Evaluator, Evaluator_Top3
from .model import BayesianRelationClassifier, FlatRelationClassifier, strip_ddp_prefix
from .optim import FusedSGD
from .pairs import flatten_scene
from .pair_loop import MinibatchLookahead, evaluate_minibatch, feed_evaluators, freeze_setup_objects, train_minibatch
from .synthetic import SceneBatch
from .train_utils import process_image_features


# ------------------------------------------------------------------------------------------------ process group
def setup(rank, world_size, backend=None):
    """``train_test.py:25-28`` on RCCL: one process per GPU, rendezvous on 127.0.0.1 (MASTER_ADDR / MASTER_PORT from the
    environment win)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12356")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(rank)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world_size)


# ------------------------------------------------------------------------------------------------ host-repository helpers
def _host(name, fallback):
    """A helper of the host repository's ``utils`` module (the reference's own, outside the hot path) or the local fall-back."""
    try:
        import utils as host_utils
        return getattr(host_utils, name)
    except Exception:
        return fallback


def _collate(batch):
    batch = [x for x in batch if x is not None]
    return tuple(zip(*batch))


class PrecomputedFeatures:
    """Feature "encoder" for datasets whose image slot already holds the ``[256,32,32]`` DETR encoder output (BASELINE config 1)."""

    def eval(self):
        return self

    def __call__(self, images):
        return images


def build_feature_encoder(args, rank):
    enc = args["models"].get("feature_encoder", "detr101")
    if callable(enc):
        return enc
    if enc == "precomputed":
        return PrecomputedFeatures()
    try:
        from utils import build_detr101                  # host repository: torch.hub DETR-101 (needs the network once, and its weights)
    except Exception:
        from .detr import build_detr101                  # stand-alone: the same architecture and state-dict names, no hub (detr.py)
    return build_detr101(args).to(rank).eval()


def _class_weight(args):
    fn = _host("get_num_each_class_reordered", None)
    if fn is not None:
        counts = fn(args).double()
    else:
        from .synthetic import HeadConfig, predicate_counts
        counts = predicate_counts(HeadConfig(dataset=args["dataset"]["dataset"], num_geometric=args["models"]["num_geometric"],
                                             num_possessive=args["models"]["num_possessive"],
                                             num_semantic=args["models"]["num_semantic"])).double()
    return (1 - counts / counts.sum()).numpy()


def _scalar(x):
    return float(x) if x is not None else None


def _record(path, record, entry):
    record.append(entry)
    with open(path, "w") as f:
        json.dump(record, f)


def _record_train(args, record, rank, epoch, batch_count, lr, recall_top3, recall, mean_recall_top3, mean_recall, recall_zs, mean_recall_zs,
                  running_losses, running_loss_relationship, running_loss_contrast, running_loss_connectivity, running_loss_commonsense,
                  connectivity_recall, num_connected, num_not_connected, connectivity_precision, num_connected_pred, wmap_rel, wmap_phrase):
    norm = args["training"]["print_freq"] * args["training"]["batch_size"]
    entry = {"rank": rank, "epoch": epoch, "batch": batch_count, "lr": lr, "recall_relationship": [float(r) for r in recall],
             "mean_recall": [float(r) for r in mean_recall], "total_losses": float(running_losses) / norm,
             "connectivity_recall": float(connectivity_recall) / (num_connected + 1e-5),
             "connectivity_precision": float(connectivity_precision) / (num_connected_pred + 1e-5),
             "num_connected": num_connected, "num_not_connected": num_not_connected}
    if recall_zs is not None:
        entry["zero_shot_recall"] = [float(r) for r in recall_zs]
    if wmap_rel is not None:
        entry.update(wmap_rel=float(wmap_rel), wmap_phrase=float(wmap_phrase))
    print("TRAIN, rank %d, epoch %d, batch %d, lr: %.7f, R@k: %.4f, %.4f, %.4f, loss: %.4f" %
          (rank, epoch, batch_count, lr, recall[0], recall[1], recall[2], entry["total_losses"]))
    _record(args["training"]["result_path"] + "train_results_" + str(rank) + ".json", record, entry)


def _record_test(args, test_record, rank, epoch, recall_top3, recall, mean_recall_top3, mean_recall, recall_zs, mean_recall_zs,
                 connectivity_recall, num_connected, num_not_connected, connectivity_precision, num_connected_pred, wmap_rel, wmap_phrase):
    entry = {"rank": rank, "epoch": epoch, "recall_relationship": [float(r) for r in recall], "mean_recall": [float(r) for r in mean_recall],
             "connectivity_recall": float(connectivity_recall) / (num_connected + 1e-5),
             "connectivity_precision": float(connectivity_precision) / (num_connected_pred + 1e-5),
             "num_connected": num_connected, "num_not_connected": num_not_connected}
    if recall_top3 is not None:
        entry["recall_top3"] = [float(r) for r in recall_top3]
    if recall_zs is not None:
        entry["zero_shot_recall"] = [float(r) for r in recall_zs]
    if wmap_rel is not None:
        entry.update(wmap_rel=float(wmap_rel), wmap_phrase=float(wmap_phrase))
    print("TEST, rank: %d, epoch: %d, R@k: %.4f, %.4f, %.4f, mR@k: %.4f, %.4f, %.4f" %
          (rank, epoch, recall[0], recall[1], recall[2], mean_recall[0], mean_recall[1], mean_recall[2]))
    _record(args["training"]["result_path"] + "test_results_" + str(rank) + ".json", test_record, entry)


# ------------------------------------------------------------------------------------------------ model / checkpoint plumbing
def build_classifier(args, rank):
    m = args["models"]
    if m["hierarchical_pred"]:
        net = BayesianRelationClassifier(args=args, input_dim=m["hidden_dim"], feature_size=m["feature_size"], num_classes=m["num_classes"],
                                         num_super_classes=m["num_super_classes"], num_geometric=m["num_geometric"],
                                         num_possessive=m["num_possessive"], num_semantic=m["num_semantic"])
    else:
        net = FlatRelationClassifier(args=args, input_dim=m["hidden_dim"], output_dim=m["num_relations"], feature_size=m["feature_size"],
                                     num_classes=m["num_classes"], num_super_classes=m["num_super_classes"])
    return net.to(rank)


def checkpoint_name(args, epoch, cs_mode):
    """``train_test.py:82-90,308-317`` / ``evaluate.py:65-73``: <path><Hier|Flat>RelationModel_<CS|Baseline>_<clustering>[_]<epoch>_0.pth.
    The reference SAVES without the underscore before the epoch and LOADS with it; both spellings are looked up when loading."""
    base = ("HierRelationModel" if args["models"]["hierarchical_pred"] else "FlatRelationModel") + ("_CS" if cs_mode else "_Baseline")
    base = args["training"]["checkpoint_path"] + base + "_" + args["dataset"]["supcat_clustering"]
    return base + str(epoch) + "_0.pth", base + "_" + str(epoch) + "_0.pth"


def save_checkpoint(model, path):
    torch.save({"module." + k: v for k, v in model.state_dict().items()}, path)


def load_checkpoint(model, args, epoch, cs_mode, rank):
    for path in reversed(checkpoint_name(args, epoch, cs_mode)):
        if os.path.exists(path):
            if rank == 0:
                print("Loading pretrained model from %s..." % path)
            model.load_state_dict(strip_ddp_prefix(torch.load(path, map_location="cuda:%d" % rank)))
            return path
    raise FileNotFoundError("no checkpoint at %s (or %s)" % checkpoint_name(args, epoch, cs_mode))


class EmptyMinibatch(Exception):
    """The dataloader tuple holds no sample (all dropped): the only condition under which a step is skipped.  Errors of the
    feature encoder are NOT caught - a rank that swallowed one would skip the step's collectives."""


def _to_batch(args, data, detr, rank, with_aug):
    """The dataloader tuple (``dataloader.py:163-165``) -> ``SceneBatch`` (+ augmented-view features)."""
    try:
        images, images_aug, image_depth, categories, super_categories, bbox, relationships, subj_or_obj, annot_path = data[:9]
    except (ValueError, IndexError):
        raise EmptyMinibatch()               # every sample was dropped by the loader: the collate gave an empty tuple
    with torch.no_grad():
        image_feature = process_image_features(args, images, detr, rank)
        feature_aug = process_image_features(args, images_aug, detr, rank) if (with_aug and images_aug[0] is not None) else None
    depth = torch.stack([d.to(rank) for d in image_depth])
    sup = None if super_categories[0] is None else [list(s) for s in super_categories]
    return SceneBatch(image_feature, depth, list(bbox), list(categories), sup, list(relationships), list(subj_or_obj)), feature_aug, annot_path


def _commonsense_keys(args):
    if args["training"]["run_mode"] != "train_cs":
        return None
    suffix = "_gpt4v" if args["models"].get("llm_model") == "gpt4v" else ""
    aligned = torch.load("triplets/commonsense_aligned_triplets%s.pt" % suffix)
    violated = torch.load("triplets/commonsense_violated_triplets%s.pt" % suffix)
    return list(aligned.keys()), list(violated.keys())


# ------------------------------------------------------------------------------------------------ training()
def training(gpu, args, train_subset, test_subset):
    """Train and evaluate the local prediction module on predicate classification (``train_test.py:31-330``)."""
    rank = gpu
    world_size = int(os.environ.get("WORLD_SIZE", 0)) or max(torch.cuda.device_count(), 1)
    setup(rank, world_size)
    print("rank", rank, "torch.distributed.is_initialized", dist.is_initialized())
    T = args["training"]

    writer = None
    if rank == 0:
        try:
            from torch.utils.tensorboard import SummaryWriter
            log_dir = "runs/train_sg"
            if os.path.exists(log_dir):
                shutil.rmtree(log_dir)
            writer = SummaryWriter(log_dir)
        except Exception:
            writer = None                                    # tensorboard is optional (not installed in the ROCm image)

    collate_fn = _host("collate_fn", _collate)
    make_loader = lambda ds: torch.utils.data.DataLoader(
        ds, batch_size=T["batch_size"], shuffle=False, collate_fn=collate_fn, num_workers=0, drop_last=True,
        sampler=torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world_size, rank=rank))
    train_loader, test_loader = make_loader(train_subset), make_loader(test_subset)
    print("Finished loading the datasets...")

    record, test_record = [], []
    paths = [T["result_path"] + "%s_results_%d.json" % (k, rank) for k in ("train", "test")]
    if not T["continue_train"]:
        for p in paths:
            with open(p, "w") as f:
                json.dump([], f)
    else:
        record, test_record = (json.load(open(p)) for p in paths)

    relation_classifier = build_classifier(args, rank)
    detr = build_feature_encoder(args, rank)
    cs_mode = T["run_mode"] == "train_cs"
    if T["continue_train"]:
        load_checkpoint(relation_classifier, args, T["start_epoch"] - 1, cs_mode, rank)
    if rank == 0:
        print(f"Total number of parameters in the model: {sum(p.numel() for p in relation_classifier.parameters())}")

    if world_size > 1 and T.get("dp_mode", "sharded") == "sharded":
        # reduce-scatter of the gradients + SGD on this rank's shard + all-gather of the parameters: reducer and optimizer in one
        optimizer = reducer = ShardedSGD(relation_classifier.named_parameters(), world_size, rank, lr=T["learning_rate"], momentum=0.9,
                                         weight_decay=T["weight_decay"], defer_gather=True).attach(relation_classifier)
    else:
        optimizer = FusedSGD(relation_classifier.parameters(), lr=T["learning_rate"], momentum=0.9, weight_decay=T["weight_decay"])
        reducer = GradReducer(world_size)
    relation_classifier.train()
    original_lr = optimizer.param_groups[0]["lr"]
    class_weight = _class_weight(args)
    commonsense = _commonsense_keys(args)
    record_train = _host("record_train_results", _record_train)

    Recall = Evaluator(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100])
    Recall_top3 = Evaluator_Top3(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100]) \
        if args["dataset"]["dataset"] == "vg" else None
    recall_top3 = recall = mean_recall_top3 = mean_recall = recall_zs = mean_recall_zs = wmap_rel = wmap_phrase = None
    running_losses = torch.zeros((), device=rank)
    running_contrast = torch.zeros((), device=rank)
    stats = torch.zeros(5, dtype=torch.int64, device=rank)     # not connected, connected, predicted connected, precision, recall numerators

    lr_decay = 1
    pending_scalar = None
    freeze_setup_objects()              # host GC: nothing built so far is garbage; keep the full collections out of the step loop
    for epoch in range(T["start_epoch"], T["num_epoch"]):
        print("Start Training... EPOCH %d / %d\n" % (epoch, T["num_epoch"]))
        if epoch == T["scheduler_param1"] or epoch == T["scheduler_param2"]:
            lr_decay *= 0.1
        for batch_count, data in enumerate(train_loader):
            try:
                batch, feature_aug, annot_path = _to_batch(args, data, detr, rank, with_aug=True)
            except EmptyMinibatch:
                # a minibatch whose samples were all dropped by the loader.  The reference just skips it - and with more than one
                # rank leaves the others waiting in DDP's all-reduce; here the rank still takes part in the step's collectives
                # (zero gradients), applies the same update as its peers and joins the print_freq barrier.
                if world_size > 1:
                    optimizer.zero_grad(set_to_none=True)
                    # lr of this update: the reference never takes this path (it hangs), so there is no golden value; the sqrt(n_max
                    # fraction) rescale of the line below needs a minibatch this rank does not have, and the lr the PREVIOUS minibatch
                    # left behind is arbitrary - take the un-rescaled epoch rate (with ShardedSGD this rank's shard uses it)
                    optimizer.param_groups[0]["lr"] = original_lr * lr_decay
                    relation_classifier.zero_gradient_step(reducer)
                    optimizer.step()
                    if batch_count % T["print_freq"] == 0 or batch_count + 1 == len(train_loader):
                        dist.barrier()
                continue
            Recall.load_annotation_paths(annot_path)
            # the reference rescales lr inside its graph_iter loop and steps once per minibatch: the value in force at
            # optimizer.step() is the one of the LAST graph_iter, whose keep_in_batch are the images with the most objects
            n = np.array([int(b.shape[0]) for b in batch.bbox])
            optimizer.param_groups[0]["lr"] = original_lr * lr_decay * math.sqrt(float((n == n.max()).sum()) / len(n))
            loss = train_minibatch(relation_classifier, batch, optimizer, reducer=reducer, class_weight=class_weight,
                                   lambda_connectivity=T["lambda_connectivity"], lambda_not_connected=T["lambda_not_connected"],
                                   image_feature_aug=feature_aug, lambda_contrast=T.get("lambda_contrast", 1.0), commonsense=commonsense,
                                   lambda_commonsense=T.get("lambda_commonsense", 1.0), lambda_cs_weak=T.get("lambda_cs_weak", 0.1),
                                   lambda_cs_strong=T.get("lambda_cs_strong", 10.0))
            running_losses += loss.detach()
            if getattr(relation_classifier, "last_contrast_loss", None) is not None and feature_aug is not None:
                running_contrast += relation_classifier.last_contrast_loss.detach()
            if relation_classifier.last_connectivity_stats is not None:
                stats += relation_classifier.last_connectivity_stats
            last = batch_count + 1 == len(train_loader)
            if (batch_count % T["eval_freq"] == 0 or last) and relation_classifier.last_outputs is None:
                pass                                         # no image of this minibatch had two objects: nothing to rank
            elif batch_count % T["eval_freq"] == 0 or last:
                # evaluator feed with the training-mode outputs of this step, no overlap filter (iou_mask all ones, train_test.py:209)
                feed_evaluators(relation_classifier, relation_classifier.last_scene, relation_classifier.last_outputs, Recall,
                                Recall_top3 if args["models"]["hierarchical_pred"] else None, overlap=None)
                if args["dataset"]["dataset"] == "vg":
                    recall, _, mean_recall, recall_zs, _, mean_recall_zs = Recall.compute(per_class=True)
                    if args["models"]["hierarchical_pred"]:
                        recall_top3, _, mean_recall_top3 = Recall_top3.compute(per_class=True)
                        Recall_top3.clear_data()
                else:
                    recall, _, mean_recall, _, _, _ = Recall.compute(per_class=True)
                    wmap_rel, wmap_phrase = Recall.compute_precision()
                Recall.clear_data()
            if rank == 0 and writer is not None:
                # the reference logs float(loss) every step (train_test.py:266): reading THIS step's value would stop the host until the
                # step has run (the device then idles ~3 ms per step while the next one is prepared); the value is written one
                # iteration later instead, under its own step index
                if pending_scalar is not None:
                    writer.add_scalar("train/running_losses", float(pending_scalar[0]), pending_scalar[1])
                pending_scalar = (running_losses.clone(), batch_count + len(train_loader) * epoch)
            if (batch_count % T["print_freq"] == 0 or last) and recall is not None:
                s = stats.tolist()
                record_train(args, record, rank, epoch, batch_count, optimizer.param_groups[0]["lr"], recall_top3, recall, mean_recall_top3,
                             mean_recall, recall_zs, mean_recall_zs, running_losses, running_losses, running_contrast,
                             torch.zeros(()), torch.zeros(()), torch.tensor(float(s[4])), s[1], s[0], torch.tensor(float(s[3])), s[2],
                             wmap_rel, wmap_phrase)
            if batch_count % T["print_freq"] == 0 or last:
                dist.barrier()
            # the reference resets the losses, connectivity_precision, num_connected and num_not_connected every minibatch and lets
            # connectivity_recall and num_connected_pred run on (train_test.py:308-309): same here, so the logged ratios agree.
            # (Per-term running losses are not kept apart on the fused path: the total is logged as running_loss_relationship,
            # connectivity and commonsense as 0.)
            running_losses.zero_(); running_contrast.zero_()
            stats[0] = 0; stats[1] = 0; stats[3] = 0
        if pending_scalar is not None:
            writer.add_scalar("train/running_losses", float(pending_scalar[0]), pending_scalar[1])
            pending_scalar = None
        if hasattr(optimizer, "wait_gathers"):
            optimizer.wait_gathers()                         # deferred all-gather of fc1.weight: land it before the parameters are read
        relation_classifier.engine().verify_checks(block=True)   # deferred device-side plan checks of the epoch's last steps: before the checkpoint
        if rank == 0:
            path = checkpoint_name(args, epoch, cs_mode)[0]
            print("Saving model to %s..." % path)
            save_checkpoint(relation_classifier, path)
        dist.barrier()
        testing(args, detr, relation_classifier, test_loader, test_record, epoch, rank, writer)
        relation_classifier.train()

    dist.destroy_process_group()
    if rank == 0 and writer is not None:
        writer.close()
    print("FINISHED TRAINING\n")


# ------------------------------------------------------------------------------------------------ testing()
def testing(args, detr, relation_classifier, test_loader, test_record, epoch, rank, writer):
    """Predicate-classification evaluation of one epoch (``train_test.py:333-465``)."""
    if hasattr(detr, "eval"):
        detr.eval()
    relation_classifier.eval()
    T = args["training"]
    stats = torch.zeros(5, dtype=torch.int64, device=rank)
    recall = mean_recall_top3 = mean_recall = recall_zs = mean_recall_zs = wmap_rel = wmap_phrase = recall_top3 = None
    Recall = Evaluator(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100])
    Recall_top3 = Evaluator_Top3(args=args, num_classes=args["models"]["num_relations"], iou_thresh=0.5, top_k=[20, 50, 100]) \
        if args["dataset"]["dataset"] == "vg" else None
    record_test = _host("record_test_results", _record_test)
    skip = bool(T.get("skip_filtered_pairs", False))          # 2x faster, identical Recall@K; connectivity counters are then not collected
    print("Start Testing PC...")
    cfg, dev = relation_classifier.head_config(), next(relation_classifier.parameters()).device

    def prepare(batch_count, data):
        """minibatch k+1 is loaded, encoded and flattened while the device scores minibatch k (pair_loop.MinibatchLookahead)"""
        if epoch < 2 and batch_count > 100:
            return MinibatchLookahead.STOP
        try:
            batch, _, _ = _to_batch(args, data, detr, rank, with_aug=False)
        except EmptyMinibatch:
            return None
        return batch, flatten_scene(cfg, batch, dev)
    with torch.no_grad():
        ahead = MinibatchLookahead(test_loader, prepare)
        for batch_count, (batch, scene) in ahead:
            last = batch_count + 1 == len(test_loader)
            feed = batch_count % T["eval_freq_test"] == 0 or last
            evaluate_minibatch(relation_classifier, batch, Recall if feed else None,
                               Recall_top3 if (feed and args["models"]["hierarchical_pred"]) else None, skip_filtered=skip,
                               scene=scene, while_running=ahead.fetch_next)
            if relation_classifier.last_connectivity_stats is not None:
                stats += relation_classifier.last_connectivity_stats
            if feed:
                if args["dataset"]["dataset"] == "vg":
                    recall, _, mean_recall, recall_zs, _, mean_recall_zs = Recall.compute(per_class=True)
                    if rank == 0 and writer is not None:
                        step = batch_count + len(test_loader) * epoch
                        for k, name in enumerate(("test/Recall@20", "test/Recall@50", "test/Recall@100")):
                            writer.add_scalar(name, recall[k], step)
                    if args["models"]["hierarchical_pred"]:
                        recall_top3, _, mean_recall_top3 = Recall_top3.compute(per_class=True)
                        Recall_top3.clear_data()
                else:
                    recall, _, mean_recall, _, _, _ = Recall.compute(per_class=True)
                    wmap_rel, wmap_phrase = Recall.compute_precision()
                Recall.clear_data()
            if batch_count % T["print_freq_test"] == 0 or last:
                s = stats.tolist()
                record_test(args, test_record, rank, epoch, recall_top3, recall, mean_recall_top3, mean_recall, recall_zs, mean_recall_zs,
                            torch.tensor(float(s[4])), s[1], s[0], torch.tensor(float(s[3])), s[2], wmap_rel, wmap_phrase)
                if dist.is_initialized():
                    dist.barrier()
    relation_classifier.engine().verify_checks(block=True)       # plan checks posted by the last evaluation passes
    print("FINISHED EVALUATING\n")
    return recall, mean_recall, stats

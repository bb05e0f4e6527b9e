"""On-disk annotation format of the reference -> the minibatch contract of the fused path (data half).

The reference's ``prepare_datasets`` writes one ``*_annotations.pkl`` per image with ``torch.save`` (``dataset_utils.py:186-196``):
``image_depth [1,F,F]``, ``categories [n]`` int64, ``super_categories`` (list of n int64 tensors), ``masks``, ``bbox [n,4]`` f32 as
(x0,x1,y0,y1) on the feature grid (objects sorted by area, descending), ``bbox_origin``, ``relationships`` / ``subj_or_obj`` (lower
triangular lists: entry g-1 has length g).  ``PrepareVisualGenomeDataset.__getitem__`` (``dataloader.py:111-147``) then drops images
with <= 1 or > 20 objects (or with a box that is empty at image resolution), truncates the boxes to int and re-indexes the
predicates (``wears`` 12 -> ``wearing`` 4, then frequency order -> super-category order).  This module does the same on the host -
it is bookkeeping on a few dozen integers per image - and collates the survivors into a ``SceneBatch``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .synthetic import SceneBatch

# dataset_utils.py:647-650 (data): predicate index in frequency order -> index in super-category order; slot 50 maps "none" to -1
RELATION_CLASS_FREQ2SCAT = torch.tensor([11, 18, 8, 20, 23, 10, 25, 0, 34, 6, 14, 44, 24, 45, 9, 26, 5, 33, 13, 16,
                                         42, 27, 30, 48, 41, 29, 35, 3, 49, 4, 7, 15, 39, 2, 36, 17, 40, 22, 19, 28,
                                         38, 43, 21, 1, 31, 46, 12, 37, 32, 47, -1])


def prepare_annotation(annot: Dict, feature_size: int = 32, use_depth: bool = True, image_hw: Optional[Tuple[int, int]] = None,
                       rel_reorder: Optional[torch.Tensor] = RELATION_CLASS_FREQ2SCAT, max_objects: int = 20) -> Optional[Dict]:
    """``dataloader.py:111-147`` for one loaded annotation dict; ``None`` when the reference's loader would return ``None``.
    ``image_hw`` = (height, width) as the reference names them (it assigns ``image.shape[0], image.shape[1]`` to ``width, height``);
    without it the empty-box-at-image-resolution test (``:123-128``) is skipped.  ``rel_reorder=None`` keeps the stored predicate ids."""
    categories = annot["categories"]
    n = int(categories.shape[0])
    if n <= 1 or n > max_objects:
        return None
    bbox = annot["bbox"]
    if image_hw is not None:
        height, width = image_hw
        raw = bbox.clone() / feature_size
        raw[:2] *= height                      # the reference scales the first two ROWS (objects), not columns - kept as is
        raw[2:] *= width
        raw = raw.ceil().int()
        if torch.any(raw[:, 1] - raw[:, 0] <= 0) or torch.any(raw[:, 3] - raw[:, 2] <= 0):
            return None
    depth = annot["image_depth"] if use_depth else torch.zeros(1, feature_size, feature_size)
    rels = []
    for rel in annot["relationships"]:
        rel = rel.clone()
        rel[rel == 12] = 4                      # wearing <- wears
        rels.append(rel_reorder[rel] if rel_reorder is not None else rel)
    return dict(image_depth=depth, categories=categories, super_categories=annot.get("super_categories"), bbox=bbox.int(),
                relationships=rels, subj_or_obj=[s.clone() for s in annot["subj_or_obj"]])


def load_annotation(path: str, **kw) -> Optional[Dict]:
    """``torch.load`` of one ``*_annotations.pkl`` + ``prepare_annotation``."""
    return prepare_annotation(torch.load(path, map_location="cpu"), **kw)


def collate(image_feature: torch.Tensor, annots: Sequence[Optional[Dict]]) -> Tuple[SceneBatch, List[int]]:
    """``utils.collate_fn`` (drop the ``None`` samples) + the tuple -> ``SceneBatch`` step.  ``image_feature`` [B,2D,F,F] holds one
    row per entry of ``annots``; returns the batch of the surviving images and their indices."""
    keep = [i for i, a in enumerate(annots) if a is not None]
    a = [annots[i] for i in keep]
    sup = [x["super_categories"] for x in a]
    batch = SceneBatch(image_feature[keep], torch.stack([x["image_depth"] for x in a]) if a else image_feature.new_zeros(0, 1, 1, 1),
                       [x["bbox"] for x in a], [x["categories"] for x in a], None if any(s is None for s in sup) else sup,
                       [x["relationships"] for x in a], [x["subj_or_obj"] for x in a], [int(x["categories"].shape[0]) for x in a])
    return batch, keep

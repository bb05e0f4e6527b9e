"""Deterministic synthetic weights and inputs for the pairwise relation path.

There are no checkpoints or Visual Genome data offline, and torch's RNG streams are not a
contract across builds, so every tensor used by the tests, ``bench.py`` and ``smoke()`` is
produced by a counter-based integer hash (``lowbias32``) over the element index.  The same
call gives bit-identical float32 values in this container (where the golden vectors were
produced from the imported reference) and on the GPU box.

Shapes follow the reference's ``state_dict`` contract (``model.py:105-136`` of the reference:
conv1_1/conv1_2 ``[D,2D+1,1,1]``, conv2_1 ``[4D,2D,3,3]``, conv3_1 ``[8D,4D,3,3]``, fc1
``[4096, 8D*(F/4)^2]``, fc2 ``[512, 4096+labels]``, fc3_1/2/3, fc4, fc5) and the input data
contract of the pair loop (``train_test.py:146-169``, ``dataloader.py:113-147``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

_CHUNK = 1 << 24


def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def hash_uniform(seed: int, n: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """``n`` float32 values in ``[lo, hi)``; element ``i`` depends only on ``(seed, i)``.  Large arrays (fc1.weight: 268 M
    values) are filled chunk by chunk on a few threads (the numpy kernels release the GIL): same values, a fraction of the time."""
    out = np.empty(n, dtype=np.float32)
    salt = np.uint32((seed * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF)

    def fill(s):
        e = min(n, s + _CHUNK)
        idx = np.arange(s, e, dtype=np.uint32)
        h = _lowbias32(idx ^ salt)
        h = _lowbias32(h + salt)
        u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))
        out[s:e] = np.float32(lo) + u * np.float32(hi - lo)

    starts = range(0, n, _CHUNK)
    if len(starts) >= 4:
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4, len(starts))) as pool:
            list(pool.map(fill, starts))
    else:
        for s in starts:
            fill(s)
    return out


def hash_normal(seed: int, n: int) -> np.ndarray:
    """Approximately N(0,1) float32 (sum of four uniforms, variance-normalised)."""
    acc = np.zeros(n, dtype=np.float32)
    for k in range(4):
        acc += hash_uniform(seed * 4 + k + 1000003, n, -1.0, 1.0)
    return acc * np.float32(math.sqrt(3.0 / 4.0))


def hash_randint(seed: int, n: int, lo: int, hi: int) -> np.ndarray:
    """``n`` int64 values in ``[lo, hi)``."""
    u = hash_uniform(seed, n, 0.0, 1.0).astype(np.float64)
    return np.minimum((u * (hi - lo)).astype(np.int64) + lo, hi - 1)


def dropout_keep_mask(seed: int, rows: int, cols: int, row0: int = 0) -> np.ndarray:
    """Host replica of the kernels' dropout keep-bit (``csrc/common.h:dropout_keep``, p = 0.5 as ``model.py:120-121``):
    element (row, col) of a ``[*, cols]`` activation is kept iff bit 16 of ``lowbias32((row*cols+col) ^ salt(seed))`` is
    set.  ``[rows, cols]`` bool for rows ``row0 .. row0+rows-1``; the parity tests hand ``2 * mask`` to the oracle."""
    salt = np.uint32((seed * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF)
    idx = (np.arange(row0 * cols, (row0 + rows) * cols, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return (((_lowbias32(idx ^ salt) >> np.uint32(16)) & np.uint32(1)) == 1).reshape(rows, cols)


@dataclass
class HeadConfig:
    """Sizes the hot path reads from ``args`` (reference ``main.py:49-85``, ``config.yaml``)."""

    dataset: str = "vg"
    hidden_dim: int = 128
    feature_size: int = 32
    num_classes: int = 150
    num_super_classes: int = 17
    num_geometric: int = 15
    num_possessive: int = 11
    num_semantic: int = 24
    hierarchical: bool = True

    @property
    def num_relations(self) -> int:
        return self.num_geometric + self.num_possessive + self.num_semantic

    @property
    def label_dim(self) -> int:
        if self.dataset == "vg":
            return 2 * (self.num_classes + self.num_super_classes)
        return 2 * self.num_classes

    def args(self, run_mode: str = "eval", fixtures: Optional[str] = None) -> dict:
        """A nested ``args`` dict with every key the hot path reads."""
        fx = fixtures or ""
        return {
            "dataset": {
                "dataset": self.dataset,
                "train_triplets": fx + "train_triplets.pt",
                "test_triplets": fx + "test_triplets.pt",
                "zero_shot_triplets": fx + "zero_shot_triplets.pt",
                "sub2super_cat_dict": fx + "sub2super_cat_dict.pt",
                "supcat_clustering": "motif",
            },
            "models": {
                "hidden_dim": self.hidden_dim,
                "feature_size": self.feature_size,
                "num_classes": self.num_classes,
                "num_super_classes": self.num_super_classes,
                "num_relations": self.num_relations,
                "num_geometric": self.num_geometric,
                "num_possessive": self.num_possessive,
                "num_semantic": self.num_semantic,
                "hierarchical_pred": self.hierarchical,
                "image_size": 1024,
                "llm_model": "gpt3.5",
                "num_img_feature": 2 * self.hidden_dim,
            },
            "training": {
                "run_mode": run_mode,
                "eval_freq": 1,
                "eval_freq_test": 1,
                "lambda_connectivity": 0.1,
                "lambda_not_connected": 1,
                "lambda_commonsense": 1,
                "lambda_contrast": 1,
                "lambda_cs_weak": 0.1,
                "lambda_cs_strong": 10,
                "learning_rate": 1e-5,
                "weight_decay": 1e-4,
            },
        }


def param_shapes(cfg: HeadConfig) -> Dict[str, Tuple[int, ...]]:
    D, F = cfg.hidden_dim, cfg.feature_size
    shapes: Dict[str, Tuple[int, ...]] = {
        "conv1_1.weight": (D, 2 * D + 1, 1, 1), "conv1_1.bias": (D,),
        "conv1_2.weight": (D, 2 * D + 1, 1, 1), "conv1_2.bias": (D,),
        "conv2_1.weight": (4 * D, 2 * D, 3, 3), "conv2_1.bias": (4 * D,),
        "conv3_1.weight": (8 * D, 4 * D, 3, 3), "conv3_1.bias": (8 * D,),
        "fc1.weight": (4096, 8 * D * (F // 4) ** 2), "fc1.bias": (4096,),
        "fc2.weight": (512, 4096 + cfg.label_dim), "fc2.bias": (512,),
    }
    if cfg.hierarchical:
        shapes.update({
            "fc3_1.weight": (cfg.num_geometric, 512), "fc3_1.bias": (cfg.num_geometric,),
            "fc3_2.weight": (cfg.num_possessive, 512), "fc3_2.bias": (cfg.num_possessive,),
            "fc3_3.weight": (cfg.num_semantic, 512), "fc3_3.bias": (cfg.num_semantic,),
            "fc4.weight": (1, 512), "fc4.bias": (1,),
            "fc5.weight": (3, 512), "fc5.bias": (3,),
        })
    else:
        shapes.update({
            "fc3.weight": (cfg.num_relations, 512), "fc3.bias": (cfg.num_relations,),
            "fc4.weight": (1, 512), "fc4.bias": (1,),
        })
    return shapes


_SD_CACHE: Dict[tuple, Dict[str, torch.Tensor]] = {}
_SD_CACHE_MAX = 3


def make_state_dict(cfg: HeadConfig, seed: int = 0, head_gain: float = 1.0,
                    trunk_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random-init weights scaled like PyTorch's default (uniform ±1/sqrt(fan_in)).

    ``head_gain`` multiplies the fc3_*/fc4/fc5 weights so the log-softmaxes are not nearly
    uniform (a trained head has O(1..10) logits); ``trunk_gain`` multiplies conv/fc weights.
    The last few results are cached (the test suites rebuild the same 1.1 GB dictionaries dozens of times); every call returns
    fresh clones, so callers may modify what they get.
    """
    key = (tuple(sorted(cfg.__dict__.items())), int(seed), float(head_gain), float(trunk_gain))
    hit = _SD_CACHE.get(key)
    if hit is None:
        hit = {}
        for k, (name, shape) in enumerate(sorted(param_shapes(cfg).items())):
            base = name.rsplit(".", 1)[0]
            wshape = param_shapes(cfg)[base + ".weight"]
            fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            gain = head_gain if base.startswith(("fc3", "fc4", "fc5")) else trunk_gain
            n = int(np.prod(shape))
            v = hash_uniform(seed * 131 + k, n, -bound * gain, bound * gain)
            hit[name] = torch.from_numpy(v.reshape(shape))
        while len(_SD_CACHE) >= _SD_CACHE_MAX:
            _SD_CACHE.pop(next(iter(_SD_CACHE)))
        _SD_CACHE[key] = hit
    else:
        _SD_CACHE[key] = _SD_CACHE.pop(key)            # most recently used last
    return {k: v.clone() for k, v in hit.items()}


@dataclass
class SceneBatch:
    """One minibatch in the reference's data contract (SURVEY §8a')."""

    image_feature: torch.Tensor            # [B, 2D, F, F] f32
    image_depth: torch.Tensor              # [B, 1, F, F] f32
    bbox: List[torch.Tensor]               # per image [n,4] int32: x0,x1,y0,y1 on the grid
    categories: List[torch.Tensor]         # per image [n] int64
    super_categories: Optional[List[List[torch.Tensor]]]  # per image, per object, 1..3 ids
    relationships: List[List[torch.Tensor]]  # per image: n-1 tensors, entry g-1 has length g
    subj_or_obj: List[List[torch.Tensor]]    # same shape, f32 in {1,0,-1}
    num_objects: List[int] = field(default_factory=list)


def default_sub2super(num_classes: int = 150, num_super: int = 17, seed: int = 7) -> Dict[int, List[int]]:
    """Synthetic stand-in for ``sub2super_cat_dict.pt`` (lists of length 1..3)."""
    ln = hash_randint(seed, num_classes, 0, 10)
    out = {}
    for c in range(num_classes):
        k = 1 if ln[c] < 7 else (2 if ln[c] < 9 else 3)
        ids = hash_randint(seed * 977 + c, k, 0, num_super).tolist()
        out[c] = [int(i) for i in ids]
    return out


_VG_PREDICATE_COUNTS = [  # reference utils.py:258-265 (class-count prior, data not code)
    47342, 1996, 3092, 3624, 3477, 9903, 41363, 3411, 251756, 13715, 96589, 712432, 1914, 9317, 22596,
    3288, 9145, 2945, 277943, 2312, 146339, 2065, 2517, 136099, 15457, 66425, 10191, 5213, 2312, 3806,
    4688, 1973, 1853, 9894, 42722, 3739, 3083, 1869, 2253, 3095, 2721, 3810, 8856, 2241, 18643, 14185,
    1925, 1740, 4613, 3490]
_OIV6_PREDICATE_COUNTS = [150983, 7665, 841, 455, 9402, 52561, 145480, 157, 175, 77, 27, 4827, 1146, 198,
                          77, 1, 12, 4, 43, 702, 8, 1111, 51, 43, 367, 10, 462, 11, 2094, 114]


def predicate_counts(cfg: HeadConfig) -> torch.Tensor:
    c = _VG_PREDICATE_COUNTS if cfg.dataset == "vg" else _OIV6_PREDICATE_COUNTS
    return torch.tensor(c[:cfg.num_relations], dtype=torch.float32)


def make_scene_batch(cfg: HeadConfig, num_objects: Sequence[int], seed: int = 0,
                     connect_frac: float = 0.02, sub2super: Optional[Dict[int, List[int]]] = None,
                     edge_boxes: bool = False) -> SceneBatch:
    """Synthetic minibatch per SURVEY §8d: N(0,1) features, U(0,1) depth, integer boxes sorted
    by area descending, uniform categories, ~``connect_frac`` of unordered pairs related with a
    predicate drawn from the class-count prior."""
    B, C, F = len(num_objects), 2 * cfg.hidden_dim, cfg.feature_size
    feat = torch.from_numpy(hash_normal(seed * 17 + 1, B * C * F * F).reshape(B, C, F, F))
    depth = torch.from_numpy(hash_uniform(seed * 17 + 2, B * F * F, 0.0, 1.0).reshape(B, 1, F, F))
    if cfg.dataset == "vg" and sub2super is None:
        sub2super = default_sub2super(cfg.num_classes, cfg.num_super_classes)
    prior = predicate_counts(cfg).double().numpy()
    cdf = np.cumsum(prior / prior.sum())
    bboxes, cats, spcats, rels, dirs = [], [], [], [], []
    for b, n in enumerate(num_objects):
        s = seed * 1009 + b * 31
        x0 = hash_randint(s + 3, n, 0, F - 4)
        y0 = hash_randint(s + 4, n, 0, F - 4)
        w = 1 + (hash_uniform(s + 5, n, 0.0, 1.0) * (F - x0)).astype(np.int64)
        h = 1 + (hash_uniform(s + 6, n, 0.0, 1.0) * (F - y0)).astype(np.int64)
        w = np.minimum(w, F - x0)
        h = np.minimum(h, F - y0)
        box = np.stack([x0, x0 + w, y0, y0 + h], axis=1)
        if edge_boxes and n >= 4:
            box[0] = [0, F, 0, F]           # full image
            box[1] = [3, 4, 5, 6]           # 1x1
            box[2] = [7, 7, 2, 9]           # zero area
            box[3] = [F - 2, F, F - 2, F]   # corner, overlaps nothing small
        order = np.argsort(-(box[:, 1] - box[:, 0]) * (box[:, 3] - box[:, 2]), kind="stable")
        box = box[order]
        bboxes.append(torch.from_numpy(box.astype(np.int32)))
        c = hash_randint(s + 7, n, 0, cfg.num_classes)
        cats.append(torch.from_numpy(c))
        if cfg.dataset == "vg":
            spcats.append([torch.tensor(sub2super[int(ci)], dtype=torch.int64) for ci in c])
        rel_i, dir_i = [], []
        for g in range(1, n):
            u = hash_uniform(s * 7 + g + 11, g, 0.0, 1.0)
            r = hash_uniform(s * 7 + g + 5003, g, 0.0, 1.0)
            d = hash_uniform(s * 7 + g + 9001, g, 0.0, 1.0)
            conn = u < connect_frac
            pred = np.searchsorted(cdf, r.astype(np.float64)).clip(0, cfg.num_relations - 1)
            rel_i.append(torch.from_numpy(np.where(conn, pred, -1).astype(np.int64)))
            dir_i.append(torch.from_numpy(np.where(conn, (d < 0.5).astype(np.float32), -1.0).astype(np.float32)))
        rels.append(rel_i)
        dirs.append(dir_i)
    return SceneBatch(feat, depth, bboxes, cats, spcats if cfg.dataset == "vg" else None, rels, dirs,
                      list(num_objects))


# ---------------------------------------------------------------------------------------------------------------- VG-like boxes
VG_IMAGE_SIZES = ((375, 500), (333, 500), (500, 375), (600, 800), (768, 1024))      # (height, width) of typical Visual Genome images


def vg_like_boxes(n: int, seed: int, median_area: float = 0.06, sigma_area: float = 1.2, sigma_aspect: float = 0.5,
                  feature_size: int = 32) -> np.ndarray:
    """[n,4] int boxes (x0,x1,y0,y1) on the feature grid, produced by the REFERENCE'S OWN box pipeline from raw pixel boxes whose
    marginals are ASSUMED (no Visual Genome annotation is available offline): area fraction log-normal around ``median_area`` with
    a heavy tail (a few near-full-image boxes: sky / building / wall), aspect ratio log-normal, position uniform, on an image of a
    typical VG size.  Pipeline, line by line: pixel box (xmin, ymin, xmax, ymax) -> ``utils.resize_boxes`` (``utils.py:38-55``:
    ``int(coordinate * new / original)`` - truncation, so small boxes can become EMPTY on the grid) -> stored as (x0, x1, y0, y1)
    (``dataset_utils.py:124-126``), objects sorted by raw area, descending (``:113-116``) -> the loader's ``bbox.int()``
    (``dataloader.py:129``).  The loader's drop rules (image dropped when a box is empty at image resolution, ``:123-128``; <= 1 or
    > 20 objects, ``:119``) are the caller's (``annotations.prepare_annotation``); empty grid boxes stay in the list exactly as the
    reference keeps them."""
    rng = np.random.default_rng(seed)
    h_img, w_img = VG_IMAGE_SIZES[int(rng.integers(len(VG_IMAGE_SIZES)))]
    area = np.clip(np.exp(rng.normal(np.log(median_area), sigma_area, n)), 0.0005, 0.98)
    asp = np.exp(rng.normal(0.0, sigma_aspect, n))
    w = np.clip(np.sqrt(area * asp), 0.01, 1.0) * w_img
    h = np.clip(np.sqrt(area / asp), 0.01, 1.0) * h_img
    xmin = rng.uniform(0, w_img - w)
    ymin = rng.uniform(0, h_img - h)
    raw = np.stack([xmin, ymin, xmin + w, ymin + h], axis=1)
    order = np.argsort(-(w * h), kind="stable")
    rh, rw = feature_size / h_img, feature_size / w_img
    out = []
    for k in order:
        b = raw[k]
        r = [int(b[0] * rw), int(b[1] * rh), int(b[2] * rw), int(b[3] * rh)]         # resize_boxes: xmin, ymin, xmax, ymax
        out.append([r[0], r[2], r[1], r[3]])                                          # x_min, x_max, y_min, y_max
    return np.asarray(out, dtype=np.int32).reshape(-1, 4)


def with_vg_like_boxes(batch: "SceneBatch", seed: int, **kw) -> "SceneBatch":
    """``batch`` with every image's boxes replaced by ``vg_like_boxes`` (same object counts; features, labels, relations untouched)."""
    batch.bbox = [torch.from_numpy(vg_like_boxes(int(b.shape[0]), seed * 7919 + i, **kw)).to(b.dtype) for i, b in enumerate(batch.bbox)]
    return batch

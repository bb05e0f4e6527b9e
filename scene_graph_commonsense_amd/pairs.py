"""Host-side pair enumeration and ragged-annotation flattening for the fused path.

The reference enumerates pairs with two nested Python loops and runs the classifier once per
(graph_iter, edge_iter, direction) on a batch of *images* (``train_test.py:189-258``).  The fused path
scores every ordered pair of every image in one pass, so the loops become index arrays, emitted in
exactly the reference's order (SURVEY §8a'): for g in 1..maxN-1, for e in 0..g-1, direction 1
(subject g, object e) then direction 2 (subject e, object g); inside a step, images in
``keep_in_batch`` order (images with more than g objects, ascending).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch


@dataclass
class PairIndex:
    """All ordered pairs of a minibatch in reference call order (host numpy arrays, length P)."""
    image: np.ndarray        # minibatch-local image index (= which_in_batch)
    g: np.ndarray            # graph_iter of the step
    e: np.ndarray            # edge_iter of the step
    first: np.ndarray        # True for direction 1 (subject = object g)
    sub: np.ndarray          # flattened object index of the subject
    obj: np.ndarray          # flattened object index of the object
    step: np.ndarray         # direction-step ordinal t (0-based) of the reference loop
    call_sizes: np.ndarray   # pairs per direction-step, in order
    obj_offset: np.ndarray   # [B+1] prefix sum of objects per image

    @property
    def n_pairs(self) -> int:
        return int(self.image.shape[0])


def enumerate_pairs(num_objects: Sequence[int]) -> PairIndex:
    """Reference call order: graph_iter g = 1.., edge_iter e < g, direction 1 then 2, and inside a direction-step the images
    with more than g objects in batch order.  One numpy block per g (the per-step Python loop cost 28 ms at 8 x 64 objects)."""
    n = np.asarray(num_objects, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(n)])
    img, gg, ee, ff, ss, oo, st, sizes = [], [], [], [], [], [], [], []
    t = 0
    for g in range(1, int(n.max()) if len(n) else 0):
        keep = np.nonzero(n > g)[0]
        k = len(keep)
        if k == 0:
            continue
        e_col = np.repeat(np.arange(g, dtype=np.int64), 2)[:, None]                  # [2g,1]: e of every direction-step
        f_col = np.tile(np.array([True, False]), g)[:, None]
        base = off[keep][None, :]
        shape = (2 * g, k)
        img.append(np.broadcast_to(keep[None, :], shape).reshape(-1)); gg.append(np.full(2 * g * k, g, dtype=np.int64))
        ee.append(np.broadcast_to(e_col, shape).reshape(-1)); ff.append(np.broadcast_to(f_col, shape).reshape(-1))
        ss.append((base + np.where(f_col, g, e_col)).reshape(-1)); oo.append((base + np.where(f_col, e_col, g)).reshape(-1))
        st.append(np.broadcast_to((t + np.arange(2 * g, dtype=np.int64))[:, None], shape).reshape(-1))
        sizes.append(np.full(2 * g, k, dtype=np.int64))
        t += 2 * g
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt))
    return PairIndex(cat(img, np.int64), cat(gg, np.int64), cat(ee, np.int64), cat(ff, bool), cat(ss, np.int32),
                     cat(oo, np.int32), cat(st, np.int64), cat(sizes, np.int64), off)


def slice_norm(i: int, F: int) -> int:
    """Python slice-bound semantics of ``mask[int(a):int(b)]`` on an axis of length F."""
    i = int(i)
    if i < 0:
        i += F
        return max(i, 0)
    return min(i, F)


def normalise_boxes(bbox: torch.Tensor, F: int) -> np.ndarray:
    """[n,4] (x0,x1,y0,y1), int or float -> int32 with int() truncation and Python slice clipping applied
    (``mask[int(y0):int(y1), int(x0):int(x1)]``, ``train_test.py:164-169``): vectorised ``slice_norm``."""
    b = bbox.detach().cpu().numpy()
    if b.dtype.kind == "f":
        b = np.trunc(b)
    b = b.astype(np.int64).reshape(-1, 4)
    return np.where(b < 0, np.maximum(b + F, 0), np.minimum(b, F)).astype(np.int32)


def super_multihot(super_categories: Optional[List[List[torch.Tensor]]], num_super: int) -> Optional[np.ndarray]:
    """Multi-hot rows in flattened object order with the reference quirk (``utils.py:136-149``): for a
    k-element list only element 0 and element k-1 are set (k = 2..4); longer lists keep element 0 only."""
    if super_categories is None:
        return None
    rows = []
    for per_img in super_categories:
        for s in per_img:
            s = [int(v) for v in (s.tolist() if hasattr(s, "tolist") else s)]
            r = np.zeros(num_super, dtype=np.float32)
            r[s[0]] += 1
            if 2 <= len(s) <= 4:
                r[s[-1]] += 1
            rows.append(r)
    return np.stack(rows) if rows else np.zeros((0, num_super), dtype=np.float32)


def pair_targets(relationships, subj_or_obj, pidx: PairIndex):
    """Directed targets per ordered pair (``train_utils.py:169-187``): predicate where the stored direction
    matches the pair's direction, else -1.  Also returns the raw (undirected) predicate row entry."""
    P = pidx.n_pairs
    directed = np.full(P, -1, dtype=np.int64)
    raw = np.full(P, -1, dtype=np.int64)
    for k in range(P):
        b, g, e = int(pidx.image[k]), int(pidx.g[k]), int(pidx.e[k])
        r = int(relationships[b][g - 1][e])
        d = float(subj_or_obj[b][g - 1][e])
        raw[k] = r
        flag = 1.0 if pidx.first[k] else 0.0
        if d == flag:
            directed[k] = r
    return directed, raw


class DeviceScene:
    """A minibatch flattened for the fused path.  Every per-pair table lives on the device and is BUILT there
    (``sgc_scene_tables``); the host pair index ``pidx`` (numpy arrays in reference call order, used by the evaluator feed and
    by tests) is derived lazily from the object counts and never touched by the training step."""

    def __init__(self, **kw):
        self._pidx = None
        self.directed = None          # [P] int32 device: directed target per pair (-1 = not connected), when targets were given
        self.raw_target = None        # [P] int32 device: stored predicate of the unordered pair (either direction)
        self.__dict__.update(kw)

    # image_feature [B,256,32,32] f32 | image_depth [B,1,32,32] f32 | obj_img [n_obj] i32 | bbox [n_obj,4] i32 (slice-normalised)
    # cats [n_obj] i64 | super_mh [n_obj,S] f32 or None | sub_idx, obj_idx, step, image [P] i32 | img_ptr [B+1] i32
    # pid [n_obj,max_n] i32 | obj_ptr [n_obj+1], sub_list, obj_list [P] i32 (CSR of the pair contraction) | step_ptr [T+1] i32
    # bbox_raw [n_obj,4] numpy as given | num_objects list | n_pairs, n_steps, max_n ints

    @property
    def pidx(self) -> PairIndex:
        if self._pidx is None:
            self._pidx = enumerate_pairs(self.num_objects)
        return self._pidx

    @property
    def sub_csr(self):
        return (self.obj_ptr, self.sub_list)

    @property
    def obj_csr(self):
        return (self.obj_ptr, self.obj_list)


class _PinnedRing:
    """A few pinned host staging buffers per device so that the one H2D copy of a minibatch's annotations is asynchronous; a
    buffer is reused only after the copy that read it has completed (which also bounds how far the host runs ahead)."""

    def __init__(self, slots=3):
        self.slots, self.bufs, self.events, self.k = slots, {}, {}, 0

    def take(self, nbytes):
        i = self.k % self.slots
        self.k += 1
        ev = self.events.get(i)
        if ev is not None:
            ev.synchronize()
        buf = self.bufs.get(i)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8).pin_memory()
            self.bufs[i] = buf
        return i, buf

    def sent(self, i):
        ev = self.events.get(i)
        if ev is None:
            ev = self.events[i] = torch.cuda.Event()
        ev.record()


_RINGS = {}


def graph_iter_offsets(n: np.ndarray):
    """goff [max_n+1]: first pair of every graph_iter block, goff[g+1] = goff[g] + 2*g*#{images with more than g objects}."""
    max_n = int(n.max()) if len(n) else 0
    g = np.arange(max(max_n, 1), dtype=np.int64)
    k = (n[None, :] > g[:, None]).sum(1) if len(n) else np.zeros(1, dtype=np.int64)
    goff = np.zeros(max(max_n, 1) + 1, dtype=np.int64)
    goff[1:] = np.cumsum(2 * g * k)
    return goff, max_n


def object_window_rects(bb: np.ndarray) -> np.ndarray:
    """[n,4] normalised boxes (x0,x1,y0,y1 on the 32-grid) -> [n,4] half-open rectangles (x0,x1,y0,y1) on the 8x8 grid of conv3's
    pooling windows outside which the object cannot influence a pair's conv3 output (host replica of
    ``csrc/kernels_shared.hip:object_windows``; zero rectangle for an empty box)."""
    bb = np.asarray(bb, dtype=np.int64).reshape(-1, 4)
    out = np.zeros_like(bb)
    for a in (0, 2):
        lo, hi = np.clip(bb[:, a], 0, None), np.clip(bb[:, a + 1], None, 32)
        ok = hi > lo
        lo, hi = np.maximum(lo - 1, 0), np.minimum(hi + 1, 32)
        lo, hi = lo >> 1, (hi + 1) >> 1
        lo, hi = np.maximum(lo - 1, 0), np.minimum(hi + 1, 16)
        out[:, a], out[:, a + 1] = np.where(ok, lo >> 1, 0), np.where(ok, (hi + 1) >> 1, 0)
    empty = (out[:, 1] <= out[:, 0]) | (out[:, 3] <= out[:, 2])
    out[empty] = 0
    return out


def count_shared_windows(bb: np.ndarray, img_ptr) -> int:
    """Number of pair-specific (X) windows over all ordered pairs of every image: sum of |R_i n R_j|."""
    r = object_window_rects(bb)
    total = 0
    for b in range(len(img_ptr) - 1):
        q = r[int(img_ptr[b]):int(img_ptr[b + 1])]
        if len(q) < 2:
            continue
        ox = np.clip(np.minimum(q[:, None, 1], q[None, :, 1]) - np.maximum(q[:, None, 0], q[None, :, 0]), 0, None)
        oy = np.clip(np.minimum(q[:, None, 3], q[None, :, 3]) - np.maximum(q[:, None, 2], q[None, :, 2]), 0, None)
        a = ox * oy
        total += int(a.sum() - np.trace(a))
    return total


def object_d16_rects(bb: np.ndarray) -> np.ndarray:
    """[n,4] normalised boxes -> [n,4] half-open rectangles (x0,x1,y0,y1) on the 16-grid of conv3's INPUT where the object's conv2
    half differs from the background's: the box, +-1 pixel (conv2_1 is 3x3), pooled 2x2 (host replica of
    ``csrc/kernels_shared.hip:axis_d16``; zero rectangle for an empty box)."""
    bb = np.asarray(bb, dtype=np.int64).reshape(-1, 4)
    out = np.zeros_like(bb)
    for a in (0, 2):
        lo, hi = np.clip(bb[:, a], 0, None), np.clip(bb[:, a + 1], None, 32)
        ok = hi > lo
        lo, hi = np.maximum(lo - 1, 0), np.minimum(hi + 1, 32)
        out[:, a], out[:, a + 1] = np.where(ok, lo >> 1, 0), np.where(ok, (hi + 1) >> 1, 0)
    return out


def count_conv2_windows(bb: np.ndarray) -> int:
    """2x2-pixel windows of the 32-grid on which an object's conv2 half can differ from the background's: sum of the D16 areas."""
    d = object_d16_rects(bb)
    return int(((d[:, 1] - d[:, 0]) * (d[:, 3] - d[:, 2])).sum())


def count_linear_windows(bb: np.ndarray, img_ptr) -> int:
    """X windows of the LINEAR pairs (``csrc/kernels_shared.hip``, sixth identity): ordered pairs whose window rectangles intersect
    while their 16-grid regions of influence do not - their conv3 pre-activation on those windows is the sum of per-object ones."""
    r, d = object_window_rects(bb), object_d16_rects(bb)
    total = 0
    for b in range(len(img_ptr) - 1):
        q, e = r[int(img_ptr[b]):int(img_ptr[b + 1])], d[int(img_ptr[b]):int(img_ptr[b + 1])]
        if len(q) < 2:
            continue
        ox = np.clip(np.minimum(q[:, None, 1], q[None, :, 1]) - np.maximum(q[:, None, 0], q[None, :, 0]), 0, None)
        oy = np.clip(np.minimum(q[:, None, 3], q[None, :, 3]) - np.maximum(q[:, None, 2], q[None, :, 2]), 0, None)
        a = ox * oy
        meet = (np.minimum(e[:, None, 1], e[None, :, 1]) > np.maximum(e[:, None, 0], e[None, :, 0])) & \
               (np.minimum(e[:, None, 3], e[None, :, 3]) > np.maximum(e[:, None, 2], e[None, :, 2]))
        np.fill_diagonal(a, 0)
        total += int(a[~meet].sum())
    return total


def count_object_windows(bb: np.ndarray) -> int:
    """Windows of the pseudo-pairs (o, background) and (background, o) that are computed per object (second level of sharing):
    2 * sum |R_o|."""
    r = object_window_rects(bb)
    return int(2 * ((r[:, 1] - r[:, 0]) * (r[:, 3] - r[:, 2])).sum())


def window_entry_counts(bb: np.ndarray, img_ptr) -> np.ndarray:
    """[64] number of pair-specific (X) entries per pooling window over all ordered pairs of every image: a window inside the
    rectangles of c objects of an image is an X window of c*(c-1) ordered pairs."""
    r = object_window_rects(bb)
    out = np.zeros(64, dtype=np.int64)
    wy, wx = np.divmod(np.arange(64), 8)
    for b in range(len(img_ptr) - 1):
        q = r[int(img_ptr[b]):int(img_ptr[b + 1])]
        if len(q) < 2:
            continue
        inside = (wx[None] >= q[:, 0:1]) & (wx[None] < q[:, 1:2]) & (wy[None] >= q[:, 2:3]) & (wy[None] < q[:, 3:4])
        c = inside.sum(0)
        out += c * (c - 1)
    return out


def window_major_layout(entry_counts: np.ndarray, n_pseudo: int):
    """Group offsets of the window-major row space (``csrc/kernels_shared.hip``): group w = n_pseudo per-object rows + its X entries,
    padded to a multiple of 256.  Returns (goff [65] int32, tile_group [rows/256] int32)."""
    size = (n_pseudo + np.asarray(entry_counts, dtype=np.int64) + 255) // 256 * 256
    goff = np.concatenate([[0], np.cumsum(size)]).astype(np.int32)
    tile_group = np.repeat(np.arange(64, dtype=np.int32), (size // 256).astype(np.int64))
    return goff, tile_group


def flatten_scene(cfg, batch, device) -> DeviceScene:
    """SceneBatch (reference data contract) -> DeviceScene with all ordered pairs in reference order.
    Host work is O(objects): concatenating the ragged annotation lists into one pinned staging buffer; one asynchronous
    H2D copy; the O(pairs) tables (indices, steps, CSR lists, directed targets) are written by ``sgc_scene_tables``."""
    from . import _lib
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("flatten_scene builds the pair tables with a HIP kernel: it needs a GPU device (no CPU fallback)")
    lib = _lib.load()
    n_list = [int(b.shape[0]) for b in batch.bbox]
    B = len(n_list)
    n = np.asarray(n_list, dtype=np.int64)
    n_obj = int(n.sum())
    goff, max_n = graph_iter_offsets(n)
    P = int(goff[max_n]) if max_n > 0 else 0
    T = max_n * (max_n - 1) if max_n > 1 else 0
    img_ptr = np.concatenate([[0], np.cumsum(n)])
    F = cfg.feature_size
    raw = np.concatenate([b.detach().cpu().numpy() for b in batch.bbox]) if B else np.zeros((0, 4))
    bb = normalise_boxes(torch.from_numpy(raw), F) if n_obj else np.zeros((0, 4), dtype=np.int32)
    cats = torch.cat([c.reshape(-1) for c in batch.categories]).to(torch.int64).cpu().numpy() if n_obj else np.zeros(0, dtype=np.int64)
    mh = super_multihot(batch.super_categories, cfg.num_super_classes) if cfg.dataset == "vg" else None
    rel = dirs = None
    if getattr(batch, "relationships", None) is not None and getattr(batch, "subj_or_obj", None) is not None:
        cat_i = lambda rows, dt: (torch.cat([torch.as_tensor(r).reshape(-1) for r in rows]).to(dt) if len(rows) else torch.zeros(0, dtype=dt))
        rel = torch.cat([cat_i(r, torch.int32) for r in batch.relationships]).cpu().numpy()
        dirs = torch.cat([cat_i(r, torch.float32) for r in batch.subj_or_obj]).cpu().numpy()
        if rel.shape[0] != int((n * (n - 1) // 2).sum()) or dirs.shape[0] != rel.shape[0]:
            raise ValueError("relationships / subj_or_obj must hold g entries for graph_iter g of every image (dataloader.py:144-147)")

    # ---- one staging buffer, one H2D copy
    parts = [("hdr", np.concatenate([n, img_ptr, goff]).astype(np.int32)), ("bbox", bb.reshape(-1)), ("cats", cats)]
    if mh is not None:
        parts.append(("mh", mh.reshape(-1).astype(np.float32)))
    if rel is not None:
        parts += [("rel", rel.astype(np.int32)), ("dir", dirs.astype(np.float32))]
    offs, total = {}, 0
    for name, a in parts:
        offs[name] = total
        total += (a.nbytes + 15) // 16 * 16
    ring = _RINGS.setdefault(dev, _PinnedRing())
    with torch.cuda.device(dev):
        slot, host = ring.take(total)
        hv = host.numpy()
        for name, a in parts:
            hv[offs[name]:offs[name] + a.nbytes] = a.view(np.uint8).reshape(-1)
        stage = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
        stage[:total].copy_(host[:total], non_blocking=True)
        ring.sent(slot)
    view = lambda name, a, dt: stage[offs[name]:offs[name] + a.nbytes].view(dt)
    hdr = view("hdr", parts[0][1], torch.int32)
    n_d, img_ptr_d, goff_d = hdr[:B], hdr[B:2 * B + 1], hdr[2 * B + 1:]
    bbox_d = view("bbox", parts[1][1], torch.int32).view(-1, 4)
    cats_d = view("cats", cats, torch.int64)
    mh_d = view("mh", parts[3][1], torch.float32).view(n_obj, -1) if mh is not None else None
    rel_d = view("rel", rel.astype(np.int32), torch.int32) if rel is not None else None
    dir_d = view("dir", dirs.astype(np.float32), torch.float32) if rel is not None else None

    # ---- O(pairs) tables on the device
    pid_ld = max(max_n, 1)
    sizes = dict(sub_idx=P, obj_idx=P, step=P, image=P, directed=P, raw=P, sub_list=P, obj_list=P, pid=max(n_obj, 1) * pid_ld,
                 obj_ptr=n_obj + 1, obj_img=max(n_obj, 1), step_ptr=T + 1)
    tab = torch.empty(sum(sizes.values()) + 4 * len(sizes), dtype=torch.int32, device=dev)
    t, o = {}, 0
    for k_, sz in sizes.items():
        t[k_] = tab[o:o + sz]
        o += (sz + 3) // 4 * 4
    if n_obj > 0:
        _lib.check(lib.sgc_scene_tables(_lib.ptr(n_d), _lib.ptr(img_ptr_d), _lib.ptr(goff_d), B, max_n, P, n_obj, pid_ld,
                                        _lib.ptr(rel_d), _lib.ptr(dir_d), _lib.ptr(t["sub_idx"]), _lib.ptr(t["obj_idx"]), _lib.ptr(t["step"]),
                                        _lib.ptr(t["image"]), _lib.ptr(t["directed"]), _lib.ptr(t["raw"]), _lib.ptr(t["pid"]),
                                        _lib.ptr(t["obj_ptr"]), _lib.ptr(t["sub_list"]), _lib.ptr(t["obj_list"]), _lib.ptr(t["obj_img"]),
                                        _lib.ptr(t["step_ptr"]), _lib.stream_ptr()), "sgc_scene_tables")
    return DeviceScene(image_feature=batch.image_feature.to(dev, torch.float32).contiguous(),
                       image_depth=batch.image_depth.to(dev, torch.float32).contiguous(),
                       obj_img=t["obj_img"][:n_obj], bbox=bbox_d, cats=cats_d, super_mh=mh_d, sub_idx=t["sub_idx"], obj_idx=t["obj_idx"],
                       step=t["step"], image=t["image"], directed=t["directed"] if rel is not None else None,
                       raw_target=t["raw"] if rel is not None else None, img_ptr=img_ptr_d, pid=t["pid"].view(max(n_obj, 1), pid_ld),
                       obj_ptr=t["obj_ptr"], sub_list=t["sub_list"], obj_list=t["obj_list"], step_ptr=t["step_ptr"], bbox_raw=raw,
                       num_objects=n_list, n_pairs=P, n_steps=T, max_n=int(max_n), _stage=stage, _tables=tab,
                       shared_windows=count_shared_windows(bb, img_ptr) if (n_obj and F == 32) else None,
                       window_entries=window_entry_counts(bb, img_ptr) if (n_obj and F == 32) else None,
                       object_windows=count_object_windows(bb) if (n_obj and F == 32) else None,
                       linear_windows=count_linear_windows(bb, img_ptr) if (n_obj and F == 32) else None,
                       conv2_windows=count_conv2_windows(bb) if (n_obj and F == 32) else None,
                       _rel_src=getattr(batch, "relationships", None) if rel is not None else None)


def pair_targets_fast(relationships, subj_or_obj, pidx: PairIndex) -> np.ndarray:
    """Vectorised ``pair_targets`` (directed targets only)."""
    n = np.diff(pidx.obj_offset).astype(np.int64)
    sq_off = np.concatenate([[0], np.cumsum(n * n)])
    rel = np.full(int(sq_off[-1]), -1, dtype=np.int64)
    dirs = np.full(int(sq_off[-1]), -1.0, dtype=np.float32)
    for b in range(len(n)):
        nb = int(n[b])
        for g in range(1, nb):
            base = int(sq_off[b]) + g * nb
            rel[base:base + g] = np.asarray(relationships[b][g - 1])
            dirs[base:base + g] = np.asarray(subj_or_obj[b][g - 1])
    idx = sq_off[pidx.image] + pidx.g * n[pidx.image] + pidx.e
    flag = np.where(pidx.first, 1.0, 0.0).astype(np.float32)
    return np.where(dirs[idx] == flag, rel[idx], -1).astype(np.int64)


def match_target_sgd(relationships, subj_or_obj, categories_target, bbox_target):
    """``utils.match_target_sgd`` (utils.py:294-350) without the per-element tensor appends: the ground-truth triplets of every
    image as (cat_subject, cat_object, bbox_subject [t,4], bbox_object [t,4], relation) lists, ``None`` where an image has none.
    The reference's loop bound is kept: ``graph_iter`` runs over ``range(len(relationships[i]))`` = 0..n-2, so relations whose
    graph object is the last object of an image are not collected."""
    out = ([], [], [], [], [])
    for b in range(len(relationships)):
        s_idx, o_idx, rel = [], [], []
        for g in range(len(relationships[b])):
            if g == 0:
                continue
            flag = np.asarray(subj_or_obj[b][g - 1])[:g]
            r = np.asarray(relationships[b][g - 1])[:g]
            for e in np.nonzero((flag == 1) | (flag == 0))[0]:
                s_i, o_i = (g, int(e)) if flag[e] == 1 else (int(e), g)
                s_idx.append(s_i); o_idx.append(o_i); rel.append(int(r[e]))
        if rel:
            cats, box = torch.as_tensor(categories_target[b]).reshape(-1), torch.as_tensor(bbox_target[b]).reshape(-1, 4)
            si, oi = torch.as_tensor(s_idx), torch.as_tensor(o_idx)
            vals = (cats[si], cats[oi], box[si], box[oi], torch.as_tensor(rel, dtype=torch.int64))
        else:
            vals = (None, None, None, None, None)
        for lst, v in zip(out, vals):
            lst.append(v)
    return out

"""The relation head for sizes OTHER than the reference's defaults (``model.py:110-111`` accepts any ``input_dim`` /
``feature_size``; every shipped configuration uses 128 / 32, which is what the tiled MFMA kernels of ``engine.RelHeadEngine`` are built
for).  ``GenericTrunkEngine`` keeps the whole engine interface - ``forward_pairs``, ``train_forward`` / ``train_backward``, the per-step
``forward()`` node, chunking, the contrastive / commonsense terms, evaluators - and replaces only the TRUNK (conv1 .. fc1,
``model.py:138-150``) by the plain-f32 per-pair kernels of ``csrc/kernels_generic.hip`` (the reference's graph literally: no sharing
identities, no 16-bit copies).  fc2 with the label gather, the head, the loss and their backward do not depend on the two sizes and run
on the ordinary kernels.  Not tuned: this path exists so that a module built with non-default sizes RUNS on the GPU (unit-test sized
configurations such as ``input_dim=16, feature_size=8``) instead of raising; there is still no CPU fallback."""
from __future__ import annotations

import ctypes
from typing import Dict

import torch

from . import _lib
from .engine import PairOutputs, RelHeadEngine, TrainContext, Workspace
from .synthetic import HeadConfig


class GenericTrunkEngine(RelHeadEngine):

    def _check_sizes(self, cfg: HeadConfig):
        if cfg.feature_size % 4 != 0 or cfg.feature_size <= 0 or cfg.hidden_dim <= 0:
            raise NotImplementedError("feature_size must be a positive multiple of 4 (two 2x2 poolings, model.py:144,147)")
        if cfg.num_relations + 4 > 64:
            raise NotImplementedError("head kernel holds one output row per wavefront lane (<= 60 relations)")

    # ------------------------------------------------------------------ weights: the f32 masters in the reference's own layouts
    def _load_trunk_weights(self, sd, g, fc1_sync):
        w, C = self.w, self.cfg.hidden_dim
        w["g_w1"] = torch.stack([g("conv1_1.weight").reshape(C, 2 * C + 1), g("conv1_2.weight").reshape(C, 2 * C + 1)]).contiguous()
        w["g_b1"] = torch.stack([g("conv1_1.bias"), g("conv1_2.bias")]).contiguous()
        w["g_w2"], w["g_b2"] = g("conv2_1.weight").contiguous(), g("conv2_1.bias").contiguous()
        w["g_w3"], w["g_b3"] = g("conv3_1.weight").contiguous(), g("conv3_1.bias").contiguous()
        if fc1_sync is not None:
            fc1_sync()
        w["g_wf1"], w["bf1"] = g("fc1.weight").contiguous(), g("fc1.bias").contiguous()

    def _prep_bwd_trunk_weights(self, sd, g, fc1_sync):
        pass                                    # the backward reads the same f32 weights

    # ------------------------------------------------------------------ trunk
    def _sides(self, image_feature, image_depth, obj_img, bbox, sub_idx, obj_idx, role_inputs):
        """What the trunk kernels read: (feat, depth, stride_feat, stride_depth, img [2,P], box [2,P,4])."""
        C, F, dev = self.cfg.hidden_dim, self.cfg.feature_size, self.device
        P = int(sub_idx.shape[0])
        if role_inputs is None:
            feat = image_feature.to(dev, torch.float32).contiguous()
            depth = image_depth.to(dev, torch.float32).contiguous()
            if tuple(feat.shape[1:]) != (2 * C, F, F) or tuple(depth.shape[1:]) != (1, F, F):
                raise ValueError("image_feature / image_depth must be [B,%d,%d,%d] / [B,1,%d,%d]" % (2 * C, F, F, F, F))
            s, o = sub_idx.long(), obj_idx.long()
            img = torch.stack([obj_img.long()[s], obj_img.long()[o]]).to(torch.int32).contiguous()
            box = torch.stack([bbox[s], bbox[o]]).to(torch.int32).contiguous()
            return [(feat, depth, 2 * C * F * F, F * F)], img, box
        # per-step forward(): row k of h_sub / h_obj is the pre-masked crop of pair k; the two sides read different tensors
        hs, ho = (h.to(dev, torch.float32).contiguous() for h in role_inputs)
        if tuple(hs.shape[1:]) != (2 * C + 1, F, F) or hs.shape != ho.shape:
            raise ValueError("h_sub / h_obj must be [b,%d,%d,%d]" % (2 * C + 1, F, F))
        ids = torch.arange(P, dtype=torch.int32, device=dev)
        full = torch.tensor([0, F, 0, F], dtype=torch.int32, device=dev).repeat(2, P, 1).contiguous()
        st = (2 * C + 1) * F * F
        both = torch.cat([hs, ho], dim=0)                 # one tensor, image index p for the subject side, P + p for the object side
        img = torch.stack([ids, ids + P]).contiguous()
        return [(both, both.view(-1)[2 * C * F * F:], st, st)], img, full

    def _trunk_forward(self, src, img, box, P, Ppad, dropout, seed, keep: bool):
        """conv1 .. fc1 for ``P`` pairs; returns (h1 f16 [Ppad*4096], saved activations for the backward or None)."""
        lib, w, cfg = self.lib, self.w, self.cfg
        C, F = cfg.hidden_dim, cfg.feature_size
        ws = self.ws if keep else self.scratch
        st = self._st
        feat, depth, sf, sd = src[0]
        a = ws.get("g_a", P * F * F * 2 * C, torch.float32)
        _lib.check(lib.sgc_generic_conv1_tanh(_lib.ptr(feat), _lib.ptr(depth), ctypes.c_long(sf), ctypes.c_long(sd), _lib.ptr(img), _lib.ptr(box),
                                              _lib.ptr(w["g_w1"]), _lib.ptr(w["g_b1"]), P, C, F, _lib.ptr(a), st()), "sgc_generic_conv1_tanh")
        H, Q = F // 2, (F // 4) ** 2
        z = ws.get("g_z", P * H * H * 4 * C, torch.float32)
        cz = ws.get("g_cz", P * H * H * 4 * C, torch.uint8)
        _lib.check(lib.sgc_generic_conv3x3_relu_pool(_lib.ptr(a), _lib.ptr(w["g_w2"]), _lib.ptr(w["g_b2"]), P, F, 2 * C, 4 * C, _lib.ptr(z), _lib.ptr(cz),
                                                     st()), "sgc_generic_conv3x3_relu_pool")
        y = ws.get("g_y", P * Q * 8 * C, torch.float32)
        cy = ws.get("g_cy", P * Q * 8 * C, torch.uint8)
        _lib.check(lib.sgc_generic_conv3x3_relu_pool(_lib.ptr(z), _lib.ptr(w["g_w3"]), _lib.ptr(w["g_b3"]), P, H, 4 * C, 8 * C, _lib.ptr(y), _lib.ptr(cy),
                                                     st()), "sgc_generic_conv3x3_relu_pool")
        h1 = ws.get("h1", Ppad * 4096, torch.float16)
        if Ppad > P:
            Workspace._zero(h1[P * 4096:])
        _lib.check(lib.sgc_generic_fc1_relu(_lib.ptr(y), _lib.ptr(w["g_wf1"]), _lib.ptr(w["bf1"]), P, Q, 8 * C, int(dropout), ctypes.c_uint(seed),
                                            _lib.ptr(h1), st()), "sgc_generic_fc1_relu")
        return h1, (dict(src=src, img=img, box=box, a=a, z=z, cz=cz, y=y, cy=cy) if keep else None)

    def _fc2_head(self, h1, lsub, lobj, sub_idx, obj_idx, P, Ppad, train, seed, iou_mask=None, keep=False):
        p = (self.ws if keep else self.scratch).get("p", Ppad * 512, torch.float32)
        _lib.check(self.lib.sgc_fc2_labels_relu(_lib.ptr(h1), _lib.ptr(self.w["w2m"]), _lib.ptr(self.w["bf2"]), _lib.ptr(lsub), _lib.ptr(lobj),
                                                _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(p), P, int(train), ctypes.c_uint(seed), self._st()),
                   "sgc_fc2_labels_relu")
        return p, self.head(p, P, iou_mask)

    # ------------------------------------------------------------------ the engine interface
    def forward_pairs(self, image_feature, image_depth, obj_img, bbox, cats, super_mh, sub_idx, obj_idx, train=False, seeds=(0, 0),
                      keep_argmax=False, iou_mask=None, dense=None, select=None, shared_windows=None, pair_order=None) -> PairOutputs:
        P = int(sub_idx.shape[0])
        Ppad = (P + 63) // 64 * 64
        src, img, box = self._sides(image_feature, image_depth, obj_img, bbox, sub_idx, obj_idx, None)
        lsub, lobj = self.label_vectors(cats, super_mh)
        h1, _ = self._trunk_forward(src, img, box, P, Ppad, train, seeds[0], keep=False)
        _, out = self._fc2_head(h1, lsub, lobj, sub_idx, obj_idx, P, Ppad, train, seeds[1], iou_mask)
        if select is not None:                     # every pair is computed (tiny shapes); the unselected ones are blanked as the tiled path does
            drop = ~select.bool()
            out.relation[drop] = 0
            if out.super_relation is not None:
                out.super_relation[drop] = 0
            out.connectivity[drop] = 0
            hidden = out.hidden.clone()
            hidden[drop] = 0
            out.cand_conf[drop] = -float("inf")
            out.cand_pred[drop] = 0
            return PairOutputs(out.relation, out.super_relation, out.connectivity, hidden, out.cand_conf, out.cand_pred)
        return out

    def compat_forward(self, hs, ho, c1, c2, mh1, mh2, train=False, seeds=(0, 0)) -> PairOutputs:
        b = int(hs.shape[0])
        Ppad = (b + 63) // 64 * 64
        ids = torch.arange(b, dtype=torch.int32, device=self.device)
        src, img, box = self._sides(None, None, None, None, ids, ids, (hs, ho))
        lsub, _ = self.label_vectors(c1, mh1)
        _, lobj = self.label_vectors(c2, mh2)
        h1, _ = self._trunk_forward(src, img, box, b, Ppad, train, seeds[0], keep=False)
        return self._fc2_head(h1, lsub, lobj, ids, ids, b, Ppad, train, seeds[1])[1]

    def train_forward(self, image_feature, image_depth, obj_img, bbox, cats, super_mh, sub_idx, obj_idx, seeds=(0, 0), dropout=True, dense=None,
                      role_inputs=None, cats_obj=None, super_mh_obj=None, shared_windows=None, pair_order=None) -> TrainContext:
        ctx = TrainContext()
        ctx.n_obj = int(obj_img.shape[0])
        ctx.P = P = int(sub_idx.shape[0])
        ctx.Ppad = Ppad = (P + 63) // 64 * 64
        ctx.obj_img, ctx.bbox, ctx.sub_idx, ctx.obj_idx = obj_img, bbox, sub_idx, obj_idx
        ctx.cats = (cats, cats if cats_obj is None else cats_obj)
        ctx.super_mh = (super_mh, super_mh if role_inputs is None else super_mh_obj)
        ctx.dropout, ctx.seeds = dropout, seeds
        ctx.n_img = int(image_feature.shape[0]) if role_inputs is None else int(role_inputs[0].shape[0])
        ctx.shared = None
        src, img, box = self._sides(image_feature, image_depth, obj_img, bbox, sub_idx, obj_idx, role_inputs)
        ctx.lsub, lobj_same = self.label_vectors(ctx.cats[0], ctx.super_mh[0])
        ctx.lobj = lobj_same if role_inputs is None else self.label_vectors(ctx.cats[1], ctx.super_mh[1])[1]
        ctx.h1, ctx.generic = self._trunk_forward(src, img, box, P, Ppad, dropout, seeds[0], keep=True)
        ctx.p, ctx.out = self._fc2_head(ctx.h1, ctx.lsub, ctx.lobj, sub_idx, obj_idx, P, Ppad, dropout, seeds[1], keep=True)
        return ctx

    def _generic_trunk_backward(self, ctx, dh1, grads: Dict[str, torch.Tensor], side):
        """From dh1 (bf16, gradient wrt fc1's pre-activation) to the gradients of fc1, conv3, conv2, conv1 (``train_backward`` has done
        the head and fc2).  Everything on the caller's stream: the shapes this path serves are launch-bound."""
        lib, w, cfg, dev, st = self.lib, self.w, self.cfg, self.device, self._st
        C, F, P = cfg.hidden_dim, cfg.feature_size, ctx.P
        H, Q = F // 2, (F // 4) ** 2
        g = ctx.generic
        sc = self.scratch
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        dy = sc.get("g_dy", P * Q * 8 * C, torch.float32)
        dwf1, dbf1 = f32(4096, 8 * C * Q), f32(4096)
        _lib.check(lib.sgc_generic_fc1_bwd(_lib.ptr(dh1), _lib.ptr(g["y"]), _lib.ptr(w["g_wf1"]), P, Q, 8 * C, _lib.ptr(dy), _lib.ptr(dwf1), _lib.ptr(dbf1),
                                           st()), "sgc_generic_fc1_bwd")
        grads["fc1.weight"], grads["fc1.bias"] = dwf1, dbf1
        dz = sc.get("g_dz", P * H * H * 4 * C, torch.float32)
        dpre3 = sc.get("g_dpre3", P * H * H * 8 * C, torch.float32)
        dw3, db3 = f32(8 * C, 4 * C, 3, 3), f32(8 * C)
        _lib.check(lib.sgc_generic_conv3x3_bwd(_lib.ptr(g["z"]), _lib.ptr(w["g_w3"]), _lib.ptr(dy), _lib.ptr(g["cy"]), P, H, 4 * C, 8 * C, _lib.ptr(dpre3),
                                               _lib.ptr(dz), _lib.ptr(dw3), _lib.ptr(db3), st()), "sgc_generic_conv3x3_bwd")
        grads["conv3_1.weight"], grads["conv3_1.bias"] = dw3, db3
        da = sc.get("g_da", P * F * F * 2 * C, torch.float32)
        dpre2 = sc.get("g_dpre2", P * F * F * 4 * C, torch.float32)
        dw2, db2 = f32(4 * C, 2 * C, 3, 3), f32(4 * C)
        _lib.check(lib.sgc_generic_conv3x3_bwd(_lib.ptr(g["a"]), _lib.ptr(w["g_w2"]), _lib.ptr(dz), _lib.ptr(g["cz"]), P, F, 2 * C, 4 * C, _lib.ptr(dpre2),
                                               _lib.ptr(da), _lib.ptr(dw2), _lib.ptr(db2), st()), "sgc_generic_conv3x3_bwd")
        grads["conv2_1.weight"], grads["conv2_1.bias"] = dw2, db2
        feat, depth, sf, sd = g["src"][0]
        dw1, db1 = f32(2, C, 2 * C + 1), f32(2, C)
        _lib.check(lib.sgc_generic_conv1_bwd(_lib.ptr(feat), _lib.ptr(depth), ctypes.c_long(sf), ctypes.c_long(sd), _lib.ptr(g["img"]), _lib.ptr(g["box"]),
                                             _lib.ptr(g["a"]), _lib.ptr(da), P, C, F, _lib.ptr(dw1), _lib.ptr(db1), st()), "sgc_generic_conv1_bwd")
        for r, nm in enumerate(("conv1_1", "conv1_2")):
            grads[nm + ".weight"] = dw1[r].reshape(C, 2 * C + 1, 1, 1)
            grads[nm + ".bias"] = db1[r]

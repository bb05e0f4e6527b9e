"""Host orchestration of the fused pairwise relation path on one MI355X.

All compute is in the HIP kernels of ``csrc/`` reached through the C-ABI of ``include/sgc_relhead.h``;
torch supplies device memory, the current HIP stream and (elsewhere) torch.distributed.  The restructuring
identities are SURVEY §7.1:

  conv1 is 1x1 and the bbox mask is per pixel   -> conv1 runs once per image and role (``sgc_conv1_tanh``)
  conv2 is linear over a channel concat          -> U_i + V_j from per-object convs (``sgc_conv2_object``)
  everything after that ReLU is per pair         -> expansion, conv3, fc1, fc2, head over all pairs at once
  one-hot label concat                           -> per-object 512-vectors gathered in fc2's epilogue
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .synthetic import HeadConfig

ELEM_F16, ELEM_BF16 = 0, 1
XC = 384            # packed input channels (2*128+1 = 257 zero-padded to a multiple of the K tile)


def _c_long(v):
    return ctypes.c_long(int(v))


@dataclass
class PairOutputs:
    relation: torch.Tensor                  # [P, R] log-probs (hier) or raw logits (flat)
    super_relation: Optional[torch.Tensor]  # [P, 3]
    connectivity: torch.Tensor              # [P] raw logit
    hidden: torch.Tensor                    # [P, 512] post-ReLU (post-dropout) fc2 output
    cand_conf: torch.Tensor                 # [P, 3] (hier) or [P, 1]
    cand_pred: torch.Tensor                 # [P, 3] int32 / [P, 1]


class Workspace:
    """Grow-only cache of device buffers keyed by name (no allocation inside the steady-state step)."""

    def __init__(self, device):
        self.device = device
        self.bufs: Dict[str, torch.Tensor] = {}

    def get(self, name, numel, dtype, zero=False):
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.zeros(int(numel), dtype=dtype, device=self.device)
            self.bufs[name] = t
        elif zero:
            t[:numel].zero_()
        return t[:numel]

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


class RelHeadEngine:
    """Forward (and, in ``engine_bwd``, backward) of the relation head over explicit pair lists."""

    def __init__(self, cfg: HeadConfig, device="cuda:0"):
        if cfg.hidden_dim != 128 or cfg.feature_size != 32:
            raise NotImplementedError("the gfx950 kernels are specialised to hidden_dim=128, feature_size=32")
        if cfg.num_relations + 4 > 64:
            raise NotImplementedError("head kernel holds one output row per wavefront lane (<= 60 relations)")
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.ws = Workspace(self.device)
        self.w: Dict[str, torch.Tensor] = {}
        self.T = (1.0, 1.0, 1.0)

    # ------------------------------------------------------------------ weights
    def load_weights(self, sd: Dict[str, torch.Tensor]):
        """Build the 16-bit compute copies (layouts of csrc/kernels_fwd.hip) from the f32 master weights."""
        cfg, dev = self.cfg, self.device
        g = lambda k: sd[k].detach().to(dev, torch.float32)
        w = self.w
        w1r = torch.zeros(2, 128, XC, dtype=torch.float16, device=dev)
        w1r[0, :, :257] = g("conv1_1.weight").view(128, 257).half()
        w1r[1, :, :257] = g("conv1_2.weight").view(128, 257).half()
        w["w1r"] = w1r
        w["b1"] = torch.stack([g("conv1_1.bias"), g("conv1_2.bias")]).contiguous()
        w["cst"] = torch.tanh(w["b1"]).half().contiguous()                  # tanh(conv1(0)) outside the box
        c2 = g("conv2_1.weight")
        w["w2r"] = torch.stack([c2[:, r * 128:(r + 1) * 128].permute(0, 2, 3, 1).reshape(512, 1152) for r in (0, 1)]
                               ).half().contiguous()
        w["b2"] = g("conv2_1.bias").contiguous()
        w["w3r"] = g("conv3_1.weight").permute(0, 2, 3, 1).reshape(1024, 4608).half().contiguous()
        w["b3"] = g("conv3_1.bias").contiguous()
        w["w1p"] = g("fc1.weight").view(4096, 1024, 64).permute(0, 2, 1).reshape(4096, 65536).half().contiguous()
        w["bf1"] = g("fc1.bias").contiguous()
        fc2 = g("fc2.weight")
        w["fc2_full"] = fc2
        w["w2m"] = fc2[:, :4096].half().contiguous()
        w["bf2"] = g("fc2.bias").contiguous()
        R = cfg.num_relations
        if cfg.hierarchical:
            rows = [g("fc3_1.weight"), g("fc3_2.weight"), g("fc3_3.weight"), g("fc5.weight"), g("fc4.weight")]
            bias = [g("fc3_1.bias"), g("fc3_2.bias"), g("fc3_3.bias"), g("fc5.bias"), g("fc4.bias")]
        else:
            rows = [g("fc3.weight"), g("fc4.weight")]
            bias = [g("fc3.bias"), g("fc4.bias")]
        Wc = torch.zeros(64, 512, device=dev)
        bc = torch.zeros(64, device=dev)
        rc = torch.cat(rows)
        Wc[:rc.shape[0]] = rc
        bc[:rc.shape[0]] = torch.cat(bias)
        w["head_wt"] = Wc.t().contiguous()
        w["head_b"] = bc
        self.head_rows = rc.shape[0]

    # ------------------------------------------------------------------ helpers
    def _st(self):
        return _lib.stream_ptr()

    def label_vectors(self, cats: torch.Tensor, super_mh: Optional[torch.Tensor]):
        """Per-object 512-vectors replacing the one-hot/multi-hot concat of ``model.py:152-168``."""
        cfg, fc2 = self.cfg, self.w["fc2_full"]
        C = cfg.num_classes
        lsub = fc2[:, 4096 + cats].t()
        lobj = fc2[:, 4096 + C + cats].t()
        if super_mh is not None and cfg.dataset == "vg":
            S = cfg.num_super_classes
            lsub = lsub + super_mh @ fc2[:, 4096 + 2 * C:4096 + 2 * C + S].t()
            lobj = lobj + super_mh @ fc2[:, 4096 + 2 * C + S:4096 + 2 * C + 2 * S].t()
        return lsub.contiguous(), lobj.contiguous()

    # ------------------------------------------------------------------ stages
    def image_maps(self, f0: torch.Tensor, f1: Optional[torch.Tensor], roles=(0, 1), tag="img"):
        """conv1 + tanh per image and role: returns {role: a_img [n_img*1024,128] f16}."""
        lib, ws = self.lib, self.ws
        n_img = f0.shape[0]
        C0 = f0.shape[1]
        C1 = 0 if f1 is None else f1.shape[1]
        x = ws.get("x_" + tag, n_img * 1024 * XC, torch.float16)
        _lib.check(lib.sgc_pack_image_nhwc(_lib.ptr(f0), C0, _lib.ptr(f1), C1, _lib.ptr(x), n_img, 1024, XC, self._st()),
                   "sgc_pack_image_nhwc")
        out = {}
        for r in roles:
            a = ws.get("a_img_%s_%d" % (tag, r), n_img * 1024 * 128, torch.float16)
            _lib.check(lib.sgc_conv1_tanh(_lib.ptr(x), _lib.ptr(self.w["w1r"][r]), _lib.ptr(self.w["b1"][r]), _lib.ptr(a),
                                          n_img * 1024, XC, self._st()), "sgc_conv1_tanh")
            out[r] = a
        self._x = x
        return out

    def object_halves(self, a_img, obj_img: torch.Tensor, bbox: torch.Tensor, roles=(0, 1)):
        """Per-object masked maps and conv2 halves U (role 0) / V (role 1, carries the bias)."""
        lib, ws = self.lib, self.ws
        n_obj = obj_img.shape[0]
        res = {}
        for r in roles:
            a_pad = ws.get("a_pad_%d" % r, n_obj * 34 * 34 * 128, torch.float16)
            _lib.check(lib.sgc_object_masked_maps(_lib.ptr(a_img[r]), _lib.ptr(obj_img), _lib.ptr(bbox),
                                                  _lib.ptr(self.w["cst"][r]), _lib.ptr(a_pad), n_obj, 32, 128,
                                                  self._st()), "sgc_object_masked_maps")
            uv = ws.get("uv_%d" % r, n_obj * 1024 * 512, torch.float16)
            _lib.check(lib.sgc_conv2_object(_lib.ptr(a_pad), _lib.ptr(self.w["w2r"][r]),
                                            _lib.ptr(self.w["b2"]) if r == 1 else None, _lib.ptr(uv), n_obj, self._st()),
                       "sgc_conv2_object")
            res[r] = uv
        return res

    def pair_trunk(self, U, V, sub_idx, obj_idx, lsub, lobj, train=False, seeds=(0, 0), keep_argmax=False,
                   iou_mask=None) -> PairOutputs:
        lib, ws, cfg = self.lib, self.ws, self.cfg
        P = int(sub_idx.shape[0])
        Ppad = (P + 63) // 64 * 64
        z = ws.get("z_pad", P * 18 * 18 * 512, torch.float16)      # border stays zero: only interiors are written
        _lib.check(lib.sgc_pair_expand(_lib.ptr(U), _lib.ptr(V), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(z), P,
                                       ELEM_F16, self._st()), "sgc_pair_expand")
        y = ws.get("y", Ppad * 65536, torch.float16)
        am = ws.get("argmax", P * 65536, torch.uint8) if keep_argmax else None
        _lib.check(lib.sgc_conv3_relu_pool(_lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(y),
                                           _lib.ptr(am), P, self._st()), "sgc_conv3_relu_pool")
        h1 = ws.get("h1", Ppad * 4096, torch.float16)
        _lib.check(lib.sgc_fc1_relu(_lib.ptr(y), _lib.ptr(self.w["w1p"]), _lib.ptr(self.w["bf1"]), _lib.ptr(h1), P, 65536,
                                    int(train), ctypes.c_uint(seeds[0]), self._st()), "sgc_fc1_relu")
        p = ws.get("p", Ppad * 512, torch.float32)
        _lib.check(lib.sgc_fc2_labels_relu(_lib.ptr(h1), _lib.ptr(self.w["w2m"]), _lib.ptr(self.w["bf2"]), _lib.ptr(lsub),
                                           _lib.ptr(lobj), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(p), P,
                                           int(train), ctypes.c_uint(seeds[1]), self._st()), "sgc_fc2_labels_relu")
        return self.head(p, P, iou_mask)

    def head(self, p, P, iou_mask=None) -> PairOutputs:
        lib, cfg, dev = self.lib, self.cfg, self.device
        R = cfg.num_relations
        hier = cfg.hierarchical
        nc = 3 if hier else 1
        rel = torch.empty(P, R, dtype=torch.float32, device=dev)
        sup = torch.empty(P, 3, dtype=torch.float32, device=dev) if hier else None
        conn = torch.empty(P, dtype=torch.float32, device=dev)
        cconf = torch.empty(P, nc, dtype=torch.float32, device=dev)
        cpred = torch.empty(P, nc, dtype=torch.int32, device=dev)
        T = self.T
        f = ctypes.c_float
        _lib.check(lib.sgc_bayes_head(_lib.ptr(p), _lib.ptr(self.w["head_wt"]), _lib.ptr(self.w["head_b"]), P,
                                      cfg.num_geometric if hier else R, cfg.num_possessive if hier else 0,
                                      cfg.num_semantic if hier else 0, int(hier), f(T[0]), f(T[1]), f(T[2]),
                                      _lib.ptr(rel), _lib.ptr(sup), _lib.ptr(conn), _lib.ptr(cconf), _lib.ptr(cpred),
                                      _lib.ptr(iou_mask), self._st()), "sgc_bayes_head")
        return PairOutputs(rel, sup, conn, p[:P * 512].view(P, 512), cconf, cpred)

    # ------------------------------------------------------------------ fused entry
    def forward_pairs(self, image_feature, image_depth, obj_img, bbox, cats, super_mh, sub_idx, obj_idx, train=False,
                      seeds=(0, 0), keep_argmax=False, iou_mask=None) -> PairOutputs:
        """One call per minibatch: image maps -> per-object halves -> all pairs."""
        a_img = self.image_maps(image_feature, image_depth)
        uv = self.object_halves(a_img, obj_img, bbox)
        lsub, lobj = self.label_vectors(cats, super_mh)
        self._lsub, self._lobj = lsub, lobj
        return self.pair_trunk(uv[0], uv[1], sub_idx, obj_idx, lsub, lobj, train, seeds, keep_argmax, iou_mask)

"""Forward of ``engine.RelHeadEngine``: image maps -> per-object conv2 halves -> pair expansion -> conv3 / fc1 over shared windows (or per pair)
-> fc2 + label gather -> head; evaluation (``forward_pairs``) and training (``train_forward`` keeps what the backward needs in a ``TrainContext``)."""
from __future__ import annotations

from .engine_core import *          # noqa: F401,F403  (TUNING, Workspace, PairOutputs, TrainContext, _lib, torch, np, ... - see engine_core.__all__)


class ForwardMixin:
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

    def object_halves(self, a_img, obj_img: torch.Tensor, bbox: torch.Tensor, roles=(0, 1), with_bg=False, regions=None):
        """Per-object masked maps and conv2 halves U (role 0) / V (role 1, carries the bias).
        ``with_bg``: one object with an EMPTY box per image is appended (index n_obj + image) - the constant map tanh(b1) every
        masked map equals outside its box; its halves are the background of ``conv3_shared``.  (One per image rather than one in
        all: the backward sums the background's gradient per image, so a step over B images stays the sum of B one-image steps.)
        ``regions`` (host count of the 2x2-pixel windows of all objects' D16 rectangles, ``DeviceScene.conv2_windows``; needs
        ``with_bg``): conv2 runs on those windows only - outside them an object's half IS its image's background half (the map is
        the constant tanh(b1) outside the box), copied row by row: same bits, 17 % of the rows on the benchmark's boxes."""
        lib, ws = self.lib, self.ws
        n_real = int(obj_img.shape[0])
        n_img = 0
        if with_bg:
            n_img = int(a_img[roles[0]].numel()) // (1024 * 128)
            obj_img = torch.cat([obj_img, torch.arange(n_img, dtype=obj_img.dtype, device=obj_img.device)])
            bbox = torch.cat([bbox, bbox.new_zeros(n_img, 4)])
        n_obj = obj_img.shape[0]
        by_region = bool(with_bg and regions and TUNING.shared_conv2 and n_real > 0)
        if by_region:
            cnt = torch.empty(n_real, dtype=torch.int32, device=self.device)
            _lib.check(lib.sgc_conv2_regions_count(_lib.ptr(bbox), n_real, _lib.ptr(cnt), self._st()), "sgc_conv2_regions_count")
            incl = torch.cumsum(cnt, 0, dtype=torch.int32)
            rlist = self.scratch.get("conv2_regions", int(regions) + 64, torch.int32)
            _lib.check(lib.sgc_conv2_regions_fill(_lib.ptr(bbox), n_real, _lib.ptr(incl), _lib.ptr(rlist), self._st()), "sgc_conv2_regions_fill")
            rn = incl[n_real - 1:]
        res = {}
        for r in roles:
            a_pad = ws.get("a_pad_%d" % r, n_obj * 34 * 34 * 128, torch.float16)
            _lib.check(lib.sgc_object_masked_maps(_lib.ptr(a_img[r]), _lib.ptr(obj_img), _lib.ptr(bbox),
                                                  _lib.ptr(self.w["cst"][r]), _lib.ptr(a_pad), n_obj, 32, 128,
                                                  self._st()), "sgc_object_masked_maps")
            uv = self.scratch.get("uv_%d" % r, n_obj * 1024 * 512, torch.float16)
            bias = _lib.ptr(self.w["b2"]) if r == 1 else None
            if by_region:
                def run(a_pad=a_pad, uv=uv, bias=bias, r=r):
                    _lib.check(lib.sgc_conv2_object(_lib.ptr(a_pad[n_real * 34 * 34 * 128:]), _lib.ptr(self.w["w2r"][r]), bias,
                                                    _lib.ptr(uv[n_real * 1024 * 512:]), n_img, self._st()), "sgc_conv2_object")
                    _lib.check(lib.sgc_conv2_object_regions(_lib.ptr(a_pad), _lib.ptr(self.w["w2r"][r]), bias, _lib.ptr(rlist), _lib.ptr(rn),
                                                            int(regions), _lib.ptr(uv), self._st()), "sgc_conv2_object_regions")
                    _lib.check(lib.sgc_conv2_fill_background(_lib.ptr(bbox), _lib.ptr(obj_img), n_real, _lib.ptr(uv), self._st()),
                               "sgc_conv2_fill_background")
                self._timed("conv2_fwd", run)
            else:
                self._timed("conv2_fwd", lambda: _lib.check(lib.sgc_conv2_object(_lib.ptr(a_pad), _lib.ptr(self.w["w2r"][r]), bias, _lib.ptr(uv),
                                                                                n_obj, self._st()), "sgc_conv2_object"))
            res[r] = uv
        return res

    def expand(self, U, V, sub_idx, obj_idx, P, z, z_bf=None, amz=None, dense=None, pixrect=None):
        """Pair expansion: dense LDS-staged kernel when the pair list is "all ordered pairs of every image"
        (dense = (img_ptr, pid, max_n)), generic pair-list kernel otherwise.  ``pixrect`` ([P] packed rectangles from
        ``shared_plan``): only the pixels conv3 over shared windows reads are written (dense kernel only)."""
        lib = self.lib
        if dense is not None and 0 < dense[2] <= 150:
            img_ptr, pid, max_n = dense
            self._timed("expand_dense", lambda: _lib.check(lib.sgc_pair_expand_dense_windows(
                _lib.ptr(U), _lib.ptr(V), _lib.ptr(img_ptr), _lib.ptr(pid), int(pid.shape[1]), int(img_ptr.shape[0]) - 1, max_n,
                _lib.ptr(z), _lib.ptr(z_bf), _lib.ptr(amz), _lib.ptr(pixrect), self._st()), "sgc_pair_expand_dense_windows"))
        elif z_bf is None and amz is None:
            self._timed("expand", lambda: _lib.check(lib.sgc_pair_expand(_lib.ptr(U), _lib.ptr(V), _lib.ptr(sub_idx), _lib.ptr(obj_idx),
                                                                       _lib.ptr(z), P, ELEM_F16, self._st()), "sgc_pair_expand"))
        else:
            self._timed("expand_train", lambda: _lib.check(lib.sgc_pair_expand_train(
                _lib.ptr(U), _lib.ptr(V), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(z), _lib.ptr(z_bf), _lib.ptr(amz), P,
                self._st()), "sgc_pair_expand_train"))

    def fc1_shared(self, wm, ywm, bbox, sub_idx, obj_idx, incl, P, n_obj, h1, dropout, seed, order=None):
        """fc1 + ReLU (+ dropout) from the window-major rows: grouped GEMM, per-object 2-D prefix sums, per-pair assembly."""
        lib, sc = self.lib, self.scratch
        w1p = self.w["w1p"]                          # deferred copy: made here (after the wait for fc1.weight's all-gather, if one is in flight)
        owm = sc.get("owm", wm["rows"] * int(lib.sgc_fc1_products_pitch()), torch.float32)
        oxh = None
        if TUNING.fc1_x16:
            oxh = sc.get("oxh", wm["rows"] * 4096, torch.float16)
            self._timed("fc1_fwd_windows", lambda: _lib.check(lib.sgc_fc1_windows_gemm_x16(
                _lib.ptr(ywm), _lib.ptr(w1p), _lib.ptr(wm["tile_group"]), _lib.ptr(wm["goff"]), wm["n2"], _lib.ptr(owm), _lib.ptr(oxh), wm["rows"],
                self._st()), "sgc_fc1_windows_gemm_x16"))
        else:
            self._timed("fc1_fwd_windows", lambda: _lib.check(lib.sgc_fc1_windows_gemm(
                _lib.ptr(ywm), _lib.ptr(w1p), _lib.ptr(wm["tile_group"]), _lib.ptr(owm), wm["rows"], self._st()), "sgc_fc1_windows_gemm"))
        S = sc.get("fc1_S", wm["n2"] * 81 * 4096, torch.float32)
        self._timed("fc1_fwd_integral", lambda: _lib.check(lib.sgc_fc1_integral(_lib.ptr(owm), _lib.ptr(wm["goff"]), wm["n2"], _lib.ptr(S), self._st()),
                                                           "sgc_fc1_integral"))
        own = None
        if TUNING.fc1_own_sums:            # S'_j[R_j] per object: read once per pair instead of four corners
            own = sc.get("fc1_own", max(n_obj, 1) * 4096, torch.float32)
            _lib.check(lib.sgc_fc1_own_rect_sums(_lib.ptr(S), _lib.ptr(bbox), n_obj, _lib.ptr(own), self._st()), "sgc_fc1_own_rect_sums")
        if order is not None and (not TUNING.assemble_by_subject or int(order.shape[0]) != P):
            order = None
        if oxh is not None:
            self._timed("fc1_fwd_assemble", lambda: _lib.check(lib.sgc_fc1_assemble_x16(
                _lib.ptr(S), _lib.ptr(oxh), _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(incl), _lib.ptr(wm["dest"]), n_obj,
                _lib.ptr(self.w["bf1"]), int(dropout), ctypes.c_uint(seed), _lib.ptr(h1), P, _lib.ptr(own), _lib.ptr(order), self._st()),
                "sgc_fc1_assemble_x16"))
        else:
            self._timed("fc1_fwd_assemble", lambda: _lib.check(lib.sgc_fc1_assemble_ordered(
                _lib.ptr(S), _lib.ptr(owm), _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(incl), _lib.ptr(wm["dest"]), n_obj,
                _lib.ptr(self.w["bf1"]), int(dropout), ctypes.c_uint(seed), _lib.ptr(h1), P, _lib.ptr(own), _lib.ptr(order), self._st()),
                "sgc_fc1_assemble_ordered"))

    def conv3_shared(self, plan, z, U, V, bbox, obj_img, sub_idx, obj_idx, P, y, am, y_bf, keep=None, wm=None):
        """conv3 + ReLU + pool with the per-object part computed once per object (``csrc/kernels_shared.hip``): U / V hold
        n_obj + n_img objects, the last n_img the empty-box backgrounds of the images; ``plan`` from ``shared_plan``.
        ``z`` [P + 2 n_obj (+ n_img)] padded maps: the real pairs' expansion is there already, the pseudo-pairs' (and background
        maps') is written here; ``keep=(z_bf, amz)`` (training): same-shaped bf16 copy and routing codes for the backward.
        ``wm`` (``window_major_rows``): ``y`` / ``y_bf`` are the window-major buffers of the shared fc1 and nothing is assembled
        per pair; with ``plan['objects']`` the pseudo-pairs are computed on their own windows only (second level).
        Returns what the backward needs."""
        lib, sc = self.lib, self.scratch
        own = self.ws if keep is not None else sc
        n_obj, n_img = int(obj_img.shape[0]), plan["n_img"]
        n2 = 2 * n_obj
        objects = bool(plan["objects"]) and wm is not None
        n_tail = n2 + (n_img if objects else 0)                                  # + the all-background map of every image
        bg_codes = raw_n = None
        if TUNING.plan_kernels and obj_img.dtype == torch.int32:
            tabs = sc.get("ps_tables", 2 * n_tail + 64 * n_img + 4, torch.int32)
            ps_sub, ps_obj = tabs[:n_tail], tabs[n_tail:2 * n_tail]
            bg_codes, raw_n = tabs[2 * n_tail:2 * n_tail + 64 * n_img], tabs[2 * n_tail + 64 * n_img:2 * n_tail + 64 * n_img + 1]
            _lib.check(lib.sgc_pseudo_pair_tables(_lib.ptr(obj_img), n_obj, n_img, P, int(objects), _lib.ptr(ps_sub), _lib.ptr(ps_obj),
                                                  _lib.ptr(bg_codes), _lib.ptr(raw_n), self._st()), "sgc_pseudo_pair_tables")
        else:
            ar = torch.arange(n_obj, dtype=torch.int32, device=self.device)
            bg = obj_img.to(torch.int32) + n_obj                                 # every object's background = its image's
            ps_sub, ps_obj = torch.cat([ar, bg]), torch.cat([bg, ar])
            if objects:
                bgs = torch.arange(n_obj, n_obj + n_img, dtype=torch.int32, device=self.device)
                ps_sub, ps_obj = torch.cat([ps_sub, bgs]), torch.cat([ps_obj, bgs])
        zt = z[P * 18 * 18 * 512:]
        if keep is None:
            _lib.check(lib.sgc_pair_expand(_lib.ptr(U), _lib.ptr(V), _lib.ptr(ps_sub), _lib.ptr(ps_obj), _lib.ptr(zt), n_tail, ELEM_F16, self._st()),
                       "sgc_pair_expand")
        else:
            _lib.check(lib.sgc_pair_expand_train(_lib.ptr(U), _lib.ptr(V), _lib.ptr(ps_sub), _lib.ptr(ps_obj), _lib.ptr(zt),
                                                 _lib.ptr(keep[0][(P - keep[2]) * 18 * 18 * 512:]), _lib.ptr(keep[1][P * 256 * 256:]), n_tail, self._st()),
                       "sgc_pair_expand_train")
        gather, incl = plan["gather"], plan["incl"]
        out = dict(plan, n2=n2, wm=wm)
        if wm is not None:
            am_ps = am[P * 65536:] if am is not None else None                   # routing codes of the pseudo-pairs: behind the real pairs'
            if objects:
                # second level: the pseudo-pairs' windows R_o are entries of the window list; the other rows are the background maps'
                zb = zt[n2 * 18 * 18 * 512:]
                y_bg = sc.get("y_bg", n_img * 65536, torch.float16)
                ybf_bg = sc.get("ybf_bg", n_img * 65536, torch.bfloat16) if y_bf is not None else None
                am_bg = own.get("am_bg", n_img * 65536, torch.uint8) if am is not None else None
                self._timed("conv3_fwd_objects", lambda: _lib.check(lib.sgc_conv3_relu_pool(
                    _lib.ptr(zb), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(y_bg), _lib.ptr(am_bg), _lib.ptr(ybf_bg), n_img,
                    self._st()), "sgc_conv3_relu_pool"))
                _lib.check(lib.sgc_shared_objects_fill_rows(_lib.ptr(bbox), _lib.ptr(obj_img), n_obj, _lib.ptr(wm["goff"]), _lib.ptr(y_bg),
                                                            _lib.ptr(ybf_bg), _lib.ptr(am_bg), _lib.ptr(y), _lib.ptr(y_bf), _lib.ptr(am_ps),
                                                            self._st()), "sgc_shared_objects_fill_rows")
                out["am_bg"] = am_bg
            else:
                self._timed("conv3_fwd_objects", lambda: _lib.check(lib.sgc_conv3_relu_pool_wm(
                    _lib.ptr(zt), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(wm["goff"]), _lib.ptr(y), _lib.ptr(am_ps), _lib.ptr(y_bf),
                    n2, self._st()), "sgc_conv3_relu_pool_wm"))
            lin = plan.get("lin") if objects else None
            if lin is None:
                self._timed("conv3_fwd_windows", lambda: _lib.check(lib.sgc_conv3_relu_pool_windows_wm(
                    _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(gather), _lib.ptr(plan["n_total"]), _lib.ptr(wm["dest_conv"]),
                    plan["bound"], _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), self._st()), "sgc_conv3_relu_pool_windows_wm"))
            else:
                # linear pairs: their X windows are combined from the pre-activations of the per-object entries (the tail of the list:
                # the same launch stores their accumulators) and of the images' background maps (a 64 n_img-window launch of their own)
                n_pe = plan["entries"] - plan["entries_real"]
                n_raw = n_pe + 64 * n_img
                raw = sc.get("raw_pre", n_raw * 4 * 1024, torch.float32)
                self._timed("conv3_fwd_windows", lambda: _lib.check(lib.sgc_conv3_relu_pool_windows_wm_raw(
                    _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(gather), _lib.ptr(plan["n_total"]), _lib.ptr(wm["dest_conv"]),
                    plan["bound"], _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), _lib.ptr(raw), plan["entries_real"], self._st()),
                    "sgc_conv3_relu_pool_windows_wm_raw"))
                if bg_codes is None:
                    bg_codes = ((P + n2 + torch.arange(n_img, device=self.device, dtype=torch.int32))[:, None] * 64
                                + torch.arange(64, device=self.device, dtype=torch.int32)[None, :]).reshape(-1).contiguous()
                    raw_n = torch.full((1,), 64 * n_img, dtype=torch.int32, device=self.device)
                self._timed("conv3_fwd_raw", lambda: _lib.check(lib.sgc_conv3_windows_raw(
                    _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(bg_codes), _lib.ptr(raw_n), 64 * n_img, _lib.ptr(raw[n_pe * 4096:]), self._st()),
                    "sgc_conv3_windows_raw"))
                self._timed("conv3_fwd_linear", lambda: _lib.check(lib.sgc_windows_linear_forward(
                    _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(obj_img), n_obj, P, _lib.ptr(lin["gather"]), _lib.ptr(lin["n"]),
                    lin["max"], _lib.ptr(plan["incl_all"]), _lib.ptr(wm["dest"]), _lib.ptr(raw), _c_long(n_pe), _lib.ptr(self.w["b3"]),
                    _lib.ptr(y), _lib.ptr(y_bf), _lib.ptr(am), _lib.ptr(lin["drow"]), self._st()), "sgc_windows_linear_forward"))
            out["am_ps"] = am_ps
            return out
        am_ps = own.get("am_ps", n2 * 65536, torch.uint8) if am is not None else None
        y_ps = sc.get("y_ps", n2 * 65536, torch.float16)
        ybf_ps = sc.get("ybf_ps", n2 * 65536, torch.bfloat16) if y_bf is not None else None
        self._timed("conv3_fwd_objects", lambda: _lib.check(lib.sgc_conv3_relu_pool(
            _lib.ptr(zt), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(y_ps), _lib.ptr(am_ps), _lib.ptr(ybf_ps), n2,
            self._st()), "sgc_conv3_relu_pool"))
        self._timed("conv3_fwd_windows", lambda: _lib.check(lib.sgc_conv3_relu_pool_windows(
            _lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]), _lib.ptr(gather), _lib.ptr(plan["n_total"]), plan["bound"], _lib.ptr(y),
            _lib.ptr(am), _lib.ptr(y_bf), self._st()), "sgc_conv3_relu_pool_windows"))
        self._timed("conv3_fwd_assemble", lambda: _lib.check(lib.sgc_shared_windows_assemble(
            _lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, n_obj, _lib.ptr(y_ps), _lib.ptr(am_ps), _lib.ptr(ybf_ps),
            _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), self._st()), "sgc_shared_windows_assemble"))
        out["am_ps"] = am_ps
        return out

    def pair_trunk(self, U, V, sub_idx, obj_idx, lsub, lobj, train=False, seeds=(0, 0), keep_argmax=False,
                   iou_mask=None, dense=None, shared=None, pair_order=None) -> PairOutputs:
        lib, ws, cfg = self.lib, self.ws, self.cfg
        P = int(sub_idx.shape[0])
        Ppad = (P + 63) // 64 * 64
        plan, Pt = None, P
        if shared is not None:
            n_obj = int(shared[1].shape[0])
            n_img = int(U.numel()) // (1024 * 512) - n_obj                 # the background objects behind the real ones
            wm_mode = shared_fc1_enabled()
            plan = self.shared_plan(shared[0], sub_idx, obj_idx, P, shared[2], n_obj=n_obj, n_img=n_img,
                                    objects=wm_mode and shared_objects_enabled(), obj_img=shared[1])
            Pt = P + 2 * n_obj + n_img
        z = ws.get("z_pad", Pt * 18 * 18 * 512, torch.float16)     # border stays zero: only interiors are written
        # the row plan (a sort, a dozen small launches, four blocking host-to-device copies) goes BEFORE the pair expansion: its
        # launches are then behind the host when the 1.4 ms expansion kernel starts, instead of leaving the GPU idle between them
        wm = self.window_major_rows(plan, P, 2 * n_obj) if shared is not None and wm_mode else None
        self.expand(U, V, sub_idx, obj_idx, P, z, dense=dense, pixrect=None if plan is None else plan["pixrect"])
        am = ws.get("argmax", Pt * 65536, torch.uint8) if keep_argmax else None
        h1 = ws.get("h1", Ppad * 4096, torch.float16)
        if shared is not None and wm_mode:
            # conv3 and fc1 over shared windows: the rows fc1 multiplies are written window-major, y [P, 65536] never exists
            ywm = ws.get("ywm", wm["rows"] * 1024, torch.float16)
            self.conv3_shared(plan, z, U, V, shared[0], shared[1], sub_idx, obj_idx, P, ywm, am, None, wm=wm)
            self.fc1_shared(wm, ywm, shared[0], sub_idx, obj_idx, plan.get("incl_all", plan["incl"]), P, n_obj, h1, train, seeds[0], order=pair_order)
        else:
            y = ws.get("y", Ppad * 65536, torch.float16)
            if shared is not None:
                self.conv3_shared(plan, z, U, V, shared[0], shared[1], sub_idx, obj_idx, P, y, am, None)
            else:
                self._timed("conv3_fwd", lambda: _lib.check(lib.sgc_conv3_relu_pool(_lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]),
                                                                                   _lib.ptr(y), _lib.ptr(am), _lib.ptr(None), P, self._st()),
                                                            "sgc_conv3_relu_pool"))
            w1p = self.w["w1p"]                               # deferred copy (Weights): made here, outside the timed launch
            self._timed("fc1_fwd", lambda: _lib.check(lib.sgc_fc1_relu(_lib.ptr(y), _lib.ptr(w1p), _lib.ptr(self.w["bf1"]), _lib.ptr(h1), P, 65536,
                                                                       int(train), ctypes.c_uint(seeds[0]), self._st()), "sgc_fc1_relu"))
        p = ws.get("p", Ppad * 512, torch.float32)
        self._timed("fc2_fwd", lambda: _lib.check(lib.sgc_fc2_labels_relu(_lib.ptr(h1), _lib.ptr(self.w["w2m"]), _lib.ptr(self.w["bf2"]), _lib.ptr(lsub),
                                           _lib.ptr(lobj), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(p), P,
                                           int(train), ctypes.c_uint(seeds[1]), self._st()), "sgc_fc2_labels_relu"))
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
                      seeds=(0, 0), keep_argmax=False, iou_mask=None, dense=None, select=None, shared_windows=None, pair_order=None) -> PairOutputs:
        """One call per minibatch: image maps -> per-object halves -> all pairs.
        ``select`` ([P] bool / uint8 device tensor): run the per-pair trunk (expansion, conv3, fc1, fc2, head) ONLY for the selected
        pairs and scatter the results into full-size outputs; the other pairs get confidence -inf (exactly what the overlap filter
        gives them in the evaluator, ``evaluator.py:131-134``), prediction 0, zero log-probs and hidden vectors."""
        self.verify_checks()
        a_img = self.image_maps(image_feature, image_depth)
        share = shared_conv3_enabled(shared_windows, int(sub_idx.shape[0]))
        uv = self.object_halves(a_img, obj_img, bbox, with_bg=share,
                                regions=shared_windows.get("conv2_windows") if (share and isinstance(shared_windows, dict)) else None)
        shared = (bbox, obj_img, shared_windows) if share else None
        lsub, lobj = self.label_vectors(cats, super_mh)
        self._lsub, self._lobj = lsub, lobj
        if select is None:
            return self.pair_trunk(uv[0], uv[1], sub_idx, obj_idx, lsub, lobj, train, seeds, keep_argmax, iou_mask, dense, shared, pair_order)
        sel = select.bool()
        if share and shared_windows is not None:
            # With conv3 / fc1 over shared windows a pair whose boxes do not overlap has (almost) no pair-specific window: skipping it
            # saves nothing, while a pair SUBSET loses the host's window counts (read-backs) and the dense expansion.  Compute every
            # pair and blank the unselected ones - the same outputs (measured 21.4 vs 34.2 ms per 8x64 minibatch).
            out = self.pair_trunk(uv[0], uv[1], sub_idx, obj_idx, lsub, lobj, train, seeds, keep_argmax, iou_mask, dense, shared, pair_order)
            drop = ~sel
            out.relation[drop] = 0
            if out.super_relation is not None:
                out.super_relation[drop] = 0
            out.connectivity[drop] = 0
            hidden = out.hidden.clone()
            hidden[drop] = 0
            out.cand_conf[drop] = -math.inf
            out.cand_pred[drop] = 0
            return PairOutputs(out.relation, out.super_relation, out.connectivity, hidden, out.cand_conf, out.cand_pred)
        idx = torch.nonzero(sel).flatten()
        P, Ps = int(sub_idx.shape[0]), int(idx.shape[0])
        cfg, dev = self.cfg, self.device
        nc = 3 if cfg.hierarchical else 1
        full = PairOutputs(torch.zeros(P, cfg.num_relations, device=dev), torch.zeros(P, 3, device=dev) if cfg.hierarchical else None,
                           torch.zeros(P, device=dev), torch.zeros(P, 512, device=dev),
                           torch.full((P, nc), -math.inf, device=dev), torch.zeros(P, nc, dtype=torch.int32, device=dev))
        if Ps == 0:
            return full
        dense_s = None
        if dense is not None:
            img_ptr, pid, max_n = dense
            rank = (torch.cumsum(sel.int(), 0) - 1).int()
            ok = pid >= 0
            pc = pid.clamp(min=0).long()
            dense_s = (img_ptr, torch.where(ok & sel[pc], rank[pc], torch.full_like(pid, -1)).contiguous(), max_n)
        if shared is not None and isinstance(shared[2], dict):
            # the counts describe the full pair list: for a subset only the object part stays exact, the rest is an upper bound
            shared = (shared[0], shared[1], None)
        out = self.pair_trunk(uv[0], uv[1], sub_idx[idx].contiguous(), obj_idx[idx].contiguous(), lsub, lobj, train, seeds, keep_argmax,
                              None if iou_mask is None else iou_mask[idx].contiguous(), dense_s, shared)
        full.relation[idx] = out.relation
        if full.super_relation is not None:
            full.super_relation[idx] = out.super_relation
        full.connectivity[idx] = out.connectivity
        full.hidden[idx] = out.hidden
        full.cand_conf[idx] = out.cand_conf
        full.cand_pred[idx] = out.cand_pred
        return full

    def compat_forward(self, hs, ho, c1, c2, mh1, mh2, train=False, seeds=(0, 0)) -> PairOutputs:
        """The reference's per-step call on PRE-MASKED inputs (``model.py:170``): row k of ``hs`` / ``ho`` [b,257,32,32] is the subject /
        object crop of pair k (inference: no context is kept)."""
        dev = self.device
        b = int(hs.shape[0])
        a_s = self.image_maps(hs, None, roles=(0,), tag="cs")
        a_o = self.image_maps(ho, None, roles=(1,), tag="co")
        F = self.cfg.feature_size
        full = torch.tensor([[0, F, 0, F]], dtype=torch.int32, device=dev).repeat(b, 1).contiguous()
        ids = torch.arange(b, dtype=torch.int32, device=dev)
        U = self.object_halves({0: a_s[0]}, ids, full, roles=(0,))[0]
        V = self.object_halves({1: a_o[1]}, ids, full, roles=(1,))[1]
        lsub, _ = self.label_vectors(c1, mh1)
        _, lobj = self.label_vectors(c2, mh2)
        return self.pair_trunk(U, V, ids, ids, lsub, lobj, train=train, seeds=seeds)

    def train_forward(self, image_feature, image_depth, obj_img, bbox, cats, super_mh, sub_idx, obj_idx, seeds=(0, 0),
                      dropout=True, dense=None, role_inputs=None, cats_obj=None, super_mh_obj=None, shared_windows=None, pair_order=None) -> "TrainContext":
        """Forward that keeps what the backward needs (pool argmaxes, expansion routing mask).
        ``role_inputs=(h_sub, h_obj)``: the reference's per-step call on PRE-MASKED ``[b,257,32,32]`` inputs (``model.py:170``):
        row k of each is the subject / object crop of pair k, with labels ``cats`` / ``cats_obj``; every crop is its own
        "image" with one full-size box."""
        lib, ws, sc = self.lib, self.ws, self.scratch
        self.verify_checks()
        ctx = TrainContext()
        ctx.n_obj = int(obj_img.shape[0])
        ctx.P = P = int(sub_idx.shape[0])
        ctx.Ppad = Ppad = (P + 63) // 64 * 64
        ctx.obj_img, ctx.bbox, ctx.sub_idx, ctx.obj_idx = obj_img, bbox, sub_idx, obj_idx
        ctx.cats = (cats, cats if cats_obj is None else cats_obj)
        ctx.super_mh = (super_mh, super_mh if role_inputs is None else super_mh_obj)
        ctx.dropout, ctx.seeds = dropout, seeds
        if role_inputs is None:
            ctx.n_img = int(image_feature.shape[0])
            ctx.a_img = self.image_maps(image_feature, image_depth)
            ctx.x = (self._x, self._x)
        else:
            ctx.n_img = int(role_inputs[0].shape[0])
            a_s = self.image_maps(role_inputs[0], None, roles=(0,), tag="s")
            x_s = self._x
            a_o = self.image_maps(role_inputs[1], None, roles=(1,), tag="o")
            ctx.a_img, ctx.x = {0: a_s[0], 1: a_o[1]}, (x_s, self._x)
        share = role_inputs is None and shared_conv3_enabled(shared_windows, P)   # per-step calls: every crop is one full-size box, nothing is shared
        ctx.uv = self.object_halves(ctx.a_img, obj_img, bbox, with_bg=share,
                                    regions=shared_windows.get("conv2_windows") if (share and isinstance(shared_windows, dict)) else None)
        ctx.lsub, lobj_same = self.label_vectors(ctx.cats[0], ctx.super_mh[0])
        ctx.lobj = lobj_same if role_inputs is None else self.label_vectors(ctx.cats[1], ctx.super_mh[1])[1]
        # the per-pair backward (TUNING.shared_bwd off, A/B) reads every pixel of z / amz; the shared one only those next to X windows
        narrow = share and TUNING.shared_bwd
        wm_mode = narrow and shared_fc1_enabled()
        plan = self.shared_plan(bbox, sub_idx, obj_idx, P, shared_windows, keep=True, n_obj=ctx.n_obj, n_img=ctx.n_img,
                                objects=wm_mode and shared_objects_enabled(), obj_img=obj_img) if share else None
        Pt = P + (2 * ctx.n_obj + ctx.n_img if share else 0)         # pseudo-pairs and background maps live behind the real pairs
        z = sc.get("z_pad", Pt * 18 * 18 * 512, torch.float16)
        # bf16 copy of z for the weight gradients.  With the shared backward in its patch form the real pairs' copy is never read: the
        # patch gather converts the f16 rows it gathers (sgc_windows_im2patch_f16), so the expansion writes 1 KB less per pixel and
        # only the pseudo-pairs / background maps behind the real pairs get a bf16 map (``z_bf_base`` = pair index of the buffer's first map)
        ctx.z_bf_base = P if (narrow and TUNING.patch_wgrad and dense is not None and 0 < dense[2] <= 150) else 0
        z_bf = ws.get("z_pad_bf", (Pt - ctx.z_bf_base) * 18 * 18 * 512, torch.bfloat16)
        amz = ws.get("amz", Pt * 256 * 256, torch.uint8)             # two 4-bit routing codes per byte
        wm = self.window_major_rows(plan, P, 2 * ctx.n_obj) if wm_mode else None     # before the expansion: see forward_pairs
        self.expand(ctx.uv[0], ctx.uv[1], sub_idx, obj_idx, P, z, None if ctx.z_bf_base else z_bf, amz, dense=dense,
                    pixrect=plan["pixrect"] if narrow else None)
        ctx.z_bf = z_bf
        am = ws.get("argmax", Pt * 65536, torch.uint8)              # conv3 routing codes (shared path: only the rows of listed windows)
        h1 = ws.get("h1", Ppad * 4096, torch.float16)
        ctx.shared, ctx.y, ctx.y_bf = None, None, None
        if wm_mode:
            # conv3 and fc1 over shared windows: y and its bf16 copy exist only as the window-major rows fc1 multiplies
            ywm = sc.get("ywm", wm["rows"] * 1024, torch.float16)
            ywm_bf = ws.get("ywm_bf", wm["rows"] * 1024, torch.bfloat16)
            ctx.shared = self.conv3_shared(plan, z, ctx.uv[0], ctx.uv[1], bbox, obj_img, sub_idx, obj_idx, P, ywm, am, ywm_bf, keep=(z_bf, amz, ctx.z_bf_base), wm=wm)
            ctx.shared["ywm_bf"] = ywm_bf
            self.fc1_shared(wm, ywm, bbox, sub_idx, obj_idx, plan.get("incl_all", plan["incl"]), P, ctx.n_obj, h1, dropout, seeds[0], order=pair_order)
        else:
            y = sc.get("y", Ppad * 65536, torch.float16)
            y_bf = ws.get("y_bf", Ppad * 65536, torch.bfloat16)     # bf16 copy for the fc1 weight gradient, written by the same epilogue
            if Ppad > P:
                Workspace._zero(y_bf[P * 65536:])
            if share:
                ctx.shared = self.conv3_shared(plan, z, ctx.uv[0], ctx.uv[1], bbox, obj_img, sub_idx, obj_idx, P, y, am, y_bf, keep=(z_bf, amz, ctx.z_bf_base))
            else:
                self._timed("conv3_fwd", lambda: _lib.check(lib.sgc_conv3_relu_pool(_lib.ptr(z), _lib.ptr(self.w["w3r"]), _lib.ptr(self.w["b3"]),
                                                                                   _lib.ptr(y), _lib.ptr(am), _lib.ptr(y_bf), P, self._st()),
                                                            "sgc_conv3_relu_pool"))
            w1p = self.w["w1p"]                               # deferred copy (Weights): made here, outside the timed launch
            self._timed("fc1_fwd", lambda: _lib.check(lib.sgc_fc1_relu(_lib.ptr(y), _lib.ptr(w1p), _lib.ptr(self.w["bf1"]), _lib.ptr(h1), P, 65536,
                                                                       int(dropout), ctypes.c_uint(seeds[0]), self._st()), "sgc_fc1_relu"))
            ctx.y, ctx.y_bf = y, y_bf
        p = ws.get("p", Ppad * 512, torch.float32)
        self._timed("fc2_fwd", lambda: _lib.check(lib.sgc_fc2_labels_relu(_lib.ptr(h1), _lib.ptr(self.w["w2m"]), _lib.ptr(self.w["bf2"]), _lib.ptr(ctx.lsub),
                                           _lib.ptr(ctx.lobj), _lib.ptr(sub_idx), _lib.ptr(obj_idx), _lib.ptr(p), P, int(dropout),
                                           ctypes.c_uint(seeds[1]), self._st()), "sgc_fc2_labels_relu"))
        ctx.z, ctx.amz, ctx.am, ctx.h1, ctx.p = z, amz, am, h1, p
        ctx.out = self.head(p, P)
        return ctx

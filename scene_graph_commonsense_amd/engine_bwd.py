"""Backward of ``engine.RelHeadEngine``: head loss gradient -> fc2 -> fc1 -> conv3 (shared windows: sparse-MFMA patch forms, per-object sums) ->
pair contraction -> conv2 -> mask backward -> conv1; two HIP streams (weight-gradient chain beside the data-gradient chain, ``_side_chain``)."""
from __future__ import annotations

from .engine_core import *          # noqa: F401,F403  (TUNING, Workspace, PairOutputs, TrainContext, _lib, torch, np, ... - see engine_core.__all__)


SLAB_CAPACITY = 48          # f32 split-K slabs of [1024][4608] in the weight-gradient chain's buffer (0.9 GB)


class BackwardMixin:
    def _slab_sum(self, slabs, n, count):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.sgc_slab_sum(_lib.ptr(slabs), _lib.ptr(out), _c_long(n), int(count), 0, self._st()), "sgc_slab_sum")
        return out


    def _colsum(self, X, rows, cols, elem=ELEM_BF16, blocks=None, part_name="colsum_part"):
        if blocks is None:
            blocks = max(1, min(512, rows // 64))
        blocks = int(max(1, min(blocks, rows)))
        part = self.scratch.get(part_name, blocks * cols, torch.float32)
        _lib.check(self.lib.sgc_colsum(elem, _lib.ptr(X), _lib.ptr(part), _c_long(rows), cols, blocks, self._st()), "sgc_colsum")
        return self._slab_sum(part, cols, blocks)


    def _to_bf16(self, name, src, n):
        dst = self.scratch.get(name, n, torch.bfloat16)
        self._timed("convert", lambda: _lib.check(self.lib.sgc_convert_f16_bf16(_lib.ptr(src), _lib.ptr(dst), _c_long(n), self._st()),
                                                  "sgc_convert_f16_bf16"))
        return dst


    def train_backward(self, ctx: "TrainContext", coefs, sub_csr, obj_csr, img_ptr, splits=8, grad_hook=None, dp_extra=None,
                       cs_coef=None, upstream=None):
        """Backward of the whole path.  Returns (loss scalar tensor, {reference parameter name: f32 gradient}).
        ``upstream=(g_rel, g_sup, g_conn, g_hidden)`` (``coefs`` None): no loss here - gradients of the outputs handed over by the
        caller's autograd (per-step ``forward()``); the returned loss is None."""
        lib, cfg, dev, w = self.lib, self.cfg, self.device, self.w
        ws = self.scratch              # everything allocated below is transient; what the forward kept lives in ``ctx`` / ``self.ws``
        P, Ppad, n_obj, n_img = ctx.P, ctx.Ppad, ctx.n_obj, ctx.n_img
        st = self._st
        hier = cfg.hierarchical
        R = cfg.num_relations
        scale = 2.0 if ctx.dropout else 1.0
        f = ctypes.c_float
        grads: Dict[str, torch.Tensor] = {}
        slabs_n = ctypes.c_int(0)

        # ---- head: loss, dlogits, d(fc2 pre-activation)
        dl = ws.get("dl", P * 64, torch.float32)
        loss_i = ws.get("loss_i", P, torch.float32)
        dpre = ws.get("dpre", Ppad * 512, torch.bfloat16)
        if Ppad > P:
            Workspace._zero(dpre[P * 512:])
        T = self.T
        if upstream is None:
            tgt, ca, cb, cc, cy = coefs
            _lib.check(lib.sgc_head_loss_bwd(_lib.ptr(ctx.out.relation), _lib.ptr(ctx.out.super_relation), _lib.ptr(ctx.out.connectivity),
                                             _lib.ptr(ctx.p), _lib.ptr(tgt), _lib.ptr(ca), _lib.ptr(cb), _lib.ptr(cc), _lib.ptr(cy),
                                             _lib.ptr(w["head_w"]), P, cfg.num_geometric if hier else R,
                                             cfg.num_possessive if hier else 0, cfg.num_semantic if hier else 0, int(hier),
                                             f(T[0]), f(T[1]), f(T[2]), f(scale), _lib.ptr(dl), _lib.ptr(loss_i), _lib.ptr(dpre),
                                             _lib.ptr(dp_extra), _lib.ptr(cs_coef), _lib.ptr(ctx.out.cand_conf),
                                             _lib.ptr(ctx.out.cand_pred), st()),
                       "sgc_head_loss_bwd")
            loss = loss_i.sum() if P > 0 else torch.zeros((), device=dev)      # P scalars: host-side glue
        else:
            g_rel, g_sup, g_conn, g_hid = (None if g is None else g.to(dev, torch.float32).contiguous() for g in upstream)
            _lib.check(lib.sgc_head_bwd_upstream(_lib.ptr(ctx.out.relation), _lib.ptr(ctx.out.super_relation), _lib.ptr(ctx.p),
                                                 _lib.ptr(g_rel), _lib.ptr(g_sup), _lib.ptr(g_conn), _lib.ptr(g_hid),
                                                 _lib.ptr(w["head_w"]), P, cfg.num_geometric if hier else R,
                                                 cfg.num_possessive if hier else 0, cfg.num_semantic if hier else 0, int(hier),
                                                 f(T[0]), f(T[1]), f(T[2]), f(scale), _lib.ptr(dl), _lib.ptr(dpre), st()),
                       "sgc_head_bwd_upstream")
            loss = None
        # The backward is two chains.  DATA gradients (fc2 -> fc1 -> un-pool -> conv3 -> pair contraction -> conv2 -> masks) run
        # on the caller's stream: each feeds the next.  WEIGHT gradients (one GEMM per layer + slab sums / transposes / bias
        # column sums) only have to be there when the optimizer runs, so they go to a side stream as soon as their two operands
        # exist: the HBM-bound kernels of the data chain (un-pool + pack, pair contraction, converts) and the tails of its GEMMs
        # then overlap with weight-gradient GEMM blocks instead of leaving the matrix cores idle.  ``side`` orders the side
        # stream after everything enqueued so far.  Measured -0.3 ... -2.2 ms per step in five alternated A/B pairs on three boxes
        # (small: every GEMM block owns its CU - 146 KiB LDS, all VGPRs - so kernels of two streams interleave block by block
        # instead of co-residing; profiles/README.md); results bit-identical either way (tests/test_configs_gpu.py).
        # ``SGC_BWD_STREAMS=0`` = one stream (used for per-kernel profiles: durations of overlapped launches mean little).
        side = self._side_chain()
        sl = ws.get("slabs", SLAB_CAPACITY * 1024 * 4608, torch.float32)      # split-K slabs of the weight-gradient chain (largest user: conv3)

        # ---- head weights
        with side():
            chunk = max(16, (P + 255) // 256)
            nb = (P + chunk - 1) // chunk
            part = ws.get("head_part", nb * 64 * 513, torch.float32)
            _lib.check(lib.sgc_head_wgrad(_lib.ptr(dl), _lib.ptr(ctx.p), _lib.ptr(part), P, chunk, st()), "sgc_head_wgrad")
            hw = self._slab_sum(part, 64 * 513, nb).view(64, 513)
            names = (["fc3_1", "fc3_2", "fc3_3", "fc5", "fc4"] if hier else ["fc3", "fc4"])
            sizes = ([cfg.num_geometric, cfg.num_possessive, cfg.num_semantic, 3, 1] if hier else [R, 1])
            r0 = 0
            for nm, sz in zip(names, sizes):
                grads[nm + ".weight"] = hw[r0:r0 + sz, :512].contiguous()
                grads[nm + ".bias"] = hw[r0:r0 + sz, 512].contiguous()
                r0 += sz

            # ---- fc2 weights: main block (split-K GEMM) + label columns (per-object row sums scattered by label)
            h1_bf = self._to_bf16("h1_bf", ctx.h1, Ppad * 4096)
            self._timed("fc2_wgrad", lambda: _lib.check(lib.sgc_fc2_wgrad(_lib.ptr(dpre), _lib.ptr(h1_bf), _lib.ptr(sl), Ppad, 32, ctypes.byref(slabs_n), st()),
                       "sgc_fc2_wgrad"))
            gfc2 = torch.empty_like(w["fc2_full"])                       # main block + label columns are written below
            ld2 = int(gfc2.shape[1])
            use_mh = ctx.super_mh[0] is not None and cfg.dataset == "vg"
            n_lab = 2 * cfg.num_classes + (2 * cfg.num_super_classes if use_mh else 0)
            if 4096 + n_lab < ld2:                                       # VG model without super-categories: nothing writes the
                gfc2[:, 4096 + n_lab:].zero_()                           # multi-hot columns - their gradient is zero, not garbage
            _lib.check(lib.sgc_slab_sum_ld(_lib.ptr(sl), _lib.ptr(gfc2), 512, 4096, _c_long(ld2), slabs_n.value, st()), "sgc_slab_sum_ld")
            dls = torch.empty(n_obj, 512, dtype=torch.float32, device=dev)
            dlo = torch.empty(n_obj, 512, dtype=torch.float32, device=dev)
            _lib.check(lib.sgc_segment_sum_rows(_lib.ptr(dpre), _lib.ptr(sub_csr[0]), _lib.ptr(sub_csr[1]), _lib.ptr(dls), n_obj, 512, st()),
                       "sgc_segment_sum_rows")
            _lib.check(lib.sgc_segment_sum_rows(_lib.ptr(dpre), _lib.ptr(obj_csr[0]), _lib.ptr(obj_csr[1]), _lib.ptr(dlo), n_obj, 512, st()),
                       "sgc_segment_sum_rows")
            mh_s, mh_o = (ctx.super_mh if use_mh else (None, None))
            cats_s, cats_o = (c if c.dtype == torch.int64 else c.long() for c in ctx.cats)
            _lib.check(lib.sgc_label_grads(_lib.ptr(dls), _lib.ptr(dlo), _lib.ptr(cats_s), _lib.ptr(cats_o), _lib.ptr(mh_s), _lib.ptr(mh_o),
                                           n_obj, cfg.num_classes, cfg.num_super_classes if use_mh else 0, _lib.ptr(gfc2), ld2, 4096, st()),
                       "sgc_label_grads")
            grads["fc2.weight"] = gfc2
            grads["fc2.bias"] = self._colsum(dpre, Ppad, 512)

        # ---- fc2 data gradient
        dh1 = ws.get("dh1", Ppad * 4096, torch.bfloat16)
        if Ppad > P:
            Workspace._zero(dh1[P * 4096:])
        self._timed("fc2_dgrad", lambda: _lib.check(lib.sgc_fc2_dgrad(_lib.ptr(dpre), _lib.ptr(w["w2mT"]), _lib.ptr(ctx.h1), _lib.ptr(dh1), P, f(scale), st()),
                   "sgc_fc2_dgrad"))

        if getattr(ctx, "generic", None) is not None:        # sizes other than 128 / 32: the f32 per-pair trunk (engine_generic.py)
            self._generic_trunk_backward(ctx, dh1, grads, side)
            side.join()
            return loss, grads

        # ---- fc1
        if getattr(ctx, "shared", None) is not None and ctx.shared.get("wm") is not None:
            dy = self._fc1_backward_rows(ctx, dh1, sub_csr, obj_csr, side, grads, grad_hook)
        else:
            dy = self._fc1_backward_pairs(ctx, dh1, side, grads, grad_hook)

        # ---- conv3
        nparts = ctypes.c_int(0)
        shared = ctx.shared if (getattr(ctx, "shared", None) is not None and TUNING.shared_bwd) else None
        n_objx = n_obj + (n_img if shared is not None else 0)      # the images' background objects take part in the conv2 backward
        if shared is not None:
            dz = self._conv3_backward_shared(ctx, shared, dy, sub_csr, obj_csr, img_ptr, side, sl, grads)
        else:
            dz = self._conv3_backward_pairs(ctx, dy, side, sl, grads)

        # ---- pair contraction + conv2 + masks + conv1 (per-role buffers: the side stream may still read role 0's while role 1 runs)
        gc2 = torch.empty(512, 256, 3, 3, dtype=torch.float32, device=dev)
        # second level of sharing on: an object's dU is zero outside the pixel rectangle of its pseudo-pair (every pair of the object
        # lives inside it), so the conv2 data gradient runs on those cells + one more ring (csrc/kernels_shared.hip: conv2_bwd_regions)
        c2_list = c2_n = None
        if (shared is not None and TUNING.conv2_bwd_regions and bool(shared.get("objects")) and shared.get("wm") is not None and n_obj > 0):
            c2_list = ws.get("c2b_list", n_objx * 256 + 64, torch.int32)
            c2_n = ws.get("c2b_n", 4, torch.int32)
            _lib.check(lib.sgc_conv2_bwd_regions(_lib.ptr(ctx.bbox), n_obj, n_objx, 1, _lib.ptr(c2_list), _lib.ptr(c2_n), st()), "sgc_conv2_bwd_regions")
        mapU, mapA = 34 * 34 * 512, 34 * 34 * 128
        # The conv2 / conv1 weight gradients (0.56 + 0.1 ms per role) and their slab / column sums.  On the side stream they queue up BEHIND
        # the sparse conv3 weight gradient, which starts when the data-gradient GEMM ends and is that stream's long pole (5.7 ms alone,
        # 8.4 beside this loop's passes): the caller's stream finished at 35.1 ms of the step and then waited 2.7 ms for the side stream's
        # tail (tools/two_stream_timeline.py, profiles/r06_two_stream_timeline.txt).  ``TUNING.small_wgrads_main``: they run in line on
        # the caller's stream - inside the time it would otherwise wait - with slabs / partial sums of their own.
        in_line = bool(TUNING.small_wgrads_main and TUNING.bwd_streams)
        small = contextlib.nullcontext if in_line else side
        sl_s = ws.get("slabs_small", 64 * 512 * 1152, torch.float32) if in_line else sl      # 64 = the most splits the launches choose (csrc/gemm_tn.h:tn_auto_splits)
        cpn = "colsum_part_main" if in_line else "colsum_part"
        for r, csr in ((0, sub_csr), (1, obj_csr)):
            dU = ws.get("dU_pad_%d" % r, n_objx * mapU, torch.bfloat16)
            if shared is not None:
                self._timed("contract", lambda: _lib.check(lib.sgc_pair_contract_windows(
                    _lib.ptr(dz), _lib.ptr(ctx.amz), _lib.ptr(csr[0]), _lib.ptr(csr[1]), _lib.ptr(shared["pixrect"]), _lib.ptr(img_ptr),
                    r, P, n_obj, n_img, int(bool(shared.get("objects")) and shared.get("wm") is not None), _lib.ptr(dU), st()),
                    "sgc_pair_contract_windows"))
            else:
                self._timed("contract", lambda: _lib.check(lib.sgc_pair_contract(_lib.ptr(dz), _lib.ptr(ctx.amz), _lib.ptr(csr[0]), _lib.ptr(csr[1]), _lib.ptr(dU), n_obj, st()),
                           "sgc_pair_contract"))
            with small():
                a_pad = self.ws.get("a_pad_%d" % r, n_objx * mapA, torch.float16)      # kept by the forward
                a_bf = self._to_bf16("a_pad_bf", a_pad, n_objx * mapA)
                self._timed("conv2_wgrad", lambda: _lib.check(lib.sgc_conv2_wgrad(_lib.ptr(dU), _lib.ptr(a_bf), _lib.ptr(sl_s), n_objx, 0, ctypes.byref(slabs_n), st()),
                           "sgc_conv2_wgrad"))
                assert slabs_n.value <= 64
                dW2r = self._slab_sum(sl_s, 512 * 1152, slabs_n.value)
                gc2[:, r * 128:(r + 1) * 128] = dW2r.view(512, 3, 3, 128).permute(0, 3, 1, 2)
                if r == 1:
                    grads["conv2_1.bias"] = self._colsum(dU, n_objx * 34 * 34, 512, part_name=cpn)
            da = ws.get("da", n_objx * 1024 * 128, torch.bfloat16)
            if c2_list is not None:
                Workspace._zero(da)                  # unlisted cells: the gradient there is exactly zero
                self._timed("conv2_dgrad", lambda: _lib.check(lib.sgc_conv2_dgrad_regions(
                    _lib.ptr(dU), _lib.ptr(w["wd2"][r]), _lib.ptr(c2_list), _lib.ptr(c2_n), n_objx * 256, _lib.ptr(da), st()), "sgc_conv2_dgrad_regions"))
            else:
                self._timed("conv2_dgrad", lambda: _lib.check(lib.sgc_conv2_dgrad(_lib.ptr(dU), _lib.ptr(w["wd2"][r]), _lib.ptr(da), n_objx, st()), "sgc_conv2_dgrad"))
            dcst_bg = None
            if shared is not None:                   # the background objects are constant everywhere: all of their gradient goes to tanh(b1)
                dcst_bg = self.ws.get("dcst_bg_%d" % r, 128, torch.float32)      # on THIS stream: ``da`` is rewritten by the next role
                torch.sum(da[n_obj * 1024 * 128:].view(-1, 128).float(), 0, out=dcst_bg)
            dA = ws.get("dA", n_img * 1024 * 128, torch.float32)
            cpart = ws.get("dcst_part_%d" % r, n_img * 64 * 128, torch.float32)
            _lib.check(lib.sgc_object_masked_maps_bwd(_lib.ptr(da), _lib.ptr(img_ptr), _lib.ptr(ctx.bbox), _lib.ptr(dA), _lib.ptr(cpart),
                                                      ctypes.byref(nparts), n_img, 32, 128, st()), "sgc_object_masked_maps_bwd")
            n_cst = nparts.value
            dp1 = ws.get("dpre1_%d" % r, n_img * 1024 * 128, torch.bfloat16)
            _lib.check(lib.sgc_tanh_bwd(_lib.ptr(dA), _lib.ptr(ctx.a_img[r]), _lib.ptr(dp1), _c_long(n_img * 1024 * 128), st()),
                       "sgc_tanh_bwd")
            with small():
                x_bf = self._to_bf16("x_bf", ctx.x[r], n_img * 1024 * XC)
                _lib.check(lib.sgc_conv1_wgrad(_lib.ptr(dp1), _lib.ptr(x_bf), _lib.ptr(sl_s), n_img * 1024, XC, 16, ctypes.byref(slabs_n), st()),
                           "sgc_conv1_wgrad")
                dW1 = self._slab_sum(sl_s, 128 * XC, slabs_n.value).view(128, XC)
                nm = "conv1_%d" % (r + 1)
                grads[nm + ".weight"] = dW1[:, :257].reshape(128, 257, 1, 1).contiguous()
                tb = torch.tanh(w["b1"][r])
                dcst = self._slab_sum(cpart, 128, n_cst)
                if dcst_bg is not None:
                    dcst = dcst + dcst_bg
                grads[nm + ".bias"] = self._colsum(dp1, n_img * 1024, 128, part_name=cpn) + dcst * (1 - tb * tb)
        with side():
            grads["conv2_1.weight"] = gc2
        side.join()                              # the caller's stream continues only after every gradient is complete
        return loss, grads

    def _fc1_finish_wgrad(self, dW1p, dh1, Ppad, grads, grad_hook):
        """dW1p [4096][(window, channel)] -> the reference's column order (channel*64 + window), bias gradient, early all-reduce hook."""
        lib, dev, st = self.lib, self.device, self._st
        if self.fc1_grad_gemm_order:
            # handed over as the GEMM wrote it (a view of the engine's scratch: valid until the next backward); the optimizer's fused
            # update un-permutes it on the fly
            grads["fc1.weight"] = dW1p[:4096 * 65536].view(4096, 65536)
            if grad_hook is not None:
                grad_hook("fc1.weight", grads["fc1.weight"])
            grads["fc1.bias"] = self._colsum(dh1, Ppad, 4096)
            return
        gfc1 = torch.empty(4096, 65536, dtype=torch.float32, device=dev)
        _lib.check(lib.sgc_transpose_cast(_lib.ptr(dW1p), _lib.ptr(gfc1), 2, 4096, 16, _c_long(65536), _c_long(64), _c_long(1024),
                                          _c_long(65536), _c_long(4096), _c_long(64), st()), "sgc_transpose_cast")
        grads["fc1.weight"] = gfc1
        if grad_hook is not None:              # largest gradient (97 % of the bytes) is ready first: overlap its all-reduce
            grad_hook("fc1.weight", grads["fc1.weight"])
        grads["fc1.bias"] = self._colsum(dh1, Ppad, 4096)

    def _fc1_backward_pairs(self, ctx, dh1, side, grads, grad_hook):
        """fc1 backward as two [pairs, 65536] GEMMs; returns dy [Ppad*64, 1024] (pair-major pooled gradient)."""
        lib, w, ws, st, P, Ppad = self.lib, self.w, self.scratch, self._st, ctx.P, ctx.Ppad
        with side():
            dW1p = ws.get("dW1p", 4096 * 65536, torch.float32)
            self._timed("fc1_wgrad", lambda: _lib.check(lib.sgc_fc1_wgrad(_lib.ptr(dh1), _lib.ptr(ctx.y_bf), _lib.ptr(dW1p), Ppad, 65536, st()),
                                                        "sgc_fc1_wgrad"))
            self._fc1_finish_wgrad(dW1p, dh1, Ppad, grads, grad_hook)
        dy = ws.get("dy", Ppad * 65536, torch.bfloat16)
        w1pT = w["w1pT"]                               # deferred copy (Weights): made here, outside the timed launch
        self._timed("fc1_dgrad", lambda: _lib.check(lib.sgc_fc1_dgrad(_lib.ptr(dh1), _lib.ptr(w1pT), _lib.ptr(dy), P, 65536, st()), "sgc_fc1_dgrad"))
        return dy

    def _fc1_backward_rows(self, ctx, dh1, sub_csr, obj_csr, side, grads, grad_hook):
        """fc1 backward over the window-major rows (``csrc/kernels_shared.hip``): per-object sums of dh1 + one copy of dh1 per X entry,
        then the grouped weight- and data-gradient GEMMs; returns dywm [rows, 1024] (window-major pooled gradient)."""
        lib, w, ws, st, sh = self.lib, self.w, self.scratch, self._st, ctx.shared
        wm = sh["wm"]
        gwm = ws.get("gwm", wm["rows"] * 4096, torch.bfloat16)
        self._timed("fc1_bwd_rows", lambda: (
            _lib.check(lib.sgc_fc1_gsum(_lib.ptr(dh1), _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sub_csr[0]),
                                        _lib.ptr(sub_csr[1]), _lib.ptr(obj_csr[0]), _lib.ptr(obj_csr[1]), _lib.ptr(wm["goff"]), ctx.n_obj,
                                        _lib.ptr(gwm), st()), "sgc_fc1_gsum"),
            _lib.check(lib.sgc_fc1_xrows(_lib.ptr(dh1), _lib.ptr(sh.get("gather_all", sh["gather"])), _lib.ptr(wm["dest"]), wm["E"], _lib.ptr(wm["goff"]),
                                         _lib.ptr(wm["gend"]), _lib.ptr(gwm), _lib.ptr(sh["ywm_bf"]), st()), "sgc_fc1_xrows")))
        dy = ws.get("dywm", wm["rows"] * 1024, torch.bfloat16)

        def wgrad():
            with side():
                dW1p = ws.get("dW1p", 4096 * 65536, torch.float32)
                self._timed("fc1_wgrad", lambda: _lib.check(lib.sgc_fc1_windows_wgrad(_lib.ptr(gwm), _lib.ptr(sh["ywm_bf"]), _lib.ptr(wm["goff"]),
                                                                                      _lib.ptr(dW1p), wm["rows"], st()), "sgc_fc1_windows_wgrad"))
                self._fc1_finish_wgrad(dW1p, dh1, ctx.Ppad, grads, grad_hook)

        def dgrad():
            w1pT = w["w1pT"]                               # deferred copy (Weights): made here, outside the timed launch
            self._timed("fc1_dgrad", lambda: _lib.check(lib.sgc_fc1_windows_dgrad(_lib.ptr(gwm), _lib.ptr(w1pT), _lib.ptr(wm["tile_group"]),
                                                                                  _lib.ptr(dy), wm["rows"], st()), "sgc_fc1_windows_dgrad"))
        # GEMM beside GEMM buys nothing on this chip (two ping-pong GEMMs on two streams: 13.5 ms against 13.4 back to back) while an
        # HBM-bound kernel beside a GEMM hides ~40 % of its time (profiles/r03_overlap_microbench.txt).  ``TUNING.gemms_apart``: the
        # weight-gradient GEMM is enqueued AFTER the data-gradient GEMM - ``side()`` orders the side stream behind everything enqueued on
        # the caller's stream so far - and so runs beside the row sums / un-pool kernels that follow the data gradient instead of
        # beside the data gradient itself.  Off: round 2's order (both GEMMs at once).
        if TUNING.gemms_apart:
            dgrad()
            wgrad()
        else:
            wgrad()
            dgrad()
        return dy

    def _conv3_backward_pairs(self, ctx, dy, side, sl, grads):
        """conv3 backward over every window of every pair (no per-object sharing): bias + weight gradient, returns dz.
        Weight gradient on the sparse matrix cores: the pooled gradient + the arg-max byte ARE the 2:4-compressed operand
        (csrc/gemm_tn_sp.h; one pass over dy packs it and writes the bias partials).  Data gradient with the un-pool inside its
        operand staging (sgc_conv3_dgrad_pooled): the 21 GB un-pooled tensor is neither written nor read.  (The dense weight
        gradient / the two-pass un-pool, sgc_conv3_wgrad / sgc_unpool_relu_bwd / sgc_conv3_dgrad, remain in the C-ABI and are tested
        against these in tests/test_gemm_gpu.py; the step no longer switches to them.)"""
        lib, w, ws, st, P = self.lib, self.w, self.scratch, self._st, ctx.P
        slabs_n = ctypes.c_int(0)
        bpart = ws.get("b3_part", 2048 * 1024, torch.float32)
        nparts = ctypes.c_int(0)
        z_bf = ctx.z_bf
        if getattr(ctx, "z_bf_base", 0):
            raise RuntimeError("the per-pair conv3 backward needs the bf16 copy of z that this forward did not write "
                               "(TUNING changed between forward and backward)")
        pack_a = ws.get("w3_pack_a", P * 4 * 1024 * 64, torch.uint8)
        pack_i = ws.get("w3_pack_i", P * 4 * 1024 * 8, torch.uint8)
        self._timed("unpool", lambda: _lib.check(lib.sgc_unpool_relu_bwd_pack(
            _lib.ptr(dy), _lib.ptr(ctx.am), None, _lib.ptr(bpart), ctypes.byref(nparts), _lib.ptr(pack_a),
            _lib.ptr(pack_i), P, st()), "sgc_unpool_relu_bwd_pack"))
        n_b3 = nparts.value
        with side():
            grads["conv3_1.bias"] = self._slab_sum(bpart, 1024, n_b3)
            self._timed("conv3_wgrad", lambda: _lib.check(lib.sgc_conv3_wgrad_sparse(
                None, None, _lib.ptr(z_bf), _lib.ptr(pack_a), _lib.ptr(pack_i), _lib.ptr(sl), P, 0,
                ctypes.byref(slabs_n), st()), "sgc_conv3_wgrad_sparse"))
            dW3r = self._slab_sum(sl, 1024 * 4608, slabs_n.value)
            grads["conv3_1.weight"] = dW3r.view(1024, 3, 3, 512).permute(0, 3, 1, 2).contiguous()
        dz = ws.get("dz", P * 256 * 512, torch.bfloat16)
        self._timed("conv3_dgrad", lambda: _lib.check(lib.sgc_conv3_dgrad_pooled(_lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(w["wd3"]), _lib.ptr(dz), P, st()),
                                                      "sgc_conv3_dgrad_pooled"))
        return dz

    def _conv3_backward_shared(self, ctx, sh, dy, sub_csr, obj_csr, img_ptr, side, sl, grads):
        """Backward of ``conv3_shared`` (autodiff of that graph).  Per-object part: the gradient rows of copied windows, summed per
        object (by the shared fc1's data gradient, or by ``sgc_shared_windows_assemble_bwd`` from a pair-major ``dy``), go through
        the ordinary conv3 backward of whole maps - the 2*n_obj pseudo-pairs, or with the second level only the n_img background
        maps, the pseudo-pairs' own windows being window-list entries like the X windows.  Listed windows: compact column form
        (un-pool -> [rows,1024]; weight gradient = rows^T x im2col(z); data gradient = rows x W^T -> col2im).
        Returns dz [(P + 2 n_obj + n_img) * 256, 512]: a pair's rows exist only inside its pixel rectangle (``plan['pixrect']``)."""
        lib, w, ws, st, P, n_obj, n_img = self.lib, self.w, self.scratch, self._st, ctx.P, ctx.n_obj, ctx.n_img
        n2, wm, objects = sh["n2"], sh.get("wm"), bool(sh.get("objects")) and sh.get("wm") is not None
        E = sh["entries"]
        Epad = (E + 15) // 16 * 16                                   # 4 rows per entry: the GEMMs want a multiple of 64 rows
        slabs_n, slabs_x, nparts, nparts_x = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        gather, gn = sh["gather"], sh["n_total"]
        z_bf = ctx.z_bf
        Pt = P + n2 + n_img
        dz = ws.get("dz", Pt * 256 * 512, torch.bfloat16)
        # ---- whole maps: the pseudo-pairs (first level) or the background maps (second level), gradient = sums of copied rows
        if objects:
            n_maps, map0 = n_img, P + n2
            dy_maps = ws.get("dy_bg", n_img * 65536, torch.bfloat16)
            _lib.check(lib.sgc_shared_objects_bg_grad(_lib.ptr(ctx.bbox), _lib.ptr(img_ptr), n_obj, n_img, _lib.ptr(wm["goff"]), _lib.ptr(dy),
                                                      _lib.ptr(dy_maps), st()), "sgc_shared_objects_bg_grad")
            am_maps = sh["am_bg"]
        else:
            n_maps, map0 = n2, P
            dy_maps = ws.get("dy_ps", n2 * 65536, torch.bfloat16)
            if wm is not None:
                # ``dy`` is the window-major gradient of the shared fc1: the per-object rows are already sums; bring them to pair-major order
                idx = (wm["goff"][:64].long()[None, :] + torch.arange(n2, device=self.device)[:, None]).reshape(-1)
                torch.index_select(dy.view(-1, 1024), 0, idx, out=dy_maps.view(-1, 1024))
            else:
                self._timed("conv3_bwd_assemble", lambda: _lib.check(lib.sgc_shared_windows_assemble_bwd(
                    _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sub_csr[0]), _lib.ptr(sub_csr[1]), _lib.ptr(obj_csr[0]),
                    _lib.ptr(obj_csr[1]), n_obj, _lib.ptr(dy), _lib.ptr(dy_maps), st()), "sgc_shared_windows_assemble_bwd"))
            am_maps = sh["am_ps"]
        dest = wm["dest_conv"] if wm is not None else None
        lin = sh.get("lin") if objects else None
        zb0 = getattr(ctx, "z_bf_base", 0)
        z_bf_maps = z_bf[(map0 - zb0) * 18 * 18 * 512:]
        dz_maps = dz[map0 * 256 * 512:]
        bpart = ws.get("b3_part", 2048 * 1024, torch.float32)
        bpart_x = ws.get("b3_part_x", 1024 * 1024, torch.float32)
        pack_a = ws.get("w3_pack_a", n_maps * 4 * 1024 * 64, torch.uint8)
        pack_i = ws.get("w3_pack_i", n_maps * 4 * 1024 * 8, torch.uint8)
        dy3_bg = bpart_l = None
        if lin is None:
            self._timed("unpool_objects", lambda: _lib.check(lib.sgc_unpool_relu_bwd_pack(
                _lib.ptr(dy_maps), _lib.ptr(am_maps), None, _lib.ptr(bpart), ctypes.byref(nparts), _lib.ptr(pack_a), _lib.ptr(pack_i), n_maps, st()),
                "sgc_unpool_relu_bwd_pack"))
        else:
            # the background maps also collect (minus) the gradient of the linear pairs' windows: their un-pooled gradient is dense,
            # so they take the two-pass un-pool and the dense conv3 backward (n_img maps)
            dy3_bg = ws.get("dy3_bg_pad", n_maps * 18 * 18 * 1024, torch.bfloat16)        # created zeroed: the halo stays zero
            self._timed("unpool_objects", lambda: _lib.check(lib.sgc_unpool_relu_bwd(
                _lib.ptr(dy_maps), _lib.ptr(am_maps), _lib.ptr(dy3_bg), _lib.ptr(bpart), ctypes.byref(nparts), n_maps, st()), "sgc_unpool_relu_bwd"))
        # ---- listed windows: compact un-pool.  The real pairs' windows in front of the list (one non-zero per window and channel in
        # their un-pooled gradient) go through the SPARSE forms of both backward GEMMs, which pack their operand from the pooled rows:
        # only the entries behind them (per-object entries: sums of several windows, + the boundary tile) are un-pooled
        e_real = sh.get("entries_real")
        sp_ok = e_real is not None and dest is not None and Epad and int(e_real) >= 4096
        e_spw = (int(e_real) // 16) * 16 if (sp_ok and TUNING.patch_wgrad and TUNING.sparse_wgrad) else 0
        e_spd = (int(e_real) // 256) * 256 if (sp_ok and TUNING.patch_dgrad and TUNING.sparse_dgrad and "w3sp" in w) else 0
        e_un0 = min(e_spw, e_spd)
        dy3x = ws.get("dy3x", max(Epad, 16) * 4 * 1024, torch.bfloat16)
        self._timed("unpool_windows", lambda: _lib.check(lib.sgc_windows_unpool_from(
            _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(gn), _lib.ptr(dest), e_un0, Epad - e_un0, _lib.ptr(dy3x[e_un0 * 4096:]),
            _lib.ptr(bpart_x), ctypes.byref(nparts_x), st()), "sgc_windows_unpool_from"))
        nparts_s, bpart_s = ctypes.c_int(0), None
        if e_spd:
            spa = ws.get("w3d_pack_a", 4 * e_spd * 1024, torch.bfloat16)
            spi = ws.get("w3d_pack_i", 4 * e_spd * 64, torch.int32)
            if e_un0 == e_spd and e_un0 > 0:          # the un-pool pass skipped these windows: their bias partial sums come from the packer
                bpart_s = ws.get("b3_part_s", 1024 * 1024, torch.float32)
            self._timed("dgrad_pack_windows", lambda: _lib.check(lib.sgc_windows_dgrad_sparse_pack(
                _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(dest), e_spd, _lib.ptr(spa), _lib.ptr(spi), _lib.ptr(bpart_s),
                ctypes.byref(nparts_s), st()), "sgc_windows_dgrad_sparse_pack"))
        elif e_un0 > 0:
            raise RuntimeError("internal: windows skipped by the un-pool pass without a sparse data gradient to count their bias")
        if lin is not None:
            # transpose of sgc_windows_linear_forward: + into the un-pooled rows of the two per-object entries, - into the background map
            e_real = sh["entries_real"]
            bpart_l = ws.get("b3_part_l", 64 * n_img * 1024, torch.float32)
            self._timed("linear_bwd", lambda: (
                _lib.check(lib.sgc_windows_linear_backward_objects(
                    _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sub_csr[0]), _lib.ptr(sub_csr[1]), _lib.ptr(obj_csr[0]),
                    _lib.ptr(obj_csr[1]), n_obj, P, _lib.ptr(gather), e_real, E - e_real, _lib.ptr(sh["incl_all"]), _lib.ptr(wm["dest"]),
                    _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(dy3x), st()), "sgc_windows_linear_backward_objects"),
                _lib.check(lib.sgc_windows_linear_backward_bg(
                    _lib.ptr(lin["gather"]), _lib.ptr(lin["drow"]), _lib.ptr(lin["order"]), _lib.ptr(lin["seg"]), n_img, _lib.ptr(dy),
                    _lib.ptr(ctx.am), _lib.ptr(dy3_bg), _lib.ptr(bpart_l), st()), "sgc_windows_linear_backward_bg")))
        # weight gradient of the sparse part with its second operand gathered from the forward's f16 maps (no patch copy): only where the
        # patch copy would be made from those f16 maps anyway (``zb0``: no bf16 copy of the real pairs' z exists) - the same bits
        gather_w = bool(TUNING.gather_wgrad and TUNING.patch_wgrad and zb0 and Epad and e_spw >= 4096)
        with side():
            gb = self._slab_sum(bpart, 1024, nparts.value)
            if nparts_x.value:
                gb = gb + self._slab_sum(bpart_x, 1024, nparts_x.value)
            if bpart_s is not None and nparts_s.value:
                gb = gb + self._slab_sum(bpart_s, 1024, nparts_s.value)
            if bpart_l is not None:
                gb = gb + self._slab_sum(bpart_l, 1024, 64 * n_img)
            grads["conv3_1.bias"] = gb
            if lin is None:
                self._timed("conv3_wgrad_objects", lambda: _lib.check(lib.sgc_conv3_wgrad_sparse(
                    None, None, _lib.ptr(z_bf_maps), _lib.ptr(pack_a), _lib.ptr(pack_i), _lib.ptr(sl), n_maps, 0, ctypes.byref(slabs_n), st()),
                    "sgc_conv3_wgrad_sparse"))
            else:
                self._timed("conv3_wgrad_objects", lambda: _lib.check(lib.sgc_conv3_wgrad(
                    _lib.ptr(dy3_bg), _lib.ptr(z_bf_maps), _lib.ptr(sl), n_maps, 0, ctypes.byref(slabs_n), st()), "sgc_conv3_wgrad"))
            zcol = None
            if Epad:
                # im2col + plain ping-pong TN GEMM.  (Rows of z gathered by the window list inside the GEMM block - no column
                # buffer, sgc_windows_wgrad_gather - measured 11.0 ms against 8.3 + 2.3 ms: the nine shifted re-reads of the z rows
                # by different N tiles cost more than the im2col pass; kept in the C-ABI, not used by the step.)
                if gather_w:
                    # the sparse launch gathers the real pairs' patches itself (gemm_tn_sp_kernel<2>): only the tail of the list behind
                    # them (per-object entries + the boundary tile, on the dense block) gets a patch copy
                    zcol = ws.get("zpatch_tail", max(Epad - e_spw, 16) * 16 * 512, torch.bfloat16)
                    self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2patch_f16_from(
                        _lib.ptr(ctx.z), _lib.ptr(gather), _lib.ptr(gn), e_spw, Epad - e_spw, _lib.ptr(zcol), st()), "sgc_windows_im2patch_f16_from"))
                elif TUNING.patch_wgrad:
                    # PATCH form: the 16 pixels of every listed window's input patch, read by the product at (own pixel + tap)
                    zcol = ws.get("zpatch", Epad * 16 * 512, torch.bfloat16)
                    if zb0:        # no bf16 copy of the real pairs' z: gather the f16 rows of the forward and convert
                        self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2patch_f16(_lib.ptr(ctx.z), _lib.ptr(gather), _lib.ptr(gn), Epad,
                                                                                                     _lib.ptr(zcol), st()), "sgc_windows_im2patch_f16"))
                    else:
                        self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2patch(_lib.ptr(z_bf), _lib.ptr(gather), _lib.ptr(gn), Epad,
                                                                                                 _lib.ptr(zcol), st()), "sgc_windows_im2patch"))
                else:
                    if zb0:
                        raise RuntimeError("the im2col form needs the bf16 copy of z that this forward did not write (TUNING changed between forward and backward)")
                    zcol = ws.get("zcol", Epad * 4 * 9 * 512, torch.bfloat16)
                    self._timed("im2col_windows", lambda: _lib.check(lib.sgc_windows_im2col(_lib.ptr(z_bf), _lib.ptr(gather), _lib.ptr(gn), Epad,
                                                                                           _lib.ptr(zcol), st()), "sgc_windows_im2col"))

        def auto_splits(k_rows, tiles=72):
            # host mirror of csrc/gemm_tn.h:tn_auto_splits (72 tiles of the [1024][4608] gradient): what a launch with splits = 0 writes
            nk, best = max(int(k_rows) >> 6, 1), 1
            for s_ in range(1, 65):
                if s_ > 1 and nk // s_ < 8:
                    break
                blocks = tiles * s_
                best = s_
                if blocks >= 256 and blocks * 100 >= ((blocks + 255) // 256) * 256 * 95:
                    break
            best = min(best, nk)
            per = (nk + best - 1) // best
            return (nk + per - 1) // per

        def wgrad_windows():
            # the second big GEMM of the window backward.  With ``TUNING.gemms_apart`` it is enqueued after the data-gradient GEMM
            # (the side stream then waits for it) and runs beside col2im / the pair contraction; the im2col above runs beside the
            # data-gradient GEMM.  Round 2 let the two GEMMs run side by side: 18.8 ms for the pair against 7.7 + 7.6 alone.
            with side():
                n_slabs = slabs_n.value
                e_sp = e_spw
                # slab capacity is checked BEFORE anything is launched into the 32-slab buffer (the launches' own counts are the mirror's)
                # K ranges per XCD (csrc/gemm_tn_sp.h, xcd_map 2): 32 ranges when the sparse part of the list has >= 2048 K tiles
                xcd_k = bool(TUNING.wgrad_xcd_k and Epad and e_sp >= 4096 and ((e_sp * 4) >> 6) >= 2048)
                sp_splits = 32 if xcd_k else auto_splits(e_sp * 4)
                need = n_slabs + ((sp_splits + (auto_splits((Epad - e_sp) * 4) if Epad > e_sp else 0)) if (Epad and e_sp >= 4096)
                                  else (auto_splits(Epad * 4) if Epad else 0))
                if need > SLAB_CAPACITY:
                    raise RuntimeError("split-K slabs of the conv3 weight gradient (%d) exceed the %d-slab buffer" % (need, SLAB_CAPACITY))
                if Epad and e_sp >= 4096:
                    # the real pairs' windows: their un-pooled gradient has ONE non-zero per window and channel (4 consecutive K indices)
                    # - the 2:4 pattern of the sparse matrix cores; packed straight from the pooled rows.  The per-object entries behind
                    # them (sums of several windows: dense) and the boundary tile stay on the dense block.
                    slabs_t = ctypes.c_int(0)
                    pack_a = ws.get("w3x_pack_a", (e_sp // 16) * 1024 * 64, torch.uint8)
                    pack_i = ws.get("w3x_pack_i", (e_sp // 16) * 1024 * 8, torch.uint8)
                    slx = sl[n_slabs * 1024 * 4608:]
                    if gather_w:
                        self._timed("conv3_wgrad_windows", lambda: _lib.check(lib.sgc_windows_wgrad_gather_sparse(
                            _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(dest), e_sp, _lib.ptr(ctx.z), _lib.ptr(pack_a), _lib.ptr(pack_i),
                            _lib.ptr(slx), -1 if xcd_k else 0, ctypes.byref(slabs_x), st()), "sgc_windows_wgrad_gather_sparse"))
                    else:
                        self._timed("conv3_wgrad_windows", lambda: _lib.check(lib.sgc_windows_wgrad_patch_sparse(
                            _lib.ptr(dy), _lib.ptr(ctx.am), _lib.ptr(gather), _lib.ptr(dest), e_sp, _lib.ptr(zcol), _lib.ptr(pack_a), _lib.ptr(pack_i),
                            _lib.ptr(slx), -1 if xcd_k else 0, ctypes.byref(slabs_x), st()), "sgc_windows_wgrad_patch_sparse"))
                    n_slabs += slabs_x.value
                    if Epad > e_sp:
                        slt = sl[n_slabs * 1024 * 4608:]
                        ztail = zcol if gather_w else zcol[e_sp * 16 * 512:]
                        self._timed("conv3_wgrad_windows_tail", lambda: _lib.check(lib.sgc_windows_wgrad_patch(
                            _lib.ptr(dy3x[e_sp * 4 * 1024:]), _lib.ptr(ztail), _lib.ptr(slt), (Epad - e_sp) * 4, 0,
                            ctypes.byref(slabs_t), st()), "sgc_windows_wgrad_patch"))
                        n_slabs += slabs_t.value
                elif Epad:
                    slx = sl[n_slabs * 1024 * 4608:]
                    self._timed("conv3_wgrad_windows", lambda: _lib.check((lib.sgc_windows_wgrad_patch if TUNING.patch_wgrad else lib.sgc_windows_wgrad)(
                        _lib.ptr(dy3x), _lib.ptr(zcol), _lib.ptr(slx), Epad * 4, 0, ctypes.byref(slabs_x), st()), "sgc_windows_wgrad"))
                    n_slabs += slabs_x.value
                assert n_slabs <= need, "split-K counts of the launches differ from their host mirror"
                dW3r = self._slab_sum(sl, 1024 * 4608, n_slabs)
                grads["conv3_1.weight"] = dW3r.view(1024, 3, 3, 512).permute(0, 3, 1, 2).contiguous()

        if not TUNING.gemms_apart:
            wgrad_windows()
        # ---- data gradients
        if lin is None:
            self._timed("conv3_dgrad_objects", lambda: _lib.check(lib.sgc_conv3_dgrad_pooled(
                _lib.ptr(dy_maps), _lib.ptr(am_maps), _lib.ptr(w["wd3"]), _lib.ptr(dz_maps), n_maps, st()), "sgc_conv3_dgrad_pooled"))
        else:
            self._timed("conv3_dgrad_objects", lambda: _lib.check(lib.sgc_conv3_dgrad(
                _lib.ptr(dy3_bg), _lib.ptr(w["wd3"]), _lib.ptr(dz_maps), n_maps, st()), "sgc_conv3_dgrad"))
        if Epad and TUNING.patch_dgrad:
            # PATCH form: the 16 pixels of every listed window's input patch leave the GEMM already summed over the taps (K = 1024 x
            # 1 / 2 per output element instead of 1024: 4.7 instead of 8.4 GB of stores per launch at the benchmark's size, and the
            # sum over a pair's windows reads 20 instead of 36 rows per window)
            slots = int(lib.sgc_windows_patch_slots())
            patch = ws.get("xpatch", Epad * slots * 512, torch.bfloat16)
            if e_spd:
                self._timed("conv3_dgrad_windows", lambda: _lib.check(lib.sgc_windows_dgrad_patches_sparse(
                    _lib.ptr(spa), _lib.ptr(spi), e_spd, _lib.ptr(w["w3sp"]), _lib.ptr(patch), st()), "sgc_windows_dgrad_patches_sparse"))
                if Epad > e_spd:          # the dense form's 20 rows per entry behind the sparse form's 16 rows per entry
                    self._timed("conv3_dgrad_windows_tail", lambda: _lib.check(lib.sgc_windows_dgrad_patches(
                        _lib.ptr(dy3x[e_spd * 4096:]), _lib.ptr(w["w3patch"]), _lib.ptr(patch[e_spd * 16 * 512:]), Epad - e_spd, st()),
                        "sgc_windows_dgrad_patches"))
            else:
                self._timed("conv3_dgrad_windows", lambda: _lib.check(lib.sgc_windows_dgrad_patches(_lib.ptr(dy3x), _lib.ptr(w["w3patch"]), _lib.ptr(patch), Epad, st()),
                                                                      "sgc_windows_dgrad_patches"))
            if TUNING.gemms_apart:
                wgrad_windows()                      # side stream: after the data-gradient GEMM, beside the patch sums / the contraction
            self._timed("col2im_windows", lambda: (
                _lib.check(lib.sgc_windows_patch_sum2(_lib.ptr(patch), e_spd, _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx),
                                                      _lib.ptr(sh["incl"]), P, _lib.ptr(dz), st()), "sgc_windows_patch_sum2"),
                _lib.check(lib.sgc_windows_patch_sum_objects2(_lib.ptr(patch), e_spd, _lib.ptr(ctx.bbox), n_obj, P, _lib.ptr(sh["incl"]), _lib.ptr(dz), st()),
                           "sgc_windows_patch_sum_objects2") if objects else None))
        elif Epad:
            col = ws.get("xcol", Epad * 4 * 9 * 512, torch.bfloat16)
            self._timed("conv3_dgrad_windows", lambda: _lib.check(lib.sgc_windows_dgrad_cols(_lib.ptr(dy3x), _lib.ptr(w["w3col"]), _lib.ptr(col), Epad * 4, st()),
                                                                  "sgc_windows_dgrad_cols"))
            if TUNING.gemms_apart:
                wgrad_windows()                      # side stream: after the data-gradient GEMM, beside col2im / the contraction
            self._timed("col2im_windows", lambda: (
                _lib.check(lib.sgc_windows_col2im(_lib.ptr(col), _lib.ptr(ctx.bbox), _lib.ptr(ctx.sub_idx), _lib.ptr(ctx.obj_idx), _lib.ptr(sh["incl"]),
                                                  P, _lib.ptr(dz), st()), "sgc_windows_col2im"),
                _lib.check(lib.sgc_windows_col2im_objects(_lib.ptr(col), _lib.ptr(ctx.bbox), n_obj, P, _lib.ptr(sh["incl"]), _lib.ptr(dz), st()),
                           "sgc_windows_col2im_objects") if objects else None))
        elif TUNING.gemms_apart:
            wgrad_windows()
        return dz

    # ------------------------------------------------------------------ two-stream backward
    def _side_chain(self):
        """Callable context manager that runs its body on this device's side stream, ordered after everything enqueued on the
        caller's stream so far; ``join()`` orders the caller's stream after the side stream.  Both are no-ops with
        ``TUNING.bwd_streams`` off."""
        import contextlib
        eng = self
        enabled = TUNING.bwd_streams

        class Chain:
            def __init__(self):
                self.main = torch.cuda.current_stream(eng.device)
                if enabled:
                    if getattr(eng, "_side_stream", None) is None:       # one per engine: image groups on concurrent lanes keep apart
                        eng._side_stream = torch.cuda.Stream(device=eng.device)
                    self.side = eng._side_stream
                    ev = torch.cuda.Event()
                    ev.record(self.main)
                    self.side.wait_event(ev)         # the side stream's previous work may not overtake buffers reused by this step

            @contextlib.contextmanager
            def __call__(self):
                if not enabled:
                    yield
                    return
                ev = torch.cuda.Event()
                ev.record(self.main)
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    yield

            def join(self):
                if enabled:
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.main.wait_event(ev)
        return Chain()

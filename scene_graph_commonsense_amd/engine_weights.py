"""Weight preparation of ``engine.RelHeadEngine``: the 16-bit compute copies in the GEMMs' layouts (forward: f16; backward: bf16, transposed /
patch / sparse forms), made by one gather + cast launch each (``TUNING.weight_kernels``; the torch chains are the reference they are tested against)."""
from __future__ import annotations

from .engine_core import *          # noqa: F401,F403  (TUNING, Workspace, PairOutputs, TrainContext, _lib, torch, np, ... - see engine_core.__all__)


class WeightsMixin:
    # ------------------------------------------------------------------ weights
    def load_weights(self, sd: Dict[str, torch.Tensor], fc1_sync=None):
        """Build the 16-bit compute copies (layouts of csrc/kernels_fwd.hip) from the f32 master weights.  ``fc1_sync``: called
        before ``fc1.weight`` is read (``Weights``: that copy is made at its first use in the step)."""
        cfg, dev = self.cfg, self.device
        g = lambda k: sd[k].detach().to(dev, torch.float32)
        w = self.w
        self._load_trunk_weights(sd, g, fc1_sync)
        self._load_head_weights(sd, g)

    def _permute_cast(self, src, dst, kind, dims, sstr, dstr=None, src_off=0, dst_off=0):
        """dst[dst_off + i.dstr] = cast(src[src_off + i.sstr]) over ``dims`` (``sgc_permute_cast``; dst contiguous when ``dstr`` is None)."""
        n = len(dims)
        if dstr is None:
            dstr, acc = [0] * n, 1
            for k in range(n - 1, -1, -1):
                dstr[k], acc = acc, acc * dims[k]
        _lib.check(self.lib.sgc_permute_cast(_lib.ptr(src), _lib.ptr(dst), kind, n, (ctypes.c_int * n)(*dims), (ctypes.c_long * n)(*sstr),
                                             (ctypes.c_long * n)(*dstr), _c_long(src_off), _c_long(dst_off), self._st()), "sgc_permute_cast")

    def _load_trunk_weights(self, sd, g, fc1_sync):
        w = self.w
        if TUNING.weight_kernels:
            # every layout below = one gather + cast launch from the f32 master (the torch forms in the else branch are the definition)
            w1r = self.ws.get("w1r", 2 * 128 * XC, torch.float16)                  # created zeroed; the channel padding stays zero
            c2, c3 = g("conv2_1.weight").contiguous(), g("conv3_1.weight").contiguous()
            w2r = self.ws.get("w2r", 2 * 512 * 1152, torch.float16)
            for r, name in enumerate(("conv1_1.weight", "conv1_2.weight")):
                self._permute_cast(g(name).contiguous(), w1r, 0, [128, 257], [257, 1], [XC, 1], dst_off=r * 128 * XC)
                self._permute_cast(c2, w2r, 0, [512, 2, 9, 64], [2304, 576, 1, 9], src_off=r * 128 * 9, dst_off=r * 512 * 1152)
            w3r = self.ws.get("w3r", 1024 * 4608, torch.float16)
            self._permute_cast(c3, w3r, 0, [1024, 8, 9, 64], [4608, 576, 1, 9])
            w["w1r"], w["w2r"], w["w3r"] = w1r.view(2, 128, XC), w2r.view(2, 512, 1152), w3r.view(1024, 4608)
            w["b1"] = torch.stack([g("conv1_1.bias"), g("conv1_2.bias")]).contiguous()
            w["cst"] = torch.tanh(w["b1"]).half().contiguous()                  # tanh(conv1(0)) outside the box
            w["b2"] = g("conv2_1.bias").contiguous()
            w["b3"] = g("conv3_1.bias").contiguous()
        else:
            self._load_trunk_weights_torch(g)

        fc1_param = sd["fc1.weight"]

        def make_w1p():
            if fc1_sync is not None:
                fc1_sync()
            fresh = getattr(self, "_w1p_fresh", None)
            if fresh is not None and fresh == (fc1_param.data_ptr(), fc1_param._version) and "w1p" in self.ws.bufs:
                return self.ws.bufs["w1p"][:fc1_param.numel()]       # written by the fused optimizer step (sgc_sgd_fc1_fused) for this version
            with torch.no_grad():
                return self._transpose_cast(g("fc1.weight").contiguous(), "w1p", torch.float16, 0, 4096, 16, 65536, 4096, 64, 65536, 64, 1024)
        w.defer("w1p", make_w1p)
        w["bf1"] = g("fc1.bias").contiguous()

    def _load_trunk_weights_torch(self, g):
        w = self.w
        w1r = self.ws.get("w1r", 2 * 128 * XC, torch.float16).view(2, 128, XC)     # created zeroed; the channel padding stays zero
        w1r[0, :, :257] = g("conv1_1.weight").view(128, 257).half()
        w1r[1, :, :257] = g("conv1_2.weight").view(128, 257).half()
        w["w1r"] = w1r
        w["b1"] = torch.stack([g("conv1_1.bias"), g("conv1_2.bias")]).contiguous()
        w["cst"] = torch.tanh(w["b1"]).half().contiguous()                  # tanh(conv1(0)) outside the box
        c2 = g("conv2_1.weight")
        w["w2r"] = torch.stack([conv_k_layout(c2[:, r * 128:(r + 1) * 128]) for r in (0, 1)]).half().contiguous()
        w["b2"] = g("conv2_1.bias").contiguous()
        w["w3r"] = conv_k_layout(g("conv3_1.weight")).half().contiguous()
        w["b3"] = g("conv3_1.bias").contiguous()

    def _load_head_weights(self, sd, g):
        """fc2 and the head: independent of hidden_dim / feature_size (fc1 always ends in 4096 features)."""
        cfg, w = self.cfg, self.w
        fc2 = g("fc2.weight")
        w["fc2_full"] = fc2
        if TUNING.weight_kernels:
            w2m = self.ws.get("w2m", 512 * 4096, torch.float16)
            self._permute_cast(fc2.contiguous(), w2m, 0, [512, 4096], [int(fc2.shape[1]), 1])
            w["w2m"] = w2m.view(512, 4096)
        else:
            w["w2m"] = fc2[:, :4096].half().contiguous()
        w["bf2"] = g("fc2.bias").contiguous()
        R = cfg.num_relations
        if cfg.hierarchical:
            rows = [g("fc3_1.weight"), g("fc3_2.weight"), g("fc3_3.weight"), g("fc5.weight"), g("fc4.weight")]
            bias = [g("fc3_1.bias"), g("fc3_2.bias"), g("fc3_3.bias"), g("fc5.bias"), g("fc4.bias")]
        else:
            rows = [g("fc3.weight"), g("fc4.weight")]
            bias = [g("fc3.bias"), g("fc4.bias")]
        Wc = self.ws.get("head_rows", 64 * 512, torch.float32).view(64, 512)         # created zeroed; rows beyond the head stay zero
        bc = self.ws.get("head_bias", 64, torch.float32)
        rc = torch.cat(rows)
        Wc[:rc.shape[0]] = rc
        bc[:rc.shape[0]] = torch.cat(bias)
        w["head_wt"] = Wc.t().contiguous()
        w["head_b"] = bc
        w["head_w"] = Wc                             # row-major copy for the head backward
        self.head_rows = rc.shape[0]

    # ------------------------------------------------------------------ helpers
    def _transpose_cast(self, src, name, dtype, kind, na, nb, sa_s, sb_s, ss_i, sa_d, sb_d, ds_j):
        dst = self.ws.get(name, src.numel(), dtype)
        _lib.check(self.lib.sgc_transpose_cast(_lib.ptr(src), _lib.ptr(dst), kind, na, nb, _c_long(sa_s), _c_long(sb_s),
                                               _c_long(ss_i), _c_long(sa_d), _c_long(sb_d), _c_long(ds_j), self._st()),
                   "sgc_transpose_cast")
        return dst

    # ====================================================================== training (forward + backward)
    def prep_bwd_weights(self, sd, fc1_sync=None):
        """bf16 transposed / flipped weight copies for the data-gradient GEMMs."""
        dev = self.device
        g = lambda k: sd[k].detach().to(dev, torch.float32)
        w = self.w
        if TUNING.weight_kernels:
            fc2 = w["fc2_full"].contiguous()
            w2mT = self.ws.get("w2mT", 4096 * 512, torch.bfloat16)
            self._permute_cast(fc2, w2mT, 1, [4096, 512], [1, int(fc2.shape[1])])
            w["w2mT"] = w2mT.view(4096, 512)
        else:
            w["w2mT"] = w["fc2_full"][:, :4096].t().contiguous().to(torch.bfloat16)
        self._prep_bwd_trunk_weights(sd, g, fc1_sync)

    # (own pixel q, tap k) combinations that reach coordinate c of a window's 4 x 4 input patch: see sgc_windows_dgrad_patches
    _PATCH_OPTS = staticmethod(lambda c: [(0, 0)] if c == 0 else ([(1, 2)] if c == 3 else [(0, c), (1, c - 1)]))

    def _prep_bwd_trunk_weights(self, sd, g, fc1_sync):
        w = self.w

        def make_w1pT():
            if fc1_sync is not None:
                fc1_sync()
            with torch.no_grad():
                return self._transpose_cast(g("fc1.weight").contiguous(), "w1pT", torch.bfloat16, 1, 64, 1024, 64 * 65536, 64, 65536,
                                            64, 4096, 1024 * 4096)
        w.defer("w1pT", make_w1pT)
        if not TUNING.weight_kernels:
            return self._prep_bwd_trunk_weights_torch(g)
        c2, c3 = g("conv2_1.weight").contiguous(), g("conv3_1.weight").contiguous()
        # conv3 with flipped taps and swapped channel roles, K order (chunk of 64 c_out, tap, c_out): the data gradient as a convolution
        wd3 = self.ws.get("wd3", 512 * 9216, torch.bfloat16)
        self._permute_cast(c3, wd3, 1, [512, 16, 9, 64], [9, 64 * 4608, -1, 4608], src_off=8)
        w["wd3"] = wd3.view(512, 9216)
        # [(tap, c_in)][c_out]: second operand of the column form of the data gradient over pair-specific windows
        # (only the column form reads it - ``TUNING.patch_dgrad`` off: made on first use, not on every step)
        def make_w3col():
            w3col = self.ws.get("w3col", 4608 * 1024, torch.bfloat16)
            with torch.no_grad():
                self._permute_cast(g("conv3_1.weight").contiguous(), w3col, 1, [9, 512, 1024], [1, 9, 4608])
            return w3col.view(4608, 1024)
        w.defer("w3col", make_w3col)
        # patch form of the same data gradient: per patch pixel pp = (py, px) the transposed tap matrices of its combinations, stacked along K
        opts = self._PATCH_OPTS
        d_off, d_ld, s_off, base = [], [], [], 0
        for py in range(4):
            for px in range(4):
                taps = [ky * 3 + kx for _, ky in opts(py) for _, kx in opts(px)]
                for j, t in enumerate(taps):
                    d_off.append(base + j * 1024); d_ld.append(len(taps) * 1024); s_off.append(t)
                base += 512 * len(taps) * 1024
        w3patch = self.ws.get("w3patch", base, torch.bfloat16)
        n = len(d_off)
        L = ctypes.c_long * n
        _lib.check(self.lib.sgc_segment_cast(_lib.ptr(c3), _lib.ptr(w3patch), 1, 512, 1024, _c_long(9), _c_long(4608), n, L(*d_off), L(*d_ld), L(*s_off),
                                             self._st()), "sgc_segment_cast")
        w["w3patch"] = w3patch
        # sparse form of the same data gradient (csrc/kernels_dgrad_sp.hip): per slot [512 c_in][(c_out, own pixel of the slot's set)]
        w3sp = self.ws.get("w3sp", 20 * 512 * 2048, torch.bfloat16)
        _lib.check(self.lib.sgc_windows_dgrad_sparse_weights(_lib.ptr(c3), _lib.ptr(w3sp), self._st()), "sgc_windows_dgrad_sparse_weights")
        w["w3sp"] = w3sp
        wd2 = self.ws.get("wd2", 2 * 128 * 4608, torch.bfloat16)
        for r in (0, 1):
            self._permute_cast(c2, wd2, 1, [128, 8, 9, 64], [9, 64 * 2304, -1, 2304], src_off=r * 128 * 9 + 8, dst_off=r * 128 * 4608)
        w["wd2"] = wd2.view(2, 128, 4608)

    def _prep_bwd_trunk_weights_torch(self, g):
        w = self.w
        w["wd3"] = conv_k_layout(g("conv3_1.weight").flip(2, 3).permute(1, 0, 2, 3)).to(torch.bfloat16).contiguous()
        # [(tap, c_in)][c_out]: second operand of the column form of the data gradient over pair-specific windows
        w["w3col"] = g("conv3_1.weight").permute(2, 3, 1, 0).reshape(9 * 512, 1024).to(torch.bfloat16).contiguous()
        # patch form of the same data gradient (sgc_windows_dgrad_patches): for every pixel pp = (py, px) of a window's 4 x 4 input patch
        # the tap matrices of the (own pixel q, tap k) combinations with q + k = pp, stacked along K: [512 c_in][combinations x 1024 c_out]
        w3 = g("conv3_1.weight")
        opts = self._PATCH_OPTS
        w["w3patch"] = torch.cat([torch.cat([w3[:, :, ky, kx].t() for _, ky in opts(py) for _, kx in opts(px)], dim=1).reshape(-1)
                                  for py in range(4) for px in range(4)]).to(torch.bfloat16).contiguous()
        c2 = g("conv2_1.weight")
        w["wd2"] = torch.stack([conv_k_layout(c2[:, r * 128:(r + 1) * 128].flip(2, 3).permute(1, 0, 2, 3))
                                for r in (0, 1)]).to(torch.bfloat16).contiguous()

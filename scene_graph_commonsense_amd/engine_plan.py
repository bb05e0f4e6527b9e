"""Shared-window planning of ``engine.RelHeadEngine``: per minibatch, which pooling windows of which pair are pair-specific (X), the
window lists / prefix sums / pixel rectangles the kernels of ``csrc/kernels_shared.hip`` walk, the window-major row space of the shared fc1."""
from __future__ import annotations

from .engine_core import *          # noqa: F401,F403  (TUNING, Workspace, PairOutputs, TrainContext, _lib, torch, np, ... - see engine_core.__all__)


class PlanMixin:
    FULL_PIXRECT = (16 << 5) | (16 << 15)          # packed pixel rectangle covering the whole 16x16 map

    def shared_plan(self, bbox, sub_idx, obj_idx, P, bound=None, keep=False, n_obj=0, n_img=0, objects=False, obj_img=None):
        """The window list of a pair list (``csrc/kernels_shared.hip``).  Pair index space: [P real pairs][2*n_obj pseudo-pairs
        (o, bg), (bg, o)][n_img all-background maps].  ``gather`` = pair*64 + window of every listed window (the X windows of the
        real pairs and - second level, ``objects`` - the windows R_o of the pseudo-pairs), ``incl`` inclusive prefix counts over
        the pair index space, ``pixrect`` the packed pixel rectangle in which a pair's z / routing codes / dz exist (whole map for
        pseudo-pairs without the second level and for the background maps).  ``bound`` = what the host knows
        (``model._shared_hint``): then nothing is read back; with ``keep`` (training: the backward's GEMMs need exact sizes) one
        sync reads the counts otherwise.  ``keep``: the lists live in buffers this engine owns."""
        lib = self.lib
        hint = bound if isinstance(bound, dict) else ({"windows": bound} if bound is not None else {})
        own = self.ws if keep else self.scratch
        n2 = 2 * n_obj
        Pt = P + n2 + n_img
        if (objects and TUNING.shared_linear and obj_img is not None and hint.get("linear_windows") and hint.get("windows") is not None
                and hint.get("object_windows") is not None and P > 0):
            return self._shared_plan_linear(bbox, sub_idx, obj_idx, P, hint, own, n_obj, n_img, obj_img)
        cnt = self.scratch.get("xw_count", Pt, torch.int32)
        pixrect = own.get("xw_pixrect", Pt, torch.int32)
        _lib.check(lib.sgc_shared_windows_count(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(cnt), _lib.ptr(pixrect),
                                                self._st()), "sgc_shared_windows_count")
        if Pt > P:
            cnt[P:].zero_()
            pixrect[P:].fill_(self.FULL_PIXRECT)
            if objects:
                _lib.check(lib.sgc_shared_objects_count(_lib.ptr(bbox), n_obj, _lib.ptr(cnt[P:]), _lib.ptr(pixrect[P:]), self._st()),
                           "sgc_shared_objects_count")
        incl = torch.cumsum(cnt, 0, dtype=torch.int32)
        gather = own.get("xw_gather", (P + (n2 if objects else 0)) * 64, torch.int32)
        _lib.check(lib.sgc_shared_windows_fill(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl), _lib.ptr(gather),
                                               self._st()), "sgc_shared_windows_fill")
        if objects:
            _lib.check(lib.sgc_shared_objects_fill(_lib.ptr(bbox), n_obj, P, _lib.ptr(incl), _lib.ptr(gather), self._st()), "sgc_shared_objects_fill")
        e_real, e_obj = hint.get("windows"), (hint.get("object_windows") if objects else 0)
        if (e_real is None or e_obj is None) and keep:
            e_real = int(incl[P - 1]) if P else 0
            e_obj = int(incl[Pt - 1]) - e_real
        exact = e_real is not None and e_obj is not None
        total = (e_real + e_obj) if exact else (P + (n2 if objects else 0)) * 64
        if not exact and e_real is not None:
            total = min(total, int(e_real) + n2 * 64)
        self._xw = (gather, incl[:P] if P else incl)
        self._xw_total = incl[Pt - 1:]
        self._xw_linear = None
        return dict(gather=gather, incl=incl, n_total=incl[Pt - 1:], pixrect=pixrect, bound=total, entries=total if exact else None,
                    entries_real=e_real if exact else None, window_entries=hint.get("per_window"), objects=objects, P=P, n_obj=n_obj,
                    n_img=n_img)

    def _shared_plan_linear(self, bbox, sub_idx, obj_idx, P, hint, own, n_obj, n_img, obj_img):
        """``shared_plan`` with the LINEAR pairs split off (csrc/kernels_shared.hip, sixth identity; full scenes only: the host knows
        every count).  Three lists over the same pair index space: ALL X windows (what fc1 multiplies: ``gather_all`` / ``incl_all``,
        the window-major destinations are per entry of this list), the CONV list (pairs that convolve their own windows + the
        per-object entries: under the plan's usual names ``gather`` / ``incl`` / ``n_total`` / ``pixrect``, so the conv3 kernels of
        both directions run on it unchanged) and the LINEAR list (``lin``: windows combined from per-object pre-activations)."""
        lib, dev = self.lib, self.device
        n2 = 2 * n_obj
        Pt = P + n2 + n_img
        e_all, e_obj, e_lin = int(hint["windows"]), int(hint["object_windows"]), int(hint["linear_windows"])
        cnt = self.scratch.get("xw_count3", 3 * Pt, torch.int32).view(3, Pt)
        pixrect = own.get("xw_pixrect", Pt, torch.int32)
        _lib.check(lib.sgc_shared_windows_count3(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(cnt[0]), _lib.ptr(cnt[1]),
                                                 _lib.ptr(cnt[2]), _lib.ptr(pixrect), self._st()), "sgc_shared_windows_count3")
        cnt[:, P:].zero_()
        pixrect[P:].fill_(self.FULL_PIXRECT)
        _lib.check(lib.sgc_shared_objects_count(_lib.ptr(bbox), n_obj, _lib.ptr(cnt[0, P:]), _lib.ptr(pixrect[P:]), self._st()),
                   "sgc_shared_objects_count")
        cnt[1, P:P + n2] = cnt[0, P:P + n2]
        if TUNING.plan_kernels:
            incl = own.get("xw_incl3", 3 * Pt, torch.int32).view(3, Pt)         # [3][Pt]: all / conv / linear
            _lib.check(lib.sgc_scan_rows(_lib.ptr(cnt), _lib.ptr(incl), 3, Pt, self._st()), "sgc_scan_rows")
            incl_all, incl_c, incl_l = incl[0], incl[1], incl[2]
        else:
            incl = torch.cumsum(cnt, 1, dtype=torch.int32)
            incl_all, incl_c, incl_l = incl[0].contiguous(), incl[1].contiguous(), incl[2].contiguous()
        # the host's counts size every buffer and list below; they are TRUSTED (no read-back) but checked: the device's own counts of
        # the boxes / pair lists actually passed must equal them (a scene whose boxes were edited after ``flatten_scene`` would
        # otherwise misplace rows silently).  Looked at by ``verify_checks`` at the next forward.
        self._post_check((incl_all[P - 1] != e_all) | (incl_l[P - 1] != e_lin) | ((incl_all[Pt - 1] - incl_all[P - 1]) != e_obj),
                         "shared-window plan: the scene's host-side window counts (windows=%d, linear_windows=%d, object_windows=%d) do not "
                         "match the boxes / pair lists on the device - was the scene modified after flatten_scene()?" % (e_all, e_lin, e_obj))
        e_c = e_all - e_lin
        gather_all = own.get("xw_gather_all", e_all + e_obj + 64, torch.int32)
        gather_c = own.get("xw_gather", e_c + e_obj + 64, torch.int32)
        gather_l = own.get("xw_gather_lin", e_lin + 64, torch.int32)
        _lib.check(lib.sgc_shared_windows_fill(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl_all), _lib.ptr(gather_all),
                                               self._st()), "sgc_shared_windows_fill")
        _lib.check(lib.sgc_shared_windows_fill_class(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl_c), _lib.ptr(gather_c),
                                                     1, self._st()), "sgc_shared_windows_fill_class")
        _lib.check(lib.sgc_shared_windows_fill_class(_lib.ptr(bbox), _lib.ptr(sub_idx), _lib.ptr(obj_idx), P, _lib.ptr(incl_l), _lib.ptr(gather_l),
                                                     2, self._st()), "sgc_shared_windows_fill_class")
        for inc, ga in ((incl_all, gather_all), (incl_c, gather_c)):
            _lib.check(lib.sgc_shared_objects_fill(_lib.ptr(bbox), n_obj, P, _lib.ptr(inc), _lib.ptr(ga), self._st()), "sgc_shared_objects_fill")
        # the linear windows ordered by (image, window) for the background side of the backward (stable: sums in list order):
        # one placement kernel instead of sort + searchsorted + gathers (sgc_bucket_place, bit-identical: tests/test_scene_gpu.py)
        # (two-level placement, sgc_bucket_place_seg: cost linear in the list at any minibatch size; the torch sort below is only the
        # reference the kernel is tested against, ``TUNING.plan_kernels`` off)
        if TUNING.plan_kernels:
            order = own.get("xw_lin_order", e_lin + 64, torch.int32)
            seg = own.get("xw_lin_seg", 64 * n_img + 1, torch.int32)
            self._bucket_place(gather_l, e_lin, sub_idx, obj_img, 1, 64 * n_img, None, order, seg, 1)
        else:
            code_l = gather_l[:e_lin].long()
            keys = obj_img.long()[sub_idx.long()[code_l >> 6]] * 64 + (code_l & 63)
            skeys, order = torch.sort(keys, stable=True)
            seg = torch.searchsorted(skeys, torch.arange(64 * n_img + 1, device=dev)).to(torch.int32).contiguous()
            order = order.to(torch.int32).contiguous()
        lin = dict(gather=gather_l, n=incl_l[P - 1:P].contiguous(), max=e_lin, order=order, seg=seg,
                   drow=own.get("xw_lin_drow", e_lin + 64, torch.int32))
        self._xw = (gather_all, incl_all[:P])
        self._xw_total = incl_c[Pt - 1:]
        self._xw_linear = (e_lin, e_obj)          # bench accounting: windows combined instead of convolved, per-object entries
        return dict(gather=gather_c, incl=incl_c, n_total=incl_c[Pt - 1:], pixrect=pixrect, bound=e_c + e_obj, entries=e_c + e_obj,
                    entries_real=e_c, window_entries=hint.get("per_window"), objects=True, P=P, n_obj=n_obj, n_img=n_img,
                    gather_all=gather_all, incl_all=incl_all, entries_all=e_all + e_obj, entries_real_all=e_all, lin=lin)

    def window_major_rows(self, plan, P, n2):
        """Window-major row space of the shared fc1 (``csrc/kernels_shared.hip``): device group offsets, tile -> group table and the
        row ``dest[e]`` of every listed window (X entries behind the per-object rows of their group; a pseudo-pair's own windows
        ARE per-object rows).  Per-window entry counts come from the host when it knows them (full scenes,
        ``DeviceScene.window_entries``); a pair subset costs one read-back."""
        from .pairs import window_major_layout
        dev = self.device
        split = "gather_all" in plan                  # linear pairs split off: the rows are those of the list of ALL X windows
        gather = plan["gather_all"] if split else plan["gather"]
        counts = plan.get("window_entries")
        if counts is None:
            E = int((plan["incl_all"] if split else plan["incl"])[P - 1]) if P else 0
            counts = torch.bincount((gather[:E] & 63).long(), minlength=64).cpu().numpy()
        E = int(np.asarray(counts).sum())
        goff, tile_group = window_major_layout(counts, n2)
        # ONE host-to-device copy for the four small tables: group offsets, first X row of every group, group ends, tile -> group
        gend_h = (goff[:64].astype(np.int64) + n2 + np.asarray(counts, dtype=np.int64)).astype(np.int32)
        tables = np.concatenate([goff.astype(np.int32), np.zeros(3, dtype=np.int32), (goff[:64].astype(np.int64) + n2).astype(np.int32), gend_h,
                                 np.asarray(tile_group, dtype=np.int32)])          # 68 + 64 + 64 + tiles: every table 16-byte aligned
        tab_d = torch.from_numpy(tables).to(dev)
        goff_d, xbase, gend, tile_group_d = tab_d[:65], tab_d[68:132], tab_d[132:196], tab_d[196:]
        Et = E
        if split:
            Et = plan["entries_all"]
        elif plan.get("objects"):
            Et = plan["entries"] if plan["entries"] is not None else int(plan["n_total"][0])
        dest = torch.empty(max(Et, 1), dtype=torch.int32, device=dev)
        # row of X entry e = first X row of its window's group + its rank among that window's entries in list order (what a stable sort
        # by window gives; sgc_bucket_place computes the ranks directly)
        kern = TUNING.plan_kernels
        if E > 0 and kern:
            self._bucket_place(gather, E, None, None, 0, 64, xbase, dest, None, 0)
        elif E > 0:
            cex = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
            skeys, order = torch.sort((gather[:E] & 63).long(), stable=True)
            base = torch.from_numpy(goff[:64].astype(np.int64) + n2 - cex).to(dev)
            dest[order] = (base[skeys] + torch.arange(E, device=dev)).int()
        if Et > E and kern:                          # the pseudo-pairs' own windows: row goff[w] + ps
            _lib.check(self.lib.sgc_window_rows_objects(_lib.ptr(gather[E:]), Et - E, _lib.ptr(goff_d), P, _lib.ptr(dest[E:]), self._st()),
                       "sgc_window_rows_objects")
        elif Et > E:
            code = gather[E:Et].long()
            dest[E:Et] = (goff_d[:64].long()[code & 63] + (code >> 6) - P).int()
        dest_conv = dest
        if split:
            # destination of every entry of the CONV list: the same (pair, window) sits at  first(pair, all) + its rank in the pair's
            # rectangle  in the list of all X windows (a pair is in the conv list with all of its windows or with none)
            Ec = plan["entries"]
            if kern:
                dest_conv = torch.empty(max(Ec, 1), dtype=torch.int32, device=dev)
                _lib.check(self.lib.sgc_window_rows_conv(_lib.ptr(plan["gather"]), Ec, _lib.ptr(plan["incl"]), _lib.ptr(plan["incl_all"]),
                                                         _lib.ptr(dest), _lib.ptr(dest_conv), self._st()), "sgc_window_rows_conv")
            else:
                pair_k = (plan["gather"][:Ec] >> 6).long()
                first = lambda inc: torch.cat([inc.new_zeros(1), inc[:-1]]).long()
                dest_conv = dest[torch.arange(Ec, device=dev) - first(plan["incl"])[pair_k] + first(plan["incl_all"])[pair_k]].contiguous()
        return dict(goff=goff_d, goff_host=goff, gend=gend, tile_group=tile_group_d, dest=dest, dest_conv=dest_conv,
                    rows=int(goff[64]), E=E, E_total=Et, n2=n2)

    def _bucket_place(self, codes, n, sub_idx, obj_img, img_key, n_keys, base, out, seg, mode):
        """Stable placement of a window list by key (``sgc_bucket_place_seg``: two-level kernels, scratch from the workspace)."""
        lib = self.lib
        lib.sgc_bucket_place_scratch_ints.restype = ctypes.c_long
        need = int(lib.sgc_bucket_place_scratch_ints(int(n), int(n_keys)))
        scratch = self.scratch.get("bucket_cnt", max(need, 1), torch.int32)
        _lib.check(lib.sgc_bucket_place_seg(_lib.ptr(codes), int(n), _lib.ptr(sub_idx), _lib.ptr(obj_img), int(img_key), int(n_keys), _lib.ptr(base),
                                            _lib.ptr(out), _lib.ptr(seg), int(mode), _lib.ptr(scratch), _c_long(need), self._st()),
                   "sgc_bucket_place_seg")

"""Row a2 on the device: the pair tables, directed targets, CSR lists, loss coefficients, label vectors and connectivity
statistics that the kernels of ``csrc/kernels_scene.hip`` build must equal - bit for bit, these are integers or the same double
arithmetic - what the host enumeration (pinned to the reference's loop order by ``tests/test_host_cpu.py``) gives."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(5, 2, 4, 1), (3,), (2, 2), (7, 1, 7), (20, 12, 20, 3, 9, 16, 2, 20), (1, 1), (64, 36, 64, 50)]


def _batch(nobj, seed, cfg=None, connect=0.4):
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = cfg or HeadConfig()
    return cfg, make_scene_batch(cfg, nobj, seed=seed, connect_frac=connect, edge_boxes=len(nobj) > 2)


@pytest.mark.parametrize("nobj", SHAPES)
def test_device_pair_tables_equal_host_enumeration(nobj):
    from scene_graph_commonsense_amd.engine import csr_by
    from scene_graph_commonsense_amd.pairs import enumerate_pairs, flatten_scene, normalise_boxes, pair_targets_fast, pair_targets
    cfg, batch = _batch(nobj, seed=sum(nobj))
    sc = flatten_scene(cfg, batch, "cuda:0")
    torch.cuda.synchronize()
    pidx = enumerate_pairs(nobj)
    P, n_obj = pidx.n_pairs, sum(nobj)
    assert sc.n_pairs == P and sc.n_steps == len(pidx.call_sizes) and sc.max_n == max(nobj)
    eq = lambda d, h: np.testing.assert_array_equal(d.cpu().numpy(), np.asarray(h))
    eq(sc.sub_idx, pidx.sub); eq(sc.obj_idx, pidx.obj); eq(sc.step, pidx.step); eq(sc.image, pidx.image)
    eq(sc.img_ptr, pidx.obj_offset)
    eq(sc.obj_img, np.concatenate([np.full(k, i) for i, k in enumerate(nobj)]))
    eq(sc.step_ptr, np.concatenate([[0], np.cumsum(pidx.call_sizes)]))
    pid = np.full((n_obj, max(max(nobj), 1)), -1, dtype=np.int32)
    pid[pidx.sub, pidx.obj - pidx.obj_offset[pidx.image]] = np.arange(P, dtype=np.int32)
    eq(sc.pid, pid)
    for (ptr_d, lst_d), idx in ((sc.sub_csr, pidx.sub), (sc.obj_csr, pidx.obj)):
        ptr, order = csr_by(idx, n_obj)
        eq(ptr_d, ptr); eq(lst_d, order)
    directed = pair_targets_fast(batch.relationships, batch.subj_or_obj, pidx)
    eq(sc.directed, directed)
    if P <= 400:
        d2, raw = pair_targets(batch.relationships, batch.subj_or_obj, pidx)
        eq(sc.directed, d2); eq(sc.raw_target, raw)
    eq(sc.bbox, np.concatenate([normalise_boxes(b, cfg.feature_size) for b in batch.bbox]))
    eq(sc.cats, torch.cat(batch.categories).numpy())


def test_box_normalisation_vectorised_equals_slice_semantics():
    from scene_graph_commonsense_amd.pairs import normalise_boxes, slice_norm
    g = torch.Generator().manual_seed(1)
    b = (torch.rand(500, 4, generator=g) - 0.3) * 60
    b[:50] = b[:50].round()
    ref = np.array([[slice_norm(int(v), 32) for v in row] for row in b.tolist()], dtype=np.int32)
    np.testing.assert_array_equal(normalise_boxes(b, 32), ref)
    bi = torch.randint(-40, 80, (300, 4), generator=g, dtype=torch.int32)
    ref = np.array([[slice_norm(int(v), 32) for v in row] for row in bi.tolist()], dtype=np.int32)
    np.testing.assert_array_equal(normalise_boxes(bi, 32), ref)


@pytest.mark.parametrize("hier", [True, False])
def test_loss_coefficient_kernel_is_the_host_arithmetic(hier):
    from scene_graph_commonsense_amd.engine import RelHeadEngine, loss_coefficients
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, predicate_counts
    cfg = HeadConfig(hierarchical=hier)
    for nobj, connect in (((6, 3, 5), 0.5), ((20, 20, 11), 0.05), ((4, 4), 0.0), ((9,), 1.0)):
        _, batch = _batch(nobj, seed=len(nobj) + int(connect * 10), cfg=cfg, connect=connect)
        sc = flatten_scene(cfg, batch, "cuda:0")
        counts = predicate_counts(cfg).numpy()
        cw = (1 - counts / counts.sum()).astype(np.float32)
        eng = RelHeadEngine(cfg, "cuda:0")
        for lam_c, lam_nc in ((0.1, 1.0), (0.5, 0.25)):
            got = eng.loss_coefficients_device(sc.step_ptr, sc.n_steps, sc.directed, torch.from_numpy(cw).cuda(), lam_c, lam_nc)
            ref = loss_coefficients(cfg, sc.pidx.step, sc.n_steps, sc.directed.cpu().numpy().astype(np.int64), cw, lam_c, lam_nc)
            for g_, r_ in zip(got, ref):
                np.testing.assert_array_equal(g_.cpu().numpy(), r_)


def test_label_vectors_and_their_gradient():
    from scene_graph_commonsense_amd import _lib
    from scene_graph_commonsense_amd.engine import RelHeadEngine
    from scene_graph_commonsense_amd.synthetic import HeadConfig
    g = torch.Generator().manual_seed(5)
    for cfg in (HeadConfig(), HeadConfig(dataset="oiv6", num_classes=601, num_super_classes=0, num_geometric=4, num_possessive=2, num_semantic=24)):
        eng = RelHeadEngine(cfg, "cuda:0")
        C, S = cfg.num_classes, cfg.num_super_classes
        W = torch.randn(512, 4096 + cfg.label_dim, generator=g).cuda()
        eng.w["fc2_full"] = W
        n_obj = 37
        cats = torch.randint(0, C, (n_obj,), generator=g).cuda()
        cats[1] = cats[0]
        mh = None
        if cfg.dataset == "vg":
            mh = torch.zeros(n_obj, S)
            for o in range(n_obj):
                mh[o, int(torch.randint(0, S, (1,), generator=g))] += 1
                if o % 3 == 0:
                    mh[o, int(torch.randint(0, S, (1,), generator=g))] += 1
            mh = mh.cuda()
        lsub, lobj = eng.label_vectors(cats, mh)
        ref_s = W[:, 4096 + cats].t().double()
        ref_o = W[:, 4096 + C + cats].t().double()
        if mh is not None:
            ref_s = ref_s + mh.double() @ W[:, 4096 + 2 * C:4096 + 2 * C + S].t().double()
            ref_o = ref_o + mh.double() @ W[:, 4096 + 2 * C + S:4096 + 2 * C + 2 * S].t().double()
        assert (lsub.double() - ref_s).abs().max() <= 1e-5 and (lobj.double() - ref_o).abs().max() <= 1e-5
        dls, dlo = torch.randn(n_obj, 512, generator=g).cuda(), torch.randn(n_obj, 512, generator=g).cuda()
        gW = torch.full_like(W, float("nan"))
        _lib.check(eng.lib.sgc_label_grads(_lib.ptr(dls), _lib.ptr(dlo), _lib.ptr(cats), _lib.ptr(cats), _lib.ptr(mh), _lib.ptr(mh), n_obj, C,
                                           S if mh is not None else 0,
                                           _lib.ptr(gW), int(W.shape[1]), 4096, _lib.stream_ptr()), "sgc_label_grads")
        ref = torch.zeros(512, cfg.label_dim, dtype=torch.float64, device="cuda")
        ref[:, :C].index_add_(1, cats, dls.t().double())
        ref[:, C:2 * C].index_add_(1, cats, dlo.t().double())
        if mh is not None:
            ref[:, 2 * C:2 * C + S] = dls.t().double() @ mh.double()
            ref[:, 2 * C + S:] = dlo.t().double() @ mh.double()
        assert torch.isnan(gW[:, :4096]).all()                            # only the label columns are written
        assert (gW[:, 4096:].double() - ref).abs().max() <= 1e-4


def test_connectivity_statistics_match_the_reference_formulas():
    """train_utils.py:66-87 / :176-184 evaluated literally per direction-step on the host against the one-pass kernel."""
    from scene_graph_commonsense_amd.engine import RelHeadEngine
    from scene_graph_commonsense_amd.pairs import flatten_scene
    cfg, batch = _batch((6, 4, 5), seed=3, connect=0.5)
    sc = flatten_scene(cfg, batch, "cuda:0")
    eng = RelHeadEngine(cfg, "cuda:0")
    g = torch.Generator().manual_seed(0)
    conn = torch.randn(sc.n_pairs, generator=g) * 2
    conn[::7] = 0.0                                                          # sigmoid == 0.5 exactly: predicted, but rounds to 0
    included = (torch.rand(sc.n_pairs, generator=g) < 0.8)
    directed, raw = sc.directed.cpu(), sc.raw_target.cpu()
    for inc in (None, included):
        got = eng.connectivity_stats(conn.cuda(), sc.directed, sc.raw_target, None if inc is None else inc.to(torch.uint8).cuda()).cpu().tolist()
        ref = [0, 0, 0, 0, 0]
        start = sc.step_ptr.cpu().tolist()
        for t in range(sc.n_steps):
            rows = [p for p in range(start[t], start[t + 1]) if inc is None or bool(inc[p])]
            if not rows:
                continue
            rows = torch.tensor(rows)
            c, d, r = conn[rows], directed[rows], raw[rows]
            connected = torch.where(d != -1)[0]
            pred = torch.nonzero(torch.sigmoid(c) >= 0.5).flatten()
            ref[0] += int((d == -1).sum()); ref[1] += len(connected); ref[2] += len(pred)
            ref[3] += int(torch.sum(r[pred] != -1))
            if len(connected) > 0:
                ref[4] += int(torch.sum(torch.round(torch.sigmoid(c[connected]))))
        assert got == ref, (got, ref)


def test_scene_edited_after_flattening_is_detected():
    """the shared-window plan trusts the scene's HOST-side window counts (no read-back in the step).  A scene whose boxes
    were changed after ``flatten_scene`` no longer matches them; the device's own counts are compared with the host's behind the
    step's work and the mismatch surfaces at the next call (``RelHeadEngine.verify_checks``) instead of silently misplaced rows."""
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=2))
    model.eval()
    batch = make_scene_batch(cfg, (9, 8), seed=11, connect_frac=0.3)
    sc = flatten_scene(cfg, batch, "cuda:0")
    if not sc.linear_windows:
        pytest.skip("no linear pair in this scene: the plan with host counts is not taken")
    model.training_step(sc, batch.relationships, batch.subj_or_obj)
    model.engine().verify_checks(block=True)                              # consistent scene: nothing to report
    sc.bbox[0] = torch.tensor([3, 4, 3, 4], dtype=torch.int32, device="cuda")      # shrink one box: FEWER windows than the host counted
    model.zero_grad(set_to_none=True)
    model.training_step(sc, batch.relationships, batch.subj_or_obj)
    with pytest.raises(RuntimeError, match="flatten_scene"):
        model.engine().verify_checks(block=True)


@pytest.mark.parametrize("n,n_img", [(1, 1), (4097, 2), (50000, 8), (237268, 8)])
def test_bucket_placement_equals_a_stable_sort(n, n_img):
    """``sgc_bucket_place`` (the row plan of conv3 / fc1 over shared windows: ranks among equal keys in list order) against
    ``torch.sort(stable=True)`` + ``searchsorted``, both modes, bit for bit."""
    import ctypes
    from scene_graph_commonsense_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n)
    n_obj, P = 16 * n_img, 3000 * n_img
    obj_img = torch.sort(torch.randint(0, n_img, (n_obj,), generator=g))[0].int().cuda()
    sub_idx = torch.randint(0, n_obj, (P,), generator=g).int().cuda()
    pair = torch.sort(torch.randint(0, P, (n,), generator=g))[0]
    codes = (pair * 64 + torch.randint(0, 64, (n,), generator=g)).int().cuda()
    # mode 0: destination = base[window] + rank among the window's entries
    base = (torch.arange(64) * 1000003 % 999983).int().cuda()
    out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sgc_bucket_place(_lib.ptr(codes), n, None, None, 0, 64, _lib.ptr(base), _lib.ptr(out), None, 0, _lib.stream_ptr()), "sgc_bucket_place")
    keys = (codes & 63).long()
    skeys, order = torch.sort(keys, stable=True)
    first = torch.searchsorted(skeys, torch.arange(64, device="cuda"))
    ref = torch.empty(n, dtype=torch.int64, device="cuda")
    ref[order] = base.long()[skeys] + torch.arange(n, device="cuda") - first[skeys]
    assert torch.equal(out.long(), ref)
    # mode 1: entries ordered by (image of the pair's subject, window), stable, + segment starts
    nk = 64 * n_img
    order1 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    seg = torch.full((nk + 1,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sgc_bucket_place(_lib.ptr(codes), n, _lib.ptr(sub_idx), _lib.ptr(obj_img), 1, nk, None, _lib.ptr(order1), _lib.ptr(seg), 1,
                                    _lib.stream_ptr()), "sgc_bucket_place")
    keys1 = obj_img.long()[sub_idx.long()[codes.long() >> 6]] * 64 + (codes.long() & 63)
    sk, ord_ref = torch.sort(keys1, stable=True)
    seg_ref = torch.searchsorted(sk, torch.arange(nk + 1, device="cuda"))
    assert torch.equal(order1.long(), ord_ref) and torch.equal(seg.long(), seg_ref)
    # the two-level kernels the engine uses (a workgroup per (key, segment of 8192 entries)): the same tables
    lib.sgc_bucket_place_scratch_ints.restype = ctypes.c_long
    need = int(lib.sgc_bucket_place_scratch_ints(n, nk))
    scratch = torch.full((max(need, 1),), -7, dtype=torch.int32, device="cuda")
    out2 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sgc_bucket_place_seg(_lib.ptr(codes), n, None, None, 0, 64, _lib.ptr(base), _lib.ptr(out2), None, 0, _lib.ptr(scratch),
                                        ctypes.c_long(need), _lib.stream_ptr()), "sgc_bucket_place_seg")
    assert torch.equal(out2, out)
    order2 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    seg2 = torch.full((nk + 1,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sgc_bucket_place_seg(_lib.ptr(codes), n, _lib.ptr(sub_idx), _lib.ptr(obj_img), 1, nk, None, _lib.ptr(order2), _lib.ptr(seg2), 1,
                                        _lib.ptr(scratch), ctypes.c_long(need), _lib.stream_ptr()), "sgc_bucket_place_seg")
    assert torch.equal(order2, order1) and torch.equal(seg2, seg)
    # an empty list: every segment start 0
    seg3 = torch.full((nk + 1,), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.sgc_bucket_place_seg(_lib.ptr(codes), 0, _lib.ptr(sub_idx), _lib.ptr(obj_img), 1, nk, None, _lib.ptr(order2), _lib.ptr(seg3), 1,
                                        _lib.ptr(scratch), ctypes.c_long(need), _lib.stream_ptr()), "sgc_bucket_place_seg")
    assert int(seg3.abs().sum()) == 0


def test_row_plan_by_kernels_equals_the_torch_form_at_benchmark_size():
    """The window-major row plan of an 8 x 64 scene (237 k pair-specific windows, linear pairs split off) built by the placement
    kernels (``TUNING.plan_kernels``, the default) against the sort / searchsorted / gather form of rounds 2-3: every table bit for bit."""
    from scene_graph_commonsense_amd.engine import RelHeadEngine, tuning
    from scene_graph_commonsense_amd.model import _shared_hint
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    batch = make_scene_batch(cfg, [64] * 8, seed=1000, connect_frac=0.02)
    sc = flatten_scene(cfg, batch, "cuda:0")
    eng = RelHeadEngine(cfg, "cuda:0")
    P, n_obj, n_img = sc.n_pairs, int(sc.obj_img.shape[0]), 8
    got = {}
    for kern in (True, False):
        with tuning(plan_kernels=kern):
            plan = eng.shared_plan(sc.bbox, sc.sub_idx, sc.obj_idx, P, _shared_hint(sc), keep=True, n_obj=n_obj, n_img=n_img, objects=True,
                                   obj_img=sc.obj_img)
            assert "lin" in plan and plan["lin"]["max"] > 0
            wm = eng.window_major_rows(plan, P, 2 * n_obj)
            torch.cuda.synchronize()
            got[kern] = dict(dest=wm["dest"][:wm["E_total"]].clone(), dest_conv=wm["dest_conv"][:plan["entries"]].clone(), goff=wm["goff"].clone(),
                             gend=wm["gend"].clone(), tile_group=wm["tile_group"].clone(), order=plan["lin"]["order"][:plan["lin"]["max"]].clone(),
                             seg=plan["lin"]["seg"].clone(), incl=plan["incl"].clone(), incl_all=plan["incl_all"].clone(),
                             n_lin=plan["lin"]["n"].clone(), n_total=plan["n_total"].clone())
    for k in got[True]:
        assert torch.equal(got[True][k].long(), got[False][k].long()), k

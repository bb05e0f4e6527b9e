"""conv3 over shared windows (csrc/kernels_shared.hip) against the plain per-pair conv3: the two must agree BIT FOR BIT - a
window outside an object's rectangle sees exactly the inputs the per-object pseudo-pair sees, in the same arithmetic - on boxes
that include full-image, 1x1, border-touching, empty and identical boxes, ragged images, pair subsets and the training outputs
(bf16 copy and routing codes the backward reads)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _always_shared(monkeypatch):
    """The edge-case boxes of these tests (full-image boxes next to tiny ones) make most windows pair-specific; the engine would
    switch such a scene to the per-pair kernels (``shared_conv3_enabled``).  Here the shared path is the subject."""
    from scene_graph_commonsense_amd import engine
    monkeypatch.setattr(engine.TUNING, "shared_max_fraction", 2.0)
    # the bit-for-bit statements of this file are about windows that are CONVOLVED; the linear pairs' windows (combined from three
    # per-object pre-activations: f32 round-off apart) have their own test, test_linear_pairs_*
    monkeypatch.setattr(engine.TUNING, "shared_linear", False)


def _model(cfg, seed=1):
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.synthetic import make_state_dict
    m = BayesianRelationClassifier(cfg.args()).cuda()
    m.load_state_dict(make_state_dict(cfg, seed=seed, head_gain=4.0))
    m.eval()
    return m


def _edge_boxes(batch):
    """Overwrite the first boxes of every image with the edge cases."""
    special = torch.tensor([[0, 32, 0, 32], [5, 6, 7, 8], [0, 3, 0, 2], [29, 32, 30, 32], [10, 10, 4, 9], [12, 20, 12, 20],
                            [12, 20, 12, 20], [0, 32, 15, 17], [3, 4, 0, 32], [-4, 6, -2, 9], [30, 40, 28, 36], [9, 3, 5, 8]])
    # the last three: negative starts (Python slice semantics: counted from the end), stops beyond the grid, a reversed (empty) box
    for b in batch.bbox:
        k = min(len(special), b.shape[0])
        b[:k] = special[:k].to(b.dtype)
    return batch


_FIELD = {"SGC_SHARED_CONV3": "shared_conv3", "SGC_SHARED_FC1": "shared_fc1", "SGC_SHARED_OBJECTS": "shared_objects",
          "SGC_SHARED_BWD": "shared_bwd"}


def _with_env(env, fn):
    """Run ``fn`` with switches of ``engine.TUNING`` overridden (keys: the historical names of the switches, values "0" / "1")."""
    from scene_graph_commonsense_amd import engine
    with engine.tuning(**{_FIELD[k]: v != "0" for k, v in env.items()}):
        return fn()


def _poison(eng):
    """NaN into every per-pair buffer the shared path writes only partly (the pair expansion next to the pair-specific windows,
    dz likewise): a read outside what was written shows up in the results instead of finding the previous run's values."""
    for ws in (eng.ws, eng.scratch):
        for name, t in ws.bufs.items():
            if name in ("z_pad", "z_pad_bf"):
                v = t[:t.numel() // (18 * 18 * 512) * (18 * 18 * 512)].view(-1, 18, 18, 512)
                v[:, 1:17, 1:17, :] = float("nan")           # the halo has to stay zero
            elif name == "amz":
                t.fill_(0x44)                                # "no route" everywhere
            elif name in ("dz", "xcol", "zcol", "dy3x"):
                t.view(torch.int16).fill_(0x7FC0)            # bf16 NaN
    torch.cuda.synchronize()


def _with(flag, fn):
    """conv3 shared on / off with fc1 as the one [pairs, 65536] GEMM (the bit-for-bit statements are about conv3)."""
    return _with_env({"SGC_SHARED_CONV3": flag, "SGC_SHARED_FC1": "0"}, fn)


@pytest.mark.parametrize("nobj,edge", [([9, 4, 12], True), ([36] * 3, False), ([64] * 2, True)])
def test_forward_is_bit_identical_to_the_per_pair_convolution(nobj, edge):
    from scene_graph_commonsense_amd.pairs import count_shared_windows, flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    model = _model(cfg)
    batch = make_scene_batch(cfg, nobj, seed=5)
    if edge:
        batch = _edge_boxes(batch)
    sc = flatten_scene(cfg, batch, "cuda:0")
    eng = model.refresh_weights()
    P = sc.n_pairs

    def run():
        out = model.forward_pairs(sc)
        torch.cuda.synchronize()
        return out, eng.ws.bufs["y"][:P * 65536].clone()
    o0, y0 = _with("0", run)
    _poison(eng)
    o1, y1 = _with("1", run)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    for a, b in ((o0.relation, o1.relation), (o0.connectivity, o1.connectivity), (o0.hidden, o1.hidden), (o0.cand_pred, o1.cand_pred)):
        assert torch.equal(a, b)
    # the device list of pair-specific windows has the length the host predicted
    gather, incl = eng._xw
    n = int(incl[-1])
    assert n == sc.shared_windows == count_shared_windows(sc.bbox.cpu().numpy(), sc.img_ptr.cpu().numpy())
    g = gather[:n].cpu().numpy()
    assert (np.diff(g) > 0).all() and g.max(initial=0) < P * 64
    if not edge:
        assert n < 0.3 * P * 64            # the point of it: most windows are shared


def test_pair_subset_and_overlap_filtered_evaluation():
    from scene_graph_commonsense_amd.pair_loop import evaluate_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    model = _model(cfg)
    batch = _edge_boxes(make_scene_batch(cfg, [14, 9], seed=8, connect_frac=0.3))

    def run():
        _, out, _, _ = evaluate_minibatch(model, batch, overlap_filtering=True, skip_filtered=True)
        torch.cuda.synchronize()
        return out
    a, b = _with("0", run), _with("1", run)
    assert torch.equal(a.relation, b.relation) and torch.equal(a.cand_conf, b.cand_conf) and torch.equal(a.cand_pred, b.cand_pred)


@pytest.mark.parametrize("nobj,edge,cfrac", [([20, 11, 16], True, 0.2), ([36] * 3, False, 0.03), ([1, 2, 5], True, 0.5)])
def test_training_step_shared_forward_is_bit_identical_and_shared_backward_agrees(nobj, edge, cfrac):
    """Forward: what the backward reads (bf16 copy, routing codes) and the loss are bit-identical, and with the per-pair backward
    (SGC_SHARED_BWD=0) so is every gradient.  The shared backward sums gradient rows per object BEFORE the conv3 backward and
    rounds the sums to bf16 once (instead of rounding every pair's rows), so it agrees to bf16 round-off, not bit for bit."""
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    model = _model(cfg)
    batch = make_scene_batch(cfg, nobj, seed=9, connect_frac=cfrac)
    if edge:
        batch = _edge_boxes(batch)
    sc = flatten_scene(cfg, batch, "cuda:0")
    eng = model.refresh_weights(backward=True)
    P = sc.n_pairs

    def run():
        model.zero_grad(set_to_none=True)
        loss = model.training_step(sc)
        torch.cuda.synchronize()
        return (float(loss), eng.ws.bufs["y_bf"][:P * 65536].clone(), eng.ws.bufs["argmax"][:P * 65536].clone(),
                {n: p.grad.clone() for n, p in model.named_parameters()})
    l0, yb0, am0, g0 = _with_env({"SGC_SHARED_CONV3": "0", "SGC_SHARED_FC1": "0"}, run)
    l1, yb1, am1, g1 = _with_env({"SGC_SHARED_CONV3": "1", "SGC_SHARED_BWD": "0", "SGC_SHARED_FC1": "0"}, run)
    _poison(eng)
    l2, yb2, am2, g2 = _with_env({"SGC_SHARED_CONV3": "1", "SGC_SHARED_BWD": "1", "SGC_SHARED_FC1": "0"}, run)
    for n in g2:
        assert torch.isfinite(g2[n]).all(), n
    assert l0 == l1 == l2
    assert torch.equal(yb0.view(torch.int16), yb1.view(torch.int16)) and torch.equal(am0, am1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    worst = {}
    for n in g0:
        ref = g0[n].double()
        err = float((g2[n].double() - ref).norm() / ref.norm().clamp(min=1e-30))
        worst[n] = err
        # layers above conv3 do not see the change at all; below it the per-object sums are rounded once to bf16 (2^-9)
        tol = 0.0 if n.startswith(("fc", "conv3_1.bias")) and n != "conv3_1.bias" else 6e-3
        if n.startswith("fc"):
            assert torch.equal(g2[n], g0[n]), n
        else:
            assert err <= tol, (n, err)
    print({k: "%.2e" % v for k, v in worst.items() if v > 0})
    # ---- the whole shipped default: fc1 over window-major rows as well.  h1 differs by f32 round-off before its f16 rounding, so
    # a few ReLU / dropout-scaled values move by one f16 ulp and every gradient sees that; nothing is bit-equal any more
    _poison(eng)
    for name in ("gwm", "dywm", "ywm_bf"):
        for ws in (eng.ws, eng.scratch):
            if name in ws.bufs:
                ws.bufs[name].view(torch.int16).fill_(0x7FC0)
    l3, _, _, g3 = _with_env({"SGC_SHARED_CONV3": "1", "SGC_SHARED_BWD": "1", "SGC_SHARED_FC1": "1"}, run)
    assert abs(l3 - l0) <= 2e-5 * abs(l0), (l3, l0)
    worst3 = {}
    for n in g0:
        assert torch.isfinite(g3[n]).all(), n
        ref = g0[n].double()
        worst3[n] = float((g3[n].double() - ref).norm() / ref.norm().clamp(min=1e-30))
        assert worst3[n] <= 8e-3, (n, worst3[n])
    print("fc1 shared:", {k: "%.2e" % v for k, v in worst3.items()})
    # ---- without the second level (every pseudo-pair a whole conv3 map): identical forward, gradients to bf16 round-off
    _poison(eng)
    l4, _, _, g4 = _with_env({"SGC_SHARED_CONV3": "1", "SGC_SHARED_BWD": "1", "SGC_SHARED_FC1": "1", "SGC_SHARED_OBJECTS": "0"}, run)
    assert l4 == l3
    for n in g3:
        assert torch.isfinite(g4[n]).all(), n
        e = float((g4[n].double() - g3[n].double()).norm() / g3[n].double().norm().clamp(min=1e-30))
        assert e <= 6e-3, (n, e)


@pytest.mark.parametrize("nobj,edge", [([9, 4, 12], True), ([36] * 3, False), ([64] * 2, True), ([1, 3], True)])
def test_fc1_over_window_major_rows_matches_the_one_gemm_form(nobj, edge):
    """fc1 as grouped GEMM + per-object prefix sums + per-pair assembly is the same sum in another order: f32 round-off before the
    f16 rounding of h1, far inside the 1e-3 parity tolerance of the forward (which tests/test_forward_gpu.py checks against the
    reference's goldens with this path on)."""
    from scene_graph_commonsense_amd.pair_loop import evaluate_minibatch
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    model = _model(cfg)
    batch = make_scene_batch(cfg, nobj, seed=11)
    if edge:
        batch = _edge_boxes(batch)
    sc = flatten_scene(cfg, batch, "cuda:0")
    eng = model.refresh_weights()
    P = sc.n_pairs

    def run():
        out = model.forward_pairs(sc)
        torch.cuda.synchronize()
        return out, eng.ws.bufs["h1"][:P * 4096].clone().float()
    o0, h0 = _with_env({"SGC_SHARED_FC1": "0"}, run)
    for name in ("h1", "owm", "oxh", "fc1_S", "ywm"):
        for ws in (eng.ws, eng.scratch):
            if name in ws.bufs:
                ws.bufs[name].view(torch.int16).fill_(0x7E00 if name in ("h1", "ywm", "oxh") else -1)      # NaN bit patterns
    from scene_graph_commonsense_amd import engine as _engine
    with _engine.tuning(fc1_x16=False):                        # every product row f32 (the form of rounds 3-5)
        o1, h1 = _with_env({"SGC_SHARED_FC1": "1"}, run)
    assert torch.isfinite(h1).all()
    # second level (pseudo-pairs computed on their own windows only, the rest from the images' background maps): the same bits
    with _engine.tuning(fc1_x16=False):
        o2, h2 = _with_env({"SGC_SHARED_FC1": "1", "SGC_SHARED_OBJECTS": "0"}, run)
    assert torch.equal(h1, h2) and torch.equal(o1.relation, o2.relation) and torch.equal(o1.cand_pred, o2.cand_pred)
    scale = float(h0.abs().max())
    assert float((h1 - h0).abs().max()) <= 2e-3 * scale, (float((h1 - h0).abs().max()), scale)     # one f16 ulp at the top of the range
    assert float((h1 - h0).norm() / h0.norm().clamp(min=1e-30)) <= 2e-4
    # engine.TUNING.fc1_x16 (the default since round 6): the pair-specific products pass through f16 rows - ~7 of a pair's ~20 addends
    # carry 2^-12 each before the sum's own f16 rounding; measured 2.2e-4 here against 1.6e-4 with f32 rows
    assert _engine.TUNING.fc1_x16
    o3, h3 = _with_env({"SGC_SHARED_FC1": "1"}, run)
    with _engine.tuning(shared_objects=False):
        o4, h4 = _with_env({"SGC_SHARED_FC1": "1"}, run)
    assert torch.equal(h3, h4) and torch.equal(o3.relation, o4.relation)
    assert torch.isfinite(h3).all()
    assert float((h3 - h0).abs().max()) <= 2e-3 * scale
    assert float((h3 - h0).norm() / h0.norm().clamp(min=1e-30)) <= 3e-4
    assert float((h3 - h1).abs().max()) <= 2e-3 * scale
    for a, b in ((o0.relation, o1.relation), (o0.connectivity, o1.connectivity), (o0.hidden, o1.hidden)):
        assert float((a - b).abs().max()) <= 1e-3 * max(float(a.abs().max()), 1.0)
    # pair subsets (overlap-filtered evaluation) take the read-back route for the per-window counts
    b2 = _edge_boxes(make_scene_batch(cfg, [14, 9], seed=8, connect_frac=0.3))
    ev = lambda: evaluate_minibatch(model, b2, overlap_filtering=True, skip_filtered=True)[1]
    a, b = _with_env({"SGC_SHARED_FC1": "0"}, ev), _with_env({"SGC_SHARED_FC1": "1"}, ev)
    fin = torch.isfinite(a.cand_conf)
    assert torch.equal(fin, torch.isfinite(b.cand_conf))
    assert float((a.cand_conf[fin] - b.cand_conf[fin]).abs().max()) <= 1e-3


def test_scenes_that_are_mostly_pair_specific_use_the_per_pair_kernels(monkeypatch):
    """Every box = the whole image: nothing can be shared, the column buffers of the shared backward would be at their largest.
    The host's window count sends the step to the per-pair kernels; the results are the per-pair results."""
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    from scene_graph_commonsense_amd import engine
    monkeypatch.setattr(engine.TUNING, "shared_max_fraction", 0.5)
    cfg = HeadConfig()
    model = _model(cfg)
    batch = make_scene_batch(cfg, [6, 5], seed=4, connect_frac=0.3)
    for b in batch.bbox:
        b[:] = torch.tensor([0, 32, 0, 32], dtype=b.dtype)
    sc = flatten_scene(cfg, batch, "cuda:0")
    assert sc.shared_windows == sc.n_pairs * 64
    eng = model.refresh_weights(backward=True)

    def run():
        model.zero_grad(set_to_none=True)
        loss = model.training_step(sc)
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.clone() for n, p in model.named_parameters()}, model.last_ctx_shared
    model.last_ctx_shared = None
    import scene_graph_commonsense_amd.engine as E
    seen = []
    orig = E.RelHeadEngine.conv3_shared
    monkeypatch.setattr(E.RelHeadEngine, "conv3_shared", lambda self, *a, **k: (seen.append(1), orig(self, *a, **k))[1])
    l1, g1, _ = run()
    assert not seen                                            # the shared path was not taken
    l0, g0, _ = _with_env({"SGC_SHARED_CONV3": "0"}, run)
    assert l0 == l1
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


def test_linear_pairs_combined_windows_match_convolved_windows(monkeypatch):
    """Sixth identity (csrc/kernels_shared.hip, "linear pairs"): a pair whose two objects' regions of influence on the 16-grid are
    disjoint gets its X windows from  pre_(i,bg) + pre_(bg,j) - pre_(bg,bg)  instead of a convolution of its own.  Exact in real
    arithmetic; on the device three f32 accumulations are added instead of one, so the training step with the identity ON must
    reproduce the step with it OFF to f32 round-off in the forward (an f16 ulp on h1 here and there, a max-pool route flipped at
    an exact near-tie) and to the bf16 noise of one more summation level in the gradients."""
    from scene_graph_commonsense_amd import engine
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    monkeypatch.setattr(engine.TUNING, "shared_max_fraction", 0.5)
    cfg = HeadConfig()
    model = _model(cfg, seed=3)
    batch = make_scene_batch(cfg, [24, 17, 9, 30], seed=31, connect_frac=0.2)
    sc = flatten_scene(cfg, batch, "cuda:0")
    assert sc.linear_windows > 0.05 * sc.shared_windows, (sc.linear_windows, sc.shared_windows)

    def run():
        eng = model.refresh_weights(backward=True)
        _poison(eng)
        model.zero_grad(set_to_none=True)
        loss = model.training_step(sc)
        torch.cuda.synchronize()
        out = model.last_outputs
        return (float(loss), out.relation.clone(), out.hidden.clone(), out.connectivity.clone(),
                {n: p.grad.clone() for n, p in model.named_parameters()}, getattr(eng, "_xw_linear", None))

    with engine.tuning(shared_linear=False):
        l0, r0, h0, c0, g0, x0 = run()
    with engine.tuning(shared_linear=True):
        l1, r1, h1, c1, g1, x1 = run()
    assert x0 is None and x1 is not None and x1[0] == sc.linear_windows
    print("linear windows %d of %d X windows; loss %.6f vs %.6f" % (sc.linear_windows, sc.shared_windows, l0, l1))
    assert torch.isfinite(r1).all() and torch.isfinite(h1).all()
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    scale = float(h0.abs().max())
    assert float((h0 - h1).abs().max()) <= 2e-3 * scale                         # a handful of f16 ulps of h1 through fc2
    assert float((r0 - r1).abs().max()) <= 1e-4 and float((c0 - c1).abs().max()) <= 1e-4 * max(1.0, float(c0.abs().max()))
    worst = {n: float((g0[n].double() - g1[n].double()).norm() / g0[n].double().norm().clamp(min=1e-30)) for n in g0}
    print({k: "%.1e" % v for k, v in worst.items()})
    for n, e in worst.items():
        assert e <= 5e-3, (n, e)


def test_conv2_halves_on_object_regions_equal_whole_maps_bitwise():
    """conv2_1 computed only on the 2x2-pixel windows of every object's D16 rectangle, the other rows copied from the image's
    background half (``engine.object_halves(regions=...)``): the same bits as the convolution of the whole masked maps, on boxes
    that include full-image, 1x1, border, empty, negative-start and out-of-grid ones."""
    from scene_graph_commonsense_amd import engine
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    model = _model(cfg, seed=2)
    batch = _edge_boxes(make_scene_batch(cfg, [14, 9, 3], seed=8))
    sc = flatten_scene(cfg, batch, "cuda:0")
    assert 0 < sc.conv2_windows < 256 * int(sc.obj_img.shape[0])
    eng = model.refresh_weights()
    a_img = eng.image_maps(sc.image_feature, sc.image_depth)
    with engine.tuning(shared_conv2=False):
        whole = {r: t.clone() for r, t in eng.object_halves(a_img, sc.obj_img, sc.bbox, with_bg=True, regions=sc.conv2_windows).items()}
    for name in ("uv_0", "uv_1"):
        eng.scratch.bufs[name].view(torch.int16).fill_(0x7E00)            # f16 NaN: a row nobody writes shows up
    with engine.tuning(shared_conv2=True):
        parts = eng.object_halves(a_img, sc.obj_img, sc.bbox, with_bg=True, regions=sc.conv2_windows)
    torch.cuda.synchronize()
    for r in (0, 1):
        assert torch.equal(whole[r].view(torch.int16), parts[r].view(torch.int16)), r


def test_fc1_assembly_with_presummed_own_rectangles_keeps_its_bits():
    """``TUNING.fc1_own_sums`` (default): the rectangle term S'_j[R_j] of the fc1 assembly, which depends on one object only, is summed
    once per object (``sgc_fc1_own_rect_sums``) and read as one vector per pair instead of four corner vectors - the same expression,
    so every output of the fused pass is bit-identical to the four-corner form."""
    from scene_graph_commonsense_amd.engine import tuning
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=5, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (24, 17, 9), seed=77, connect_frac=0.1, edge_boxes=True)
    sc = flatten_scene(cfg, batch, "cuda:0")
    outs = []
    for on in (True, False):
        with tuning(fc1_own_sums=on):
            o = model.forward_pairs(sc)
            torch.cuda.synchronize()
            outs.append((o.relation.clone(), o.hidden.clone(), o.connectivity.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_window_weight_gradient_on_the_sparse_matrix_cores_equals_the_dense_block():
    """``TUNING.sparse_wgrad`` (default): the conv3 weight gradient over the real pairs' listed windows runs on
    ``v_smfmac_f32_32x32x32_bf16`` (one non-zero per window and channel = 4 consecutive K indices = the 2:4 pattern; operand packed
    from the pooled rows), the per-object entries and the boundary tile on the dense block.  Same products, f32 accumulation in another
    order: conv3_1.weight's gradient agrees to 1e-5, every other gradient bit for bit."""
    from scene_graph_commonsense_amd.engine import tuning
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=9, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (40, 33, 27), seed=91, connect_frac=0.05)
    sc = flatten_scene(cfg, batch, "cuda:0")
    assert sc.shared_windows >= 8192 and sc.linear_windows > 0
    grads = []
    for on in (True, False):
        with tuning(sparse_wgrad=on, sparse_dgrad=False):          # (the sparse data gradient moves the conv3 bias sums to its packer: own test below)
            model.zero_grad(set_to_none=True)
            loss = model.training_step(sc, batch.relationships, batch.subj_or_obj)
            torch.cuda.synchronize()
            grads.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    (la, a), (lb, b) = grads
    assert la == lb
    for n in a:
        if n == "conv3_1.weight":
            e = float((a[n].double() - b[n].double()).norm() / b[n].double().norm())
            print("conv3_1.weight sparse vs dense", e)
            assert e <= 1e-5, e
        else:
            assert torch.equal(a[n], b[n]), n


def test_window_weight_gradient_gathered_from_the_f16_maps_keeps_every_bit():
    """``TUNING.gather_wgrad`` (default, round 6): the sparse weight-gradient block reads the 4 x 4 input patches of the listed windows
    straight from the forward's f16 maps through the window list (per-window bases by scalar loads, f16 -> bf16 in registers) instead of
    from a 16 KB-per-window patch copy.  Same values (the copy pass converted the same way), same products in the same order: EVERY
    gradient bit for bit, on ragged images with edge / duplicate / empty boxes, and with the contrastive branch's second engine."""
    from scene_graph_commonsense_amd.engine import tuning
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=9, head_gain=4.0))
    model.eval()
    for nobj, seed, edge in (((40, 33, 27), 91, False), ((48, 41, 9, 36), 17, True)):
        batch = make_scene_batch(cfg, nobj, seed=seed, connect_frac=0.05)
        if edge:
            batch = _edge_boxes(batch)
        sc = flatten_scene(cfg, batch, "cuda:0")
        assert sc.shared_windows >= 6000, sc.shared_windows          # the sparse (gathering) launch takes lists of >= 4096 real windows
        grads = []
        for on in (True, False):
            with tuning(gather_wgrad=on):
                _poison(model.engine())
                model.zero_grad(set_to_none=True)
                loss = model.training_step(sc, batch.relationships, batch.subj_or_obj)
                torch.cuda.synchronize()
                grads.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
        (la, a), (lb, b) = grads
        assert la == lb
        for n in a:
            assert torch.isfinite(a[n]).all(), n
            assert torch.equal(a[n], b[n]), n


def test_window_weight_gradient_with_k_ranges_per_xcd_is_the_same_sum():
    """``TUNING.wgrad_xcd_k`` (default, round 6): for lists of >= 32 768 real windows the sparse weight-gradient launch cuts K into 32 ranges
    and gives every XCD all 36 tiles of one channel half for every fourth range (``csrc/gemm_tn_sp.h``, xcd_map 2) instead of 7 ranges with one
    M tile per XCD.  Same products, another partition of the f32 sums: conv3_1.weight's gradient to 1e-5, every other gradient bit for bit."""
    from scene_graph_commonsense_amd.engine import tuning
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=9, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (64, 64, 64), seed=23, connect_frac=0.03)
    sc = flatten_scene(cfg, batch, "cuda:0")
    assert sc.shared_windows >= 40000, sc.shared_windows
    grads = []
    for on in (True, False):
        with tuning(wgrad_xcd_k=on):
            model.zero_grad(set_to_none=True)
            loss = model.training_step(sc, batch.relationships, batch.subj_or_obj)
            torch.cuda.synchronize()
            grads.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    (la, a), (lb, b) = grads
    assert la == lb
    for n in a:
        if n == "conv3_1.weight":
            e = float((a[n].double() - b[n].double()).norm() / b[n].double().norm())
            print("conv3_1.weight, K ranges per XCD vs per M tile", e)
            assert 0 < e <= 1e-5, e                      # 0 would mean the switch did not take (the list is long enough for it)
        else:
            assert torch.equal(a[n], b[n]), n


def test_window_data_gradient_on_the_sparse_matrix_cores_equals_the_dense_patch_form():
    """``TUNING.sparse_dgrad`` (default): the conv3 data gradient over the real pairs' listed windows runs on
    ``v_smfmac_f32_32x32x32_bf16`` (csrc/kernels_dgrad_sp.hip: the pooled rows masked to the own-pixel sets are the compressed operand),
    the per-object entries and the boundary tile on the dense patch block.  Same products per output element, f32 accumulation in
    another order before the bf16 rounding of the patch rows: everything above conv3's input (conv3 weight, fc1, fc2, head) is bit
    for bit the same, the conv3 bias sum is taken in another order (1e-6), conv2 / conv1 gradients agree to the bf16 noise of one
    re-rounded tensor (measured 1e-3; the oracle-routed bars of tests/test_sampled_oracle_gpu.py hold with it on)."""
    from scene_graph_commonsense_amd.engine import tuning
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=9, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (40, 33, 27), seed=91, connect_frac=0.05)
    sc = flatten_scene(cfg, batch, "cuda:0")
    assert sc.shared_windows >= 8192 and sc.linear_windows > 0
    grads = []
    for on in (True, False):
        with tuning(sparse_dgrad=on):
            model.zero_grad(set_to_none=True)
            loss = model.training_step(sc, batch.relationships, batch.subj_or_obj)
            torch.cuda.synchronize()
            grads.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    (la, a), (lb, b) = grads
    assert la == lb
    worst = {}
    for n in a:
        if n.startswith(("conv1", "conv2")):
            worst[n] = float((a[n].double() - b[n].double()).norm() / b[n].double().norm())
            assert worst[n] <= 3e-3, (n, worst[n])
        elif n == "conv3_1.bias":
            assert float((a[n].double() - b[n].double()).abs().max() / b[n].double().abs().max()) <= 1e-5
        else:
            assert torch.equal(a[n], b[n]), n
    print("sparse vs dense window data gradient:", {k: "%.1e" % v for k, v in worst.items()})


def test_conv2_backward_on_the_objects_gradient_regions_is_bit_identical():
    """``TUNING.conv2_bwd_regions`` (default): the conv2 data gradient runs only on the 2x2-pixel cells where an object's gradient can be
    non-zero - the pixel rectangle of its pseudo-pair (every pair of the object writes its dz inside it) + one ring for the 3x3 transposed
    convolution - and the rest of ``da`` is zero-filled.  The skipped cells hold exact zeros either way: every gradient is bit for bit
    the whole-map result, on boxes that include full-image, 1x1, zero-area and corner boxes."""
    from scene_graph_commonsense_amd.engine import tuning
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=9, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (40, 33, 27), seed=92, connect_frac=0.05, edge_boxes=True)
    sc = flatten_scene(cfg, batch, "cuda:0")
    grads = []
    for on in (True, False):
        with tuning(conv2_bwd_regions=on):
            model.zero_grad(set_to_none=True)
            loss = model.training_step(sc, batch.relationships, batch.subj_or_obj)
            torch.cuda.synchronize()
            grads.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    (la, a), (lb, b) = grads
    assert la == lb
    for n in a:
        assert torch.equal(a[n], b[n]), n

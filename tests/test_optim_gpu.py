"""The fused fc1.weight optimizer step (``sgc_sgd_fc1_fused``, ``optim.FusedSGD.fuse_fc1``): the gradient leaves the backward in GEMM
order, one pass un-permutes it, applies the SGD-momentum update of ``/root/reference/train_test.py:100,277`` and writes the forward's f16
compute copy.  Bar: BIT-identical to the three passes it replaces (gradient transposition, ``sgc_sgd_momentum_step``, f16 transposition)
- at the kernel level on random data and end to end over several ``train_minibatch`` steps."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_fc1_update_equals_transpose_sgd_transpose_bit_for_bit():
    from scene_graph_commonsense_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    rows = 96
    g = torch.Generator(device=dev).manual_seed(5)
    f, L = ctypes.c_float, ctypes.c_long
    st = _lib.stream_ptr
    for first, (lr, mom, wd) in ((1, (1e-3, 0.9, 1e-4)), (0, (3e-4, 0.9, 1e-4)), (0, (0.5, 0.0, 0.0))):
        w = torch.randn(rows, 65536, device=dev, generator=g) * 0.01
        m = torch.randn(rows, 65536, device=dev, generator=g) * 0.1
        gg = torch.randn(rows, 65536, device=dev, generator=g)                   # GEMM order [n][window*1024 + channel]
        # the three passes
        g_ref = torch.empty_like(gg)
        _lib.check(lib.sgc_transpose_cast(_lib.ptr(gg), _lib.ptr(g_ref), 2, rows, 16, L(65536), L(64), L(1024), L(65536), L(4096), L(64), st()), "t2")
        assert torch.equal(g_ref.view(rows, 1024, 64), gg.view(rows, 64, 1024).transpose(1, 2))      # the definition of the two orders
        w_ref, m_ref = w.clone(), m.clone()
        _lib.check(lib.sgc_sgd_momentum_step(_lib.ptr(w_ref), _lib.ptr(g_ref), _lib.ptr(m_ref), L(w.numel()), f(lr), f(mom), f(wd), first, st()), "sgd")
        c_ref = torch.empty(rows, 65536, dtype=torch.float16, device=dev)
        _lib.check(lib.sgc_transpose_cast(_lib.ptr(w_ref), _lib.ptr(c_ref), 0, rows, 16, L(65536), L(4096), L(64), L(65536), L(64), L(1024), st()), "t0")
        # the fused pass
        w_f, m_f = w.clone(), m.clone()
        c_f = torch.zeros(rows, 65536, dtype=torch.float16, device=dev)
        _lib.check(lib.sgc_sgd_fc1_fused(_lib.ptr(w_f), _lib.ptr(gg), _lib.ptr(m_f), rows, f(lr), f(mom), f(wd), first, _lib.ptr(c_f), st()), "fused")
        torch.cuda.synchronize()
        assert torch.equal(w_f, w_ref) and torch.equal(m_f, m_ref) and torch.equal(c_f, c_ref)
        assert not torch.equal(w_f, w)
        # no compute copy (the sharded optimizer's form): same weights
        w_g, m_g = w.clone(), m.clone()
        _lib.check(lib.sgc_sgd_fc1_fused(_lib.ptr(w_g), _lib.ptr(gg), _lib.ptr(m_g), rows, f(lr), f(mom), f(wd), first, None, st()), "fused")
        assert torch.equal(w_g, w_ref) and torch.equal(m_g, m_ref)


def test_train_minibatch_with_the_fused_fc1_step_equals_the_unfused_steps():
    """Three optimisation steps from the raw minibatch (dropout on): every parameter, the momentum state and the next forward's
    outputs (which read the f16 copy the fused step wrote) are bit-identical to the run whose optimizer declines the fusion."""
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pair_loop import evaluate_minibatch, train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=11, head_gain=4.0)
    batches = [make_scene_batch(cfg, (7, 5, 6), seed=50 + k, connect_frac=0.5) for k in range(3)]
    runs = []
    for fused in (True, False):
        model = BayesianRelationClassifier(cfg.args(run_mode="train")).cuda()
        model.load_state_dict(sd)
        model.train()
        opt = FusedSGD(model.parameters(), lr=2e-5, momentum=0.9, weight_decay=1e-4)
        if not fused:
            opt.fuse_fc1 = lambda m: None
        losses = []
        for b in batches:
            losses.append(float(train_minibatch(model, b, opt)))
            assert model.fc1.weight.grad is None or not fused          # the fused step consumes the GEMM-order gradient
            assert not getattr(model.fc1.weight, "_sgc_grad_gemm_order", False)
        eng = model.engine()
        assert (eng._w1p_fresh is not None) == fused and not eng.fc1_grad_gemm_order
        model.eval()
        _, out, _, _ = evaluate_minibatch(model, batches[0], overlap_filtering=False)
        torch.cuda.synchronize()
        runs.append(dict(losses=losses, params={n: p.detach().clone() for n, p in model.named_parameters()},
                         mom={n: opt.state[p]["momentum_buffer"].clone() for n, p in model.named_parameters()},
                         rel=out.relation.clone(), hid=out.hidden.clone()))
        del model, opt
        torch.cuda.empty_cache()
    a, b = runs
    assert a["losses"] == b["losses"]
    for n in a["params"]:
        assert torch.equal(a["params"][n], b["params"][n]), n
        assert torch.equal(a["mom"][n], b["mom"][n]), n
    assert torch.equal(a["rel"], b["rel"]) and torch.equal(a["hid"], b["hid"])
    assert not torch.equal(a["params"]["fc1.weight"], make_state_dict(cfg, seed=11, head_gain=4.0)["fc1.weight"].cuda())


def test_direct_training_step_keeps_the_reference_column_order():
    """Outside ``train_minibatch`` nothing changes: ``training_step`` leaves fc1.weight.grad in the reference's [4096, channel*64 + window]
    order (what the gradient parity tests compare with the oracle), untagged, and a plain FusedSGD step takes the ordinary kernel."""
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args(run_mode="train")).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=12, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (5, 4), seed=60, connect_frac=0.5)
    opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9)
    sc = flatten_scene(cfg, batch, "cuda:0")
    model.training_step(sc, batch.relationships, batch.subj_or_obj)
    g1 = model.fc1.weight.grad.clone()
    assert not getattr(model.fc1.weight, "_sgc_grad_gemm_order", False) and not model.engine().fc1_grad_gemm_order
    # the same gradient in GEMM order, as train_minibatch's fused step would see it
    eng = model.engine()
    opt.zero_grad()
    eng.fc1_grad_gemm_order = True
    try:
        model.training_step(sc, batch.relationships, batch.subj_or_obj)
    finally:
        eng.fc1_grad_gemm_order = False
    g2 = model.fc1.weight.grad
    assert getattr(model.fc1.weight, "_sgc_grad_gemm_order", False)
    assert torch.equal(g2.view(4096, 64, 1024).transpose(1, 2).reshape(4096, 65536), g1)
    with pytest.raises(RuntimeError, match="two different column orders"):
        model.training_step(sc, batch.relationships, batch.subj_or_obj)       # accumulating the reference order onto the GEMM order


@pytest.mark.parametrize("kind", ["fused_groups", "sharded_world1", "fused_contrast"])
def test_gemm_order_gradient_through_image_groups_sharded_sgd_and_the_contrastive_branch(kind):
    """The GEMM-order gradient on the other paths of ``train_minibatch``: a minibatch cut into image groups (gradient sums of several
    engine passes), ``distributed.ShardedSGD`` as reducer + optimizer (world 1: its pieces, accumulators and the row-shard update
    ``sgc_sgd_fc1_fused`` without the f16 copy), the augmented view's second engine.  Two steps; every parameter bit-identical to
    the same run with the fusion off."""
    from scene_graph_commonsense_amd import distributed as D
    from scene_graph_commonsense_amd import engine as E
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=13, head_gain=4.0)
    batches = [make_scene_batch(cfg, (6, 5, 7, 4), seed=70 + k, connect_frac=0.5) for k in range(2)]
    aug = [torch.randn(4, 256, 32, 32, generator=torch.Generator().manual_seed(k)).cuda() for k in range(2)] if kind == "fused_contrast" else [None, None]
    runs = []
    for fused in (True, False):
        with E.tuning(fused_sgd=fused):
            model = BayesianRelationClassifier(cfg.args(run_mode="train")).cuda()
            model.load_state_dict(sd)
            model.train()
            kw = {}
            if kind == "sharded_world1":
                opt = D.ShardedSGD(model.named_parameters(), 1, 0, lr=2e-5, momentum=0.9, weight_decay=1e-4).attach(model)
                kw["reducer"] = opt
            else:
                opt = FusedSGD(model.parameters(), lr=2e-5, momentum=0.9, weight_decay=1e-4)
            if kind == "fused_groups":
                kw["workspace_budget"] = 2.5e8                       # one or two images per group
            for b, a in zip(batches, aug):
                if a is not None:
                    kw["image_feature_aug"] = a
                train_minibatch(model, b, opt, **kw)
                if kind == "fused_groups":
                    assert len(model.last_image_groups) > 1
            if kind == "sharded_world1":
                assert opt.fc1_gemm_order == fused
            torch.cuda.synchronize()
            runs.append({n: p.detach().clone() for n, p in model.named_parameters()})
            del model, opt
            torch.cuda.empty_cache()
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), n
    assert not torch.equal(runs[0]["fc1.weight"], sd["fc1.weight"].cuda())


def test_sharded_sgd_through_image_groups_with_the_step_taken_by_the_caller():
    """a ``ShardedSGD`` that ``attach`` switched to GEMM order must get that order on EVERY path - also a minibatch cut into
    image groups with ``optimizer=None`` (the caller steps: gradient accumulation over two minibatches).  Bit-identical parameters
    with the fusion off."""
    from scene_graph_commonsense_amd import distributed as D
    from scene_graph_commonsense_amd import engine as E
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=17, head_gain=4.0)
    batches = [make_scene_batch(cfg, (6, 5, 7, 4), seed=90 + k, connect_frac=0.5) for k in range(2)]
    runs = []
    for fused in (True, False):
        with E.tuning(fused_sgd=fused):
            model = BayesianRelationClassifier(cfg.args(run_mode="train")).cuda()
            model.load_state_dict(sd)
            model.eval()
            opt = D.ShardedSGD(model.named_parameters(), 1, 0, lr=2e-5, momentum=0.9, weight_decay=1e-4).attach(model)
            assert opt.fc1_gemm_order == fused
            opt.zero_grad()
            for b in batches:
                train_minibatch(model, b, optimizer=None, reducer=opt, workspace_budget=2.5e8)
                assert len(model.last_image_groups) > 1
                assert not model.engine().fc1_grad_gemm_order          # the switch is back off once the call returns
            opt.step()
            torch.cuda.synchronize()
            runs.append({n: p.detach().clone() for n, p in model.named_parameters()})
            del model, opt
            torch.cuda.empty_cache()
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), n
    assert not torch.equal(runs[0]["fc1.weight"], sd["fc1.weight"].cuda())
    with pytest.raises(ValueError):
        model = BayesianRelationClassifier(cfg.args(run_mode="train")).cuda()
        from scene_graph_commonsense_amd.optim import FusedSGD
        red = D.ShardedSGD(model.named_parameters(), 1, 0, lr=2e-5).attach(model)
        train_minibatch(model, batches[0], FusedSGD(model.parameters(), lr=2e-5), reducer=red)

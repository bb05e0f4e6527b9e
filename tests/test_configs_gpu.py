"""BASELINE.json's configurations AT THEIR SIZE through forward + backward on one MI355X, checked with properties that need no
CPU oracle (the oracle takes ~1 h for one such step):

  configs[1]  VG PredCLS, 8 images x 36 objects      (10 080 ordered pairs)
  metric      VG PredCLS, 8 images x 64 objects      (32 256 ordered pairs)
  configs[4]  OpenImages V6, 4 images x 100 objects  (39 600 ordered pairs, 601 classes, (4,2,24) head, no super-classes)
  configs[2]  SGCLS / SGDET relation-head half, 16 images, predicted objects from synthetic DETR decoder outputs

* gradient additivity: images are independent, so the gradients of the B-image step equal the SUM of B single-image steps run
  with that step's own per-pair loss coefficients - to f32 summation round-off (the per-pair arithmetic is row-independent, only
  the split-K / reduction order of the weight gradients differs);
* linearity: connectivity-only parameters scale exactly with lambda_connectivity;
* outputs finite and normalised, candidate predicates inside their super-category, connectivity counters consistent."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OIV6 = dict(dataset="oiv6", num_classes=601, num_super_classes=0, num_geometric=4, num_possessive=2, num_semantic=24)


def _model(cfg, seed=1):
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.synthetic import make_state_dict
    m = BayesianRelationClassifier(cfg.args(), num_classes=cfg.num_classes, num_super_classes=cfg.num_super_classes,
                                   num_geometric=cfg.num_geometric, num_possessive=cfg.num_possessive, num_semantic=cfg.num_semantic).cuda()
    m.load_state_dict(make_state_dict(cfg, seed=seed, head_gain=4.0))
    m.eval()
    return m


def _single(batch, k):
    from scene_graph_commonsense_amd.synthetic import SceneBatch
    return SceneBatch(batch.image_feature[k:k + 1], batch.image_depth[k:k + 1], [batch.bbox[k]], [batch.categories[k]],
                      None if batch.super_categories is None else [batch.super_categories[k]], [batch.relationships[k]],
                      [batch.subj_or_obj[k]], [int(batch.bbox[k].shape[0])])


@pytest.mark.parametrize("name,kw,nobj", [("configs1_vg_8x36", {}, [36] * 8), ("metric_vg_8x64", {}, [64] * 8),
                                          ("configs4_oiv6_4x100", OIV6, [100] * 4)])
def test_baseline_config_forward_backward_at_full_size(name, kw, nobj):
    from scene_graph_commonsense_amd.model import _shared_hint          # the host's window counts: same organisation as the whole batch
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, predicate_counts
    cfg = HeadConfig(**kw)
    model = _model(cfg)
    eng = model.refresh_weights(backward=True)
    batch = make_scene_batch(cfg, nobj, seed=17, connect_frac=0.03)
    sc = flatten_scene(cfg, batch, "cuda:0")
    P = sc.n_pairs
    assert P == sum(n * (n - 1) for n in nobj)

    # ---- whole-batch step through the product entry point
    grads_l = []
    for lam in (0.1, 0.2):
        model.zero_grad(set_to_none=True)
        loss = model.training_step(sc, lambda_connectivity=lam)
        assert torch.isfinite(loss) and float(loss) > 0
        grads_l.append({n: p.grad.clone() for n, p in model.named_parameters()})
    out = model.last_outputs
    assert torch.isfinite(out.relation).all() and torch.isfinite(out.connectivity).all() and torch.isfinite(out.hidden).all()
    tot = out.relation.exp().sum(1)
    assert torch.allclose(tot, torch.ones_like(tot), atol=1e-4)                       # three conditional blocks x super probability
    ng, npos = cfg.num_geometric, cfg.num_possessive
    cp = out.cand_pred
    assert int(cp[:, 0].min()) >= 0 and int(cp[:, 0].max()) < ng and int(cp[:, 1].min()) >= ng and int(cp[:, 1].max()) < ng + npos
    assert int(cp[:, 2].min()) >= ng + npos and int(cp[:, 2].max()) < cfg.num_relations
    st = model.last_connectivity_stats.tolist()
    n_conn = int((sc.directed >= 0).sum())
    assert st[0] + st[1] == P and st[1] == n_conn and st[4] <= st[1] and st[3] <= st[2] <= P
    for n, g in grads_l[1].items():
        assert torch.isfinite(g).all(), n
    r = float(grads_l[1]["fc4.weight"].norm() / grads_l[0]["fc4.weight"].norm())
    assert abs(r - 2.0) < 1e-3                                                         # fc4 sees only the BCE term
    # lambda = 0.2 step again through the explicit engine calls, to get the coefficients this step used
    counts = predicate_counts(cfg).numpy()
    cw = torch.from_numpy((1 - counts / counts.sum()).astype(np.float32)).cuda()
    coefs = eng.loss_coefficients_device(sc.step_ptr, sc.n_steps, sc.directed, cw, 0.2, 1.0)
    whole = grads_l[1]

    # ---- the same step image by image, each with ITS rows of the batch's coefficients
    image_of = sc.image.long()
    acc = None
    for k in range(len(nobj)):
        rows = torch.nonzero(image_of == k).flatten()
        s1 = flatten_scene(cfg, _single(batch, k), "cuda:0")
        assert s1.n_pairs == rows.numel()
        c1 = tuple(c[rows].contiguous() for c in coefs)
        ctx = eng.train_forward(s1.image_feature, s1.image_depth, s1.obj_img, s1.bbox, s1.cats, s1.super_mh, s1.sub_idx, s1.obj_idx,
                                dropout=False, dense=(s1.img_ptr, s1.pid, s1.max_n), shared_windows=_shared_hint(s1))
        assert torch.equal(ctx.out.relation, out.relation[rows])                        # forward rows do not depend on the batch
        _, g1 = eng.train_backward(ctx, c1, s1.sub_csr, s1.obj_csr, s1.img_ptr)
        acc = {n: g.double() for n, g in g1.items()} if acc is None else {n: acc[n] + g1[n].double() for n in acc}
    worst = {}
    for n, g in whole.items():
        a = acc[n].view_as(g)
        worst[n] = float((g.double() - a).norm() / a.norm().clamp(min=1e-30))
    print(name, "additivity, relative Frobenius:", {k: "%.1e" % v for k, v in worst.items()})
    for n, e in worst.items():
        assert e <= 2e-4, (n, e)


def test_configs2_relation_head_half_16_images_predicted_objects():
    """configs[2] (SGCLS / SGDET end to end, B = 16) without the DETR backbone: synthetic decoder outputs for 16 images -> HIP
    object front-end -> fused pair path over the predicted objects -> Evaluator(predcls=False); the skip of filtered pairs
    gives the same recall counters and ranked indices."""
    from scene_graph_commonsense_amd.evaluator import Evaluator
    from scene_graph_commonsense_amd.object_frontend import DetrFrontEnd
    from scene_graph_commonsense_amd.pair_loop import evaluate_sgdet_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, default_sub2super, make_scene_batch
    from tests import sgdet_case
    from tests.golden_cases import GOLDEN
    import os
    cfg = HeadConfig()
    model = _model(cfg, seed=5)
    B = 16
    nobj = [12 + (3 * i) % 9 for i in range(B)]
    batch = make_scene_batch(cfg, nobj, seed=23, connect_frac=0.15)
    alp = sgdet_case.alp2fre_table()
    inv = np.argsort(alp)
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(B, 100, 151, generator=g) * 0.3
    logits[:, :, 150] += 10.0
    boxes = torch.rand(B, 100, 4, generator=g) * 0.2 + 0.4
    for b in range(B):
        for o in range(nobj[b]):
            q = 2 * o + 1
            c = int(batch.categories[b][o])
            logits[b, q, 150] -= 10.0
            logits[b, q, int(inv[c])] += 14.0
            logits[b, q, int(inv[(c * 7 + 3) % 150])] += 9.0
            x0, x1, y0, y1 = [float(v) for v in batch.bbox[b][o]]
            boxes[b, q] = torch.tensor([(x0 + x1) / 64.0, (y0 + y1) / 64.0, (x1 - x0) / 32.0, (y1 - y0) / 32.0])
    fe = DetrFrontEnd(alp.tolist())
    cats, confs, bxs, kept = fe.sgdet(logits.cuda(), boxes.clamp(0, 1).cuda())
    assert kept == list(range(B)) and all(len(c) >= n for c, n in zip(cats, nobj))
    args = cfg.args(fixtures=os.path.join(GOLDEN, "ref_fixtures") + os.sep)
    res = []
    for skip in (False, True):
        ev = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
        scene, out, included = evaluate_sgdet_minibatch(model, batch.image_feature.cuda(), batch.image_depth.cuda(), cats, confs, bxs, ev,
                                                        sub2super=default_sub2super(cfg.num_classes, cfg.num_super_classes),
                                                        targets=(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox),
                                                        skip_filtered=skip)
        r = ev.compute(per_class=True, predcls=False)
        res.append(([float(x) for x in r[0]], float(ev.num_connected_target), dict(ev.last_topk)))
        assert scene.n_pairs == sum(len(c) * (len(c) - 1) for c in cats)
        assert all(0.0 <= x <= 1.0 for x in res[-1][0]) and res[-1][1] > 0
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    for k in res[0][2]:
        np.testing.assert_array_equal(res[0][2][k], res[1][2][k])


def test_two_stream_backward_is_bitwise_the_single_stream_backward(monkeypatch):
    """The weight-gradient chain runs on a side stream (engine.train_backward); every kernel is deterministic and the chains
    only meet through events, so the gradients must be bit-identical to the one-stream order (TUNING.bwd_streams off), also when the
    step is repeated back to back (buffer reuse across steps)."""
    from scene_graph_commonsense_amd import engine
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    model = _model(cfg)
    model.train()                                               # dropout on: seeds advance per step, reset below
    batch = make_scene_batch(cfg, [36] * 8, seed=29, connect_frac=0.05)
    sc = flatten_scene(cfg, batch, "cuda:0")
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setattr(engine.TUNING, "bwd_streams", mode == "1")
        model._step = 0
        out = []
        for rep in range(3):
            model.zero_grad(set_to_none=True)
            loss = model.training_step(sc)
            out.append((loss.clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
        torch.cuda.synchronize()
        res[mode] = out
    for (l0, g0), (l1, g1) in zip(res["0"], res["1"]):
        assert torch.equal(l0, l1)
        for n in g0:
            assert torch.equal(g0[n], g1[n]), n

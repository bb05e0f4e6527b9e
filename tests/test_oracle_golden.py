"""Pin the CPU oracle (oracle/relhead_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU only.  Tolerances: the oracle restates the same
float32 PyTorch ops in the same order, so log-probs agree to ~1e-6; integer outputs are exact."""
import numpy as np
import pytest
import torch

from oracle import relhead_oracle as O
from scene_graph_commonsense_amd.synthetic import predicate_counts
from tests.golden_cases import FULL, SMALL, load_case, zero_shot_list

RTOL, ATOL = 2e-5, 2e-5


def _run_eval(name):
    cfg, sd, batch, gold = load_case(name)
    ev = O.OracleEvaluator(cfg, zero_shot_triplets=zero_shot_list() if cfg.dataset == "vg" else None)
    t3 = O.OracleEvaluatorTop3(cfg) if (cfg.dataset == "vg" and cfg.hierarchical) else None
    with torch.no_grad():
        out = O.run_pair_loop(sd, batch, cfg, mode="eval", evaluator=ev, evaluator_top3=t3)
    return cfg, gold, out, ev, t3


def _check_eval(name):
    cfg, gold, out, ev, t3 = _run_eval(name)
    recs = out["records"]
    steps = [(r["g"], r["e"]) for r in recs if r["first"]]
    assert steps == [tuple(x) for x in gold["eval_steps"].tolist()]
    assert [len(r["keep"]) for r in recs] == gold["eval_call_sizes"].tolist()
    rel = torch.cat([r["relation"] for r in recs]).numpy()
    conn = torch.cat([r["connectivity"] for r in recs]).numpy()
    hid = torch.cat([r["hidden"] for r in recs]).numpy()
    if cfg.hierarchical:
        gold_rel = np.concatenate([gold["eval_rel1"], gold["eval_rel2"], gold["eval_rel3"]], axis=1)
        sup = torch.cat([r["super_relation"] for r in recs]).numpy()
        np.testing.assert_allclose(sup, gold["eval_super"], rtol=RTOL, atol=ATOL)
    else:
        gold_rel = gold["eval_rel"]
    np.testing.assert_allclose(rel, gold_rel, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(conn, gold["eval_conn"][:, 0], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(hid, gold["eval_hidden"], rtol=RTOL, atol=ATOL)
    # evaluator state (order included) and ranking
    s = ev.flat_state()
    np.testing.assert_array_equal(s["which"].numpy(), gold["ev_which_in_batch"])
    np.testing.assert_array_equal(s["pred"].numpy(), gold["ev_relation_pred"])
    np.testing.assert_array_equal(s["rel_t"].numpy(), gold["ev_relation_target"])
    np.testing.assert_allclose(s["conf"].numpy(), gold["ev_confidence"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(s["conn"].numpy(), gold["ev_connectivity"], rtol=RTOL, atol=ATOL)
    res = ev.compute(per_class=True)
    np.testing.assert_allclose(np.array(res[0]), gold["ev_recall"], atol=1e-12)
    np.testing.assert_allclose(np.array([float(x) for x in res[2]]), gold["ev_mean_recall"], atol=1e-6, equal_nan=True)
    np.testing.assert_allclose(torch.stack(res[1]).numpy(), gold["ev_recall_per_class"], atol=1e-6, equal_nan=True)
    if cfg.dataset == "vg":
        np.testing.assert_allclose(np.array(res[3]), gold["ev_recall_zs"], atol=1e-12)
    assert ev.num_connected_target == gold["ev_num_connected_target"][0]
    for row, image in enumerate(sorted(ev.last_sorted)):
        mine = ev.last_sorted[image].numpy()
        ref = gold["ev_top100_stable"][row]
        ref = ref[ref >= 0]
        # top-K indices: exact (both sides use a stable descending order)
        np.testing.assert_array_equal(mine, ref)
    if t3 is not None:
        r3 = t3.compute(per_class=True)
        np.testing.assert_allclose(np.array(r3[0]), gold["top3_recall"], atol=1e-12)
        np.testing.assert_allclose(np.array([float(x) for x in r3[2]]), gold["top3_mean_recall"], atol=1e-6,
                                   equal_nan=True)


def _check_train(name):
    cfg, sd, batch, gold = load_case(name)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    w = O.class_weights(predicate_counts(cfg))
    out = O.run_pair_loop(sd, batch, cfg, mode="train", weights=w)
    loss = out["losses"]
    np.testing.assert_allclose(float(loss), gold["train_loss"][0], rtol=1e-5)
    loss.backward()
    for pname, p in sd.items():
        g = p.grad.flatten()
        key = pname.replace(".", "__")
        stride = max(1, g.numel() // 509)
        ref_l2 = gold["grad_l2__" + key][0]
        np.testing.assert_allclose(float(g.double().norm()), ref_l2, rtol=1e-4)
        np.testing.assert_allclose(g[::stride][:509].numpy(), gold["grad_sample__" + key], rtol=1e-3,
                                   atol=1e-5 * max(ref_l2, 1e-12))


@pytest.mark.parametrize("name", SMALL)
def test_oracle_eval_small(name):
    _check_eval(name)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_train_small(name):
    _check_train(name)


@pytest.mark.slow
@pytest.mark.parametrize("name", FULL)
def test_oracle_eval_full(name):
    _check_eval(name)


@pytest.mark.slow
def test_oracle_train_full():
    _check_train("vg_flat")


def test_super_class_quirk():
    # reference utils.py:136-149: for [3,4,6] only 3 and 6 are set
    m = O.super_class_multihot([torch.tensor([3, 4, 6]), torch.tensor([2]), torch.tensor([5, 1])], 17)
    assert m[0].nonzero().flatten().tolist() == [3, 6]
    assert m[1].nonzero().flatten().tolist() == [2]
    assert sorted(m[2].nonzero().flatten().tolist()) == [1, 5]


def test_overlap_filter_and_masks():
    bb = torch.tensor([[0, 4, 0, 4], [2, 6, 2, 6], [10, 12, 10, 12], [7, 7, 2, 9]], dtype=torch.int32)
    m = O.build_masks(bb, 32)
    assert m[0].sum() == 16 and m[3].sum() == 0
    f = O.overlap_filter(m[[0, 0, 0]].unsqueeze(1), m[[1, 2, 3]].unsqueeze(1))
    assert f.tolist() == [True, False, False]


def _aug_features(batch, seed):
    from scene_graph_commonsense_amd.synthetic import hash_normal
    f = batch.image_feature
    noise = torch.from_numpy(hash_normal(seed * 31 + 99, f.numel()).reshape(f.shape))
    return 0.9 * f + 0.3 * noise


def _check_contrast(name):
    import os
    from tests.golden_cases import CASES, GOLDEN
    cfg, sd, batch, _ = load_case(name)
    gold = dict(np.load(os.path.join(GOLDEN, name + "_contrast.npz")))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    w = O.class_weights(predicate_counts(cfg))
    out = O.run_pair_loop(sd, batch, cfg, mode="train", weights=w, image_feature_aug=_aug_features(batch, CASES[name][2]))
    np.testing.assert_allclose(float(out["loss_contrast"]), gold["trainc_contrast"][0], rtol=2e-5)
    np.testing.assert_allclose(float(out["losses"]), gold["trainc_loss"][0], rtol=1e-5)
    out["losses"].backward()
    for pname, p in sd.items():
        g = p.grad.flatten()
        key = pname.replace(".", "__")
        stride = max(1, g.numel() // 509)
        ref_l2 = gold["gradc_l2__" + key][0]
        np.testing.assert_allclose(float(g.double().norm()), ref_l2, rtol=1e-4)
        np.testing.assert_allclose(g[::stride][:509].numpy(), gold["gradc_sample__" + key], rtol=1e-3, atol=1e-5 * max(ref_l2, 1e-12))


@pytest.mark.parametrize("name", SMALL)
def test_oracle_contrastive_small(name):
    """Supervised-contrastive branch (second augmented view + SupConLossHierar) against the reference."""
    _check_contrast(name)


@pytest.mark.parametrize("name", SMALL)
def test_oracle_train_cs_small(name):
    """Training loss with the commonsense penalty (run_mode train_cs) against the reference."""
    import os
    from tests.golden_cases import GOLDEN
    cfg, sd, batch, _ = load_case(name)
    gold = dict(np.load(os.path.join(GOLDEN, name + "_traincs.npz")))
    fx = os.path.join(GOLDEN, "ref_fixtures") + os.sep
    aligned = torch.load(fx + "commonsense_aligned_triplets.pt")
    violated = torch.load(fx + "commonsense_violated_triplets.pt")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.run_pair_loop(sd, batch, cfg, mode="train", weights=O.class_weights(predicate_counts(cfg)),
                          commonsense=(aligned, violated))
    np.testing.assert_allclose(float(out["losses"]), gold["traincs_loss"][0], rtol=1e-5)
    out["losses"].backward()
    for pname, p in sd.items():
        g = p.grad.flatten()
        key = pname.replace(".", "__")
        stride = max(1, g.numel() // 509)
        ref_l2 = gold["gradcs_l2__" + key][0]
        np.testing.assert_allclose(float(g.double().norm()), ref_l2, rtol=1e-4)
        np.testing.assert_allclose(g[::stride][:509].numpy(), gold["gradcs_sample__" + key], rtol=1e-3, atol=1e-5 * max(ref_l2, 1e-12))


def test_oracle_bayes_head_and_aug_forward_match_reference_boundary_golden():
    """oracle.bayes_head with temperatures, and classifier_forward's hidden for the plain / augmented view, against the
    REAL reference (tests/golden/boundary.npz, made by tests/golden/make_boundary_golden.py)."""
    import os
    from tests.boundary_cases import aug_features, head_inputs
    from tests.golden_cases import CASES, GOLDEN, load_case
    gold = dict(np.load(os.path.join(GOLDEN, "boundary.npz")))
    h, sd = head_inputs()
    r1, r2, r3, sup = O.bayes_head(sd, h, T=(1.0, 2.0, 0.5))
    for got, key in ((r1, "head_rel1"), (r2, "head_rel2"), (r3, "head_rel3"), (sup, "head_super")):
        np.testing.assert_allclose(got.numpy(), gold[key], rtol=2e-5, atol=2e-5)
    cfg, sdm, batch, _ = load_case("vg_full")
    masks = [O.build_masks(b, 32) for b in batch.bbox]
    fa = aug_features(batch, CASES["vg_full"][2])
    gm = torch.stack([masks[i][1].unsqueeze(0) for i in range(3)]); em = torch.stack([masks[i][0].unsqueeze(0) for i in range(3)])
    cs = torch.tensor([int(batch.categories[i][1]) for i in range(3)]); co = torch.tensor([int(batch.categories[i][0]) for i in range(3)])
    ss = [batch.super_categories[i][1] for i in range(3)]; so = [batch.super_categories[i][0] for i in range(3)]
    with torch.no_grad():
        for feat, key in ((batch.image_feature, "aug_pred"), (fa, "aug_pred_aug")):
            hs = torch.cat((feat * gm, batch.image_depth * gm), dim=1); ho = torch.cat((feat * em, batch.image_depth * em), dim=1)
            out = O.classifier_forward(sdm, hs, ho, cs, co, ss, so)
            np.testing.assert_allclose(out[5].numpy(), gold[key], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("with_cs", [False, True])
def test_streamed_step_losses_equal_the_literal_running_sums(with_cs):
    """``run_pair_loop(step_loss_hook=...)`` (one call's autograd graph alive at a time: what the full-size sampled-step jobs of
    ``tests/oracle_worker.py`` use) against the literal running-sum form pinned to the reference above: same loss, same gradients."""
    import os
    from tests.golden_cases import GOLDEN
    cfg, sd, batch, _ = load_case("vg_small")
    kw = {}
    if with_cs:
        fx = os.path.join(GOLDEN, "ref_fixtures") + os.sep
        kw["commonsense"] = (torch.load(fx + "commonsense_aligned_triplets.pt"), torch.load(fx + "commonsense_violated_triplets.pt"))
    keep = lambda g, e: (g + e) % 3 != 1                                    # a restricted loop, as the sampled-step jobs run it
    w = O.class_weights(predicate_counts(cfg))
    a = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lit = O.run_pair_loop(a, batch, cfg, mode="train", weights=w, step_filter=keep, **kw)
    lit["losses"].backward()
    T = len(lit["records"])
    b = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    total = [0.0]

    def own(k, loss):
        ((T - k) * loss).backward()
        total[0] += (T - k) * float(loss.detach())

    out = O.run_pair_loop(b, batch, cfg, mode="train", weights=w, step_filter=keep, step_loss_hook=own, **kw)
    assert out["losses"] is None and len(out["records"]) == T
    np.testing.assert_allclose(total[0], float(lit["losses"]), rtol=1e-5)
    for k in a:
        ga, gb = a[k].grad, b[k].grad
        assert float((ga - gb).norm()) <= 2e-5 * max(float(ga.norm()), 1e-12), k

"""GPU parity of the fused forward+backward (loss with the reference's running-sum quirk, all parameter
gradients) against the CPU oracle and against gradient fingerprints stored from the real reference.
Tolerances (written here because BASELINE.json's north_star gives one only for the forward outputs):
  * loss: 2e-3 relative (it is a function of the forward outputs only);
  * head gradients (fc3_*, fc4, fc5: above every ReLU/max-pool mask): 5e-3 relative Frobenius;
  * fc2 / fc1 / conv3 / conv2 / conv1 gradients: 6e-2 relative Frobenius and cosine >= 0.997.  These sit below
    ReLU / dropout / max-pool routing masks; a forward value that differs by the f16 forward error
    (~5e-4 of scale) flips the mask of the ~5e-4 fraction of units whose pre-activation is that close to 0,
    and a fraction f of flipped routes moves a gradient by ~sqrt(f) = 2-4e-2 in Frobenius norm however
    exact the backward arithmetic is (bf16 gradient tensors with f32 accumulation add ~3e-3)."""
import numpy as np
import pytest
import torch

from tests.golden_cases import load_case

pytestmark = pytest.mark.gpu
GRAD_TOL = 6e-2
HEAD_TOL = 5e-3


def _tol(name):
    return HEAD_TOL if name.split('.')[0] in ('fc3', 'fc3_1', 'fc3_2', 'fc3_3', 'fc4', 'fc5') else GRAD_TOL


# The reference's goldens hold 509 strided samples of every gradient: an un-routed comparison like the whole-tensor one, same bound
# (round 2 allowed twice that).  What pins the arithmetic of the contrastive / commonsense steps are the ROUTED comparisons below.
SAMPLE_TOL = _tol
ROUTED_TOL = 5e-3        # tests/test_training_mode_gpu.py: head, fc2, fc1, conv3 with the device's routes injected; 7e-3 below conv3


def _routed_tol(name):
    return 7e-3 if name.split(".")[0] in ("conv2_1", "conv1_1", "conv1_2") else ROUTED_TOL


def run_train(cfg, sd, batch, dropout=False):
    from scene_graph_commonsense_amd.engine import RelHeadEngine, csr_by, loss_coefficients
    from scene_graph_commonsense_amd.pairs import flatten_scene, pair_targets
    from scene_graph_commonsense_amd.synthetic import predicate_counts
    dev = "cuda:0"
    eng = RelHeadEngine(cfg, dev)
    eng.load_weights(sd)
    eng.prep_bwd_weights(sd)
    sc = flatten_scene(cfg, batch, dev)
    pidx = sc.pidx
    directed, _ = pair_targets(batch.relationships, batch.subj_or_obj, pidx)
    counts = predicate_counts(cfg).numpy()
    cw = 1 - counts / counts.sum()
    coefs = loss_coefficients(cfg, pidx.step, len(pidx.call_sizes), directed, cw)
    coefs_d = tuple(torch.from_numpy(c).to(dev) for c in coefs)
    n_obj = int(sc.obj_img.shape[0])
    sub_csr = tuple(torch.from_numpy(a).to(dev) for a in csr_by(pidx.sub, n_obj))
    obj_csr = tuple(torch.from_numpy(a).to(dev) for a in csr_by(pidx.obj, n_obj))
    img_ptr = torch.from_numpy(pidx.obj_offset.astype(np.int32)).to(dev)
    from scene_graph_commonsense_amd.model import _shared_hint
    ctx = eng.train_forward(sc.image_feature, sc.image_depth, sc.obj_img, sc.bbox, sc.cats, sc.super_mh, sc.sub_idx, sc.obj_idx,
                            dropout=dropout, shared_windows=_shared_hint(sc))
    loss, grads = eng.train_backward(ctx, coefs_d, sub_csr, obj_csr, img_ptr)
    torch.cuda.synchronize()
    return float(loss), {k: v.float().cpu() for k, v in grads.items()}


def _fro(a, b):
    return float((a.double() - b.double()).norm() / max(b.double().norm(), 1e-30))


@pytest.mark.oracle_heavy
@pytest.mark.parametrize("hier", [True, False])
def test_backward_matches_oracle(hier):
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict, predicate_counts
    cfg = HeadConfig(hierarchical=hier)
    sd = make_state_dict(cfg, seed=21, head_gain=4.0)
    batch = make_scene_batch(cfg, (4, 3, 2), seed=21, connect_frac=0.6, edge_boxes=True)
    loss, grads = run_train(cfg, sd, batch)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=O.class_weights(predicate_counts(cfg)))
    out["losses"].backward()
    ref_loss = float(out["losses"])
    print("loss", loss, ref_loss)
    assert abs(loss - ref_loss) <= 2e-3 * abs(ref_loss)
    worst = {}
    for k, p in sdr.items():
        e = _fro(grads[k], p.grad)
        worst[k] = e
    print({k: "%.1e" % v for k, v in worst.items()})
    for k, e in worst.items():
        assert e <= _tol(k), (k, e)
        a, b = grads[k].double().flatten(), sdr[k].grad.double().flatten()
        assert float(a @ b / (a.norm() * b.norm())) >= 0.997, k


def fingerprint_chain(gold, l2_key, sample_key, grads, g_routed, g_ref=None):
    """Device gradients against the fingerprints stored from the REAL reference (L2 norm + 509 strided samples per tensor) with a bound
    DERIVED in the test (see ``test_backward_matches_reference_fingerprints``): with ``g_routed`` = the oracle's gradients under the
    device's own routing decisions,
        |device - stored|  <=  |device - g_routed|  +  |g_routed - stored|
                               arithmetic: <= the     what the (justified) routing flips put between two runs of the SAME
                               routed bound           reference graph - computed here, not assumed
    ``g_ref`` (the oracle's un-routed gradients, optional): also checks that the oracle reproduces the stored samples (1e-3, as
    ``tests/test_oracle_golden.py`` does on the CPU).  Returns {tensor: text} for the log."""
    report = {}
    for k, g in grads.items():
        key = k.replace(".", "__")
        stride = max(1, g.numel() // 509)
        take = lambda t: t.flatten()[::stride][:509].double().cpu().numpy()
        ref, ref_l2 = gold[sample_key + key].astype(np.float64), float(gold[l2_key + key][0])
        nref = max(np.linalg.norm(ref), 1e-30)
        if g_ref is not None:
            d_host = np.linalg.norm(take(g_ref[k]) - ref) / nref
            assert d_host <= 1e-3, (k, d_host)
        tol = _routed_tol(k) if _tol(k) == GRAD_TOL else HEAD_TOL
        arith = np.linalg.norm(take(g) - take(g_routed[k])) / nref
        assert arith <= 2 * tol, (k, arith)          # on 509 samples (twice the whole-tensor bound: a sample's share of the error fluctuates)
        e_full = _fro(g, g_routed[k])
        assert e_full <= tol, (k, e_full)
        d_route = np.linalg.norm(take(g_routed[k]) - ref) / nref
        err = np.linalg.norm(take(g) - ref) / nref
        assert err <= 2 * tol + d_route + 1e-4, (k, err, d_route)                    # triangle inequality
        n_dev, n_rt = float(g.double().norm()), float(g_routed[k].double().norm())
        assert abs(n_dev - ref_l2) <= tol * n_rt + abs(n_rt - ref_l2) + 1e-4 * ref_l2, (k, n_dev, n_rt, ref_l2)
        report[k] = "%.1e (flips %.1e, arithmetic %.1e)" % (err, d_route, arith)
    return report


@pytest.mark.oracle_heavy
@pytest.mark.parametrize("name", ["vg_full", "vg_flat", "oiv6_full"])
def test_backward_matches_reference_fingerprints(name):
    """The device's gradients against the fingerprints stored from the REAL reference (L2 norm + 509 strided samples per tensor), with
    a bound that is derived in the test instead of calibrated (``profiles/r06_backward_attribution.txt``: an
    exact float64 backward behind the f16 forward is already 2.6-4.5e-2 away on ``vg_full`` - routing flips, not arithmetic).  The chain:

      (a) the oracle WITHOUT injected routes reproduces the stored samples (it is the reference; 1e-3 as in ``tests/test_oracle_golden.py``:
          two f32 runs of one graph on different hosts / thread counts);
      (b) every routing decision the device takes differently from the oracle lies within the forward tolerance of its decision boundary
          (``train_case.route_flip_margins``: |pre-activation| <= 1e-3 of the layer's scale for a ReLU, the routed value within 2e-3 of the
          window's maximum for a max-pool) - the forward parity statement of ``BASELINE.json`` applied to the routes;
      (c) with those routes injected the oracle's gradient and the device's agree to the ARITHMETIC bound (5e-3; 7e-3 under conv3) on the
          same samples and norms;
      (d) therefore device vs stored fingerprint <= (c) + the distance the justified flips of (b) put between the oracle's two runs:
          asserted as such - the un-routed distance itself (printed) needs no bar of its own, and a single flipped unit of an 18-pair
          fixture (4.4e-2 of fc2's gradient) no longer decides whether an exact kernel change may ship."""
    from tests.train_case import FORWARD_TOL, oracle_train, route_flip_margins, run_train_gpu
    cfg, sd, batch, gold = load_case(name)
    loss, grads, routes, sc = run_train_gpu(cfg, sd, batch, keep_ctx=True)
    assert abs(loss - gold["train_loss"][0]) <= 2e-3 * abs(gold["train_loss"][0])
    pre = {}
    _, g_ref, _ = oracle_train(cfg, sd, batch, sc, capture=pre)
    _, g_routed, _ = oracle_train(cfg, sd, batch, sc, routes=routes)
    margins = route_flip_margins(pre, routes)
    print(name, {k: "%d of %d flipped, worst margin %.1e" % v for k, v in margins.items()})
    for kind, (n_flip, n_all, worst) in margins.items():
        assert worst <= (2 if kind.startswith("pool") else 1) * FORWARD_TOL, (kind, n_flip, worst)          # (b)
        assert n_flip <= 0.01 * n_all, (kind, n_flip, n_all)
    print(fingerprint_chain(gold, "grad_l2__", "grad_sample__", grads, g_routed, g_ref))                         # (a), (c), (d)


def _aug_features(batch, seed):
    from scene_graph_commonsense_amd.synthetic import hash_normal
    f = batch.image_feature
    noise = torch.from_numpy(hash_normal(seed * 31 + 99, f.numel()).reshape(f.shape))
    return 0.9 * f + 0.3 * noise


def test_supcon_kernel_matches_oracle():
    """SupConLossHierar kernel (loss + feature gradient) against the oracle's literal restatement with autograd."""
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.engine import RelHeadEngine
    from scene_graph_commonsense_amd.synthetic import HeadConfig
    eng = RelHeadEngine(HeadConfig(), "cuda:0")
    g = torch.Generator().manual_seed(0)
    for M in (1, 7, 300):
        feats = (torch.rand(M, 2, 512, generator=g) * 0.3).requires_grad_(True)
        labels = torch.randint(0, 50, (M,), generator=g)
        if M > 2:
            labels[1] = labels[0]
        loss = O.supcon_hierar_loss(feats, labels)
        loss.backward()
        flat = torch.cat((feats[:, 0], feats[:, 1]), dim=0).detach().contiguous().cuda()
        l2, dF = eng.supcon_loss(flat, labels.int().cuda(), grad_scale=1.0)
        torch.cuda.synchronize()
        ref_dF = torch.cat((feats.grad[:, 0], feats.grad[:, 1]), dim=0)
        assert abs(float(l2) - float(loss)) <= 1e-4 * max(1.0, abs(float(loss))), (M, float(l2), float(loss))
        assert _fro(dF.cpu(), ref_dF) <= 1e-3, (M, _fro(dF.cpu(), ref_dF))


@pytest.mark.parametrize("name", ["vg_full_hit", "vg_full"])
def test_contrastive_training_step_matches_reference(name):
    """Whole training step with the supervised-contrastive branch against the reference's loss / gradient fingerprints."""
    import os
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from tests.golden_cases import CASES, GOLDEN
    path = os.path.join(GOLDEN, name + "_contrast.npz")
    if not os.path.exists(path):
        pytest.skip("no contrastive golden for " + name)
    gold = dict(np.load(path))
    cfg, sd, batch, _ = load_case(name)
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(sd)
    model.eval()
    sc = flatten_scene(cfg, batch, "cuda:0")
    loss = model.training_step(sc, batch.relationships, batch.subj_or_obj, image_feature_aug=_aug_features(batch, CASES[name][2]))
    torch.cuda.synchronize()
    lc = float(model.last_contrast_loss)
    print(name, "contrastive", lc, gold["trainc_contrast"][0], "total", float(loss), gold["trainc_loss"][0])
    assert abs(lc - gold["trainc_contrast"][0]) <= 2e-2 * max(1.0, abs(gold["trainc_contrast"][0]))
    assert abs(float(loss) - gold["trainc_loss"][0]) <= 2e-3 * abs(gold["trainc_loss"][0])
    # the gradient fingerprints of this step: ``test_contrastive_step_with_device_routes_is_arithmetic_exact`` (bound derived there)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_train_cs_step_matches_reference():
    """Training step with the commonsense penalty (run_mode train_cs) against the reference's loss / gradient fingerprints."""
    import os
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from tests.golden_cases import GOLDEN
    name = "vg_full_hit"
    gold = dict(np.load(os.path.join(GOLDEN, name + "_traincs.npz")))
    fx = os.path.join(GOLDEN, "ref_fixtures") + os.sep
    aligned = torch.load(fx + "commonsense_aligned_triplets.pt")
    violated = torch.load(fx + "commonsense_violated_triplets.pt")
    cfg, sd, batch, _ = load_case(name)
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(sd)
    model.eval()
    sc = flatten_scene(cfg, batch, "cuda:0")
    loss = model.training_step(sc, batch.relationships, batch.subj_or_obj, commonsense=(list(aligned.keys()), list(violated.keys())))
    torch.cuda.synchronize()
    print("train_cs loss", float(loss), gold["traincs_loss"][0])
    assert abs(float(loss) - gold["traincs_loss"][0]) <= 2e-3 * abs(gold["traincs_loss"][0])
    # the gradient fingerprints of this step: ``test_train_cs_step_with_device_routes_is_arithmetic_exact`` (bound derived there)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.oracle_heavy
def test_commonsense_penalty_alone_matches_oracle():
    """Only the train_cs penalty is active (no connected pair, lambda_connectivity = 0): loss and head gradients must match
    the oracle tightly - this isolates the max-softmax gradient and the per-step mean / running-sum coefficients."""
    import os
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict, predicate_counts
    from tests.golden_cases import GOLDEN
    fx = os.path.join(GOLDEN, "ref_fixtures") + os.sep
    aligned = torch.load(fx + "commonsense_aligned_triplets.pt")
    violated = torch.load(fx + "commonsense_violated_triplets.pt")
    for hier in (True, False):
        cfg = HeadConfig(hierarchical=hier)
        sd = make_state_dict(cfg, seed=31, head_gain=5.0)
        batch = make_scene_batch(cfg, (4, 3), seed=31, connect_frac=0.0)
        # make a few triplets aligned / violated so that both penalties and the "aligned -> no penalty" case occur
        al = dict(aligned); vi = dict(violated)
        if hier:
            from scene_graph_commonsense_amd.model import BayesianRelationClassifier as M
        else:
            from scene_graph_commonsense_amd.model import FlatRelationClassifier as M
        model = (M(cfg.args()) if hier else M(cfg.args(), output_dim=cfg.num_relations)).cuda()
        model.load_state_dict(sd)
        model.eval()
        sc = flatten_scene(cfg, batch, "cuda:0")
        out0 = model.forward_pairs(sc)
        pred = out0.cand_pred.cpu().numpy()
        cats = sc.cats.cpu().numpy()
        for k in range(0, sc.pidx.n_pairs, 3):                       # every third pair: first candidate aligned
            al[(int(cats[sc.pidx.sub[k]]), int(pred[k, 0]), int(cats[sc.pidx.obj[k]]))] = 1
        for k in range(1, sc.pidx.n_pairs, 4):                       # every fourth: last candidate violated
            vi[(int(cats[sc.pidx.sub[k]]), int(pred[k, -1]), int(cats[sc.pidx.obj[k]]))] = 1
        loss = model.training_step(sc, batch.relationships, batch.subj_or_obj, lambda_connectivity=0.0,
                                   commonsense=(list(al.keys()), list(vi.keys())))
        torch.cuda.synchronize()
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=O.class_weights(predicate_counts(cfg)),
                              lambda_connectivity=0.0, commonsense=(al, vi))
        ref["losses"].backward()
        print("cs-only loss", float(loss), float(ref["losses"]))
        assert abs(float(loss) - float(ref["losses"])) <= 1e-3 * abs(float(ref["losses"]))
        for n, p in model.named_parameters():
            if n.split(".")[0] in ("fc3", "fc3_1", "fc3_2", "fc3_3"):
                e = _fro(p.grad.float().cpu(), sdr[n].grad)
                assert e <= HEAD_TOL, (n, e)
            if n == "fc5.weight":      # softmax is shift invariant: the super-category head gets no gradient (autograd: round-off)
                assert float(p.grad.abs().max()) == 0.0
                assert float(sdr[n].grad.norm()) <= 1e-4 * float(sdr["fc3_1.weight"].grad.norm())


def test_fused_sgd_matches_torch_sgd():
    """optim.FusedSGD against torch.optim.SGD (momentum 0.9, weight decay 1e-4) over three steps on odd-sized tensors: the
    one-pass kernel may contract a + b*c into an fma, so agreement is to 2 ulp of the weights, not bitwise."""
    from scene_graph_commonsense_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(0)
    shapes = [(1031,), (257, 129), (4, 3, 3, 5), (1,)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    odd = torch.randn(64, generator=g).cuda()                                  # gradients may be unaligned views of larger buffers
    o_ref = torch.optim.SGD(ref, lr=0.05, momentum=0.9, weight_decay=1e-4)
    o_mine = FusedSGD(mine, lr=0.05, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        for a, b in zip(ref, mine):
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
            if a.numel() == 1:
                odd[5] = gr.reshape(())
                b.grad = odd[5:6]
        if step == 2:
            o_ref.param_groups[0]["lr"] = o_mine.param_groups[0]["lr"] = 0.01
        versions = [b._version for b in mine]
        o_ref.step(); o_mine.step()
        assert all(b._version > v for b, v in zip(mine, versions))           # in-place update is visible to version-based caches
        for a, b in zip(ref, mine):
            assert (a - b).abs().max().item() <= 3e-7 * max(1.0, a.abs().max().item())
            assert (o_ref.state[a]["momentum_buffer"] - o_mine.state[b]["momentum_buffer"]).abs().max().item() <= 1e-6


@pytest.mark.oracle_heavy
@pytest.mark.parametrize("name", ["vg_full_hit", "vg_full"])
def test_contrastive_step_with_device_routes_is_arithmetic_exact(name):
    """The contrastive training step against the ORACLE with the device's routing injected in both trunks (main view: every pair;
    augmented view: the connected pairs): loss, contrastive term and every parameter gradient at the routed bounds."""
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from tests.golden_cases import CASES
    from tests.train_case import fro, routed_model_step
    cfg, sd, batch, _ = load_case(name)
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(sd)
    model.eval()
    sc = flatten_scene(cfg, batch, "cuda:0")
    aug = _aug_features(batch, CASES[name][2])
    loss, grads, ref_loss, ref_grads, out = routed_model_step(model, sc, batch, aug=aug)
    lc, ref_lc = float(model.last_contrast_loss), float(out["loss_contrast"])
    print(name, "routed contrastive", lc, ref_lc, "total", loss, ref_loss)
    assert abs(lc - ref_lc) <= 2e-3 * max(1.0, abs(ref_lc))
    assert abs(loss - ref_loss) <= 2e-3 * abs(ref_loss)
    errs = {k: fro(grads[k], ref_grads[k]) for k in ref_grads}
    print({k: "%.1e" % v for k, v in errs.items()})
    for k, e in errs.items():
        assert e <= _routed_tol(k), (k, e)
    # and against the REAL reference's stored fingerprints of this step, bound derived from the routed run (``fingerprint_chain``)
    import os
    from tests.golden_cases import GOLDEN
    path = os.path.join(GOLDEN, name + "_contrast.npz")
    if os.path.exists(path):
        print(fingerprint_chain(dict(np.load(path)), "gradc_l2__", "gradc_sample__", grads, ref_grads))


@pytest.mark.oracle_heavy
def test_train_cs_step_with_device_routes_is_arithmetic_exact():
    """The train_cs step (commonsense penalty on top of the hierarchical loss) against the oracle with the device's routes."""
    import os
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from tests.golden_cases import GOLDEN
    from tests.train_case import fro, routed_model_step
    fx = os.path.join(GOLDEN, "ref_fixtures") + os.sep
    aligned = torch.load(fx + "commonsense_aligned_triplets.pt")
    violated = torch.load(fx + "commonsense_violated_triplets.pt")
    cfg, sd, batch, _ = load_case("vg_full_hit")
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(sd)
    model.eval()
    sc = flatten_scene(cfg, batch, "cuda:0")
    loss, grads, ref_loss, ref_grads, _ = routed_model_step(
        model, sc, batch, step_kw=dict(commonsense=(list(aligned.keys()), list(violated.keys()))),
        oracle_kw=dict(commonsense=(aligned, violated)))
    print("routed train_cs loss", loss, ref_loss)
    assert abs(loss - ref_loss) <= 2e-3 * abs(ref_loss)
    errs = {k: fro(grads[k], ref_grads[k]) for k in ref_grads}
    print({k: "%.1e" % v for k, v in errs.items()})
    for k, e in errs.items():
        assert e <= _routed_tol(k), (k, e)
    print(fingerprint_chain(dict(np.load(os.path.join(GOLDEN, "vg_full_hit_traincs.npz"))), "gradcs_l2__", "gradcs_sample__", grads, ref_grads))

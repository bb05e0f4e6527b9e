"""GPU parity of the fused forward path (HIP kernels via the C-ABI) against
  (a) golden vectors produced by the real reference (tests/golden/*.npz) and
  (b) the CPU oracle on the same seeded inputs.
Tolerance (BASELINE.json north_star): 1e-3 relative on the super-category and fine-relation outputs.  The
contractions run in f16 with f32 accumulation, so we bound the max abs error by 1e-3 x the output's max
magnitude (relative to scale) and report the per-element relative error on the log-probs as well."""
import numpy as np
import pytest
import torch

from tests.golden_cases import load_case

pytestmark = pytest.mark.gpu
REL_TOL = 1e-3


def _engine(cfg, sd):
    from scene_graph_commonsense_amd.engine import RelHeadEngine
    eng = RelHeadEngine(cfg, "cuda:0")
    eng.load_weights(sd)
    return eng


def _forward(cfg, sd, batch):
    from scene_graph_commonsense_amd.pairs import flatten_scene
    eng = _engine(cfg, sd)
    sc = flatten_scene(cfg, batch, "cuda:0")
    out = eng.forward_pairs(sc.image_feature, sc.image_depth, sc.obj_img, sc.bbox, sc.cats, sc.super_mh, sc.sub_idx,
                            sc.obj_idx)
    torch.cuda.synchronize()
    return sc, out


def _rel_err(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-12)


@pytest.mark.parametrize("name", ["vg_full", "oiv6_full", "vg_flat", "vg_full_hit"])
def test_forward_matches_reference_golden(name):
    cfg, sd, batch, gold = load_case(name)
    sc, out = _forward(cfg, sd, batch)
    pidx = sc.pidx
    rows = []
    for (g, e) in gold["eval_steps"].tolist():
        for first in (True, False):
            sel = np.nonzero((pidx.g == g) & (pidx.e == e) & (pidx.first == first))[0]
            rows.append(sel)
    rows = np.concatenate(rows)
    assert len(rows) == int(gold["eval_call_sizes"].sum())
    rel = out.relation.cpu().numpy()[rows]
    conn = out.connectivity.cpu().numpy()[rows]
    hid = out.hidden.cpu().numpy()[rows]
    if cfg.hierarchical:
        gold_rel = np.concatenate([gold["eval_rel1"], gold["eval_rel2"], gold["eval_rel3"]], axis=1)
        sup = out.super_relation.cpu().numpy()[rows]
        e_sup = _rel_err(sup, gold["eval_super"])
        assert e_sup <= REL_TOL, e_sup
        per_sup = np.abs(sup - gold["eval_super"]) / np.maximum(np.abs(gold["eval_super"]), 1.0)
        print(name, "per-element relative error of the super-category log-probs: max %.2e" % per_sup.max())
        assert per_sup.max() <= REL_TOL                                   # per element (floor 1 on the magnitude, as below)
    else:
        gold_rel = gold["eval_rel"]
    e_rel, e_conn, e_hid = _rel_err(rel, gold_rel), _rel_err(conn, gold["eval_conn"][:, 0]), _rel_err(hid, gold["eval_hidden"])
    print(name, "rel err (rel-to-scale): relation %.2e conn %.2e hidden %.2e" % (e_rel, e_conn, e_hid))
    assert e_rel <= REL_TOL and e_conn <= REL_TOL and e_hid <= REL_TOL
    # per element: every fine-relation log-prob within 1e-3 relative, with an absolute floor of 1e-3 x 1 for the entries
    # whose magnitude is below 1 (a log-prob near 0 is a probability near 1; its relative error is not meaningful)
    per_elem = np.abs(rel - gold_rel) / np.maximum(np.abs(gold_rel), 1.0)
    print(name, "per-element relative error of the fine-relation log-probs: max %.2e, 99.9%% %.2e" % (per_elem.max(), np.quantile(per_elem, 0.999)))
    assert per_elem.max() <= REL_TOL, per_elem.max()
    per_conn = np.abs(conn - gold["eval_conn"][:, 0]) / np.maximum(np.abs(gold["eval_conn"][:, 0]), 1.0)
    assert per_conn.max() <= REL_TOL, per_conn.max()
    # integer outputs: per-super-category argmax must be exact wherever the reference's top-2 gap is resolvable
    if cfg.hierarchical:
        segs = [(0, cfg.num_geometric), (cfg.num_geometric, cfg.num_geometric + cfg.num_possessive),
                (cfg.num_geometric + cfg.num_possessive, cfg.num_relations)]
        pred = out.cand_pred.cpu().numpy()[rows]
        for s, (lo, hi) in enumerate(segs):
            ref_seg = gold_rel[:, lo:hi]
            ref_arg = ref_seg.argmax(1) + lo
            top2 = np.sort(ref_seg, axis=1)[:, -2:]
            resolvable = (top2[:, 1] - top2[:, 0]) > 2 * REL_TOL * np.abs(gold_rel).max()
            assert (pred[resolvable, s] == ref_arg[resolvable]).all()
            print(name, "segment %d: %d rows, %d resolvable, %d argmax flips among the unresolvable"
                  % (s, len(ref_arg), int(resolvable.sum()), int((pred[:, s] != ref_arg).sum())))
            assert (pred[:, s] == ref_arg).mean() >= 0.95


def test_forward_matches_oracle_all_pairs():
    """All ordered pairs of a ragged minibatch (incl. edge-case boxes) against the CPU oracle."""
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=11, head_gain=6.0)
    batch = make_scene_batch(cfg, (4, 2, 3), seed=11, connect_frac=0.5, edge_boxes=True)
    sc, out = _forward(cfg, sd, batch)
    with torch.no_grad():
        ref = O.run_pair_loop(sd, batch, cfg, mode="eval", overlap_filtering=False)
    rel = torch.cat([r["relation"] for r in ref["records"]]).numpy()
    sup = torch.cat([r["super_relation"] for r in ref["records"]]).numpy()
    conn = torch.cat([r["connectivity"] for r in ref["records"]]).numpy()
    assert rel.shape[0] == sc.pidx.n_pairs
    for nm, a, b in (("relation", out.relation, rel), ("super", out.super_relation, sup), ("conn", out.connectivity, conn)):
        e = _rel_err(a.cpu().numpy(), b)
        print(nm, "%.2e" % e)
        assert e <= REL_TOL, (nm, e)


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_ragged_batches_with_float_boxes_match_oracle(seed):
    """Ragged minibatches (an image with a single object included) whose boxes are FLOATS as the SGDET front-end produces them -
    fractional, slightly negative and beyond the grid - so that int() truncation and Python slice clipping of the mask build
    (``train_test.py:164-169``) matter: forward outputs of every ordered pair and the training loss against the CPU oracle."""
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pairs import flatten_scene
    from scene_graph_commonsense_amd.synthetic import HeadConfig, hash_uniform, make_scene_batch, make_state_dict, predicate_counts
    cfg = HeadConfig()
    nobj = [(3, 1, 4), (2, 5), (4, 4, 2)][seed % 3]
    sd = make_state_dict(cfg, seed=seed, head_gain=5.0)
    batch = make_scene_batch(cfg, nobj, seed=seed, connect_frac=0.5)
    for b in range(len(nobj)):
        jitter = torch.from_numpy(hash_uniform(seed * 7 + b, 4 * nobj[b], -1.6, 1.6).reshape(nobj[b], 4))
        fb = batch.bbox[b].float() + jitter
        fb[0, 0] = -0.4                      # int(-0.4) = 0
        if nobj[b] > 1:
            fb[1, 1] = 33.7                  # beyond the grid: slice clips to 32
            fb[1, 2] = -2.3                  # int(-2.3) = -2 -> slice start 30
        batch.bbox[b] = fb
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(sd)
    model.eval()
    sc = flatten_scene(cfg, batch, "cuda:0")
    out = model.forward_pairs(sc)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=O.class_weights(predicate_counts(cfg)))
    rel = torch.cat([r["relation"] for r in ref["records"]]).numpy()
    conn = torch.cat([r["connectivity"] for r in ref["records"]]).numpy()
    assert rel.shape[0] == sc.n_pairs == sum(n * (n - 1) for n in nobj)
    assert _rel_err(out.relation.cpu().numpy(), rel) <= REL_TOL and _rel_err(out.connectivity.cpu().numpy(), conn) <= REL_TOL
    loss = model.training_step(sc)
    assert abs(float(loss) - float(ref["losses"])) <= 2e-3 * abs(float(ref["losses"]))


@pytest.mark.parametrize("hier", [True, False])
def test_head_kernel_against_float64_and_independent_of_a_pairs_position(hier):
    """sgc_bayes_head (reference model.py:176-184 + the evaluator's per-range max / first argmax, evaluator.py:160-174) on random hidden
    vectors: (a) log-probabilities against the same head in float64; (b) a pair's outputs do not depend on where it sits in the batch -
    the kernel walks four pairs per wavefront pass, the batch sizes below put every pair in every slot and exercise the ragged last
    group and both workgroup sizes - bit for bit."""
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_state_dict
    cfg = HeadConfig(hierarchical=hier)
    sd = make_state_dict(cfg, seed=5, head_gain=6.0)
    eng = _engine(cfg, sd)
    eng.T = (1.0, 2.0, 0.5)
    P = 8 * 256 * 4 + 7                                     # above the eight-wavefront threshold, ragged
    g = torch.Generator().manual_seed(9)
    hid = torch.relu(torch.randn(P, 512, generator=g)).cuda().contiguous()
    mask = (torch.rand(P, generator=g) > 0.2).to(torch.uint8).cuda()
    full = eng.head(hid.flatten(), P, mask)
    torch.cuda.synchronize()
    # (a) float64
    W = {k: v.double().cuda() for k, v in sd.items() if k.startswith("fc")}
    h = hid.double()
    if hier:
        sup = torch.log_softmax(h @ W["fc5.weight"].T + W["fc5.bias"], dim=1)
        rels = [torch.log_softmax((h @ W["fc3_%d.weight" % (k + 1)].T + W["fc3_%d.bias" % (k + 1)]) / eng.T[k], dim=1) + sup[:, k:k + 1]
                for k in range(3)]
        rel = torch.cat(rels, dim=1)
        assert float((full.super_relation.double() - sup).abs().max()) <= 1e-5
        lo = 0
        for k, r in enumerate(rels):
            cm, am = r.max(dim=1)
            ok = mask.bool()
            assert float((full.cand_conf[ok, k].double() - cm[ok]).abs().max()) <= 1e-5
            assert bool(torch.isinf(full.cand_conf[~ok, k]).all())
            agree = (full.cand_pred[:, k].long() == am + lo)
            assert float(agree.float().mean()) >= 0.999          # near-ties of two log-probabilities in f32
            lo += r.shape[1]
    else:
        rel = h @ W["fc3.weight"].T + W["fc3.bias"]
    assert float((full.relation.double() - rel).abs().max()) <= 1e-5 * max(1.0, float(rel.abs().max()))
    conn = h @ W["fc4.weight"].T + W["fc4.bias"]
    assert float((full.connectivity.double() - conn[:, 0]).abs().max()) <= 1e-5 * max(1.0, float(conn.abs().max()))
    # (b) position independence
    for start, n in ((1, 5), (2, 9), (3, 1030), (5, 4)):
        part = eng.head(hid[start:start + n].contiguous().flatten(), n, mask[start:start + n].contiguous())
        torch.cuda.synchronize()
        for name in ("relation", "connectivity", "cand_conf", "cand_pred") + (("super_relation",) if hier else ()):
            a, b = getattr(part, name), getattr(full, name)[start:start + n]
            assert torch.equal(a, b), (name, start, n)

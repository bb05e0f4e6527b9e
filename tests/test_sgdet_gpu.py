"""GPU: end-to-end SGDET evaluation - DETR outputs -> HIP object front-end -> fused pair path over the PREDICTED objects ->
drop-in Evaluator with predcls=False - against the golden vectors of the real reference (tests/golden/make_sgdet_golden.py)."""
import os

import numpy as np
import pytest
import torch

from tests import sgdet_case
from tests.golden_cases import GOLDEN

pytestmark = pytest.mark.gpu
FX = os.path.join(GOLDEN, "ref_fixtures") + os.sep
GOLD = np.load(os.path.join(GOLDEN, "sgdet_vg.npz"))


def test_sgdet_end_to_end_matches_reference():
    from scene_graph_commonsense_amd.evaluator import Evaluator
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.object_frontend import DetrFrontEnd
    from scene_graph_commonsense_amd.pair_loop import evaluate_sgdet_minibatch
    from scene_graph_commonsense_amd.synthetic import default_sub2super
    cfg, sd, batch, logits, boxes = sgdet_case.make_case()
    sgdet_case.apply_stored_targets(batch, GOLD)
    args = cfg.args(fixtures=FX)
    model = BayesianRelationClassifier(args).cuda()
    model.load_state_dict(sd)
    model.eval()
    fe = DetrFrontEnd(sgdet_case.alp2fre_table().tolist())
    cats, confs, bxs, kept = fe.sgdet(logits.cuda(), boxes.cuda())
    assert kept == [0, 1, 2]
    for i in kept:                                               # the front end reproduces the reference's object lists
        np.testing.assert_array_equal(cats[i].cpu().numpy(), GOLD["fe_cat_%d" % i])
        np.testing.assert_array_equal(bxs[i].cpu().numpy(), GOLD["fe_box_%d" % i])
        np.testing.assert_allclose(confs[i].cpu().numpy(), GOLD["fe_conf_%d" % i], rtol=2e-6)
    ev = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    evaluate_sgdet_minibatch(model, batch.image_feature.cuda(), batch.image_depth.cuda(), cats, confs, bxs, ev,
                             sub2super=default_sub2super(cfg.num_classes, cfg.num_super_classes),
                             targets=(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox))
    np.testing.assert_array_equal(ev.which_in_batch.cpu().numpy(), GOLD["ev_which"])
    np.testing.assert_array_equal(ev.subject_cat_pred.cpu().numpy(), GOLD["ev_scat"])
    np.testing.assert_array_equal(ev.object_cat_pred.cpu().numpy(), GOLD["ev_ocat"])
    conf, ref = ev.confidence.cpu().numpy(), GOLD["ev_conf"]
    assert (np.isinf(conf) == np.isinf(ref)).all()
    fin = np.isfinite(ref)
    assert np.abs(conf[fin] - ref[fin]).max() <= 1e-3 * np.abs(ref[fin]).max()        # log-prob + category confidences
    assert (ev.relation_pred.cpu().numpy() == GOLD["ev_pred"]).mean() >= 0.95
    res = ev.compute(per_class=True, predcls=False)
    assert float(ev.num_connected_target) == float(GOLD["num_connected_target"])
    np.testing.assert_allclose(np.array([float(r) for r in res[0]]), GOLD["recall"], atol=0.1)
    assert [float(ev.result_dict[k]) for k in (20, 50, 100)] == GOLD["hits"].tolist()
    # the skip of filtered pairs (images here are below the top-K limit, so everything is still computed) gives the same state
    ev2 = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    evaluate_sgdet_minibatch(model, batch.image_feature.cuda(), batch.image_depth.cuda(), cats, confs, bxs, ev2,
                             sub2super=default_sub2super(cfg.num_classes, cfg.num_super_classes),
                             targets=(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox), skip_filtered=True)
    np.testing.assert_array_equal(ev2.confidence.cpu().numpy(), conf)


def test_sgcls_path_matches_oracle():
    """SGCLS (evaluate.py:590-690): ground-truth boxes carrying the MATCHED predicted categories go through the same fused path;
    checked against the oracle's literal loop on the same matched lists."""
    from oracle import relhead_oracle as ro
    from scene_graph_commonsense_amd.evaluator import Evaluator
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.object_frontend import DetrFrontEnd
    from scene_graph_commonsense_amd.pair_loop import evaluate_sgdet_minibatch
    from scene_graph_commonsense_amd.synthetic import default_sub2super
    cfg, sd, batch, logits, boxes = sgdet_case.make_case()
    sgdet_case.apply_stored_targets(batch, GOLD)
    args = cfg.args(fixtures=FX)
    model = BayesianRelationClassifier(args).cuda()
    model.load_state_dict(sd)
    model.eval()
    fe = DetrFrontEnd(sgdet_case.alp2fre_table().tolist())
    cats, confs, bxs, kept = fe.sgdet(logits.cuda(), boxes.cuda())
    m, mc, tm = fe.match_object_categories(cats, confs, bxs, [b.float().cuda() for b in batch.bbox])
    assert sum(len(x) for x in m) > sum(sgdet_case.NOBJ)           # repeated-box ties duplicate some ground-truth boxes
    ev = Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    evaluate_sgdet_minibatch(model, batch.image_feature.cuda(), batch.image_depth.cuda(), m, mc, tm, ev,
                             sub2super=default_sub2super(cfg.num_classes, cfg.num_super_classes),
                             targets=(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox))
    mcpu, mccpu, tmcpu = [x.cpu() for x in m], [x.cpu() for x in mc], [x.cpu() for x in tm]
    oev = ro.OracleEvaluator(cfg, zero_shot_triplets=[])
    with torch.no_grad():
        ro.run_sgdet_loop(sd, batch.image_feature, batch.image_depth, mcpu, mccpu, tmcpu, sgdet_case.super_categories_of(mcpu, cfg), cfg, oev)
    oev.accumulate_target(*[ro.match_target_sgd(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox)[i] for i in (4, 0, 1, 2, 3)])
    st = oev.flat_state()
    np.testing.assert_array_equal(ev.which_in_batch.cpu().numpy(), st["which"].numpy())
    np.testing.assert_array_equal(ev.subject_cat_pred.cpu().numpy(), st["scat"].numpy())
    conf, ref = ev.confidence.cpu().numpy(), st["conf"].numpy()
    assert (np.isinf(conf) == np.isinf(ref)).all()
    fin = np.isfinite(ref)
    assert np.abs(conf[fin] - ref[fin]).max() <= 1e-3 * np.abs(ref[fin]).max()
    res = ev.compute(per_class=True, predcls=False)
    ores = oev.compute(per_class=True, predcls=False)
    assert float(ev.num_connected_target) == float(oev.num_connected_target)
    np.testing.assert_allclose(np.array([float(r) for r in res[0]]), np.array([float(r) for r in ores[0]]), atol=0.1)

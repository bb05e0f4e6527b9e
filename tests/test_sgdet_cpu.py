"""CPU: the SGDET end-to-end oracle (front-end restatement + pair loop over predicted objects + predcls=False evaluator) against
the golden vectors produced by the real reference classifier / Evaluator / match_target_sgd (tests/golden/make_sgdet_golden.py)."""
import os

import numpy as np
import torch

from oracle import frontend_oracle as fo
from oracle import relhead_oracle as ro
from tests import sgdet_case

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgdet_vg.npz"))


def test_sgdet_oracle_matches_reference():
    cfg, sd, batch, logits, boxes = sgdet_case.make_case()
    sgdet_case.apply_stored_targets(batch, GOLD)
    cats, confs, bxs, kept = fo.frontend_sgdet(logits, boxes, sgdet_case.alp2fre_table().tolist())
    assert kept == [0, 1, 2]
    for i in kept:
        assert np.array_equal(cats[i].numpy(), GOLD["fe_cat_%d" % i]) and np.array_equal(bxs[i].numpy(), GOLD["fe_box_%d" % i])
        assert np.array_equal(confs[i].numpy(), GOLD["fe_conf_%d" % i])
        assert len(cats[i]) < 2 * (sgdet_case.NOBJ[i] + 3)               # the near-duplicate was suppressed
    sp = sgdet_case.super_categories_of(cats, cfg)
    ev = ro.OracleEvaluator(cfg, zero_shot_triplets=[])
    with torch.no_grad():
        ro.run_sgdet_loop(sd, batch.image_feature, batch.image_depth, cats, confs, bxs, sp, cfg, ev)
    cs, co, bs, bo, rt = ro.match_target_sgd(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox)
    for i in kept:
        assert np.array_equal(rt[i].numpy(), GOLD["mt_rel_%d" % i])
    ev.accumulate_target(rt, cs, co, bs, bo)
    st = ev.flat_state()
    assert np.array_equal(st["pred"].numpy(), GOLD["ev_pred"]) and np.array_equal(st["which"].numpy(), GOLD["ev_which"])
    assert np.array_equal(st["scat"].numpy(), GOLD["ev_scat"]) and np.array_equal(st["ocat"].numpy(), GOLD["ev_ocat"])
    fin = np.isfinite(GOLD["ev_conf"])
    assert np.array_equal(np.isfinite(st["conf"].numpy()), fin)
    assert np.allclose(st["conf"].numpy()[fin], GOLD["ev_conf"][fin], rtol=1e-4, atol=1e-4)
    rk, rpc, mrk, rk_zs, _, _ = ev.compute(per_class=True, predcls=False)
    assert float(ev.num_connected_target) == float(GOLD["num_connected_target"])
    assert [ev.result_dict[k] for k in (20, 50, 100)] == GOLD["hits"].tolist()
    assert np.allclose([float(r) for r in rk], GOLD["recall"])
    assert np.allclose(np.stack([r.numpy() for r in rpc]), GOLD["recall_per_class"], equal_nan=True)


def test_compare_object_cat_and_targets():
    assert ro.compare_object_cat(5, 149) and ro.compare_object_cat(123, 14) and ro.compare_object_cat(14, 123)
    assert not ro.compare_object_cat(14, 63) and not ro.compare_object_cat(2, 3)
    rel = [[torch.tensor([7]), torch.tensor([-1, 3])]]
    sd = [[torch.tensor([1.0]), torch.tensor([-1.0, 0.0])]]
    cats = [torch.tensor([10, 11, 12])]
    box = [torch.tensor([[0, 4, 0, 4], [1, 5, 1, 5], [2, 6, 2, 6]])]
    cs, co, bs, bo, rt = ro.match_target_sgd(rel, sd, cats, box)
    # reference quirk kept on purpose: the loop runs graph_iter over range(len(relationships)) = 0..n-2, so the relations
    # whose "graph" object is the LAST object of the image (here the (1,2) relation) are never collected
    assert cs[0].tolist() == [11] and co[0].tolist() == [10] and rt[0].tolist() == [7]
    assert bs[0].tolist() == [[1, 5, 1, 5]] and bo[0].tolist() == [[0, 4, 0, 4]]
    none = ro.match_target_sgd([[torch.tensor([-1])]], [[torch.tensor([-1.0])]], [torch.tensor([1, 2])], [torch.zeros(2, 4)])
    assert none[4] == [None]

"""CPU tests (no GPU): C-ABI exports, host-side logic of the fused path against the oracle, loud failure
without a GPU, and the world_size-2 data-parallel glue over gloo."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import relhead_oracle as O
from scene_graph_commonsense_amd import pairs as PR
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict, predicate_counts
from tests.golden_cases import GOLDEN, load_case

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_library_exports_every_declared_symbol():
    from scene_graph_commonsense_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "sgc_relhead.h")).read()
    names = re.findall(r"\bint\s+(sgc_\w+)\s*\(", hdr)
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n


def test_pair_enumeration_matches_reference_order():
    cfg = HeadConfig(hidden_dim=16, feature_size=8)
    batch = make_scene_batch(cfg, (5, 2, 4, 1), seed=9, connect_frac=0.5)
    sd = make_state_dict(cfg, seed=9)
    with torch.no_grad():
        ref = O.run_pair_loop(sd, batch, cfg, mode="eval", overlap_filtering=False)
    pidx = PR.enumerate_pairs([5, 2, 4, 1])
    recs = ref["records"]
    assert pidx.call_sizes.tolist() == [len(r["keep"]) for r in recs]
    assert pidx.image.tolist() == torch.cat([r["keep"] for r in recs]).tolist()
    assert pidx.g.tolist() == sum([[r["g"]] * len(r["keep"]) for r in recs], [])
    assert pidx.first.tolist() == sum([[r["first"]] * len(r["keep"]) for r in recs], [])
    directed = PR.pair_targets_fast(batch.relationships, batch.subj_or_obj, pidx)
    assert directed.tolist() == torch.cat([r["target"] for r in recs]).tolist()
    d2, _ = PR.pair_targets(batch.relationships, batch.subj_or_obj, pidx)
    assert (d2 == directed).all()


def test_box_slice_semantics_and_super_multihot():
    b = torch.tensor([[-3, 40, 2.9, 31.2], [5, 3, -40, 0]])
    n = PR.normalise_boxes(b, 32)
    m = torch.zeros(2, 32, 32, dtype=torch.bool)
    for j in range(2):
        m[j, int(b[j][2]):int(b[j][3]), int(b[j][0]):int(b[j][1])] = 1
        area = max(n[j][1] - n[j][0], 0) * max(n[j][3] - n[j][2], 0)
        assert int(m[j].sum()) == area
    mh = PR.super_multihot([[torch.tensor([3, 4, 6]), torch.tensor([2])]], 17)
    assert mh[0].nonzero()[0].tolist() == [3, 6] and mh[1].nonzero()[0].tolist() == [2]
    assert (mh == O.super_class_multihot([torch.tensor([3, 4, 6]), torch.tensor([2])], 17).numpy()).all()


@pytest.mark.parametrize("name", ["vg_small", "vg_bert_small"])
def test_loss_coefficients_reproduce_reference_loss(name):
    """Folding the per-step loss bookkeeping (running-sum quirk, BCE overwrite, weighted NLL) into per-pair
    coefficients must give the reference's scalar loss when applied to the oracle's forward outputs."""
    from scene_graph_commonsense_amd.engine import loss_coefficients
    cfg, sd, batch, gold = load_case(name)
    with torch.no_grad():
        ref = O.run_pair_loop(sd, batch, cfg, mode="eval", overlap_filtering=False)
    recs = ref["records"]
    rel = torch.cat([r["relation"] for r in recs]).double().numpy()
    sup = torch.cat([r["super_relation"] for r in recs]).double().numpy()
    conn = torch.cat([r["connectivity"] for r in recs]).double().numpy()
    pidx = PR.enumerate_pairs(batch.num_objects)
    directed = PR.pair_targets_fast(batch.relationships, batch.subj_or_obj, pidx)
    counts = predicate_counts(cfg).numpy()
    tgt, a, b, c, y = loss_coefficients(cfg, pidx.step, len(pidx.call_sizes), directed, 1 - counts / counts.sum())
    ng, npos = cfg.num_geometric, cfg.num_possessive
    t = np.maximum(tgt, 0)
    st = np.where(t < ng, 0, np.where(t < ng + npos, 1, 2))
    idx = np.arange(len(t))
    softplus = lambda x: np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))
    loss = np.where(tgt >= 0, -a * sup[idx, st] - b * rel[idx, t], 0.0) + c * np.where(y > 0.5, softplus(-conn), softplus(conn))
    np.testing.assert_allclose(loss.sum(), gold["train_loss"][0], rtol=2e-5)


def test_grid_iou_closed_form_equals_rasterised():
    from scene_graph_commonsense_amd.evaluator import first_hit, grid_iou_matrix
    rng = np.random.RandomState(0)
    bt = rng.randint(-2, 36, size=(12, 4))
    bp = rng.randint(-2, 36, size=(9, 4))
    m = grid_iou_matrix(bt, bp, 32)
    for i in range(12):
        for j in range(9):
            assert abs(m[i, j] - O.grid_iou(torch.tensor(bt[i]), torch.tensor(bp[j]), 32)) < 1e-12
    ok = np.array([[0, 1, 1], [0, 0, 0]], dtype=bool)
    assert first_hit(ok, ok, ok).tolist() == [1, 3]


@pytest.mark.parametrize("name", ["vg_small", "vg_bert_small"])
def test_evaluator_host_matching_against_reference(name, monkeypatch):
    """Evaluator host logic (append order, blocked candidates, target selection, counters) with the two device steps - ranking
    and hit test - replaced by CPU stand-ins FOR THIS TEST ONLY (the product refuses CPU tensors; the kernels are pinned to
    the same goldens by tests/test_evaluator_gpu.py)."""
    from scene_graph_commonsense_amd import evaluator as EV

    def cpu_rank(conf, which, K):
        """CPU stand-in of ``rank_topk_device``: stable descending sort per image."""
        images, counts = torch.unique(which, return_counts=True)
        order = torch.sort(which, stable=True)[1]
        seg = torch.zeros(len(images) + 1, dtype=torch.int32)
        seg[1:] = torch.cumsum(counts, 0).int()
        top = torch.full((len(images), K), -1, dtype=torch.int32)
        cnt = torch.zeros(len(images), dtype=torch.int32)
        keep_pos = torch.full((len(images), K), -1, dtype=torch.int32)
        for r in range(len(images)):
            pos = order[int(seg[r]):int(seg[r + 1])]
            o = torch.sort(conf[pos], descending=True, stable=True)[1][:K]
            top[r, :len(o)] = o.int()
            keep_pos[r, :len(o)] = pos[o].int()
            cnt[r] = len(o)
        return images, keep_pos, cnt, top, seg, order

    def cpu_hits(cand, keep_pos, keep_cnt, targets, K, feature_size, iou_thresh, equiv=None):
        """CPU stand-in of ``recall_hits`` built on the module's numpy helpers (the kernel itself is pinned by the GPU tests)."""
        n_t = int(targets["rel"].shape[0])
        hit = np.full(n_t, K, dtype=np.int32)
        pred = cand["pred"].numpy().reshape(len(cand["scat"]), -1)
        for t in range(n_t):
            r = int(targets["row"][t])
            keep = keep_pos[r, :int(keep_cnt[r])].long().numpy()
            label = (cand["scat"].numpy()[keep] == int(targets["scat"][t])) & (cand["ocat"].numpy()[keep] == int(targets["ocat"][t]))
            iou = (EV.grid_iou_matrix(targets["sbox"][t:t + 1].numpy(), cand["sbox"].numpy()[keep], feature_size)[0] >= iou_thresh) & \
                (EV.grid_iou_matrix(targets["obox"][t:t + 1].numpy(), cand["obox"].numpy()[keep], feature_size)[0] >= iou_thresh)
            ok = label & iou & (pred[keep] == int(targets["rel"][t])).any(1)
            if ok.any():
                hit[t] = int(np.argmax(ok))
        return hit
    monkeypatch.setattr(EV, "rank_topk_device", cpu_rank)
    monkeypatch.setattr(EV, "recall_hits", cpu_hits)
    cfg, sd, batch, gold = load_case(name)
    args = cfg.args(fixtures=os.path.join(GOLDEN, "ref_fixtures") + os.sep)
    ev = EV.Evaluator(args, cfg.num_relations, 0.5, [20, 50, 100])
    t3 = EV.Evaluator_Top3(args, cfg.num_relations, 0.5, [20, 50, 100])
    with torch.no_grad():
        O.run_pair_loop(sd, batch, cfg, mode="eval", evaluator=ev, evaluator_top3=t3)
    res = ev.compute(per_class=True)
    np.testing.assert_allclose(np.array(res[0]), gold["ev_recall"], atol=1e-12)
    np.testing.assert_allclose(np.array(res[3]), gold["ev_recall_zs"], atol=1e-12)
    np.testing.assert_allclose(torch.stack(res[1]).numpy(), gold["ev_recall_per_class"], atol=1e-6, equal_nan=True)
    np.testing.assert_allclose(np.array(t3.compute()[0]), gold["top3_recall"], atol=1e-12)


def test_product_path_fails_loudly_without_gpu():
    from scene_graph_commonsense_amd.evaluator import rank_topk
    from scene_graph_commonsense_amd.model import BayesianHead, BayesianRelationClassifier
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args())
    x = torch.zeros(1, 257, 32, 32)
    with pytest.raises(RuntimeError, match="GPU"):
        model(x, x, torch.tensor([1]), torch.tensor([2]), [torch.tensor([0])], [torch.tensor([1])], "cpu")
    with pytest.raises(RuntimeError, match="GPU"):
        BayesianHead()(torch.zeros(2, 512))
    with pytest.raises(RuntimeError, match="GPU"):
        rank_topk(torch.zeros(4), torch.zeros(4, dtype=torch.long), 3)
    # parameter names / shapes follow the reference state_dict contract
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes["fc1.weight"] == (4096, 65536) and shapes["fc2.weight"] == (512, 4430)
    assert shapes["conv1_1.weight"] == (128, 257, 1, 1) and shapes["fc3_3.weight"] == (24, 512)
    assert sum(v.numel() for v in model.state_dict().values()) == 276701750


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scene_graph_commonsense_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    params = [("fc1.weight", torch.nn.Parameter(torch.zeros(8, 4))), ("fc2.bias", torch.nn.Parameter(torch.zeros(5))),
              ("conv1_1.weight", torch.nn.Parameter(torch.zeros(3)))]
    red = D.GradReducer(w)
    big = torch.full((8, 4), float(rank + 1))
    red.hook("fc1.weight", big)                   # early, asynchronous
    params[0][1].grad = big
    params[1][1].grad = torch.arange(5.0) * (rank + 1)
    params[2][1].grad = torch.full((3,), 10.0 * rank)
    red.finish(params)
    cnt = D.allreduce_counters(torch.tensor([rank + 1, 7]))
    # the form training_step uses: the step's gradient dict is reduced BEFORE accumulation into param.grad; a non-contiguous
    # view (conv weights come out of a permute) must survive the flat bucket, and the early tensor must not be reduced twice
    red2 = D.GradReducer(w)
    g_fc1 = torch.full((8, 4), 10.0 * (rank + 1))
    red2.hook("fc1.weight", g_fc1)
    grads = {"fc1.weight": g_fc1, "conv3_1.weight": (torch.arange(12.0).view(3, 4) * (rank + 1)).t(), "fc2.bias": torch.ones(5) * rank}
    red2.finish_grads(grads)
    mean1 = (w + 1) / 2.0                         # mean over ranks of (rank + 1)
    assert abs(grads["fc1.weight"][0, 0].item() - 10.0 * mean1) < 1e-5 and grads["conv3_1.weight"].is_contiguous()
    assert torch.allclose(grads["conv3_1.weight"], torch.arange(12.0).view(3, 4).t() * mean1) and torch.allclose(grads["fc2.bias"], torch.full((5,), mean1 - 1))
    assert not red2.pending and not red2.done
    q.put((rank, params[0][1].grad[0, 0].item(), params[1][1].grad.tolist(), params[2][1].grad.tolist(), cnt.tolist()))
    dist.destroy_process_group()


def _spawn(target, world, extra=(), timeout=240):
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:                       # a port the OS knows to be free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    [p.start() for p in procs]
    try:
        res = sorted((q.get(timeout=timeout) for _ in range(world)), key=lambda t: t[0])
    finally:
        [p.join(60) for p in procs]
        [p.kill() for p in procs if p.is_alive()]
    return res


@pytest.mark.parametrize("world", [2, 3, 8])
def test_data_parallel_gradient_allreduce_gloo(world):
    """``GradReducer`` over gloo at 2, 3 and 8 ranks (8 = the node the path is built for: ``/root/reference/train_test.py:72-80``)."""
    res = _spawn(_dp_worker, world)
    m = (world + 1) / 2.0
    for rank, g0, g1, g2, cnt in res:
        assert abs(g0 - m) < 1e-6                                    # mean of 1 .. world
        assert np.allclose(g1, [m * i for i in range(5)])
        assert np.allclose(g2, [10.0 * (m - 1)] * 3)
        assert cnt == [world * (world + 1) // 2, 7 * world]


def test_commonsense_bitmap_packing_host():
    from scene_graph_commonsense_amd.commonsense import TripletBitmaps
    keys = [(0, 0, 0), (1, 2, 3), (149, 49, 149), (5, 7, 31), (5, 7, 32)]
    bm = TripletBitmaps(keys, [(1, 2, 3)], 150, 50, "cpu")
    words = bm.aligned.numpy().view(np.uint32)
    def bit(w, s, r, o):
        b = (s * 50 + r) * 150 + o
        return (int(w[b >> 5]) >> (b & 31)) & 1
    for k in keys:
        assert bit(words, *k) == 1
    assert bit(words, 2, 2, 3) == 0 and bit(words, 5, 7, 33) == 0
    assert int(sum(bin(int(x)).count("1") for x in words)) == len(keys)
    assert bit(bm.violated.numpy().view(np.uint32), 1, 2, 3) == 1


def test_match_target_sgd_host_matches_oracle():
    """pairs.match_target_sgd (vectorised) against the literal oracle restatement of utils.py:294-350, incl. its loop-bound quirk."""
    from oracle import relhead_oracle as ro
    from scene_graph_commonsense_amd.pairs import match_target_sgd
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    batch = make_scene_batch(HeadConfig(), (6, 5, 2, 1), seed=9, connect_frac=0.5)
    got = match_target_sgd(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox)
    want = ro.match_target_sgd(batch.relationships, batch.subj_or_obj, batch.categories, batch.bbox)
    n_rel = 0
    for g_l, w_l in zip(got, want):
        assert len(g_l) == len(w_l) == 4
        for g, w in zip(g_l, w_l):
            assert (g is None) == (w is None)
            if g is not None:
                assert torch.equal(g.to(w.dtype), w)
                n_rel += len(g)
    assert n_rel > 0 and want[4][3] is None


def test_annotation_files_round_trip(tmp_path):
    """annotations.py against a literal restatement of dataloader.py:111-147 on files written in the reference's on-disk format
    (dataset_utils.py:186-196), incl. the drop rules and the predicate re-indexing table taken from the reference."""
    import numpy as np
    from scene_graph_commonsense_amd import annotations as AN
    table = torch.from_numpy(np.load(os.path.join(GOLDEN, "ref_fixtures", "relation_class_freq2scat.npy")))
    assert torch.equal(AN.RELATION_CLASS_FREQ2SCAT, table)
    cfg = HeadConfig()
    g = torch.Generator().manual_seed(3)
    paths, want = [], []
    for k, n in enumerate((5, 1, 21, 3)):
        b = make_scene_batch(cfg, (n,), seed=40 + k, connect_frac=0.5)
        rels = [torch.randint(0, 50, r.shape, generator=g) * (r >= 0) + r * (r < 0) for r in b.relationships[0]]   # frequency-order ids
        if rels:
            rels[0][0] = 12
        annot = dict(image_depth=b.image_depth[0], curr_instance=list(range(n)), num_relations=0, categories=b.categories[0],
                     super_categories=b.super_categories[0], masks=torch.zeros(n, 32, 32, dtype=torch.uint8),
                     bbox=b.bbox[0].float() + 0.4, bbox_origin=b.bbox[0].float() * 10, relationships=rels, subj_or_obj=b.subj_or_obj[0])
        p = str(tmp_path / ("img%d_annotations.pkl" % k))
        torch.save(annot, p)
        paths.append(p)
        # literal restatement of the loader
        if n <= 1 or n > 20:
            want.append(None)
        else:
            rr = []
            for rel in [r.clone() for r in rels]:
                rel[rel == 12] = 4
                rr.append(table[rel])
            want.append(dict(bbox=(b.bbox[0].float() + 0.4).int(), relationships=rr))
    got = [AN.load_annotation(p, image_hw=(480, 640)) for p in paths]
    assert [x is None for x in got] == [False, True, True, False]
    for gt, w in zip(got, want):
        if w is not None:
            assert torch.equal(gt["bbox"], w["bbox"]) and gt["bbox"].dtype == torch.int32
            assert all(torch.equal(a, b_) for a, b_ in zip(gt["relationships"], w["relationships"]))
    assert int(got[0]["relationships"][0][0]) == int(table[4])            # wears -> wearing -> super-category order
    feat = torch.randn(4, 256, 32, 32)
    batch, keep = AN.collate(feat, got)
    assert keep == [0, 3] and batch.image_feature.shape[0] == 2 and batch.num_objects == [5, 3]
    assert torch.equal(batch.image_feature[1], feat[3]) and len(batch.relationships[0]) == 4
    # an object that is empty at image resolution drops the image (dataloader.py:123-128)
    a = torch.load(paths[0]); a["bbox"][0] = torch.tensor([3.0, 3.0, 4.0, 9.0])
    assert AN.prepare_annotation(a, image_hw=(480, 640)) is None and AN.prepare_annotation(a) is not None


def test_oracle_injection_hooks_are_neutral_and_exact():
    """The test-only hooks of the oracle: injecting the true routes / all-ones masks reproduces the plain forward, and the
    host replica of the dropout hash is deterministic with rate 1/2."""
    import torch.nn.functional as F
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.synthetic import dropout_keep_mask
    g = torch.Generator().manual_seed(3)
    c = torch.randn(2, 5, 8, 6, generator=g)
    ref, idx = F.max_pool2d(F.relu(c), 2, 2, return_indices=True)
    yy, xx = idx // 6, idx % 6
    code = (yy % 2) * 2 + (xx % 2)
    code[ref <= 0] = 4
    assert torch.equal(O.routed_relu_pool(c, code), ref)
    m = dropout_keep_mask(77, 64, 4096)
    assert abs(m.mean() - 0.5) < 0.01 and np.array_equal(m[10:20], dropout_keep_mask(77, 10, 4096, row0=10))
    assert not np.array_equal(m, dropout_keep_mask(78, 64, 4096))
    # independent evaluation of the hash for a few elements (csrc/common.h:dropout_keep)
    def keep(seed, idx):
        x = (idx ^ ((seed * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF)) & 0xFFFFFFFF
        x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16
        return bool((x >> 16) & 1)
    for r, cc in ((0, 0), (3, 17), (63, 4095)):
        assert m[r, cc] == keep(77, r * 4096 + cc)


def test_bench_launcher_starts_every_rank_gloo_world2():
    """``bench.py --gpus 2`` without torchrun is its own launcher (the flag used to be dead): two child ranks
    rendezvous over gloo on 127.0.0.1, rank 0 prints ONE JSON line with n_gpus == 2; a world size that disagrees with --gpus
    is refused instead of silently running one rank."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run",
                        "--backend", "gloo"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_launcher_eight_ranks_dry_run_both_launch_forms():
    """The driver's 8-GPU invocation without the GPUs: ``bench.py --gpus 8 --dry-run`` as its own launcher AND under
    ``python -m torch.distributed.run --nproc-per-node 8`` (how the driver starts it): eight ranks rendezvous on 127.0.0.1, one JSON
    line from rank 0 with n_gpus = ranks_seen = 8, MAX-reduced time (rank r sleeps r + 1 ms per step)."""
    import json
    import socket
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run",
                        "--backend", "gloo"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["steps"] == 3 and out["ms_per_step"] >= 8.0
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run",
                        "--backend", "gloo"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8


def test_dropin_shims_resolve_like_main_py(tmp_path):
    """``from train_test import training`` / ``from evaluate import eval_pc, eval_sgc, eval_sgd`` / ``from model import *`` exactly as
    main.py / train_test.py of the reference write them, with scene_graph_commonsense_amd/dropin first on sys.path."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from train_test import training, testing, setup\n"
            "from evaluate import eval_pc, eval_sgc, eval_sgd\n"
            "from model import BayesianRelationClassifier, FlatRelationClassifier, BayesianHead\n"
            "from evaluator import Evaluator, Evaluator_Top3\n"
            "from train_utils import train_one_direction, evaluate_one_direction, process_image_features\n"
            "import inspect\n"
            "assert list(inspect.signature(training).parameters) == ['gpu', 'args', 'train_subset', 'test_subset']\n"
            "assert list(inspect.signature(eval_pc).parameters) == ['gpu', 'args', 'test_subset', 'curr_dataset', 'prepare_cs_step']\n"
            "assert list(inspect.signature(testing).parameters) == ['args', 'detr', 'relation_classifier', 'test_loader', 'test_record', 'epoch', 'rank', 'writer']\n"
            "print('ok')\n") % (REPO, os.path.join(REPO, "scene_graph_commonsense_amd", "dropin"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_graph_iter_offsets_describe_the_reference_enumeration():
    """``pairs.graph_iter_offsets`` is all the host tells ``sgc_scene_tables`` about the pair order: block g starts at goff[g] and
    holds 2g rows of k_g pairs (k_g = #{images with more than g objects}); it must agree with the explicit enumeration that
    ``test_pair_enumeration_matches_reference_order`` pins to the reference's loops - for random ragged shapes."""
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = rng.integers(0, 9, size=int(rng.integers(1, 7)))
        goff, max_n = PR.graph_iter_offsets(np.asarray(n, dtype=np.int64))
        pidx = PR.enumerate_pairs(n.tolist())
        assert max_n == int(n.max())
        assert int(goff[max(max_n, 1)]) == pidx.n_pairs == int(sum(k * (k - 1) for k in n))
        if pidx.n_pairs:
            # every pair's block, row and column follow from goff alone
            g = np.searchsorted(goff, np.arange(pidx.n_pairs), side="right") - 1
            assert np.array_equal(g, pidx.g)
            k = np.array([(n > gg).sum() for gg in g])
            local = np.arange(pidx.n_pairs) - goff[g]
            row = local // k
            assert np.array_equal(row // 2, pidx.e) and np.array_equal(row % 2 == 0, pidx.first)
            assert np.array_equal(g * (g - 1) + row, pidx.step)


def test_object_window_rects_match_a_brute_force_influence_propagation():
    """The closed-form rectangle of conv3 pooling windows an object can influence (pairs.object_window_rects, replicated in
    csrc/kernels_shared.hip) against the literal chain: box mask -> 3x3 dilation (conv2_1) -> 2x2 any (max-pool) -> 3x3 dilation
    (conv3_1) -> 2x2 any (max-pool)."""
    from scene_graph_commonsense_amd import pairs as PR
    rng = np.random.default_rng(3)

    def dil(m):
        p = np.pad(m, 1)
        o = np.zeros_like(m)
        for dy in range(3):
            for dx in range(3):
                o |= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
        return o

    boxes = [[0, 32, 0, 32], [5, 6, 7, 8], [0, 1, 0, 1], [31, 32, 31, 32], [10, 10, 4, 9], [3, 4, 0, 32], [0, 0, 0, 0], [2, 3, 29, 31]]
    for _ in range(300):
        x0, y0 = rng.integers(0, 32, 2)
        boxes.append([x0, rng.integers(x0, 33), y0, rng.integers(y0, 33)])
    bb = np.asarray(boxes)
    rects = PR.object_window_rects(bb)
    for (x0, x1, y0, y1), r in zip(bb, rects):
        m = np.zeros((32, 32), dtype=bool)
        m[y0:y1, x0:x1] = True
        d16 = dil(m).reshape(16, 2, 16, 2).any(axis=(1, 3)) if m.any() else np.zeros((16, 16), dtype=bool)
        d8 = dil(d16).reshape(8, 2, 8, 2).any(axis=(1, 3)) if m.any() else np.zeros((8, 8), dtype=bool)
        want = np.zeros((8, 8), dtype=bool)
        want[r[2]:r[3], r[0]:r[1]] = True
        assert (want == d8).all(), (x0, x1, y0, y1, r)
    # pair count: sum over ordered pairs of the rectangle intersections
    img_ptr = [0, 100, 180, len(bb)]
    total = 0
    for b in range(3):
        q = rects[img_ptr[b]:img_ptr[b + 1]]
        for i in range(len(q)):
            for j in range(len(q)):
                if i != j:
                    total += max(0, min(q[i, 1], q[j, 1]) - max(q[i, 0], q[j, 0])) * max(0, min(q[i, 3], q[j, 3]) - max(q[i, 2], q[j, 2]))
    assert total == PR.count_shared_windows(bb, img_ptr)


def test_image_group_planner_covers_every_image_within_the_budget():
    """``pair_loop.plan_image_groups`` (host logic of the automatic chunking): consecutive ranges that cover every image once,
    each within the budget unless it is a single image, one group when everything fits, estimates that grow with the overlap
    of the boxes and fall back to the per-pair cost when most windows are pair-specific."""
    from scene_graph_commonsense_amd.pair_loop import _COST, image_workspace_bytes, plan_image_groups, slice_batch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    cfg = HeadConfig()
    nobj = [64, 3, 40, 40, 1, 64, 20, 20, 20, 64]
    batch = make_scene_batch(cfg, nobj, seed=9)
    for train in (True, False):
        cost = [1.15 * image_workspace_bytes(cfg, b, train) for b in batch.bbox]
        assert plan_image_groups(cfg, batch, train, 1e15) == [(0, len(nobj))]
        for budget in (sum(cost) / 2.5, max(cost) * 1.01, max(cost) / 3):
            groups = plan_image_groups(cfg, batch, train, budget)
            assert groups[0][0] == 0 and groups[-1][1] == len(nobj) and all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
            assert len(groups) >= 2
            for a, b in groups:
                assert b > a and (b - a == 1 or sum(cost[a:b]) <= budget)
                assert a == 0 or sum(cost[groups[[g[0] for g in groups].index(a) - 1][0]:a]) + cost[a] > budget   # greedy: the previous group was full
    # boxes: disjoint small boxes are cheaper than full-image boxes, which are priced on the per-pair kernels
    small = torch.tensor([[k % 8 * 4, k % 8 * 4 + 2, k // 8 * 4, k // 8 * 4 + 2] for k in range(40)], dtype=torch.int32)
    full = torch.tensor([[0, 32, 0, 32]] * 40, dtype=torch.int32)
    c_small, c_full = image_workspace_bytes(cfg, small, True), image_workspace_bytes(cfg, full, True)
    assert c_full == _COST[True][2] * 40 * 39 + _COST[True][3] * 40 and c_small < 0.6 * c_full
    sub = slice_batch(batch, 2, 5)
    assert sub.num_objects == [40, 40, 1] and sub.image_feature.shape[0] == 3 and len(sub.relationships) == 3


def _torch_sgd_update(p, g, m, lr, momentum, weight_decay, first):
    """The update rule of torch.optim.SGD (dampening 0) on a flat slice - what sgc_sgd_momentum_step computes on the GPU; injected
    into ShardedSGD on the CPU so that its partitioning / collective / accumulation logic can be tested without a GPU."""
    d = g + weight_decay * p
    if first:
        m.copy_(d)
    else:
        m.mul_(momentum).add_(d)
    p.sub_(lr * m)


def _sharded_worker(rank, world, port, q, defer=False, fc1_shape=(64, 128), want_pieces=4):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scene_graph_commonsense_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(0)                       # same initial parameters on every rank
    shapes = [("conv1_1.weight", (3, 5, 1, 1)), ("fc1.weight", tuple(fc1_shape)), ("fc1.bias", (7,)), ("fc2.weight", (5, 13))]
    params = [(n, torch.nn.Parameter(torch.randn(s, generator=g))) for n, s in shapes]
    ref = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in params]
    ref_opt = torch.optim.SGD([p for _, p in ref], lr=0.05, momentum=0.9, weight_decay=1e-3)
    opt = D.ShardedSGD(params, w, r, lr=0.05, momentum=0.9, weight_decay=1e-3, buckets=4, update_fn=_torch_sgd_update, defer_gather=defer)
    n_fc1 = fc1_shape[0] * fc1_shape[1]
    if want_pieces:          # row blocks of the big parameter: ``buckets`` halved until a block splits into world x 4-element pieces
        assert len(opt.pieces["fc1.weight"]) == want_pieces and opt.pieces["fc1.weight"][-1].length == n_fc1 // want_pieces // w
        assert sum(pc.length for pc in opt.pieces["fc1.weight"]) * w == n_fc1
        assert opt.small_pad % (4 * w) == 0 and opt.small_pad >= 15 + 7 + 65
    else:                    # a size world x 4 does not divide: the parameter rides in the padded flat bucket of the small ones
        assert opt.big == [] and "fc1.weight" not in opt.pieces and opt.small_pad >= 15 + 7 + 65 + n_fc1
        assert opt.small_pad % (4 * w) == 0 and opt.small_pad - (15 + 7 + 65 + n_fc1) < 4 * w
    worst = 0.0
    for step in range(4):
        # every rank's own gradients (what its images give); the reference applies their MEAN with a plain torch SGD
        gr = [torch.Generator().manual_seed(100 * step + k) for k in range(w)]
        all_g = [{n: torch.randn(p.shape, generator=gr[k]) for n, p in params} for k in range(w)]
        accumulate = step == 2                                # step 2 accumulates two backward passes before stepping
        opt.zero_grad()
        mine = {n: t.clone() for n, t in all_g[r].items()}
        if step % 2 == 0:
            opt.hook("fc1.weight", mine["fc1.weight"])        # early hand-over, as the backward does
        mine["fc2.weight"] = mine["fc2.weight"].t().contiguous().t()      # a non-contiguous gradient (conv weights come out of a permute)
        opt.finish_grads(mine)
        assert not opt.pending and all(p.grad is None for _, p in params)
        if accumulate:
            opt.finish_grads({n: 0.5 * t for n, t in all_g[r].items()})
        for n, p in ref:
            p.grad = sum(all_g[k][n] for k in range(w)) / w * (1.5 if accumulate else 1.0)
        if step == 3:
            opt.param_groups[0]["lr"] = ref_opt.param_groups[0]["lr"] = 0.01
        v0 = [p._version for _, p in params]
        opt.step(); ref_opt.step()
        assert all(p._version > v for (_, p), v in zip(params, v0))
        if defer:
            # the small parameters are final when step() returns; fc1.weight's gathers may still be in flight until a reader waits
            for (n, p), (_, q_) in zip(params, ref):
                if n != "fc1.weight":
                    worst = max(worst, float((p.detach() - q_.detach()).abs().max()))
            assert all(big for _, _, big in opt.gathers) and (len(opt.gathers) > 0) == bool(want_pieces)
            opt.wait_gathers()
        assert not opt.gathers
        for (n, p), (_, q_) in zip(params, ref):
            worst = max(worst, float((p.detach() - q_.detach()).abs().max()))
    # the momentum state is sharded: this rank holds 1/world of every bucket
    mom = sum(pc.mom.numel() for pcs in opt.pieces.values() for pc in pcs)
    total = sum(p.numel() for _, p in params)
    q.put((rank, worst, mom, total, [p.detach().numpy().copy() for _, p in params]))     # numpy: no shared-memory handles
    dist.destroy_process_group()


@pytest.mark.parametrize("world,defer,fc1_shape,want_pieces", [
    (2, False, (64, 128), 4), (2, True, (64, 128), 4),
    (8, True, (64, 128), 4),            # the node size the path is built for: 4 row blocks x 8 shards of 256
    (8, False, (48, 4), 2),             # 192 elements: 4 blocks x 8 ranks x 4 does not divide -> halved to 2 blocks of 96 (12 per rank)
    (3, True, (48, 8), 4),              # a world that is not a power of two: 384 = 4 blocks x 3 ranks x 32
    (3, False, (64, 128), 0),           # 8192 is not a multiple of 3 x 4: fc1.weight goes through the padded flat bucket
])
def test_sharded_sgd_reduce_scatter_update_all_gather_gloo(world, defer, fc1_shape, want_pieces):
    """distributed.ShardedSGD under gloo at 2, 3 and 8 ranks: four steps (early hook or not, gradient accumulation, lr change) leave
    EVERY rank with the parameters a single torch.optim.SGD gets from the mean gradients; optimizer state is 1/world per rank (+ the
    padding of the flat bucket).  ``defer``: the all-gather of the big parameter is waited for by its next reader, not by ``step``.
    Covers the piece / padding arithmetic of ``distributed.py`` (``buckets`` halving, non-dividing sizes) at the world sizes the
    world-2 tests never reach (VERDICT r4 missing 1)."""
    res = _spawn(_sharded_worker, world, (defer, fc1_shape, want_pieces))
    for rank, worst, mom, total, _ in res:
        assert worst <= 1e-6, worst
        assert mom <= total // world + 4 * world + 8
    for other in res[1:]:
        for a, b in zip(res[0][4], other[4]):
            assert np.array_equal(a, b)                                    # replicas stay bit-identical


def test_sharded_sgd_world1_is_plain_sgd():
    from scene_graph_commonsense_amd import distributed as D
    g = torch.Generator().manual_seed(3)
    params = [("fc1.weight", torch.nn.Parameter(torch.randn(16, 32, generator=g))), ("fc4.bias", torch.nn.Parameter(torch.randn(1, generator=g)))]
    ref = [torch.nn.Parameter(p.detach().clone()) for _, p in params]
    ro = torch.optim.SGD(ref, lr=0.1, momentum=0.9, weight_decay=1e-4)
    opt = D.ShardedSGD(params, 1, 0, lr=0.1, momentum=0.9, weight_decay=1e-4, update_fn=_torch_sgd_update)
    for step in range(3):
        gs = {n: torch.randn(p.shape, generator=g) for n, p in params}
        opt.zero_grad()
        opt.hook("fc1.weight", gs["fc1.weight"])
        opt.finish_grads(gs)
        for q_, (n, _) in zip(ref, params):
            q_.grad = gs[n].clone()
        opt.step(); ro.step()
    for (_, p), q_ in zip(params, ref):
        assert float((p.detach() - q_.detach()).abs().max()) <= 1e-6


def test_tuning_environment_variables(monkeypatch):
    """``engine.Tuning.from_env``: the three plain variables and the ``SGC_TUNING`` field list; unknown fields are an error."""
    from scene_graph_commonsense_amd.engine import Tuning
    for k in ("SGC_SHARED_LEVEL", "SGC_SHARED_MAX_FRACTION", "SGC_BWD_STREAMS", "SGC_TUNING"):
        monkeypatch.delenv(k, raising=False)
    t = Tuning.from_env()
    assert t == Tuning() and t.patch_dgrad and t.patch_wgrad and t.shared_linear and t.gemms_apart
    monkeypatch.setenv("SGC_SHARED_LEVEL", "2")
    monkeypatch.setenv("SGC_BWD_STREAMS", "0")
    monkeypatch.setenv("SGC_TUNING", "gemms_apart=0, patch_dgrad=0,shared_max_fraction=0.25")
    t = Tuning.from_env()
    assert (t.shared_conv3, t.shared_fc1, t.shared_objects, t.shared_linear) == (True, True, False, False)
    assert not t.bwd_streams and not t.gemms_apart and not t.patch_dgrad and t.patch_wgrad and t.shared_max_fraction == 0.25
    monkeypatch.setenv("SGC_TUNING", "no_such_field=1")
    with pytest.raises(ValueError):
        Tuning.from_env()


def test_deferred_weight_entries_are_made_at_first_use():
    """``engine.Weights``: a deferred entry is built by the first ``w[key]`` (after the registered sync), once."""
    from scene_graph_commonsense_amd.engine import Weights
    calls = []
    w = Weights()
    w["a"] = 1
    w.defer("b", lambda: calls.append("made") or 2)
    assert w["a"] == 1 and calls == []
    assert w["b"] == 2 and w["b"] == 2 and calls == ["made"]
    w.defer("b", lambda: calls.append("again") or 3)          # the next weight version replaces the entry
    assert w["b"] == 3 and calls == ["made", "again"]


def test_minibatch_lookahead_order_skip_stop():
    """pair_loop.MinibatchLookahead (the evaluation drivers prepare minibatch k+1 while the device scores minibatch k): every prepared
    item is used exactly once and in order, skipped items never surface, STOP ends the iteration without touching the rest of the
    loader, and a consumer that never looks ahead still sees everything."""
    from scene_graph_commonsense_amd.pair_loop import MinibatchLookahead as L
    log = []

    def prep(i, d):
        log.append(("prep", i))
        return None if d == "skip" else (L.STOP if d == "stop" else d.upper())
    a = L(["a", "skip", "b", "c", "stop", "d"], prep)
    assert log == [("prep", 0)]                                  # one ahead from the start
    for i, x in a:
        log.append(("use", i, x))
        if x != "B":                                             # the consumer of B forgets to look ahead
            a.fetch_next()
            a.fetch_next()                                       # idempotent until the item is taken
    assert log == [("prep", 0), ("use", 0, "A"), ("prep", 1), ("prep", 2), ("use", 2, "B"), ("prep", 3), ("use", 3, "C"), ("prep", 4)]
    assert list(L([], prep)) == [] and list(L(["skip"], prep)) == []
    assert [x for _, x in L(["x", "y"], prep)] == ["X", "Y"]

"""Two data-parallel ranks sharing the one GPU of the test box (gloo transport, CUDA tensors): the gradients that
``train_minibatch`` + ``GradReducer`` leave in ``param.grad`` must equal the mean of the ranks' local gradients,
including the early (asynchronous) fc1.weight reduction.  RCCL itself needs one GPU per rank and is exercised by the
driver's multi-GPU bench; this test pins the integration logic."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _detection_shaped_batch(cfg, rank):
    """BASELINE.json configs[3] per rank: 8 images, ragged object lists of detector size (12-30 objects), FLOAT boxes as the SGDET
    front-end produces them (``evaluate.py:332``: x0, x1, y0, y1 on the grid, truncated by ``int()`` when the masks are built)."""
    from scene_graph_commonsense_amd.synthetic import hash_uniform, make_scene_batch
    nobj = [18 + rank, 25, 12, 30 - rank, 22, 16, 27, 20]
    batch = make_scene_batch(cfg, nobj, seed=300 + rank, connect_frac=0.05)
    for k, b in enumerate(batch.bbox):
        jit = torch.from_numpy(hash_uniform(977 * rank + k, b.numel(), 0.0, 0.95).reshape(b.shape))
        batch.bbox[k] = torch.minimum(b.float() + jit, torch.tensor(32.0))
    return batch


def _worker(rank, world, port, q, shape="small"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from scene_graph_commonsense_amd import distributed as D
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = HeadConfig()
    model = BayesianRelationClassifier(cfg.args()).cuda()
    model.load_state_dict(make_state_dict(cfg, seed=5, head_gain=4.0))
    model.eval()
    batch = make_scene_batch(cfg, (3, 2), seed=100 + rank, connect_frac=0.6) if shape == "small" else _detection_shaped_batch(cfg, rank)
    train_minibatch(model, batch)                                                 # local gradients (every rank its own images)
    local = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    train_minibatch(model, batch, reducer=D.GradReducer(world))                   # reduced gradients
    worst = 0.0
    for n, p in model.named_parameters():
        gathered = [torch.empty_like(local[n]) for _ in range(world)]
        dist.all_gather(gathered, local[n].contiguous())
        mean = torch.stack(gathered).mean(0)
        err = float((p.grad - mean).abs().max() / mean.abs().max().clamp(min=1e-30))
        worst = max(worst, err)
    # gradient accumulation (param.grad already set): a second reduced step must ADD the mean gradient - the early
    # asynchronous fc1.weight reduction may not be read before it has finished and been divided
    train_minibatch(model, batch, reducer=D.GradReducer(world))
    for n, p in model.named_parameters():
        gathered = [torch.empty_like(local[n]) for _ in range(world)]
        dist.all_gather(gathered, local[n].contiguous())
        mean2 = 2 * torch.stack(gathered).mean(0)
        worst = max(worst, float((p.grad - mean2).abs().max() / mean2.abs().max().clamp(min=1e-30)))
    q.put((rank, worst))
    dist.destroy_process_group()


@pytest.mark.parametrize("shape", ["small", "configs3_sgdet_8_images_per_rank"])
def test_two_ranks_mean_gradients_on_one_gpu(shape):
    """``configs3...``: BASELINE.json configs[3]'s shape per rank (8 images of detector-sized ragged object lists with float boxes,
    gradients reduced across the ranks) - two of its eight ranks, sharing this box's one GPU over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                       # a port the OS knows to be free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, shape)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=600) for _ in range(2))
    [p.join(120) for p in procs]
    for rank, worst in res:
        assert worst <= 1e-5, (rank, worst)


def _sharded_worker(rank, world, port, q):
    """Two steps (the second accumulates two backward passes) with ``ShardedSGD`` as reducer + optimizer on the real HIP path, against
    the all-reduce path (``GradReducer`` + ``FusedSGD``) started from the same weights on the same images: same parameters."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from scene_graph_commonsense_amd import distributed as D
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=5, head_gain=4.0)
    batch = make_scene_batch(cfg, (3, 2), seed=100 + rank, connect_frac=0.6)
    models = []
    for mode in ("allreduce", "sharded"):
        model = BayesianRelationClassifier(cfg.args()).cuda()
        model.load_state_dict(sd)
        model.eval()                                                               # dropout off: both modes see the same step
        if mode == "sharded":
            opt = red = D.ShardedSGD(model.named_parameters(), world, rank, lr=1e-5, momentum=0.9, weight_decay=1e-4,
                                     defer_gather=True).attach(model)      # fc1.weight's gather is waited for where fc1's copies are made
        else:
            opt, red = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4), D.GradReducer(world)
        train_minibatch(model, batch, opt, reducer=red)                            # step 1
        opt.zero_grad(set_to_none=True)                                            # step 2: two backward passes, one update
        model.training_step(model.last_scene, batch.relationships, batch.subj_or_obj, reducer=red)
        model.training_step(model.last_scene, batch.relationships, batch.subj_or_obj, reducer=red)
        opt.step()
        torch.cuda.synchronize()
        models.append(model)
        if mode == "sharded":
            mom = sum(pc.mom.numel() for pcs in opt.pieces.values() for pc in pcs if pc.mom is not None)
    worst = 0.0
    moved = 0.0
    for (n, a), (_, b) in zip(models[0].named_parameters(), models[1].named_parameters()):
        worst = max(worst, float((a.detach() - b.detach()).abs().max() / a.detach().abs().max().clamp(min=1e-30)))
        moved = max(moved, float((a.detach() - sd[n].cuda()).abs().max()))
    # replicas agree bit for bit across ranks
    same = True
    for n, p in models[1].named_parameters():
        if p.numel() > (1 << 22):
            continue
        other = [torch.empty_like(p.detach()) for _ in range(world)]
        dist.all_gather(other, p.detach().contiguous())
        same = same and torch.equal(other[0], other[1])
    total = sum(p.numel() for p in models[1].parameters())
    q.put((rank, worst, moved, same, mom, total))
    dist.destroy_process_group()


def test_two_ranks_sharded_sgd_equals_allreduce_path_on_one_gpu():
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=900) for _ in range(2))
    [p.join(120) for p in procs]
    for rank, worst, moved, same, mom, total in res:
        assert moved > 0 and worst <= 1e-6, (rank, worst, moved)                   # same update (f32 fma / summation order only)
        assert same and mom <= total // 2 + 64


def _rccl_worker(port, q):
    """ONE rank on a real RCCL communicator (backend "nccl" on ROCm) with the collectives forced on: every call the multi-GPU step
    makes - the asynchronous all-reduce / the eight reduce-scatters of fc1.weight's gradient launched from the backward's SIDE
    stream, the flat bucket, the all-gathers into views of the parameters' storage - goes through RCCL's own stream and its event
    ordering.  With one rank a sum over ranks is the identity, so the results must equal the local path's bit for bit."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from scene_graph_commonsense_amd import distributed as D
    from scene_graph_commonsense_amd.model import BayesianRelationClassifier
    from scene_graph_commonsense_amd.optim import FusedSGD
    from scene_graph_commonsense_amd.pair_loop import train_minibatch
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=5, head_gain=4.0)
    batch = make_scene_batch(cfg, (6, 5, 4), seed=100, connect_frac=0.6)
    out = {}
    for mode in ("local", "allreduce", "sharded"):
        model = BayesianRelationClassifier(cfg.args()).cuda()
        model.load_state_dict(sd)
        model.eval()
        if mode == "sharded":
            opt = red = D.ShardedSGD(model.named_parameters(), 1, 0, lr=1e-5, momentum=0.9, weight_decay=1e-4, force_collectives=True,
                                     defer_gather=True).attach(model)      # the gather of step k is in flight when step k+1 starts
        else:
            opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4)
            red = D.GradReducer(1, force_collectives=True) if mode == "allreduce" else None
        for _ in range(3):                                                         # momentum, reused workspace and RCCL's stream across steps
            train_minibatch(model, batch, opt, reducer=red)
        torch.cuda.synchronize()
        out[mode] = {n: p.detach().clone() for n, p in model.named_parameters()}
        if red is not None:
            out[mode + "_exposed_ms"] = red.pop_exposed_ms()
    res = {}
    for mode in ("allreduce", "sharded"):
        res[mode] = all(torch.equal(out["local"][n], out[mode][n]) for n in out["local"])
    moved = max(float((out["local"][n] - sd[n].cuda()).abs().max()) for n in out["local"])
    q.put((res, moved, dist.get_backend(), out["allreduce_exposed_ms"], out["sharded_exposed_ms"]))
    dist.destroy_process_group()


def test_one_rank_over_rccl_runs_every_collective_of_the_step():
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    res, moved, backend, ex_a, ex_s = q.get(timeout=900)
    p.join(120)
    assert backend == "nccl" and moved > 0
    assert res == {"allreduce": True, "sharded": True}, res
    assert ex_a >= 0 and ex_s >= 0

"""Shared driver of the training-path parity tests: one fused forward+backward on the device that also hands back the
device's own routing decisions, and the matching oracle run with injected dropout masks / routes."""
import numpy as np
import torch


def run_train_gpu(cfg, sd, batch, dropout=False, seeds=(0, 0), keep_ctx=False, coefs=None, route_rows=None):
    """(loss, grads {name: f32 cpu}, routes, scene) of ``RelHeadEngine.train_forward`` + ``train_backward``.
    ``coefs``: per-pair loss coefficients (tgt, a, b, c, y) instead of the whole minibatch's (sampled steps of a full-size
    minibatch: zero outside them); ``route_rows``: the pairs whose routes are returned (default: all)."""
    from scene_graph_commonsense_amd.engine import RelHeadEngine, csr_by, loss_coefficients
    from scene_graph_commonsense_amd.model import _shared_hint          # the host's window counts: the organisation the product path takes
    from scene_graph_commonsense_amd.pairs import flatten_scene, pair_targets
    from scene_graph_commonsense_amd.synthetic import predicate_counts
    dev = "cuda:0"
    eng = RelHeadEngine(cfg, dev)
    eng.load_weights(sd)
    eng.prep_bwd_weights(sd)
    sc = flatten_scene(cfg, batch, dev)
    pidx = sc.pidx
    if coefs is None:
        directed, _ = pair_targets(batch.relationships, batch.subj_or_obj, pidx)
        counts = predicate_counts(cfg).numpy()
        cw = 1 - counts / counts.sum()
        coefs = loss_coefficients(cfg, pidx.step, len(pidx.call_sizes), directed, cw)
    coefs_d = tuple(torch.from_numpy(c).to(dev) for c in coefs)
    n_obj = int(sc.obj_img.shape[0])
    sub_csr = tuple(torch.from_numpy(a).to(dev) for a in csr_by(pidx.sub, n_obj))
    obj_csr = tuple(torch.from_numpy(a).to(dev) for a in csr_by(pidx.obj, n_obj))
    img_ptr = torch.from_numpy(pidx.obj_offset.astype(np.int32)).to(dev)
    ctx = eng.train_forward(sc.image_feature, sc.image_depth, sc.obj_img, sc.bbox, sc.cats, sc.super_mh, sc.sub_idx, sc.obj_idx,
                            dropout=dropout, seeds=seeds, dense=(sc.img_ptr, sc.pid, sc.max_n), shared_windows=_shared_hint(sc))
    routes = device_routes(ctx, route_rows) if keep_ctx else None
    loss, grads = eng.train_backward(ctx, coefs_d, sub_csr, obj_csr, img_ptr)
    torch.cuda.synchronize()
    return float(loss), {k: v.float().cpu() for k, v in grads.items()}, routes, sc


def device_routes(ctx, rows=None):
    """The routing decisions the device's backward follows, in the oracle's layouts (CPU tensors, one row per ordered pair):
    pool2 [P,512,16,16] / pool3 [P,1024,8,8] window codes (dy*2+dx, 4 = killed by the ReLU), relu1 [P,4096] / relu2 [P,512]
    pass masks (an element the dropout removed reads as "not passed": the injected dropout mask zeroes it anyway).
    With conv3 / fc1 over shared windows (``csrc/kernels_shared.hip``) a pair's own codes exist only next to its pair-specific
    windows; everywhere else its gradient flows through the pseudo-pair (subject, background) or (background, object), whose
    codes are what the reference's graph has there too (identical inputs), so the full per-pair tables are put together from them.
    ``rows`` (int array of pair indices): tables of those pairs only, in that order (full-size minibatches: the sampled steps)."""
    P = ctx.P
    sel = np.arange(P) if rows is None else np.asarray(rows, dtype=np.int64)
    sel_d = torch.from_numpy(sel).to(ctx.h1.device)
    n = len(sel)
    sh = getattr(ctx, "shared", None)
    unpack = lambda a: torch.stack((a & 15, a >> 4), dim=3).reshape(a.shape[0], 256, 512)      # channel 2k low nibble, 2k+1 high
    if sh is None:
        codes = unpack(ctx.amz[:P * 256 * 256].view(P, 256, 256)[sel_d].cpu())
        am3 = ctx.am[:P * 65536].view(P, 64, 1024)[sel_d].cpu()
    else:
        from scene_graph_commonsense_amd.pairs import object_window_rects
        n_obj, n2 = ctx.n_obj, sh["n2"]
        real = unpack(ctx.amz[:P * 256 * 256].view(P, 256, 256)[sel_d].cpu())
        pseudo = unpack(ctx.amz[P * 256 * 256:(P + n2) * 256 * 256].view(n2, 256, 256).cpu())
        bb = ctx.bbox.cpu().numpy()
        sub, obj = ctx.sub_idx.cpu().numpy(), ctx.obj_idx.cpu().numpy()
        pr = sh["pixrect"].cpu().numpy()
        R = object_window_rects(bb)
        # D16_o: the 16-grid pixels where object o's conv2 half differs from the background's (box, +-1 on the 32-grid, /2)
        d16 = np.zeros((len(bb), 16, 16), dtype=bool)
        for o, (x0, x1, y0, y1) in enumerate(bb):
            x0, x1, y0, y1 = max(x0, 0), min(x1, 32), max(y0, 0), min(y1, 32)
            if x1 > x0 and y1 > y0:
                lx, hx, ly, hy = max(x0 - 1, 0), min(x1 + 1, 32), max(y0 - 1, 0), min(y1 + 1, 32)
                d16[o, ly >> 1:(hy + 1) >> 1, lx >> 1:(hx + 1) >> 1] = True
        am_real = ctx.am[:P * 65536].view(P, 64, 1024)[sel_d].cpu()
        am_ps = sh["am_ps"][:n2 * 65536].view(n2, 64, 1024).cpu()
        codes = torch.empty(n, 256, 512, dtype=real.dtype)
        am3 = torch.empty(n, 64, 1024, dtype=am_real.dtype)
        wy, wx = np.divmod(np.arange(64), 8)
        for k, p in enumerate(sel):
            i, j = int(sub[p]), int(obj[p])
            r = int(pr[p])
            own = np.zeros((16, 16), dtype=bool)
            own[r & 31:(r >> 5) & 31, (r >> 10) & 31:(r >> 15) & 31] = True
            assert not (d16[i] & d16[j] & ~own).any()                 # pair-specific pixels lie inside what the pair computed itself
            from_j = torch.from_numpy((~own & d16[j]).reshape(256))
            from_i = torch.from_numpy((~own & ~d16[j]).reshape(256))
            codes[k] = real[k]
            codes[k][from_i] = pseudo[i][from_i]
            codes[k][from_j] = pseudo[n_obj + j][from_j]
            in_i = (wx >= R[i, 0]) & (wx < R[i, 1]) & (wy >= R[i, 2]) & (wy < R[i, 3])
            in_j = (wx >= R[j, 0]) & (wx < R[j, 1]) & (wy >= R[j, 2]) & (wy < R[j, 3])
            am3[k] = am_real[k]
            am3[k][torch.from_numpy(~in_j)] = am_ps[i][torch.from_numpy(~in_j)]
            s_ = torch.from_numpy(in_j & ~in_i)
            am3[k][s_] = am_ps[n_obj + j][s_]
    pool2 = codes.permute(0, 2, 1).reshape(n, 512, 16, 16).contiguous()
    pool3 = am3.permute(0, 2, 1).reshape(n, 1024, 8, 8).contiguous()
    relu1 = (ctx.h1[:P * 4096].view(P, 4096)[sel_d] != 0).float().cpu()
    relu2 = (ctx.p[:P * 512].view(P, 512)[sel_d] != 0).float().cpu()
    return dict(pool2=pool2, pool3=pool3, relu1=relu1, relu2=relu2)


def oracle_train(cfg, sd, batch, scene, dropout_seeds=None, routes=None, capture=None):
    """Oracle loss + autograd gradients; ``dropout_seeds`` injects the kernels' keep masks (x2), ``routes`` the device routing.
    ``capture`` (a dict): filled with the oracle's PRE-activations in front of its four routing decisions, one row per ordered pair in
    pair order - pool2 [P,512,32,32], pool3 [P,1024,16,16], relu1 [P,4096], relu2 [P,512] (``route_flip_margins``)."""
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.synthetic import dropout_keep_mask, predicate_counts
    start = np.concatenate([[0], np.cumsum(scene.pidx.call_sizes)])
    caps = []

    def hook(t, b):
        r0 = int(start[t])
        assert int(start[t + 1]) - r0 == b
        inj = {}
        if capture is not None:
            caps.append({})
            inj["capture"] = caps[-1]
        if dropout_seeds is not None:
            inj["drop1"] = torch.from_numpy(dropout_keep_mask(dropout_seeds[0], b, 4096, r0)).float() * 2
            inj["drop2"] = torch.from_numpy(dropout_keep_mask(dropout_seeds[1], b, 512, r0)).float() * 2
        if routes is not None:
            inj["routes"] = {k: v[r0:r0 + b] for k, v in routes.items()}
        return inj

    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=O.class_weights(predicate_counts(cfg)), call_hook=hook)
    out["losses"].backward()
    if capture is not None:
        for kind in ("pool2", "pool3", "relu1", "relu2"):
            capture[kind] = torch.cat([c[kind] for c in caps])
    return float(out["losses"]), {k: p.grad for k, p in sdr.items()}, out


def fro(a, b):
    return float((a.double() - b.double()).norm() / max(b.double().norm(), 1e-30))


FORWARD_TOL = 1e-3          # BASELINE.json north_star: forward values within 1e-3 relative


def route_flip_margins(pre, routes, keep1=None, keep2=None):
    """How far from its decision boundary is every routing decision the device takes differently from the reference?

    ``pre``: the oracle's pre-activations (``oracle_train(capture=...)``), ``routes``: the device's decisions (``device_routes``).
    A ReLU unit the device passes and the reference kills (or the other way round) is a legitimate outcome of a forward that is
    within the tolerance only if the reference's pre-activation lies within that tolerance of zero; a max-pool window routed to another
    position only if the reference's value there is within (twice) the tolerance of the window's maximum.  Returns per decision kind
    (flipped, total, worst margin / max |pre-activation| of the layer): the worst margin is what a test holds against ``FORWARD_TOL``
    (ReLU) or 2 x ``FORWARD_TOL`` (arg-max: a difference of two values).  ``keep1`` / ``keep2``: dropout keep masks (bool [P,4096] /
    [P,512]) - a dropped unit reads as "not passed" on the device whatever its pre-activation was, so it is not a decision."""
    out = {}
    for kind in ("pool2", "pool3"):
        c = pre[kind].double()
        Pn, C, H, W = c.shape
        win = c.reshape(Pn, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(Pn, C, H // 2, W // 2, 4)
        m, am = win.max(dim=4)
        own = torch.where(m > 0, am, torch.full_like(am, 4))
        dev = routes[kind].long()
        at_dev = torch.gather(win, 4, dev.clamp(max=3).unsqueeze(-1)).squeeze(-1)
        # device killed the window: right if the maximum is <= 0; device routed to q: right if win[q] is the maximum and > 0
        margin = torch.where(dev == 4, m.clamp(min=0), torch.maximum(m - at_dev, (-at_dev).clamp(min=0)))
        flipped = (own != dev) & (margin > 0)             # equal maxima at two positions: either choice is the reference's
        out[kind] = (int(flipped.sum()), flipped.numel(), float(margin[flipped].max() / c.abs().max()) if bool(flipped.any()) else 0.0)
    for kind, keep in (("relu1", keep1), ("relu2", keep2)):
        x = pre[kind].double()
        dev = routes[kind].bool()
        flipped = (x > 0) != dev
        if keep is not None:
            flipped &= torch.as_tensor(keep, dtype=torch.bool)
        out[kind] = (int(flipped.sum()), flipped.numel(), float(x[flipped].abs().max() / x.abs().max()) if bool(flipped.any()) else 0.0)
    return out


def routed_model_step(model, scene, batch, step_kw=None, oracle_kw=None, aug=None):
    """One ``model.training_step`` (the product entry point: contrastive branch, commonsense penalty, ...) with the device's own
    routing captured from the engine contexts, and the oracle's step with those routes injected - so that what is compared is
    arithmetic, not which near-zero pre-activations happened to pass a ReLU / win a max-pool.
    Returns (loss, grads {name: f32 cpu}, ref_loss, ref_grads, oracle output dict)."""
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.engine import RelHeadEngine
    from scene_graph_commonsense_amd.synthetic import predicate_counts
    cfg = model.head_config()
    captured = []
    orig = RelHeadEngine.train_backward

    def spy(self, ctx, *a, **k):
        captured.append((ctx, device_routes(ctx)))           # before the backward reuses the buffers
        return orig(self, ctx, *a, **k)

    RelHeadEngine.train_backward = spy
    try:
        model.zero_grad(set_to_none=True)
        kw = dict(step_kw or {})
        if aug is not None:
            kw["image_feature_aug"] = aug
        loss = model.training_step(scene, batch.relationships, batch.subj_or_obj, **kw)
        torch.cuda.synchronize()
    finally:
        RelHeadEngine.train_backward = orig
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}
    routes = captured[0][1]
    routes_aug = None
    if aug is not None and len(captured) > 1:
        # the augmented trunk ran for the connected pairs only, in pair order: scatter its routes to full-size tables
        P = scene.pidx.n_pairs
        conn = torch.nonzero(scene.directed.cpu() >= 0).flatten()
        ra = captured[1][1]
        assert ra["relu2"].shape[0] == conn.numel()
        routes_aug = {}
        for k, v in ra.items():
            full = torch.zeros((P,) + tuple(v.shape[1:]), dtype=v.dtype)
            full[conn] = v
            routes_aug[k] = full
    start = np.concatenate([[0], np.cumsum(scene.pidx.call_sizes)])

    def hook(t, b):
        r0 = int(start[t])
        inj = {"routes": {k: v[r0:r0 + b] for k, v in routes.items()}}
        if routes_aug is not None:
            inj["routes_aug"] = {k: v[r0:r0 + b] for k, v in routes_aug.items()}
        return inj

    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sdr = {k: v.requires_grad_(True) for k, v in sd.items()}
    out = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=O.class_weights(predicate_counts(cfg)), call_hook=hook,
                          image_feature_aug=None if aug is None else aug.cpu(), **(oracle_kw or {}))
    out["losses"].backward()
    return float(loss), grads, float(out["losses"]), {k: p.grad for k, p in sdr.items()}, out

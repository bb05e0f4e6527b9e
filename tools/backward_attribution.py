#!/usr/bin/env python3
"""Where does the BACKWARD's error come from?  (the forward's table is ``tools/rounding_attribution.py``.)  CPU only.

The reference's training step (``/root/reference/train_test.py:164-258`` loop, ``train_utils.py:116-157`` loss, ``model.py:138-184``
classifier; restated in ``oracle/relhead_oracle.py``) on the golden case ``vg_full`` in FLOAT64, with the roundings of the device path
injected ONE SOURCE AT A TIME.  The device computes the forward with f16 operands (f32 accumulation) and the backward with bf16
gradient tensors, bf16 weight copies in the data-gradient products and bf16 activation copies in the weight-gradient products:

  routes         the exact (float64) forward VALUES, but every ReLU mask and 2x2 max-pool arg-max taken from the f16 forward: the part of the
                 f16 forward's effect on the gradient that no backward, however exact, can undo (a flipped route moves a whole gradient path)
  fwd16          the f16 forward (x, w1, a, w2, U / V, z, w3, y, w_fc1, h1, w_fc2 rounded to f16: the eleven sources of
                 ``profiles/r05_rounding_attribution.txt``; straight-through in the backward), exact backward behind it: routes + values
  dy_bf16        exact forward; the gradient w.r.t. every layer's output rounded to bf16 before it enters that layer's two products
  w_bf16         exact forward; weights rounded to bf16 in the data-gradient products only
  act_bf16       exact forward; activations rounded to bf16 in the weight-gradient products only
  bwd_bf16       the three backward sources together behind the exact forward
  device_model   fwd16 + bwd_bf16: the model of what the kernels do (accumulation exact here: float64)
  dy_f16_scaled  fwd16 + w_bf16 + act_bf16 with the gradient tensors in f16 under ideal loss scaling (11-bit significand, no range limit):
                 what "f16 gradient tensors instead of bf16" would buy
  f32            the reference's own arithmetic (everything float32, no injection): its distance from float64 is the floor

Reported per variant: (1) step 1 - relative Frobenius error and cosine of every parameter tensor's gradient against float64;
(2) the K-step SGD trajectory at the reference's learning rate (``config.yaml:51``: 1e-5, momentum 0.9, weight decay 1e-4, the dropout
masks of ``tests/trajectory_case.py``): the smallest update cosine over the parameter tensors per step against the float64 trajectory -
the quantity ``tests/test_trajectory_gpu.py`` bounds.

    python tools/backward_attribution.py [K=8] [case=vg_full] [lr=1e-5]      ->  profiles/r06_backward_attribution.txt
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import relhead_oracle as O                                                    # noqa: E402
from scene_graph_commonsense_amd.pairs import enumerate_pairs                             # noqa: E402
from scene_graph_commonsense_amd.synthetic import dropout_keep_mask, predicate_counts     # noqa: E402
from tests.golden_cases import load_case                                                  # noqa: E402
from tests.trajectory_case import DROPOUT_SEED, MOMENTUM, WEIGHT_DECAY, param_names       # noqa: E402


class Flags:
    fwd16 = False          # f16 operand roundings in the forward (values and, through them, routes)
    routes_only = False    # exact values, routes of the f16 forward
    dy = None              # None | "bf16" | "f16s": rounding of the gradient tensors
    w_bf16 = False         # bf16 weights in data-gradient products
    act_bf16 = False       # bf16 activations in weight-gradient products


FLAGS = Flags()
FC1_DEFERRED = []          # (dy [b,4096], x [b,65536]) of every fc1 call of the step: ONE weight-gradient product per step instead of 2.1 GB per call


def bf16(t):
    return t.float().bfloat16().to(t.dtype)


def f16(t):
    return t.float().half().to(t.dtype)


def sig11(t):
    """Round to an 11-bit significand at unlimited range: f16 under ideal (per-value) loss scaling."""
    m, e = torch.frexp(t.double())
    return torch.ldexp(torch.round(m * 2048.0) / 2048.0, e).to(t.dtype)


def round_dy(g):
    if FLAGS.dy == "bf16":
        return bf16(g)
    if FLAGS.dy == "f16s":
        return sig11(g)
    return g


class RoundST(torch.autograd.Function):
    """f16 rounding in the forward, identity in the backward (the device keeps f32 master weights and differentiates the rounded graph)."""
    @staticmethod
    def forward(ctx, t):
        return f16(t)

    @staticmethod
    def backward(ctx, g):
        return g


def r16(t):
    return RoundST.apply(t) if FLAGS.fwd16 else t


W16 = {}                   # rounded weights of the current optimisation step: ONE copy per step, not one per classifier call (fc1: 2.1 GB each)


def w16(sd, name, cols=None):
    if not FLAGS.fwd16:
        return sd[name] if cols is None else sd[name][:, cols]
    key = (name, cols.start if cols is not None else None, torch.is_grad_enabled())
    if key not in W16:
        W16[key] = RoundST.apply(sd[name] if cols is None else sd[name][:, cols])
    return W16[key]


class Layer(torch.autograd.Function):
    """conv2d / linear whose backward applies the flagged roundings to its three operands."""
    @staticmethod
    def forward(ctx, x, w, b, pad, defer):
        ctx.save_for_backward(x, w)
        ctx.pad, ctx.has_b, ctx.defer = pad, b is not None, defer
        return F.linear(x, w, b) if x.dim() == 2 else F.conv2d(x, w, b, padding=pad)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        g = round_dy(gy)
        wq = bf16(w) if FLAGS.w_bf16 else w
        xq = bf16(x) if FLAGS.act_bf16 else x
        gb = None
        if x.dim() == 2:
            gx = g @ wq if ctx.needs_input_grad[0] else None
            if ctx.defer:
                FC1_DEFERRED.append((g, xq))
                gw = None
            else:
                gw = g.t() @ xq
            if ctx.has_b:
                gb = gy.sum(0)
        else:
            gx = torch.nn.grad.conv2d_input(x.shape, wq, g, padding=ctx.pad) if ctx.needs_input_grad[0] else None
            gw = torch.nn.grad.conv2d_weight(xq, w.shape, g, padding=ctx.pad)
            if ctx.has_b:
                gb = gy.sum((0, 2, 3))
        return gx, gw, gb, None, None


def layer(x, w, b, pad=0, defer=False):
    return Layer.apply(x, w, b, pad, defer)


def pool_codes(c):
    """Routing of ReLU + 2x2 max-pool: per window the index (dy*2+dx) of its first maximum, 4 where the ReLU kills it."""
    b, C, H, W = c.shape
    win = c.reshape(b, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, C, H // 2, W // 2, 4)
    v, i = win.max(dim=4)
    return torch.where(v > 0, i, torch.full_like(i, 4))


def trunk_and_fc2(sd, hs, ho, lab_fn, drop1, drop2, routes=None, collect=None):
    """(p, routes seen).  ``routes``: impose them (exact values, foreign decisions); ``collect``: dict filled with this pass's decisions."""
    hs, ho = r16(hs), r16(ho)
    a = r16(torch.tanh(layer(hs, w16(sd, "conv1_1.weight"), sd["conv1_1.bias"])))
    b = r16(torch.tanh(layer(ho, w16(sd, "conv1_2.weight"), sd["conv1_2.bias"])))
    w2 = w16(sd, "conv2_1.weight")
    U = r16(layer(a, w2[:, :128], None, 1))
    V = r16(layer(b, w2[:, 128:], sd["conv2_1.bias"], 1))
    c2 = U + V
    if collect is not None:
        collect["pool2"] = pool_codes(c2.detach())
    z = r16(F.max_pool2d(F.relu(c2), 2, 2) if routes is None else O.routed_relu_pool(c2, routes["pool2"]))
    c3 = layer(z, w16(sd, "conv3_1.weight"), sd["conv3_1.bias"], 1)
    if collect is not None:
        collect["pool3"] = pool_codes(c3.detach())
    y = r16(F.max_pool2d(F.relu(c3), 2, 2) if routes is None else O.routed_relu_pool(c3, routes["pool3"]))
    f1 = layer(y.reshape(y.shape[0], -1), w16(sd, "fc1.weight"), sd["fc1.bias"], 0, True)
    if collect is not None:
        collect["relu1"] = (f1.detach() > 0).to(f1.dtype)
    h1 = F.relu(f1) if routes is None else f1 * routes["relu1"]
    if drop1 is not None:
        h1 = h1 * drop1
    h1 = r16(h1)
    w = sd["fc2.weight"]
    hc = lab_fn(h1)
    wq = torch.cat((w16(sd, "fc2.weight", slice(0, 4096)), w[:, 4096:]), dim=1)             # the label columns are gathered in f32 on the device
    p = layer(hc, wq, sd["fc2.bias"])
    if collect is not None:
        collect["relu2"] = (p.detach() > 0).to(p.dtype)
    p = F.relu(p) if routes is None else p * routes["relu2"]
    if drop2 is not None:
        p = p * drop2
    return p


def classifier_forward(sd, h_sub, h_obj, c1, c2, s1, s2, num_classes=150, num_super=17, hierarchical=True, drop1=None, drop2=None,
                       T=(1.0, 1.0, 1.0), routes=None):
    """Drop-in for ``oracle.relhead_oracle.classifier_forward`` with the injections of ``FLAGS``."""
    dt = sd["fc1.weight"].dtype
    h_sub, h_obj = h_sub.to(dt), h_obj.to(dt)
    drop1 = None if drop1 is None else drop1.to(dt)
    drop2 = None if drop2 is None else drop2.to(dt)
    lab = lambda h: O.concat_labels(h, c1, c2, s1, s2, num_classes, num_super).to(dt)
    if FLAGS.routes_only:
        seen = {}
        FLAGS.fwd16 = True
        with torch.no_grad():
            trunk_and_fc2(sd, h_sub, h_obj, lab, drop1, drop2, collect=seen)
        FLAGS.fwd16 = False
        p = trunk_and_fc2(sd, h_sub, h_obj, lab, drop1, drop2, routes=seen)
    else:
        p = trunk_and_fc2(sd, h_sub, h_obj, lab, drop1, drop2)
    conn = F.linear(p, sd["fc4.weight"], sd["fc4.bias"])
    if hierarchical:
        r1, r2, r3, sup = O.bayes_head(sd, p, T)
        return r1, r2, r3, sup, conn, p
    return F.linear(p, sd["fc3.weight"], sd["fc3.bias"]), conn, p


VARIANTS = {
    "exact_f64": dict(),
    "f32": dict(dtype=torch.float32),
    "routes": dict(routes_only=True),
    "fwd16": dict(fwd16=True),
    "dy_bf16": dict(dy="bf16"),
    "w_bf16": dict(w_bf16=True),
    "act_bf16": dict(act_bf16=True),
    "bwd_bf16": dict(dy="bf16", w_bf16=True, act_bf16=True),
    "device_model": dict(fwd16=True, dy="bf16", w_bf16=True, act_bf16=True),
    "dy_f16_scaled": dict(fwd16=True, dy="f16s", w_bf16=True, act_bf16=True),
}


def set_flags(v):
    FLAGS.fwd16, FLAGS.routes_only = bool(v.get("fwd16")), bool(v.get("routes_only"))
    FLAGS.dy, FLAGS.w_bf16, FLAGS.act_bf16 = v.get("dy"), bool(v.get("w_bf16")), bool(v.get("act_bf16"))


def run(case, variant, K, lr, names, baseline=None):
    """K SGD steps; returns (losses, step-1 gradients, per-step updates [K][name]) - gradients / updates kept only for the baseline
    (as f32 for the big tensors); for a variant the comparison against ``baseline`` is made on the fly."""
    v = VARIANTS[variant]
    dt = v.get("dtype", torch.float64)
    set_flags(v)
    cfg, sd, batch, _ = load_case(case)
    batch.image_feature, batch.image_depth = batch.image_feature.to(dt), batch.image_depth.to(dt)
    nobj = [int(b.shape[0]) for b in batch.bbox]
    pidx = enumerate_pairs(nobj)
    start = np.concatenate([[0], np.cumsum(pidx.call_sizes)])
    sdr = {k_: v_.to(dt).clone().requires_grad_(True) for k_, v_ in sd.items()}
    opt = torch.optim.SGD([sdr[n] for n in names], lr=lr, momentum=MOMENTUM, weight_decay=WEIGHT_DECAY)
    weights = O.class_weights(predicate_counts(cfg)).to(dt)
    losses, grads1, updates, stats = [], None, [], dict(grad={}, step_cos=[])
    keep32 = lambda t: t.detach().to(torch.float32 if t.numel() > (1 << 22) else torch.float64).clone()
    for k in range(K):
        s1 = (DROPOUT_SEED * 2654435761 + 2 * (k + 1)) & 0xFFFFFFFF
        s2 = (DROPOUT_SEED * 2654435761 + 2 * (k + 1) + 1) & 0xFFFFFFFF

        def hook(t, b, s1=s1, s2=s2):
            r0 = int(start[t])
            return dict(drop1=torch.from_numpy(dropout_keep_mask(s1, b, 4096, r0)).to(dt) * 2,
                        drop2=torch.from_numpy(dropout_keep_mask(s2, b, 512, r0)).to(dt) * 2)

        before = {n: sdr[n].detach().clone() for n in names}
        FC1_DEFERRED.clear()
        W16.clear()
        out = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=weights, call_hook=hook)
        opt.zero_grad(set_to_none=True)
        out["losses"].backward()
        with torch.no_grad():
            G = torch.cat([g for g, _ in FC1_DEFERRED])
            X = torch.cat([x.detach() for _, x in FC1_DEFERRED])
            sdr["fc1.weight"].grad = G.t() @ X
            del G, X
        FC1_DEFERRED.clear()
        if k == 0:
            if baseline is None:
                grads1 = {n: keep32(sdr[n].grad) for n in names}
            else:
                for n in names:
                    ref = baseline["grads1"][n].double()
                    g = sdr[n].grad.double()
                    nr = float(ref.norm())
                    if nr <= 1e-300:
                        continue
                    stats["grad"][n] = (float((g - ref).norm()) / nr, float((g * ref).sum()) / max(nr * float(g.norm()), 1e-300))
        opt.step()
        losses.append(float(out["losses"].detach()))
        worst = (1.0, None)
        upd = {}
        for n in names:
            u = sdr[n].detach() - before[n]
            if baseline is None:
                upd[n] = keep32(u)
            else:
                r = baseline["updates"][k][n].double()
                u = u.double()
                na, nb = float(u.norm()), float(r.norm())
                if na <= 1e-30 and nb <= 1e-30:
                    continue
                c = float((u * r).sum()) / max(na * nb, 1e-300)
                if c < worst[0]:
                    worst = (c, n)
        if baseline is None:
            updates.append(upd)
        else:
            stats["step_cos"].append(worst)
        del before
    return dict(losses=losses, grads1=grads1, updates=updates, stats=stats)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    case = sys.argv[2] if len(sys.argv) > 2 else "vg_full"
    lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    torch.set_num_threads(int(os.environ.get("SGC_ATTR_THREADS", "8")))
    O.classifier_forward = classifier_forward          # the loop of oracle.run_pair_loop, the classifier of this file
    names = param_names(case)
    out = []
    p = lambda s: (print(s, flush=True), out.append(s))
    p("# python tools/backward_attribution.py %d %s %g   (CPU, float64 unless stated; reference lr config.yaml:51 = 1e-5)" % (K, case, lr))
    t0 = time.time()
    base = run(case, "exact_f64", K, lr, names)
    p("# float64 trajectory: losses " + " ".join("%.2f" % l for l in base["losses"]) + "   (%.0f s)" % (time.time() - t0))
    show = ["conv1_1.weight", "conv2_1.weight", "conv3_1.weight", "fc1.weight", "fc2.weight", "fc3_1.weight"]
    p("#")
    p("# (1) step 1: relative Frobenius error of the gradient against float64 (cosine in brackets), worst tensor last")
    p("%-14s " % "variant" + " ".join("%-22s" % n for n in show) + " worst")
    rows = {}
    for name in VARIANTS:
        if name == "exact_f64" or (only and name not in only):
            continue
        t0 = time.time()
        r = run(case, name, K, lr, names, baseline=base)
        rows[name] = r
        g = r["stats"]["grad"]
        wn = max(g, key=lambda n: g[n][0])
        p("%-14s " % name + " ".join("%-22s" % ("%.2e (%.5f)" % g[n]) for n in show) + " %.2e %s   [%.0f s]" % (g[wn][0], wn, time.time() - t0))
    p("#")
    p("# (2) %d SGD steps at lr %g: smallest update cosine over the parameter tensors per step against the float64 trajectory (tensor of the last step)" % (K, lr))
    for name, r in rows.items():
        sc = r["stats"]["step_cos"]
        p("%-14s " % name + " ".join("%.4f" % c for c, _ in sc) + "   %s | loss(last) %.2f vs %.2f" % (sc[-1][1], r["losses"][-1], base["losses"][-1]))
    if "device_model" in rows and "fwd16" in rows and "bwd_bf16" in rows:
        d = lambda n: 1.0 - rows[n]["stats"]["step_cos"][-1][0]
        p("#")
        p("# drift (1 - cosine) at step %d: device_model %.4f = forward (routes + f16 values) %.4f, of which routes alone %.4f; bf16 backward alone %.4f"
          % (K, d("device_model"), d("fwd16"), d("routes") if "routes" in rows else float("nan"), d("bwd_bf16")))
        if "dy_f16_scaled" in rows:
            p("# with f16 (ideally scaled) instead of bf16 gradient tensors: %.4f" % d("dy_f16_scaled"))
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_backward_attribution%s.txt" % ("" if case == "vg_full" and lr == 1e-5 else "_%s_%g" % (case, lr)))
    if not only:
        with open(dst, "w") as f:
            f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()

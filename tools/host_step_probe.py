"""Per-step wall times (device synchronised after every step) of the benchmark step from process start, with CPython's collector
watched: which steps a full collection lands in and what it costs, and whether the step time drifts because the weights move
(profiles/r05_host_gc.txt).  argv: lr_scale n_steps [freeze]
  lr_scale  multiplies the round-4 benchmark rate 1e-5 * (380/T)^2 (0 = weights fixed; bench.py now uses 1e-3)
  freeze    pair_loop.freeze_setup_objects() before the loop, as train_test.training and bench.py do"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
from scene_graph_commonsense_amd.model import BayesianRelationClassifier
from scene_graph_commonsense_amd.pair_loop import train_minibatch
from scene_graph_commonsense_amd.pairs import flatten_scene
from scene_graph_commonsense_amd.optim import FusedSGD
from scene_graph_commonsense_amd import distributed as sgd_dist
lr_scale, n = float(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
cfg = HeadConfig()
model = BayesianRelationClassifier(cfg.args(run_mode="train")).to(dev)
model.load_state_dict(make_state_dict(cfg, seed=0)); model.train()
batch = make_scene_batch(cfg, [64] * 8, seed=1000, connect_frac=0.02)
batch.image_feature = batch.image_feature.to(dev); batch.image_depth = batch.image_depth.to(dev)
opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4)
reducer = sgd_dist.GradReducer(1)
sc = flatten_scene(cfg, batch, dev)
opt.param_groups[0]["lr"] = lr_scale * 1e-5 * min(1.0, (380.0 / max(sc.n_steps, 1)) ** 2)
ts, losses, host = [], [], []
import gc
gcl = []
def _cb(phase, info, _t=[0.0]):
    if phase == "start": _t[0] = time.time()
    else: gcl.append((len(ts), info["generation"], (time.time() - _t[0]) * 1e3, info["collected"]))
gc.callbacks.append(_cb)
if len(sys.argv) > 3 and sys.argv[3] == "freeze":
    from scene_graph_commonsense_amd.pair_loop import freeze_setup_objects
    freeze_setup_objects()
for i in range(n):
    torch.cuda.synchronize(); t0 = time.time()
    l = train_minibatch(model, batch, opt, reducer=reducer)
    host.append((time.time() - t0) * 1e3)                   # the call returns when everything is enqueued
    torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3); losses.append(float(l))
print("lr_scale", lr_scale, "ms:", " ".join("%.1f" % t for t in ts))
print("   host ms until train_minibatch returns (enqueue only): median %.1f, max after the first %.1f" % (float(np.median(host[1:])), max(host[1:])))
print("   loss:", " ".join("%.4g" % l for l in losses[::5]))
print("   gc events (step, gen, ms, collected) with ms > 2:", [(a, b, round(c, 1), d) for a, b, c, d in gcl if c > 2], "count", len(gcl), "tracked objects", len(gc.get_objects()))
eng = model.engine()
h1 = eng.ws.bufs["h1"][:sc.n_pairs * 4096].float()
print("   h1 zero fraction %.3f  z zero fraction %.3f" % (float((h1 == 0).float().mean()), float((eng.ws.bufs["z_pad"][:4000000].float() == 0).float().mean()) if "z_pad" in eng.ws.bufs else -1))

"""Does the GPU idle between two steps of the un-synchronised step loop (what ``bench.py`` times)?  Events are recorded at the end of
step k and where step k + 1 enters ``model.training_step`` (after ``flatten_scene``'s launches); the GPU time between them is the
scene kernels (~10 us) plus whatever the device waits for the host.  Also: how long the host needs to enqueue a step (if that is the
step time, something in the step synchronises).  argv: n_steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
from scene_graph_commonsense_amd.model import BayesianRelationClassifier
from scene_graph_commonsense_amd.pair_loop import train_minibatch, freeze_setup_objects
from scene_graph_commonsense_amd.pairs import flatten_scene
from scene_graph_commonsense_amd.optim import FusedSGD
from scene_graph_commonsense_amd import distributed as sgd_dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
cfg = HeadConfig()
model = BayesianRelationClassifier(cfg.args(run_mode="train")).to(dev)
model.load_state_dict(make_state_dict(cfg, seed=0)); model.train()
batch = make_scene_batch(cfg, [64] * 8, seed=1000, connect_frac=0.02)
batch.image_feature = batch.image_feature.to(dev); batch.image_depth = batch.image_depth.to(dev)
opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4)
reducer = sgd_dist.GradReducer(1)
sc = flatten_scene(cfg, batch, dev)
opt.param_groups[0]["lr"] = 1e-3 * 1e-5 * min(1.0, (380.0 / max(sc.n_steps, 1)) ** 2)
freeze_setup_objects()
enter = []
orig = model.training_step
def spy(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); enter.append((e, time.time()))
    return orig(*a, **k)
model.training_step = spy
for _ in range(5):
    train_minibatch(model, batch, opt, reducer=reducer)
torch.cuda.synchronize()
enter.clear()
ends, host, t_call = [], [], []
t00 = time.time()
for i in range(n):
    t0 = time.time()
    train_minibatch(model, batch, opt, reducer=reducer)
    host.append((time.time() - t0) * 1e3); t_call.append(t0)
    e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
torch.cuda.synchronize()
wall = (time.time() - t00) * 1e3 / n
gaps = [ends[i].elapsed_time(enter[i + 1][0]) for i in range(n - 1)]
steps = [ends[i].elapsed_time(ends[i + 1]) for i in range(n - 1)]
pre = [(enter[i][1] - t_call[i]) * 1e3 for i in range(n)]
print("wall %.2f ms/step; GPU end(k) -> end(k+1): median %.2f" % (wall, float(np.median(steps))))
print("GPU time from the end of step k to step k+1 entering training_step (scene kernels + idle): median %.3f ms, min %.3f, max %.3f" % (float(np.median(gaps)), min(gaps), max(gaps)))
print("host: enqueue of a whole step median %.2f ms (max %.2f); of which before training_step (zero_grad, plan, flatten_scene) %.2f ms" % (float(np.median(host)), max(host), float(np.median(pre))))

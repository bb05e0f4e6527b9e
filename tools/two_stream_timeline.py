"""The REAL concurrent schedule of one training step (rocprofv3's kernel trace serialises dispatches, so ``tools/timeline.sh`` cannot show
it): the engine's HIP-event timers (``RelHeadEngine._timed``: one event pair around every timed launch, recorded on the stream the
launch runs on) placed on a common clock - elapsed time from an origin event recorded on the caller's stream at the start of the step.
Prints the timed launches of the last step in start order with their stream and where both streams are busy / the main stream waits.
argv: n_steps (default 6).  Only the launches wrapped in ``_timed`` appear (all GEMMs and the larger passes: ~90 % of the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict
from scene_graph_commonsense_amd.model import BayesianRelationClassifier
from scene_graph_commonsense_amd.pair_loop import train_minibatch, freeze_setup_objects
from scene_graph_commonsense_amd.pairs import flatten_scene
from scene_graph_commonsense_amd.optim import FusedSGD
from scene_graph_commonsense_amd import engine as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
cfg = HeadConfig()
model = BayesianRelationClassifier(cfg.args(run_mode="train")).to(dev)
model.load_state_dict(make_state_dict(cfg, seed=0)); model.train()
batch = make_scene_batch(cfg, [64] * 8, seed=1000, connect_frac=0.02)
batch.image_feature = batch.image_feature.to(dev); batch.image_depth = batch.image_depth.to(dev)
opt = FusedSGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4)
sc = flatten_scene(cfg, batch, dev)
opt.param_groups[0]["lr"] = 1e-3 * 1e-5 * min(1.0, (380.0 / max(sc.n_steps, 1)) ** 2)
freeze_setup_objects()
eng = model.engine()
main_id = torch.cuda.current_stream(dev).cuda_stream
rec = []
orig = E.RelHeadEngine._timed
def timed(self, name, fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(dev).cuda_stream
    a.record(); r = fn(); b.record()
    rec.append((name, a, b, st))
    return r
E.RelHeadEngine._timed = timed
for i in range(n):
    rec.clear()
    o = torch.cuda.Event(enable_timing=True); o.record()
    train_minibatch(model, batch, opt)
    e = torch.cuda.Event(enable_timing=True); e.record()
torch.cuda.synchronize()
rows = sorted(((o.elapsed_time(a), o.elapsed_time(b), name, "main" if st == main_id else "side") for name, a, b, st in rec))
print("# step %.2f ms; timed launches in start order (ms from the step's first event)" % o.elapsed_time(e))
for s0, s1, name, st in rows:
    print("%8.3f %8.3f  %6.3f  %-5s %s" % (s0, s1, s1 - s0, st, name))
for which in ("main", "side"):
    iv = sorted((a, b) for a, b, _, st in rows if st == which)
    busy = sum(b - a for a, b in iv)
    print("# %s stream: %.2f ms inside timed launches, first %.2f, last end %.2f" % (which, busy, iv[0][0] if iv else 0, max(b for _, b in iv) if iv else 0))

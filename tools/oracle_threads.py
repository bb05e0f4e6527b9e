"""How fast is the CPU oracle on this host, and how should the tests use the cores?  (GPU box: 128 threads.)
Times 4 reference calls of b = 8 pairs (fwd + loss + bwd, f32) at several torch thread counts, then K concurrent processes of
T threads each.  Output feeds tests/oracle_pool.py's worker / thread split.   python tools/oracle_threads.py [max_procs]"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one(threads, steps=2):
    import torch
    torch.set_num_threads(threads)
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch, make_state_dict, predicate_counts
    cfg = HeadConfig()
    sd = make_state_dict(cfg, seed=3)
    batch = make_scene_batch(cfg, [6] * 8, seed=123, connect_frac=0.3)
    w = O.class_weights(predicate_counts(cfg))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    best = 1e9
    for it in range(3):
        for p in sdr.values():
            p.grad = None
        t0 = time.time()
        out = O.run_pair_loop(sdr, batch, cfg, mode="train", weights=w, max_steps=steps)
        out["losses"].backward()
        dt = time.time() - t0
        if it:
            best = min(best, dt)
    pairs = sum(len(r["keep"]) for r in out["records"])
    return pairs / best


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        print("%.2f" % one(int(sys.argv[2])))
        sys.exit(0)
    ncpu = os.cpu_count()
    print("host threads", ncpu)
    for t in (4, 8, 16, 32, 64, 128):
        if t > ncpu:
            break
        r = subprocess.run([sys.executable, __file__, "--worker", str(t)], capture_output=True, text=True)
        print("1 process x %3d threads: %s pairs/s" % (t, r.stdout.strip() or r.stderr[-300:]))
    for procs, t in ((2, 32), (4, 16), (4, 32), (8, 8), (8, 16), (16, 8)):
        if procs * t > ncpu:
            continue
        t0 = time.time()
        ps = [subprocess.Popen([sys.executable, __file__, "--worker", str(t)], stdout=subprocess.PIPE, text=True) for _ in range(procs)]
        rates = [float(p.communicate()[0].strip() or 0) for p in ps]
        print("%2d processes x %3d threads: %.1f pairs/s in total (%s)" % (procs, t, sum(rates), " ".join("%.1f" % r for r in rates)))

"""Host-side description of the SAMPLED-STEP cases at BASELINE.json's sizes (shared by ``tests/test_sampled_oracle_gpu.py`` and the
oracle-job pre-launch of ``tests/conftest.py``): which (graph_iter, edge_iter) steps of the reference loop the CPU oracle
recomputes, and which rows of the device's pair tables they are.  Pure host code (no GPU, no kernel library)."""
import numpy as np

OIV6 = dict(dataset="oiv6", num_classes=601, num_super_classes=0, num_geometric=4, num_possessive=2, num_semantic=24)
N_FEW, N_MANY, N_SPREAD = 16, 24, 24
SD_SEED, HEAD_GAIN, BATCH_SEED, CONNECT_FRAC = 3, 6.0, 29, 0.3

CASES = {
    "metric_vg_8x64": ({}, [64] * 8),
    "configs1_vg_8x36": ({}, [36] * 8),
    "configs4_oiv6_4x100": (OIV6, [100] * 4),
    "configs0_vg_10x20": ({}, [20] * 10),
}
# the backward of these is compared with the oracle's: the three sizes the bench-style kernels dominate
BACKWARD_CASES = ("metric_vg_8x64", "configs1_vg_8x36", "configs4_oiv6_4x100")


def x_windows_per_pair(bbox_norm, pidx):
    from scene_graph_commonsense_amd.pairs import object_window_rects
    r = object_window_rects(bbox_norm)
    a, b = r[pidx.sub], r[pidx.obj]
    ox = np.clip(np.minimum(a[:, 1], b[:, 1]) - np.maximum(a[:, 0], b[:, 0]), 0, None)
    oy = np.clip(np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 2], b[:, 2]), 0, None)
    return ox * oy


def pick_steps(pidx, xw):
    """(g, e) steps of the loop: fewest / most pair-specific windows on average over the step's pairs + an even spread."""
    key = pidx.g * 4096 + pidx.e
    uniq, inv = np.unique(key, return_inverse=True)
    mean_x = np.bincount(inv, weights=xw.astype(np.float64)) / np.bincount(inv)
    order = np.argsort(mean_x, kind="stable")
    chosen = list(order[:N_FEW]) + list(order[-N_MANY:])
    rest = [k for k in np.linspace(0, len(uniq) - 1, N_SPREAD + 8).astype(int) if k not in set(chosen)][:N_SPREAD]
    chosen = sorted(set(chosen + rest))
    return {(int(uniq[k]) // 4096, int(uniq[k]) % 4096) for k in chosen}, mean_x[order[:N_FEW]].mean(), mean_x[order[-N_MANY:]].mean()


_CACHE = {}


def host_case(name):
    """dict(cfg, sd_args, batch, pidx, xw, steps, records=[(g, e, first)], rows=[row arrays, record order], few, many)."""
    if name in _CACHE:
        return _CACHE[name]
    import torch
    from scene_graph_commonsense_amd.pairs import enumerate_pairs, normalise_boxes
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_scene_batch
    kw, nobj = CASES[name]
    cfg = HeadConfig(**kw)
    batch = make_scene_batch(cfg, nobj, seed=BATCH_SEED, connect_frac=CONNECT_FRAC)
    pidx = enumerate_pairs(nobj)
    bb = normalise_boxes(torch.cat([b for b in batch.bbox]), cfg.feature_size)
    xw = x_windows_per_pair(bb, pidx)
    steps, few, many = pick_steps(pidx, xw)
    records, rows = [], []
    for g, e in sorted(steps):                       # the reference loop visits them in (g, e) order, direction 1 then 2
        for first in (True, False):
            records.append((g, e, first))
            rows.append(np.nonzero((pidx.g == g) & (pidx.e == e) & (pidx.first == first))[0])
    _CACHE[name] = dict(cfg=cfg, cfg_kw=kw, nobj=nobj, batch=batch, pidx=pidx, bbox=bb, xw=xw, steps=steps, records=records, rows=rows,
                        few=few, many=many)
    return _CACHE[name]


DROPOUT_SEEDS = (0xC0FFEE, 0xBADC0DE)          # the routed comparison runs in TRAINING mode (dropout on): what bench.py times


def job_spec(name, backward, routes=None, dropout_seeds=None):
    """``dropout_seeds``: the oracle gets the kernels' keep masks of the sampled pairs injected (``synthetic.dropout_keep_mask`` over the
    pair's row in the FULL minibatch: the keep bit is indexed by the position in the pass)."""
    hc = host_case(name)
    return dict(kind="sampled", cfg_kw=hc["cfg_kw"], sd_seed=SD_SEED, head_gain=HEAD_GAIN, nobj=hc["nobj"], batch_seed=BATCH_SEED,
                connect_frac=CONNECT_FRAC, steps=sorted(hc["steps"]), backward=backward, routes=routes,
                call_sizes=[len(r) for r in hc["rows"]], dropout_seeds=dropout_seeds,
                call_row0=[int(r[0]) for r in hc["rows"]])


def job_name(name, routed=False):
    return "sampled_%s%s" % (name, "_routed" if routed else "")


def prelaunch(names=None):
    """Submit the oracle jobs that need nothing from the device (called when collection ends)."""
    from tests import oracle_pool
    for name in (names or CASES):
        oracle_pool.submit(job_name(name), job_spec(name, backward=name in BACKWARD_CASES))

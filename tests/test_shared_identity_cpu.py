"""The identity behind conv3 / fc1 over shared windows, on the REFERENCE's literal graph in float64 on the CPU:
the reference masks the features with the box before conv1 (train_test.py:194-195), so for every ordered pair (i, j) the pooled conv3
output at a pooling window outside R_j equals the output of the pair (i, empty box), inside R_j but outside R_i that of
(empty box, j) - with R_o the closed-form rectangle of pairs.object_window_rects - and consequently fc1's pre-activation is
b + S_i[all] - S_i[R_j] + S'_j[R_j] - S'_j[X] + sum over X of the pair's own window products.  No GPU, no kernels: this pins the
mathematics the HIP path relies on against the oracle's restatement of model.py:138-149."""
import numpy as np
import torch
import torch.nn.functional as F


def _pooled_conv3(sd, h_sub, h_obj):
    """model.py:139-147 up to the second max-pool, as oracle.relhead_oracle.conv_trunk does it: [b, 1024, 8, 8]."""
    a = torch.tanh(F.conv2d(h_sub, sd["conv1_1.weight"], sd["conv1_1.bias"]))
    b = torch.tanh(F.conv2d(h_obj, sd["conv1_2.weight"], sd["conv1_2.bias"]))
    h = F.conv2d(torch.cat((a, b), dim=1), sd["conv2_1.weight"], sd["conv2_1.bias"], padding=1)
    h = F.max_pool2d(F.relu(h), 2, 2)
    h = F.conv2d(h, sd["conv3_1.weight"], sd["conv3_1.bias"], padding=1)
    return F.max_pool2d(F.relu(h), 2, 2)


def test_windows_outside_the_intersection_are_per_object_and_fc1_is_the_rectangle_sum():
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd import pairs as PR
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * k)
    # a narrow trunk (64 / 96 channels instead of 512 / 1024, fc1 -> 32) keeps the CPU time at seconds; geometry is what matters
    c2, c3, nf = 64, 96, 32
    sd = {"conv1_1.weight": r(128, 257, 1, 1, k=0.05), "conv1_1.bias": r(128, k=0.3), "conv1_2.weight": r(128, 257, 1, 1, k=0.05),
          "conv1_2.bias": r(128, k=0.3), "conv2_1.weight": r(c2, 256, 3, 3, k=0.03), "conv2_1.bias": r(c2, k=0.1),
          "conv3_1.weight": r(c3, c2, 3, 3, k=0.05), "conv3_1.bias": r(c3, k=0.1)}
    W1, b1 = r(nf, c3 * 64, k=0.02), r(nf, k=0.1)
    feat = torch.cat((r(1, 256, 32, 32), torch.rand(1, 1, 32, 32, generator=g, dtype=torch.float64)), dim=1)
    boxes = torch.tensor([[0, 32, 0, 32], [5, 6, 7, 8], [2, 13, 20, 31], [10, 22, 9, 14], [29, 32, 0, 3], [12, 20, 12, 20], [0, 0, 4, 9]])
    n = boxes.shape[0]
    masks = O.build_masks(boxes, 32).to(torch.float64)                      # the reference's rectangle rasterisation
    empty = torch.zeros(1, 32, 32, dtype=torch.float64)
    h = lambda m: feat * m.view(1, 1, 32, 32)
    R = PR.object_window_rects(PR.normalise_boxes(boxes, 32))
    wy, wx = np.divmod(np.arange(64), 8)
    inside = lambda o: (wx >= R[o, 0]) & (wx < R[o, 1]) & (wy >= R[o, 2]) & (wy < R[o, 3])
    with torch.no_grad():
        y_sub = [_pooled_conv3(sd, h(masks[i]), h(empty)).reshape(c3, 64) for i in range(n)]       # pseudo-pair (i, background)
        y_obj = [_pooled_conv3(sd, h(empty), h(masks[j])).reshape(c3, 64) for j in range(n)]       # pseudo-pair (background, j)
        W1w = W1.view(nf, c3, 64)                                            # reference column order of fc1: channel-major
        T_sub = [torch.einsum("fcw,cw->wf", W1w, y) for y in y_sub]          # per-object per-window products [64, nf]
        T_obj = [torch.einsum("fcw,cw->wf", W1w, y) for y in y_obj]
        seen = {"I": 0, "J": 0, "X": 0}
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                y = _pooled_conv3(sd, h(masks[i]), h(masks[j])).reshape(c3, 64)
                in_i, in_j = inside(i), inside(j)
                I, J, X = ~in_j, in_j & ~in_i, in_j & in_i
                seen["I"] += int(I.sum()); seen["J"] += int(J.sum()); seen["X"] += int(X.sum())
                assert torch.equal(y[:, I], y_sub[i][:, I]), (i, j)          # identical inputs -> identical values, bit for bit
                assert torch.equal(y[:, J], y_obj[j][:, J]), (i, j)
                ref = F.linear(y.reshape(1, -1), W1, b1)[0]                  # model.py:148: fc1 on the flattened [c, 8, 8] map
                own = torch.einsum("fcw,cw->wf", W1w, y)
                mine = b1 + T_sub[i][I].sum(0) + T_obj[j][J].sum(0) + own[X].sum(0)
                assert torch.allclose(mine, ref, rtol=0, atol=1e-11 * float(ref.abs().max())), (i, j)
        assert min(seen.values()) > 0                                        # every window type occurred


def test_linear_pairs_preactivation_is_the_sum_of_per_object_preactivations():
    """Sixth identity ("linear pairs"), again on the reference's literal graph in float64: when the 16-grid regions where
    the two objects' conv2 halves differ from the background's (pairs.object_d16_rects) are disjoint, the pair's conv3 PRE-activation
    equals  pre(i, bg) + pre(bg, j) - pre(bg, bg)  everywhere (conv3_1 bias counted once) - so its pooled output on the pair's X
    windows needs no convolution of its own.  Pairs whose regions meet must NOT satisfy it (the test has both), and the host's
    window count (pairs.count_linear_windows, what sizes the device lists) must be the brute-force count."""
    from oracle import relhead_oracle as O
    from scene_graph_commonsense_amd import pairs as PR
    g = torch.Generator().manual_seed(11)
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * k)
    c2, c3 = 48, 40
    sd = {"conv1_1.weight": r(128, 257, 1, 1, k=0.05), "conv1_1.bias": r(128, k=0.3), "conv1_2.weight": r(128, 257, 1, 1, k=0.05),
          "conv1_2.bias": r(128, k=0.3), "conv2_1.weight": r(c2, 256, 3, 3, k=0.03), "conv2_1.bias": r(c2, k=0.1),
          "conv3_1.weight": r(c3, c2, 3, 3, k=0.05), "conv3_1.bias": r(c3, k=0.1)}

    def pre(h_sub, h_obj):                                      # model.py:139-145 up to conv3_1 (before its ReLU): [c3, 16, 16]
        a = torch.tanh(F.conv2d(h_sub, sd["conv1_1.weight"], sd["conv1_1.bias"]))
        b = torch.tanh(F.conv2d(h_obj, sd["conv1_2.weight"], sd["conv1_2.bias"]))
        h = F.conv2d(torch.cat((a, b), dim=1), sd["conv2_1.weight"], sd["conv2_1.bias"], padding=1)
        h = F.max_pool2d(F.relu(h), 2, 2)
        return F.conv2d(h, sd["conv3_1.weight"], sd["conv3_1.bias"], padding=1)[0]

    feat = torch.cat((r(1, 256, 32, 32), torch.rand(1, 1, 32, 32, generator=g, dtype=torch.float64)), dim=1)
    boxes = torch.tensor([[2, 9, 3, 8], [14, 20, 2, 9], [3, 8, 15, 24], [13, 17, 14, 18], [24, 31, 20, 30], [0, 32, 0, 32], [8, 14, 5, 12],
                          [19, 25, 9, 16], [6, 6, 4, 9]])
    n = boxes.shape[0]
    bb = PR.normalise_boxes(boxes, 32)
    R, D = PR.object_window_rects(bb), PR.object_d16_rects(bb)
    masks = O.build_masks(boxes, 32).to(torch.float64)
    empty = torch.zeros(1, 32, 32, dtype=torch.float64)
    h = lambda m: feat * m.view(1, 1, 32, 32)
    n_lin = n_not = brute = 0
    with torch.no_grad():
        p_sub = [pre(h(masks[i]), h(empty)) for i in range(n)]
        p_obj = [pre(h(empty), h(masks[j])) for j in range(n)]
        p_bg = pre(h(empty), h(empty))
        bias = sd["conv3_1.bias"].view(-1, 1, 1)
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                xw = max(0, min(R[i, 1], R[j, 1]) - max(R[i, 0], R[j, 0])) * max(0, min(R[i, 3], R[j, 3]) - max(R[i, 2], R[j, 2]))
                meet = min(D[i, 1], D[j, 1]) > max(D[i, 0], D[j, 0]) and min(D[i, 3], D[j, 3]) > max(D[i, 2], D[j, 2])
                if xw == 0:
                    continue
                mine = p_sub[i] + p_obj[j] - p_bg                   # each term carries the bias once: +1 +1 -1 = once
                ref = pre(h(masks[i]), h(masks[j]))
                err = float((mine - ref).abs().max() / ref.abs().max())
                if not meet:
                    assert err <= 1e-12, (i, j, err)
                    n_lin += 1
                    brute += int(xw)
                else:
                    n_not += err > 1e-6                              # pairs whose regions meet are genuinely pair-specific
        assert n_lin >= 6 and n_not >= 6, (n_lin, n_not)
        assert float(bias.abs().max()) > 0
    assert PR.count_linear_windows(bb, [0, n]) == brute


def test_d16_rectangles_match_a_brute_force_influence_propagation():
    """pairs.object_d16_rects (host replica of csrc/kernels_shared.hip:axis_d16) against the literal chain: box mask -> 3x3 dilation on
    the 32-grid (conv2_1) -> 2x2 pooling, on random boxes including degenerate, border and out-of-grid ones."""
    from scene_graph_commonsense_amd import pairs as PR
    rng = np.random.default_rng(4)
    boxes = []
    for _ in range(300):
        x0, y0 = rng.integers(-3, 33, 2)
        boxes.append([x0, x0 + rng.integers(-2, 20), y0, y0 + rng.integers(-2, 20)])
    boxes = np.array(boxes + [[0, 32, 0, 32], [0, 1, 0, 1], [31, 32, 31, 32], [5, 5, 3, 9]])
    bb = PR.normalise_boxes(torch.from_numpy(boxes[boxes.min(1) >= 0]), 32)          # (negative starts: Python slice semantics, tested elsewhere)
    D = PR.object_d16_rects(bb)
    for k, (x0, x1, y0, y1) in enumerate(bb):
        m = np.zeros((32, 32), dtype=bool)
        m[y0:y1, x0:x1] = True
        p = np.pad(m, 1)
        d = np.zeros_like(m)
        for dy in range(3):
            for dx in range(3):
                d |= p[dy:dy + 32, dx:dx + 32]
        d16 = d.reshape(16, 2, 16, 2).any(axis=(1, 3))
        want = np.zeros((16, 16), dtype=bool)
        want[D[k, 2]:D[k, 3], D[k, 0]:D[k, 1]] = True
        assert np.array_equal(d16, want), (k, bb[k], D[k])

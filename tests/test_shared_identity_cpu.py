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

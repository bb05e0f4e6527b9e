"""The individual entry points of csrc/kernels_shared.hip against numpy / torch restatements on random inputs (the path as a whole
is checked in tests/test_shared_conv3_gpu.py): window plan, packed pixel rectangles, un-pool of listed windows, im2col / col2im of
the column forms, prefix sums and the per-object gradient sums of the shared fc1."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from scene_graph_commonsense_amd import _lib as L
    return L, L.load()


def _boxes(rng, n):
    x0, y0 = rng.integers(0, 30, n), rng.integers(0, 30, n)
    b = np.stack([x0, x0 + rng.integers(1, 33 - x0), y0, y0 + rng.integers(1, 33 - y0)], axis=1)
    b[0] = [0, 32, 0, 32]
    b[1] = [7, 7, 3, 9]                                     # empty box
    return b.astype(np.int32)


def _plan(rng, n=9):
    """boxes, all ordered pairs, device plan (count / incl / gather / pixrect) + the numpy rectangles."""
    from scene_graph_commonsense_amd.pairs import object_window_rects
    L, lib = _lib()
    bb = _boxes(rng, n)
    sub, obj = np.array([(i, j) for i in range(n) for j in range(n) if i != j], dtype=np.int32).T
    P = len(sub)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    bb_d, sub_d, obj_d = d(bb), d(sub), d(obj)
    cnt = torch.empty(P, dtype=torch.int32, device=DEV)
    pix = torch.empty(P, dtype=torch.int32, device=DEV)
    L.check(lib.sgc_shared_windows_count(L.ptr(bb_d), L.ptr(sub_d), L.ptr(obj_d), P, L.ptr(cnt), L.ptr(pix), L.stream_ptr()), "count")
    incl = torch.cumsum(cnt, 0, dtype=torch.int32)
    gather = torch.full((P * 64,), -1, dtype=torch.int32, device=DEV)
    L.check(lib.sgc_shared_windows_fill(L.ptr(bb_d), L.ptr(sub_d), L.ptr(obj_d), P, L.ptr(incl), L.ptr(gather), L.stream_ptr()), "fill")
    R = object_window_rects(bb)
    return dict(bb=bb, sub=sub, obj=obj, P=P, bb_d=bb_d, sub_d=sub_d, obj_d=obj_d, cnt=cnt, incl=incl, gather=gather, pix=pix, R=R)


def test_window_plan_and_pixel_rectangles():
    pl = _plan(np.random.default_rng(0))
    R, want_g, want_pix = pl["R"], [], []
    for p, (i, j) in enumerate(zip(pl["sub"], pl["obj"])):
        x0, x1, y0, y1 = max(R[i, 0], R[j, 0]), min(R[i, 1], R[j, 1]), max(R[i, 2], R[j, 2]), min(R[i, 3], R[j, 3])
        if x1 <= x0 or y1 <= y0:
            want_pix.append(0)
            continue
        want_g += [p * 64 + wy * 8 + wx for wy in range(y0, y1) for wx in range(x0, x1)]
        want_pix.append(max(2 * y0 - 1, 0) | (min(2 * y1 + 1, 16) << 5) | (max(2 * x0 - 1, 0) << 10) | (min(2 * x1 + 1, 16) << 15))
    E = int(pl["incl"][-1])
    assert E == len(want_g) and pl["gather"][:E].cpu().tolist() == want_g and int(pl["gather"][E]) == -1
    assert pl["pix"].cpu().tolist() == want_pix


def test_unpool_im2col_col2im_of_the_listed_windows():
    L, lib = _lib()
    rng = np.random.default_rng(1)
    pl = _plan(rng, n=6)
    P, E = pl["P"], int(pl["incl"][-1])
    Epad = (E + 15) // 16 * 16
    g = pl["gather"][:E].long()
    gn = pl["incl"][P - 1:]
    # ---- un-pool: dy3x[4e+q][c] = (code == q) ? dy[gather[e]][c] : 0, zero rows behind the list, bias partials over live routes
    dy = torch.randn(P * 64, 1024, device=DEV).bfloat16()
    am = torch.randint(0, 5, (P * 64, 1024), device=DEV, dtype=torch.uint8)
    dy3x = torch.full((Epad * 4, 1024), float("nan"), device=DEV).bfloat16()
    bpart = torch.zeros(1024, 1024, device=DEV)
    nparts = ctypes.c_int(0)
    L.check(lib.sgc_windows_unpool(L.ptr(dy), L.ptr(am), L.ptr(pl["gather"]), L.ptr(gn), None, Epad, L.ptr(dy3x), L.ptr(bpart), ctypes.byref(nparts),
                                   L.stream_ptr()), "unpool")
    want = torch.zeros(Epad, 4, 1024, device=DEV)
    for q in range(4):
        want[:E, q] = torch.where(am[g] == q, dy[g].float(), torch.zeros((), device=DEV))
    assert torch.equal(dy3x.float().view(Epad, 4, 1024), want)
    live = torch.where(am[g] < 4, dy[g].float(), torch.zeros((), device=DEV)).sum(0)
    assert torch.allclose(bpart[:nparts.value].sum(0), live, atol=1e-3, rtol=1e-4)
    # ---- im2col: zcol[4e+q][tap] = z[pair][y+ky][x+kx]
    z = torch.randn(P, 18, 18, 512, device=DEV).bfloat16()
    zcol = torch.full((Epad * 4, 9, 512), float("nan"), device=DEV).bfloat16()
    L.check(lib.sgc_windows_im2col(L.ptr(z), L.ptr(pl["gather"]), L.ptr(gn), Epad, L.ptr(zcol), L.stream_ptr()), "im2col")
    pair, w = (g >> 6), (g & 63)
    for q in range(4):
        y, x = 2 * (w >> 3) + (q >> 1), 2 * (w & 7) + (q & 1)
        for tap in range(9):
            assert torch.equal(zcol.view(Epad, 4, 9, 512)[:E, q, tap], z[pair, y + tap // 3, x + tap % 3])
    if Epad > E:
        assert float(zcol.view(Epad, 4, 9, 512)[E:].float().abs().max()) == 0.0          # padding entries are zero rows
    # ---- col2im: dz[pair][pixel] = sum over taps of col[row of (pixel - tap + 1)][tap], only inside the pair's pixel rectangle
    col = torch.randn(Epad * 4, 9, 512, device=DEV).bfloat16()
    dz = torch.full((P * 256, 512), float("nan"), device=DEV).bfloat16()
    L.check(lib.sgc_windows_col2im(L.ptr(col), L.ptr(pl["bb_d"]), L.ptr(pl["sub_d"]), L.ptr(pl["obj_d"]), L.ptr(pl["incl"]), P, L.ptr(dz),
                                   L.stream_ptr()), "col2im")
    acc = torch.zeros(P, 18, 18, 512, device=DEV)                 # padded scatter form
    colf = col.float().view(Epad, 4, 9, 512)
    for q in range(4):
        y, x = 2 * (w >> 3) + (q >> 1), 2 * (w & 7) + (q & 1)
        for tap in range(9):
            acc.index_put_((pair, y + tap // 3, x + tap % 3), colf[:E, q, tap], accumulate=True)
    pix = pl["pix"].cpu().numpy()
    dzv = dz.float().view(P, 256, 512)
    for p in range(P):
        r = int(pix[p])
        Y0, Y1, X0, X1 = r & 31, (r >> 5) & 31, (r >> 10) & 31, (r >> 15) & 31
        for Y in range(16):
            for X in range(16):
                m3 = 4 * ((Y >> 1) * 8 + (X >> 1)) + (Y & 1) * 2 + (X & 1)
                if Y0 <= Y < Y1 and X0 <= X < X1:
                    assert torch.allclose(dzv[p, m3], acc[p, Y + 1, X + 1], atol=0.07, rtol=2e-2)       # bf16 output of up to 9 terms
                else:
                    assert torch.isnan(dzv[p, m3]).all()                                             # never written


def test_fc1_prefix_sums_and_per_object_gradient_sums():
    from scene_graph_commonsense_amd.pairs import window_major_layout
    L, lib = _lib()
    rng = np.random.default_rng(2)
    pl = _plan(rng, n=7)
    n_obj, P, n2 = 7, pl["P"], 14
    goff, _ = window_major_layout(np.zeros(64, dtype=np.int64), n2)
    goff_d = torch.from_numpy(goff).to(DEV)
    rows = int(goff[64])
    owm = torch.randn(rows, int(lib.sgc_fc1_products_pitch()), device=DEV)       # padded row pitch; columns 0..4095 are the products
    S = torch.full((n2, 9, 9, 4096), float("nan"), device=DEV)
    L.check(lib.sgc_fc1_integral(L.ptr(owm), L.ptr(goff_d), n2, L.ptr(S), L.stream_ptr()), "integral")
    T = torch.stack([owm[int(goff[w]):int(goff[w]) + n2, :4096] for w in range(64)], dim=1).view(n2, 8, 8, 4096)
    want = torch.zeros(n2, 9, 9, 4096, device=DEV)
    want[:, 1:, 1:] = T.cumsum(1).cumsum(2)
    assert torch.allclose(S, want, atol=1e-4, rtol=1e-5)
    # ---- per-object sums of dh1: role 0 = windows outside the partner's rectangle, role 1 = inside the own, outside the partner's
    from scene_graph_commonsense_amd.engine import csr_by
    dh = torch.randn(P, 4096, device=DEV).bfloat16()
    sp, sl = (torch.from_numpy(a).to(DEV) for a in csr_by(pl["sub"], n_obj))
    op, ol = (torch.from_numpy(a).to(DEV) for a in csr_by(pl["obj"], n_obj))
    gwm = torch.zeros(rows, 4096, device=DEV).bfloat16()
    L.check(lib.sgc_fc1_gsum(L.ptr(dh), L.ptr(pl["bb_d"]), L.ptr(pl["sub_d"]), L.ptr(pl["obj_d"]), L.ptr(sp), L.ptr(sl), L.ptr(op), L.ptr(ol),
                             L.ptr(goff_d), n_obj, L.ptr(gwm), L.stream_ptr()), "gsum")
    R = pl["R"]
    wy, wx = np.divmod(np.arange(64), 8)
    ins = lambda o: (wx >= R[o, 0]) & (wx < R[o, 1]) & (wy >= R[o, 2]) & (wy < R[o, 3])
    want = torch.zeros(2, n_obj, 64, 4096, device=DEV)
    for p, (i, j) in enumerate(zip(pl["sub"], pl["obj"])):
        want[0, i][torch.from_numpy(~ins(j)).to(DEV)] += dh[p].float()
        want[1, j][torch.from_numpy(ins(j) & ~ins(i)).to(DEV)] += dh[p].float()
    got = torch.stack([gwm[int(goff[w]):int(goff[w]) + n2].float() for w in range(64)], dim=1).view(2, n_obj, 64, 4096)
    assert torch.allclose(got, want, atol=0.05, rtol=1e-2)                 # bf16 output of sums of up to 6 terms


def test_second_level_rows_and_their_gradient():
    """sgc_shared_objects_fill_rows copies the background map's rows to the pseudo-pairs' window-major rows outside R_o (and nowhere
    else); sgc_shared_objects_bg_grad is its transpose; sgc_shared_objects_count / _fill list exactly the windows of R_o."""
    from scene_graph_commonsense_amd.pairs import count_object_windows, window_major_layout
    L, lib = _lib()
    rng = np.random.default_rng(3)
    n_obj, n_img = 7, 2
    pl = _plan(rng, n=n_obj)
    R, bb_d = pl["R"], pl["bb_d"]
    obj_img = torch.tensor([0, 0, 0, 1, 1, 1, 1], dtype=torch.int32, device=DEV)
    img_ptr = torch.tensor([0, 3, 7], dtype=torch.int32, device=DEV)
    n2 = 2 * n_obj
    goff, _ = window_major_layout(np.zeros(64, dtype=np.int64), n2)
    goff_d = torch.from_numpy(goff).to(DEV)
    rows = int(goff[64])
    wy, wx = np.divmod(np.arange(64), 8)
    ins = lambda o: (wx >= R[o, 0]) & (wx < R[o, 1]) & (wy >= R[o, 2]) & (wy < R[o, 3])
    # ---- plan of the pseudo-pairs
    cnt = torch.zeros(n2, dtype=torch.int32, device=DEV)
    pix = torch.zeros(n2, dtype=torch.int32, device=DEV)
    L.check(lib.sgc_shared_objects_count(L.ptr(bb_d), n_obj, L.ptr(cnt), L.ptr(pix), L.stream_ptr()), "count")
    assert cnt.cpu().tolist() == [int(ins(o % n_obj).sum()) for o in range(n2)] and int(cnt.sum()) == count_object_windows(pl["bb"])
    P = 5                                                       # pretend five real pairs without X windows in front
    incl = torch.cumsum(torch.cat([torch.zeros(P, dtype=torch.int32, device=DEV), cnt]), 0, dtype=torch.int32)
    gather = torch.full((int(incl[-1]) + 3,), -1, dtype=torch.int32, device=DEV)
    L.check(lib.sgc_shared_objects_fill(L.ptr(bb_d), n_obj, P, L.ptr(incl), L.ptr(gather), L.stream_ptr()), "fill")
    want = [(P + ps) * 64 + int(w) for ps in range(n2) for w in np.nonzero(ins(ps % n_obj))[0]]
    assert gather[:len(want)].cpu().tolist() == want and int(gather[len(want)]) == -1
    # ---- rows
    y_bg = torch.randn(n_img * 64, 1024, device=DEV).half()
    ybf_bg = y_bg.bfloat16()
    am_bg = torch.randint(0, 5, (n_img * 64, 1024), device=DEV, dtype=torch.uint8)
    ywm = torch.full((rows, 1024), -7.0, device=DEV).half()
    ywm_bf = torch.full((rows, 1024), -7.0, device=DEV).bfloat16()
    am_ps = torch.full((n2 * 64, 1024), 9, device=DEV, dtype=torch.uint8)
    L.check(lib.sgc_shared_objects_fill_rows(L.ptr(bb_d), L.ptr(obj_img), n_obj, L.ptr(goff_d), L.ptr(y_bg), L.ptr(ybf_bg), L.ptr(am_bg),
                                             L.ptr(ywm), L.ptr(ywm_bf), L.ptr(am_ps), L.stream_ptr()), "fill_rows")
    for ps in range(n2):
        o = ps % n_obj
        for w in range(64):
            row = int(goff[w]) + ps
            if ins(o)[w]:
                assert float(ywm[row, 0]) == -7.0 and int(am_ps[ps * 64 + w, 0]) == 9          # left for the window-list entry
            else:
                src = int(obj_img[o]) * 64 + w
                assert torch.equal(ywm[row], y_bg[src]) and torch.equal(ywm_bf[row], ybf_bg[src]) and torch.equal(am_ps[ps * 64 + w], am_bg[src])
    # ---- gradient of the copies
    dywm = torch.randn(rows, 1024, device=DEV).bfloat16()
    dy_bg = torch.zeros(n_img * 64, 1024, device=DEV).bfloat16()
    L.check(lib.sgc_shared_objects_bg_grad(L.ptr(bb_d), L.ptr(img_ptr), n_obj, n_img, L.ptr(goff_d), L.ptr(dywm), L.ptr(dy_bg), L.stream_ptr()),
            "bg_grad")
    want = torch.zeros(n_img * 64, 1024, device=DEV)
    for ps in range(n2):
        o = ps % n_obj
        for w in range(64):
            if not ins(o)[w]:
                want[int(obj_img[o]) * 64 + w] += dywm[int(goff[w]) + ps].float()
    assert torch.allclose(dy_bg.float(), want, atol=0.05, rtol=1e-2)


def test_patch_form_of_the_conv3_data_gradient_equals_the_transposed_convolution():
    """``sgc_windows_dgrad_patches`` + ``sgc_windows_patch_sum`` (16 patch pixels per listed window, the taps summed inside the
    GEMM) against the f32 transposed convolution of the un-pooled gradient (reference: autograd of model.py:144-146's conv3 on the
    pair's own pixels), and against the column form + col2im the step used before."""
    L, lib = _lib()
    rng = np.random.default_rng(5)
    pl = _plan(rng, n=7)
    P, E = pl["P"], int(pl["incl"][-1])
    Epad = (E + 63) // 64 * 64
    g = pl["gather"][:E].long()
    pair, w = (g >> 6), (g & 63)
    gen = torch.Generator(device=DEV).manual_seed(3)
    w3 = torch.randn(1024, 512, 3, 3, device=DEV, generator=gen) * 0.02
    dy3x = torch.zeros(Epad * 4, 1024, device=DEV)
    dy3x[:E * 4] = torch.randn(E * 4, 1024, device=DEV, generator=gen) * (torch.rand(E * 4, 1024, device=DEV, generator=gen) < 0.25)
    dy3x = dy3x.bfloat16()
    # weights in the two layouts of engine.prep_bwd_weights
    opts = lambda c: [(0, 0)] if c == 0 else ([(1, 2)] if c == 3 else [(0, c), (1, c - 1)])
    w3patch = torch.cat([torch.cat([w3[:, :, ky, kx].t() for _, ky in opts(py) for _, kx in opts(px)], dim=1).reshape(-1)
                         for py in range(4) for px in range(4)]).bfloat16().contiguous()
    w3col = w3.permute(2, 3, 1, 0).reshape(9 * 512, 1024).bfloat16().contiguous()
    slots = int(lib.sgc_windows_patch_slots())                                    # 20: the centre pixels take two rows (K <= 2048 per row)
    patch = torch.full((Epad * slots, 512), float("nan"), device=DEV).bfloat16()
    L.check(lib.sgc_windows_dgrad_patches(L.ptr(dy3x), L.ptr(w3patch), L.ptr(patch), Epad, L.stream_ptr()), "dgrad_patches")
    # ---- every patch pixel against f32: patch[e][py][px] = sum_{q + k = (py, px)} dy3x[4e + q] @ W[:, :, ky, kx]
    wb = w3.bfloat16().float()
    dyf = dy3x.float().view(Epad, 4, 1024)
    want = torch.zeros(Epad, 4, 4, 512, device=DEV)
    for q in range(4):
        for ky in range(3):
            for kx in range(3):
                want[:, (q >> 1) + ky, (q & 1) + kx] += dyf[:, q] @ wb[:, :, ky, kx]
    rows, k = [], 0
    for pp in range(16):
        c = (pp >> 2) in (1, 2) and (pp & 3) in (1, 2)
        rows.append((k, k + 1) if c else (k,))
        k += 2 if c else 1
    assert k == slots
    pf = patch.float().view(Epad, slots, 512)
    got = torch.stack([sum(pf[:, j] for j in t) for t in rows], dim=1).view(Epad, 4, 4, 512)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-2 * scale                       # one bf16 rounding of an f32 sum
    assert float(got[E:].abs().max()) == 0.0                                     # entries behind the list: zero rows in, zero rows out
    # ---- the sum over a pair's windows against the column form + col2im on the same operands
    dz_p = torch.full((P * 256, 512), float("nan"), device=DEV).bfloat16()
    L.check(lib.sgc_windows_patch_sum(L.ptr(patch), L.ptr(pl["bb_d"]), L.ptr(pl["sub_d"]), L.ptr(pl["obj_d"]), L.ptr(pl["incl"]), P, L.ptr(dz_p),
                                      L.stream_ptr()), "patch_sum")
    col = torch.empty(Epad * 4, 9 * 512, device=DEV).bfloat16()
    L.check(lib.sgc_windows_dgrad_cols(L.ptr(dy3x), L.ptr(w3col), L.ptr(col), Epad * 4, L.stream_ptr()), "dgrad_cols")
    dz_c = torch.full((P * 256, 512), float("nan"), device=DEV).bfloat16()
    L.check(lib.sgc_windows_col2im(L.ptr(col), L.ptr(pl["bb_d"]), L.ptr(pl["sub_d"]), L.ptr(pl["obj_d"]), L.ptr(pl["incl"]), P, L.ptr(dz_c),
                                   L.stream_ptr()), "col2im")
    a, b = dz_p.float(), dz_c.float()
    assert torch.equal(torch.isnan(a), torch.isnan(b))                            # the same pixels are written
    live = ~torch.isnan(a)
    assert live.any()
    # f32 reference of the whole thing: scatter the exact patches
    acc = torch.zeros(P, 18, 18, 512, device=DEV)
    for py in range(4):
        for px in range(4):
            acc.index_put_((pair, 2 * (w >> 3) + py, 2 * (w & 7) + px), want[:E, py, px], accumulate=True)
    m3 = torch.arange(256, device=DEV)
    Y, X = 2 * ((m3 >> 2) >> 3) + ((m3 & 3) >> 1), 2 * ((m3 >> 2) & 7) + (m3 & 1)
    ref = acc[:, Y + 1, X + 1].reshape(P * 256, 512)     # padded frame: patch pixel py of window wy is row 2 wy + py, image pixel y is row y + 1
    s2 = float(ref[live].abs().max())
    assert float((a[live] - ref[live]).abs().max()) <= 2e-2 * s2
    assert float((b[live] - ref[live]).abs().max()) <= 4e-2 * s2                  # the column form rounds nine terms to bf16 before adding
    assert float((a[live] - ref[live]).abs().mean()) <= float((b[live] - ref[live]).abs().mean())   # the patch form is the more accurate one


def test_patch_form_of_the_conv3_weight_gradient_is_bitwise_the_im2col_form():
    """``sgc_windows_im2patch`` + ``sgc_windows_wgrad_patch`` (16 patch rows per listed window, read at own pixel + tap) against
    ``sgc_windows_im2col`` + ``sgc_windows_wgrad`` (36 column rows): the same products in the same order of accumulation."""
    L, lib = _lib()
    rng = np.random.default_rng(6)
    pl = _plan(rng, n=8)
    P, E = pl["P"], int(pl["incl"][-1])
    Epad = (E + 63) // 64 * 64
    gn = pl["incl"][P - 1:]
    g = pl["gather"][:E].long()
    gen = torch.Generator(device=DEV).manual_seed(4)
    z = torch.randn(P, 18, 18, 512, device=DEV, generator=gen).bfloat16()
    dy3x = torch.zeros(Epad * 4, 1024, device=DEV)
    dy3x[:E * 4] = torch.randn(E * 4, 1024, device=DEV, generator=gen)
    dy3x = dy3x.bfloat16()
    zpatch = torch.full((Epad, 16, 512), float("nan"), device=DEV).bfloat16()
    L.check(lib.sgc_windows_im2patch(L.ptr(z), L.ptr(pl["gather"]), L.ptr(gn), Epad, L.ptr(zpatch), L.stream_ptr()), "im2patch")
    pair, w = (g >> 6), (g & 63)
    for py in range(4):
        for px in range(4):
            assert torch.equal(zpatch[:E, py * 4 + px], z[pair, 2 * (w >> 3) + py, 2 * (w & 7) + px])
    if Epad > E:
        assert float(zpatch[E:].float().abs().max()) == 0.0
    zcol = torch.empty(Epad * 4, 9 * 512, device=DEV).bfloat16()
    L.check(lib.sgc_windows_im2col(L.ptr(z), L.ptr(pl["gather"]), L.ptr(gn), Epad, L.ptr(zcol), L.stream_ptr()), "im2col")
    out = []
    for fn, B in ((lib.sgc_windows_wgrad, zcol), (lib.sgc_windows_wgrad_patch, zpatch)):
        slabs = torch.full((64, 1024, 9 * 512), float("nan"), device=DEV)
        n = ctypes.c_int(0)
        L.check(fn(L.ptr(dy3x), L.ptr(B), L.ptr(slabs), Epad * 4, 0, ctypes.byref(n), L.stream_ptr()), "wgrad")
        out.append(slabs[:n.value].sum(0))
        assert n.value >= 1
    assert torch.equal(out[0], out[1])
    ref = dy3x.float().t() @ zcol.float()
    assert float((out[1] - ref).abs().max()) <= 2e-3 * float(ref.abs().max())

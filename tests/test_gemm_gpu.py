"""GPU tests of the two MFMA GEMM engines (csrc/gemm_nt.h, csrc/gemm_tn.h) through the debug C-ABI hooks.
Reference = torch fp32 matmul of the same 16-bit-rounded operands; tolerance covers f32 accumulation order."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from scene_graph_commonsense_amd import _lib
    return _lib.load(), _lib


def _rand(shape, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1).to(dtype).cuda()


def _window_major_index(S):
    """row m = 4*W + q  ->  (y, x) of the SxS image."""
    m = torch.arange(S * S)
    W, q = m // 4, m % 4
    py, px = W // (S // 2), W % (S // 2)
    return 2 * py + q // 2, 2 * px + q % 2


@pytest.mark.parametrize("dtype,elem", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (1000, 512, 4096), (256, 256, 64), (300, 256, 128),
                                   (520, 512, 256), (700, 256, 320), (512, 768, 1088),
                                   (9700, 8192, 128)])      # 38 x 32 tiles: the patch-aligned XCD walk with a partial band
def test_gemm_nt_plain(dtype, elem, M, N, K):
    lib, L = _lib()
    A, B = _rand((M, K), dtype, 1), _rand((N, K), dtype, 2)
    bias = _rand((N,), torch.float32, 3)
    C = torch.empty(M, N, dtype=dtype, device="cuda")
    st = lib.sgc_dbg_gemm_nt(elem, L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, ctypes.c_long(K), ctypes.c_long(K),
                             ctypes.c_long(N), L.ptr(bias), L.stream_ptr())
    assert st == 0
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t() + bias
    err = (C.float() - ref).abs().max().item()
    tol = (2e-3 if elem == 0 else 1.6e-2) * ref.abs().max().item()
    assert err <= tol, (err, tol)


def _conv_inputs(n_img, S, Cin, N, dtype):
    x = _rand((n_img, Cin, S, S), dtype, 5)
    w = (_rand((N, Cin, 3, 3), dtype, 6).float() * 0.1).to(dtype)
    xp = torch.zeros(n_img, S + 2, S + 2, Cin, dtype=dtype, device="cuda")
    xp[:, 1:-1, 1:-1, :] = x.permute(0, 2, 3, 1)
    wr = w.reshape(N, Cin // 64, 64, 9).permute(0, 1, 3, 2).contiguous().reshape(N, 9 * Cin)   # K = (c/64, tap, c%64)
    return x, w, xp.contiguous(), wr


@pytest.mark.parametrize("dtype,elem", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("n_img,lgS,Cin,N", [(3, 4, 128, 128), (2, 5, 128, 256), (5, 4, 512, 128), (3, 4, 128, 256),
                                             (2, 4, 512, 512), (2, 4, 1024, 256), (9, 4, 64, 256), (11, 4, 192, 512)])
def test_conv_nt(dtype, elem, n_img, lgS, Cin, N):
    lib, L = _lib()
    S = 1 << lgS
    x, w, xp, wr = _conv_inputs(n_img, S, Cin, N, dtype)
    bias = _rand((N,), torch.float32, 7)
    C = torch.empty(n_img * S * S, N, dtype=dtype, device="cuda")
    st = lib.sgc_dbg_conv_nt(elem, L.ptr(xp), L.ptr(wr), L.ptr(C), n_img, lgS, Cin, N, L.ptr(bias), L.stream_ptr())
    assert st == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float(), w.float(), bias, padding=1)      # [n,N,S,S]
    yy, xx = _window_major_index(S)
    ref_rows = ref[:, :, yy, xx].permute(0, 2, 1).reshape(n_img * S * S, N)
    err = (C.float() - ref_rows).abs().max().item()
    tol = (2e-3 if elem == 0 else 1.6e-2) * ref_rows.abs().max().item()
    assert err <= tol, (err, tol)


def test_tr_probe_layout():
    """Document what ds_read_b64_tr_b16 returns for the linear address pattern the TN kernel uses."""
    lib, L = _lib()
    lane = torch.arange(64)
    t, grp = lane % 16, lane // 16
    addr = (grp * 128 + (t // 4) * 32 + (t % 4) * 8).int().cuda()
    out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
    assert lib.sgc_dbg_tr_probe(L.ptr(addr), L.ptr(out), L.stream_ptr()) == 0
    torch.cuda.synchronize()
    got = out.view(64, 4).cpu()
    # expectation: lane t of group g receives column t of the 4x16 block g: elements g*64 + j*16 + t
    exp = torch.stack([grp * 64 + j * 16 + t for j in range(4)], dim=1).to(torch.int16)
    print("tr probe lanes 0..3, 16..17:", got[:4].tolist(), got[16:18].tolist())
    assert torch.equal(got, exp), got[:20].tolist()


@pytest.mark.parametrize("dtype,elem", [(torch.float16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 64, 1), (256, 384, 640, 3), (512, 128, 4096, 8), (512, 512, 4096, 8),
                                          (1024, 768, 2048, 16), (256, 256, 64, 1), (256, 512, 192, 1), (512, 256, 1088, 3), (512, 1024, 1088, 3),
                                          (1024, 2048, 640, 2)])
def test_gemm_tn_plain(dtype, elem, M, N, K, splits):
    lib, L = _lib()
    A, B = _rand((K, M), dtype, 11), _rand((K, N), dtype, 12)
    C = torch.zeros(splits, M, N, dtype=torch.float32, device="cuda")
    slabs = ctypes.c_int(0)
    st = lib.sgc_dbg_gemm_tn(elem, L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, ctypes.c_long(M), ctypes.c_long(N), splits,
                             ctypes.byref(slabs), L.stream_ptr())
    assert st == 0
    torch.cuda.synchronize()
    got = C[:slabs.value].sum(0)
    ref = A.float().t() @ B.float()
    err = (got - ref).abs().max().item()
    assert err <= 1e-3 * ref.abs().max().item() + 1e-4, err


@pytest.mark.parametrize("n_img,lgS,Cin,M,splits", [(2, 4, 128, 128, 2), (3, 5, 128, 256, 4), (2, 4, 512, 128, 1),
                                                    (8, 4, 512, 1024, 8), (16, 4, 256, 512, 16), (3, 4, 256, 256, 1),
                                                    (5, 4, 512, 256, 3), (4, 5, 128, 512, 5), (3, 4, 128, 256, 2), (2, 5, 384, 256, 3)])
def test_conv_tn(n_img, lgS, Cin, M, splits):
    lib, L = _lib()
    dtype, elem = torch.bfloat16, 1
    S = 1 << lgS
    x, _, xp, _ = _conv_inputs(n_img, S, Cin, 128, dtype)
    dy = _rand((n_img, M, S, S), dtype, 21)                       # grad wrt conv output [n, M, S, S]
    yy, xx = _window_major_index(S)
    dy_rows = dy[:, :, yy, xx].permute(0, 2, 1).reshape(n_img * S * S, M).contiguous()
    C = torch.zeros(splits, M, 9 * Cin, dtype=torch.float32, device="cuda")
    slabs = ctypes.c_int(0)
    st = lib.sgc_dbg_conv_tn(elem, L.ptr(dy_rows), L.ptr(xp), L.ptr(C), M, n_img, lgS, Cin, splits,
                             ctypes.byref(slabs), L.stream_ptr())
    assert st == 0
    torch.cuda.synchronize()
    got = C[:slabs.value].sum(0).view(M, 3, 3, Cin).permute(0, 3, 1, 2)       # -> [M, Cin, 3, 3]
    ref = torch.nn.grad.conv2d_weight(x.float(), (M, Cin, 3, 3), dy.float(), padding=1)
    err = (got - ref).abs().max().item()
    assert err <= 1e-3 * ref.abs().max().item() + 1e-4, err


def test_pingpong_blocks_repeatable():
    """Race screen for the counted-vmcnt ping-pong blocks (csrc/gemm_nt_pp.h hazard rules): the same launch repeated under
    memory load must give bit-identical results every time and match the f32 reference."""
    lib, L = _lib()
    dtype, elem = torch.bfloat16, 1
    M, N, K = 2048, 1024, 4096
    A, B = _rand((M, K), dtype, 31), _rand((N, K), dtype, 32)
    ref = A.float() @ B.float().t()
    noise = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    first = None
    for it in range(12):
        C = torch.empty(M, N, dtype=dtype, device="cuda")
        noise.normal_()                                            # evict L2 / perturb timing between launches
        assert lib.sgc_dbg_gemm_nt(elem, L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, ctypes.c_long(K), ctypes.c_long(K),
                                   ctypes.c_long(N), None, L.stream_ptr()) == 0
        torch.cuda.synchronize()
        if first is None:
            first = C.clone()
            assert (C.float() - ref).abs().max().item() <= 1.6e-2 * ref.abs().max().item()
        else:
            assert torch.equal(C, first), "ping-pong NT block is not repeatable (iteration %d)" % it
    # TN engine, 2 K splits
    At, Bt = _rand((K, 512), dtype, 33), _rand((K, 1024), dtype, 34)
    reft = At.float().t() @ Bt.float()
    first = None
    for it in range(12):
        Cs = torch.zeros(2, 512, 1024, dtype=torch.float32, device="cuda")
        slabs = ctypes.c_int(0)
        noise.normal_()
        assert lib.sgc_dbg_gemm_tn(elem, L.ptr(At), L.ptr(Bt), L.ptr(Cs), 512, 1024, K, ctypes.c_long(512), ctypes.c_long(1024), 2,
                                   ctypes.byref(slabs), L.stream_ptr()) == 0
        torch.cuda.synchronize()
        got = Cs[:slabs.value].sum(0)
        if first is None:
            first = got.clone()
            assert (got - reft).abs().max().item() <= 1e-3 * reft.abs().max().item() + 1e-4
        else:
            assert torch.equal(got, first), "ping-pong TN block is not repeatable (iteration %d)" % it
    # halo conv
    n_img, Cin, Nc = 64, 512, 512
    x, w, xp, wr = _conv_inputs(n_img, 16, Cin, Nc, dtype)
    first = None
    for it in range(12):
        C = torch.empty(n_img * 256, Nc, dtype=dtype, device="cuda")
        noise.normal_()
        assert lib.sgc_dbg_conv_nt(elem, L.ptr(xp), L.ptr(wr), L.ptr(C), n_img, 4, Cin, Nc, None, L.stream_ptr()) == 0
        torch.cuda.synchronize()
        if first is None:
            first = C.clone()
        else:
            assert torch.equal(C, first), "ping-pong halo conv block is not repeatable (iteration %d)" % it


@pytest.mark.parametrize("n_pairs,splits", [(8, 1), (40, 0), (37, 3)])
def test_conv3_wgrad_sparse_matches_dense(n_pairs, splits):
    """Sparse-MFMA conv3 weight gradient (pooled gradient + arg-max byte as the 2:4 operand) against the dense path
    (un-pool, then the ping-pong TN block) on the same data: same products, different summation order."""
    lib, L = _lib()
    g = torch.Generator().manual_seed(n_pairs)
    P = n_pairs
    dy = (torch.randn(P * 64, 1024, generator=g) * 0.5).bfloat16().cuda()
    am = torch.randint(0, 5, (P * 64, 1024), generator=g, dtype=torch.uint8).cuda()
    z = torch.zeros(P, 18, 18, 512, dtype=torch.bfloat16)
    z[:, 1:17, 1:17] = (torch.randn(P, 16, 16, 512, generator=g) * 0.5).bfloat16()
    z = z.cuda()
    # dense reference path: un-pool into dy3_pad, then the dense weight gradient
    dy3 = torch.zeros(P, 18, 18, 1024, dtype=torch.bfloat16, device="cuda")
    bpart = torch.zeros(2048, 1024, device="cuda")
    nparts = ctypes.c_int(0)
    assert lib.sgc_unpool_relu_bwd(L.ptr(dy), L.ptr(am), L.ptr(dy3), L.ptr(bpart), ctypes.byref(nparts), P, L.stream_ptr()) == 0
    sl = torch.zeros(32, 1024, 4608, device="cuda")
    ns = ctypes.c_int(0)
    assert lib.sgc_conv3_wgrad(L.ptr(dy3), L.ptr(z), L.ptr(sl), P, 4, ctypes.byref(ns), L.stream_ptr()) == 0
    torch.cuda.synchronize()
    ref = sl[:ns.value].sum(0)
    ac = torch.empty(P * 4 * 1024 * 64, dtype=torch.uint8, device="cuda")
    ic = torch.empty(P * 4 * 1024 * 8, dtype=torch.uint8, device="cuda")
    sl2 = torch.zeros(32, 1024, 4608, device="cuda")
    ns2 = ctypes.c_int(0)
    assert lib.sgc_conv3_wgrad_sparse(L.ptr(dy), L.ptr(am), L.ptr(z), L.ptr(ac), L.ptr(ic), L.ptr(sl2), P, splits, ctypes.byref(ns2),
                                      L.stream_ptr()) == 0
    torch.cuda.synchronize()
    got = sl2[:ns2.value].sum(0)
    err = (got - ref).abs().max().item()
    assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (err, ref.abs().max().item())
    # fused un-pool + pack pass: identical dy3_pad and packed operand (bit-exact), bias partial sums to f32 rounding
    dy3b = torch.zeros(P, 18, 18, 1024, dtype=torch.bfloat16, device="cuda")
    bpart2 = torch.zeros(2048, 1024, device="cuda")
    nparts2 = ctypes.c_int(0)
    ac2 = torch.empty_like(ac)
    ic2 = torch.empty_like(ic)
    assert lib.sgc_unpool_relu_bwd_pack(L.ptr(dy), L.ptr(am), L.ptr(dy3b), L.ptr(bpart2), ctypes.byref(nparts2), L.ptr(ac2),
                                        L.ptr(ic2), P, L.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dy3b.view(torch.int16), dy3.view(torch.int16))
    assert torch.equal(ac2, ac) and torch.equal(ic2, ic)
    b_ref = bpart[:nparts.value].sum(0)
    b_got = bpart2[:nparts2.value].sum(0)
    assert (b_got - b_ref).abs().max().item() <= 1e-4 * b_ref.abs().max().item() + 1e-5
    sl3 = torch.zeros(32, 1024, 4608, device="cuda")
    ns3 = ctypes.c_int(0)
    assert lib.sgc_conv3_wgrad_sparse(None, None, L.ptr(z), L.ptr(ac2), L.ptr(ic2), L.ptr(sl3), P, splits, ctypes.byref(ns3),
                                      L.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert ns3.value == ns2.value and torch.equal(sl3[:ns3.value], sl2[:ns2.value])


@pytest.mark.parametrize("P", [1, 5, 160, 17000])          # 17000 images: byte offsets beyond 2^31 (base pointers must be 64-bit)
def test_conv3_dgrad_with_fused_unpool_equals_materialised_unpool(P):
    """``sgc_conv3_dgrad_pooled`` (the block un-pools the pooled gradient + routing byte into its LDS patch) against the two-pass
    form (``sgc_unpool_relu_bwd_pack`` writes the un-pooled tensor, ``sgc_conv3_dgrad`` reads it): the operand values and the
    accumulation order are the same, so the results must be bit-identical whenever both run the halo block (P >= 128), and
    equal to f32 summation order otherwise (small P takes the generic block in the two-pass form)."""
    import ctypes
    from scene_graph_commonsense_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(P)
    dy = (torch.randn(P * 64, 1024, generator=g) * 0.5).bfloat16().cuda()
    am = torch.randint(0, 5, (P * 64, 1024), generator=g, dtype=torch.uint8).cuda()              # 0..3 position, 4 = killed by the ReLU
    wd3 = (torch.randn(512, 9 * 1024, generator=g) * 0.02).bfloat16().cuda()
    dy3 = torch.zeros(P, 18, 18, 1024, dtype=torch.bfloat16, device="cuda")
    bpart = torch.zeros(2048, 1024, device="cuda")
    pack_a = torch.empty(P * 4 * 1024 * 64, dtype=torch.uint8, device="cuda")
    pack_i = torch.empty(P * 4 * 1024 * 8, dtype=torch.uint8, device="cuda")
    nparts = ctypes.c_int(0)
    st = _lib.stream_ptr()
    _lib.check(lib.sgc_unpool_relu_bwd_pack(_lib.ptr(dy), _lib.ptr(am), _lib.ptr(dy3), _lib.ptr(bpart), ctypes.byref(nparts), _lib.ptr(pack_a),
                                            _lib.ptr(pack_i), P, st), "unpool")
    # the un-pooled tensor is what the routing says: value at the coded pixel of every window, zero elsewhere
    if P <= 200:
        ref3 = torch.zeros(P, 16, 16, 1024)
        dyc, amc = dy.float().cpu().view(P, 8, 8, 1024), am.cpu().view(P, 8, 8, 1024)
        for q in range(4):
            ref3[:, (q >> 1)::2, (q & 1)::2] = torch.where(amc == q, dyc, torch.zeros(()))
        assert torch.equal(dy3[:, 1:17, 1:17].float().cpu(), ref3)
    dz_a = torch.empty(P * 256, 512, dtype=torch.bfloat16, device="cuda")
    dz_b = torch.full((P * 256, 512), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.sgc_conv3_dgrad(_lib.ptr(dy3), _lib.ptr(wd3), _lib.ptr(dz_a), P, st), "dgrad")
    _lib.check(lib.sgc_conv3_dgrad_pooled(_lib.ptr(dy), _lib.ptr(am), _lib.ptr(wd3), _lib.ptr(dz_b), P, st), "dgrad pooled")
    # a second pass with the packed operand only (dy3 == NULL) leaves the same packed bytes
    pa2, pi2 = torch.empty_like(pack_a), torch.empty_like(pack_i)
    _lib.check(lib.sgc_unpool_relu_bwd_pack(_lib.ptr(dy), _lib.ptr(am), None, _lib.ptr(bpart), ctypes.byref(nparts), _lib.ptr(pa2), _lib.ptr(pi2), P, st),
               "unpool, pack only")
    torch.cuda.synchronize()
    assert torch.equal(pa2, pack_a) and torch.equal(pi2, pack_i)
    assert torch.isfinite(dz_b.float()).all()
    if P >= 128:
        assert torch.equal(dz_a, dz_b)
    else:
        err = (dz_a.float() - dz_b.float()).abs().max() / dz_a.float().abs().max()
        assert float(err) <= 1e-2                               # bf16 outputs of two accumulation orders: at most a rounding step apart


@pytest.mark.parametrize("oiv6", [False, True])
def test_weight_layouts_by_gather_kernels_equal_the_torch_chains(oiv6):
    """Every 16-bit compute copy of the conv1 / conv2 / conv3 / fc2 weights built by ``sgc_permute_cast`` / ``sgc_segment_cast``
    (``TUNING.weight_kernels``, the default: one launch per layout) against the torch view / permute / flip / cat / cast chains that
    define the layouts (rounds 1-3), bit for bit - including the flipped-tap data-gradient forms and the stacked tap matrices of the
    patch form."""
    from scene_graph_commonsense_amd.engine import RelHeadEngine, tuning
    from scene_graph_commonsense_amd.synthetic import HeadConfig, make_state_dict
    cfg = HeadConfig(dataset="oiv6", num_classes=601, num_super_classes=0, num_geometric=4, num_possessive=2, num_semantic=24) if oiv6 else HeadConfig()
    sd = {k: v.cuda() for k, v in make_state_dict(cfg, seed=17, head_gain=3.0).items()}
    got = {}
    for kern in (True, False):
        eng = RelHeadEngine(cfg, "cuda:0")
        with tuning(weight_kernels=kern):
            eng.load_weights(sd)
            eng.prep_bwd_weights(sd)
        torch.cuda.synchronize()
        got[kern] = {k: eng.w[k].clone() for k in ("w1r", "w2r", "w3r", "w2m", "w2mT", "wd3", "w3col", "w3patch", "wd2", "b1", "cst", "b2", "b3")}
    for k, a in got[True].items():
        b = got[False][k]
        assert a.dtype == b.dtype and a.numel() == b.numel(), k
        assert torch.equal(a.reshape(-1).view(torch.int16 if a.element_size() == 2 else torch.int32),
                           b.reshape(-1).view(torch.int16 if b.element_size() == 2 else torch.int32)), k

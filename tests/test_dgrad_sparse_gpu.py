"""conv3 data gradient over listed windows on the sparse matrix cores (``csrc/kernels_dgrad_sp.hip``: ``sgc_windows_dgrad_patches_sparse``)
against (a) the float64 definition - the backward of ``/root/reference/model.py:145-147`` restricted to a window's 4 x 4 input patch - and
(b) the dense patch form it replaces (``sgc_windows_unpool`` + ``sgc_windows_dgrad_patches``): same products, the structural zeros not
issued, so the two agree to the bf16 rounding of f32 sums taken in another order."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _slots():
    """[(pp, [(own pixel q, tap)])] in the order of the 20 output slots (centre pixels: two slots of two combinations)."""
    opts = lambda c: [(0, 0)] if c == 0 else ([(1, 2)] if c == 3 else [(0, c), (1, c - 1)])
    out = []
    for py in range(4):
        for px in range(4):
            combos = [(qy * 2 + qx, ky * 3 + kx) for qy, ky in opts(py) for qx, kx in opts(px)]
            if len(combos) == 4:
                out.append((py * 4 + px, combos[:2]))
                out.append((py * 4 + px, combos[2:]))
            else:
                out.append((py * 4 + px, combos))
    assert len(out) == 20
    return out


def _w3patch(w3):
    """The dense form's weight operand (engine._prep_bwd_trunk_weights_torch)."""
    opts = lambda c: [(0, 0)] if c == 0 else ([(1, 2)] if c == 3 else [(0, c), (1, c - 1)])
    return torch.cat([torch.cat([w3[:, :, ky, kx].t() for _, ky in opts(py) for _, kx in opts(px)], dim=1).reshape(-1)
                      for py in range(4) for px in range(4)]).to(torch.bfloat16).contiguous()


@pytest.mark.parametrize("n_sparse,n_rows", [(256, 300), (768, 1000)])
def test_sparse_patch_dgrad_matches_the_definition_and_the_dense_form(n_sparse, n_rows):
    from scene_graph_commonsense_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(7 + n_sparse)
    st = _lib.stream_ptr
    E = n_sparse
    dy = (torch.randn(n_rows, 1024, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    am = torch.randint(0, 5, (n_rows, 1024), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)        # 4 = killed by the ReLU
    am[:3] = 4                                                                                                  # whole windows without a route
    am[3] = 2
    gather = torch.randperm(n_rows, device=dev, generator=g)[:E].to(torch.int32).contiguous()
    dest = torch.randperm(n_rows, device=dev, generator=g)[:E].to(torch.int32).contiguous()
    w3 = (torch.randn(1024, 512, 3, 3, device=dev, generator=g) * 0.02).contiguous()
    # ---- definition in float64 (operands as the kernels see them: bf16 values)
    dyf = dy.double()[dest.long()]
    code = am[gather.long()].long()
    w3b = w3.to(torch.bfloat16).double()
    ref = torch.zeros(E, 20, 512, dtype=torch.float64, device=dev)
    for s, (pp, combos) in enumerate(_slots()):
        for q, tap in combos:
            ref[:, s] += (dyf * (code == q)) @ w3b[:, :, tap // 3, tap % 3]
    bias_ref = (dyf * (code < 4)).sum(0)
    # ---- sparse form
    w3sp = torch.empty(20 * 512 * 2048, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.sgc_windows_dgrad_sparse_weights(_lib.ptr(w3), _lib.ptr(w3sp), st()), "weights")
    pack_a = torch.empty(4 * E * 1024, dtype=torch.bfloat16, device=dev)
    pack_i = torch.empty(4 * E * 64, dtype=torch.int32, device=dev)
    patch16 = torch.full((E, 16, 512), float("nan"), dtype=torch.bfloat16, device=dev)      # one row per patch pixel: centre halves summed
    bpart = torch.zeros(1024, 1024, dtype=torch.float32, device=dev)
    nparts = ctypes.c_int(0)
    _lib.check(lib.sgc_windows_dgrad_sparse_pack(_lib.ptr(dy), _lib.ptr(am), _lib.ptr(gather), _lib.ptr(dest), E, _lib.ptr(pack_a), _lib.ptr(pack_i),
                                                 _lib.ptr(bpart), ctypes.byref(nparts), st()), "sparse pack")
    _lib.check(lib.sgc_windows_dgrad_patches_sparse(_lib.ptr(pack_a), _lib.ptr(pack_i), E, _lib.ptr(w3sp), _lib.ptr(patch16), st()), "sparse dgrad")
    torch.cuda.synchronize()
    assert nparts.value > 0
    bias = bpart[:nparts.value].double().sum(0)
    assert float((bias - bias_ref).abs().max()) <= 1e-3 * float(bias_ref.abs().max())
    pp_of = [pp for pp, _ in _slots()]
    ref16 = torch.zeros(E, 16, 512, dtype=torch.float64, device=dev)
    for s_, pp in enumerate(pp_of):
        ref16[:, pp] += ref[:, s_]
    scale = float(ref16.abs().max())
    err = float((patch16.double() - ref16).abs().max())
    assert torch.isfinite(patch16.float()).all()
    assert err <= 2.0 ** -8 * scale, (err, scale)                       # bf16 rounding of the f32 sums (2^-9 relative to the value)
    # ---- dense form on the same inputs
    Epad = E
    gn = torch.tensor([E], dtype=torch.int32, device=dev)
    dy3x = torch.empty(Epad * 4 * 1024, dtype=torch.bfloat16, device=dev)
    bpart_d = torch.zeros(1024, 1024, dtype=torch.float32, device=dev)
    nparts_d = ctypes.c_int(0)
    _lib.check(lib.sgc_windows_unpool(_lib.ptr(dy), _lib.ptr(am), _lib.ptr(gather), _lib.ptr(gn), _lib.ptr(dest), Epad, _lib.ptr(dy3x), _lib.ptr(bpart_d),
                                      ctypes.byref(nparts_d), st()), "unpool")
    patch_d = torch.empty(E, 20, 512, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.sgc_windows_dgrad_patches(_lib.ptr(dy3x), _lib.ptr(_w3patch(w3)), _lib.ptr(patch_d), E, st()), "dense dgrad")
    torch.cuda.synchronize()
    single = [s_ for s_, pp in enumerate(pp_of) if pp_of.count(pp) == 1]                    # the 12 pixels both forms keep in one row
    d = (patch16[:, [pp_of[s_] for s_ in single]].float() - patch_d[:, single].float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * scale                           # two bf16 roundings apart at most
    assert float((d > 0).float().mean()) < 0.2                           # and mostly the same bits
    dense16 = torch.zeros(E, 16, 512, dtype=torch.float32, device=dev)
    for s_, pp in enumerate(pp_of):
        dense16[:, pp] += patch_d[:, s_].float()
    assert float((patch16.float() - dense16).abs().max()) <= 2.0 ** -6 * scale             # centre pixels: one rounding here, two + a sum there
    # partial un-pool (entries behind the sparse ones): rows and bias partials of the tail equal the full pass's
    e0 = E // 2
    dy3x_t = torch.empty((E - e0) * 4 * 1024, dtype=torch.bfloat16, device=dev)
    bpart_t = torch.zeros(1024, 1024, dtype=torch.float32, device=dev)
    nparts_t = ctypes.c_int(0)
    _lib.check(lib.sgc_windows_unpool_from(_lib.ptr(dy), _lib.ptr(am), _lib.ptr(gather), _lib.ptr(gn), _lib.ptr(dest), e0, E - e0, _lib.ptr(dy3x_t),
                                           _lib.ptr(bpart_t), ctypes.byref(nparts_t), st()), "unpool_from")
    torch.cuda.synchronize()
    assert torch.equal(dy3x_t, dy3x[e0 * 4096:])
    tail_ref = (dyf[e0:] * (code[e0:] < 4)).sum(0)
    assert float((bpart_t[:nparts_t.value].double().sum(0) - tail_ref).abs().max()) <= 1e-3 * float(bias_ref.abs().max())


def test_sparse_patch_dgrad_beyond_two_to_the_twenty_windows():
    """the packed rows of a list with more than 2^20 windows lie beyond the 2 GiB a buffer descriptor based at the set's
    first row could address (the loads returned zeros: a silent zero gradient).  The tile's rows are now addressed from the tile's own
    base; checked on the LAST M tile of a 1 049 088-window list (an OpenImages 4 x 100 minibatch with wide boxes gets there) against the
    float64 definition.  Few distinct source rows, gathered many times: the list is long, the inputs small."""
    from scene_graph_commonsense_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    st = _lib.stream_ptr
    E, n_rows = (1 << 20) + 512, 4096
    dy = (torch.randn(n_rows, 1024, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    am = torch.randint(0, 5, (n_rows, 1024), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    gather = torch.randint(0, n_rows, (E,), device=dev, generator=g, dtype=torch.int32)
    dest = torch.randint(0, n_rows, (E,), device=dev, generator=g, dtype=torch.int32)
    w3 = (torch.randn(1024, 512, 3, 3, device=dev, generator=g) * 0.02).contiguous()
    w3sp = torch.empty(20 * 512 * 2048, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.sgc_windows_dgrad_sparse_weights(_lib.ptr(w3), _lib.ptr(w3sp), st()), "weights")
    pack_a = torch.empty(4 * E * 1024, dtype=torch.bfloat16, device=dev)                    # 8.6 GB
    pack_i = torch.empty(4 * E * 64, dtype=torch.int32, device=dev)
    patch16 = torch.full((E, 16, 512), float("nan"), dtype=torch.bfloat16, device=dev)      # 17 GB
    _lib.check(lib.sgc_windows_dgrad_sparse_pack(_lib.ptr(dy), _lib.ptr(am), _lib.ptr(gather), _lib.ptr(dest), E, _lib.ptr(pack_a), _lib.ptr(pack_i),
                                                 None, None, st()), "sparse pack")
    _lib.check(lib.sgc_windows_dgrad_patches_sparse(_lib.ptr(pack_a), _lib.ptr(pack_i), E, _lib.ptr(w3sp), _lib.ptr(patch16), st()), "sparse dgrad")
    torch.cuda.synchronize()
    w3b = w3.to(torch.bfloat16).double()
    pp_of = [pp for pp, _ in _slots()]
    for e0 in (0, (1 << 20) - 256, E - 256):                    # first tile, the last one below the old limit, the last one
        sl = slice(e0, e0 + 256)
        dyf = dy.double()[dest[sl].long()]
        code = am[gather[sl].long()].long()
        ref16 = torch.zeros(256, 16, 512, dtype=torch.float64, device=dev)
        for s_, (pp, combos) in enumerate(_slots()):
            for q, tap in combos:
                ref16[:, pp] += (dyf * (code == q)) @ w3b[:, :, tap // 3, tap % 3]
        scale = float(ref16.abs().max())
        got = patch16[sl].double()
        assert torch.isfinite(got).all()
        assert scale > 0 and float((got - ref16).abs().max()) <= 2.0 ** -8 * scale, e0

"""Weight-gradient GEMMs of the exact (f32) tier against float64, GEMM by GEMM (-m gpu).

The backward of every nn.Linear the decoder evaluates (decoder.py:291-349, 109-134) is one GEMM over the sample points:
dW[out, in] = sum_p dy[out, p] * act[in, p], db[out] = sum_p dy[out, p].  The library runs them from two tile-major arrays
(dy_T, act_T: [tile][rows][32 points]) according to a plan (dfn_wgrad_plan) - here every GEMM of that plan, every bias row sum and
the scatter into the flat parameter vector are recomputed in float64 from random arrays, for point counts that fill the slices
of the points unevenly, leave slices empty, or are the training step's own."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plan(lib, field, what):
    n = lib.dfn_wgrad_plan(field, what, None, 0)
    assert n > 0, lib.dfn_last_error()
    out = np.empty(n, dtype=np.int32)
    assert lib.dfn_wgrad_plan(field, what, out.ctypes.data_as(C.POINTER(C.c_int32)), n) == n
    return out


@pytest.mark.parametrize("field", [0, 1])
@pytest.mark.parametrize("NP", [32 * 5, 32 * 37, 32 * 1027, 2048 * 64])
def test_f32_weight_and_bias_gradients_vs_float64(field, NP):
    from dfanerf import engine
    from dfanerf._lib import lib, check
    engine.require_gpu()
    dev = torch.device("cuda")
    rows_a, rows_g = check(lib.dfn_train_rows(field, 0), "rows"), check(lib.dfn_train_rows(field, 1), "rows")
    T = NP // 32
    g = torch.Generator(device="cpu").manual_seed(1000 * field + NP)
    dy = torch.randn(T, rows_g, 32, generator=g).to(dev)
    act = torch.randn(T, rows_a, 32, generator=g).to(dev)
    n_par = 955242
    nb = check(lib.dfn_bias_floats(0, field), "bias")
    ws = torch.empty(check(lib.dfn_train_rows(field, 3), "ws"), dtype=torch.float32, device=dev)
    ws.fill_(float("nan"))                       # a slice the reduction must not read stays poisoned
    grad = torch.zeros(n_par, dtype=torch.float32, device=dev)
    dbias = torch.full((nb,), float("nan"), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.dfn_weight_bias_grad(0, field, p(dy), p(act), NP, p(ws), p(grad), p(dbias), st), "dfn_weight_bias_grad")
    torch.cuda.synchronize()

    ops = _plan(lib, field, 0).reshape(-1, 6)
    cmap = torch.from_numpy(_plan(lib, field, 1)).to(dev).long()
    bias_rows = torch.from_numpy(_plan(lib, field, 2)).to(dev).long()
    assert bias_rows.numel() == nb
    dy64, act64 = dy.double(), act.double()
    ref = torch.zeros(n_par, dtype=torch.float64, device=dev)
    scale = torch.zeros(n_par, dtype=torch.float64, device=dev)       # sum of |products|: what the rounding error is relative to
    for a_row, M, b_row, N, c_off, _ in ops.tolist():
        if N == 0:
            continue
        A, B = dy64[:, a_row:a_row + M, :], act64[:, b_row:b_row + N, :]
        Cm = torch.einsum("tmp,tnp->mn", A, B).reshape(-1)
        Sm = torch.einsum("tmp,tnp->mn", A.abs(), B.abs()).reshape(-1)
        idx = cmap[c_off:c_off + M * N]
        ok = idx >= 0
        ref.index_add_(0, idx[ok], Cm[ok])
        scale.index_add_(0, idx[ok], Sm[ok])
    err = (grad.double() - ref).abs()
    touched = scale > 0
    assert int(touched.sum()) > 500000
    # f32 accumulation of NP products: a few ulps of the sum of magnitudes, growing like sqrt(steps)
    tol = 2e-6 * scale + 1e-6
    worst = float((err / tol).max())
    assert worst < 1.0, f"field {field}, NP {NP}: worst |dW error| / tolerance = {worst:.2f}"
    assert float(grad[~touched].abs().max()) == 0.0          # nothing outside the map is written
    has = bias_rows >= 0
    rows64 = dy64.permute(1, 0, 2).reshape(rows_g, -1)
    ref_b = torch.zeros(nb, dtype=torch.float64, device=dev)
    ref_b[has] = rows64[bias_rows[has]].sum(1)
    sc_b = torch.zeros(nb, dtype=torch.float64, device=dev)
    sc_b[has] = rows64[bias_rows[has]].abs().sum(1)
    assert bool(torch.isfinite(dbias).all())
    worst_b = float(((dbias.double() - ref_b).abs() / (2e-6 * sc_b + 1e-6)).max())
    assert worst_b < 1.0, f"field {field}, NP {NP}: worst |db error| / tolerance = {worst_b:.2f}"
    assert float(dbias[~has].abs().max() if bool((~has).any()) else 0.0) == 0.0

    # the two stages apart (what the training step calls) = the one call, bit for bit; and a second run reproduces them
    grad2 = torch.zeros_like(grad)
    dbias2 = torch.full_like(dbias, float("nan"))
    ws.fill_(float("nan"))
    check(lib.dfn_weight_bias_grad_partials(0, field, 0, p(dy), p(act), NP, p(ws), p(dbias2), st), "partials")
    check(lib.dfn_weight_bias_grad_reduce(0, field, NP, p(ws), p(grad2), p(dbias2), st), "reduce")
    torch.cuda.synchronize()
    assert torch.equal(grad2, grad) and torch.equal(dbias2, dbias)

    # ... and the GEMM stage in its two launches, the narrow one first and on a stream of its own (what the f32 training step does with
    # its last field): the launches write disjoint pieces of the workspace
    grad3 = torch.zeros_like(grad)
    dbias3 = torch.full_like(dbias, float("nan"))
    ws.fill_(float("nan"))
    torch.cuda.synchronize()
    other = torch.cuda.Stream()
    check(lib.dfn_weight_bias_grad_partials_part(0, field, 0, p(dy), p(act), NP, p(ws), p(dbias3), 2, C.c_void_p(other.cuda_stream)), "part 2")
    check(lib.dfn_weight_bias_grad_partials_part(0, field, 0, p(dy), p(act), NP, p(ws), p(dbias3), 1, st), "part 1")
    torch.cuda.current_stream().wait_stream(other)
    check(lib.dfn_weight_bias_grad_reduce(0, field, NP, p(ws), p(grad3), p(dbias3), st), "reduce")
    torch.cuda.synchronize()
    assert torch.equal(grad3, grad) and torch.equal(dbias3, dbias)
    # the selector is the f32 tier's: anything else is refused
    assert lib.dfn_weight_bias_grad_partials_part(0, field, 0, p(dy), p(act), NP, p(ws), p(dbias3), 0, st) != 0
    assert lib.dfn_weight_bias_grad_partials_part(1, field, 0, p(dy), p(act), NP, p(ws), p(dbias3), 1, st) != 0

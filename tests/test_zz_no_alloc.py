"""The allocation-free column sums (`rlgan_column_sums_noalloc`) against `rlgan_column_sums`: same kernels and
order of additions, so the outputs must agree bit for bit.  Two calls in a row on the library's fixed scratch,
the second smaller than the first (what the first left behind in the scratch must not leak into it)."""
import numpy as np
import pytest
import torch

from rl_gan_amd import _lib

pytestmark = pytest.mark.gpu


def _both(rows, cols, relu, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    dy = torch.randn(rows, cols, generator=g).cuda()
    y = torch.randn(rows, cols, generator=g).cuda() if relu else None
    yp = y.data_ptr() if relu else None
    part = torch.full((512 * cols,), float("nan"), device="cuda")
    ref = torch.full((cols,), float("nan"), device="cuda")
    got = torch.full((cols,), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib.rlgan_column_sums(dy.data_ptr(), yp, rows, cols, part.data_ptr(), ref.data_ptr(), st), "rlgan_column_sums")
    _lib.check(_lib.lib.rlgan_column_sums_noalloc(dy.data_ptr(), yp, rows, cols, got.data_ptr(), st), "rlgan_column_sums_noalloc")
    torch.cuda.synchronize()
    r, o = ref.cpu().numpy(), got.cpu().numpy()
    assert np.all(np.isfinite(r))
    np.testing.assert_array_equal(o, r)
    # and the sums are the sums (float64 on the host, loose: the order of additions differs)
    d = dy.double().cpu().numpy()
    if relu:
        d = d * (y.cpu().numpy() > 0)
    np.testing.assert_allclose(o, d.sum(0), rtol=0, atol=2e-3 * np.sqrt(rows))


def test_two_calls_in_a_row_second_smaller():
    _both(4096 * 361, 32, True, 1)      # the decoder's third layer at batch 4096: all 512 partial rows in use
    _both(1000, 7, False, 2)            # fewer rows, fewer columns, fewer partial rows


def test_widest_and_narrowest():
    _both(70000, 256, False, 3)
    _both(5, 1, True, 4)
    _both(513, 200, True, 5)


def test_beyond_the_scratch_is_refused():
    x = torch.zeros(8, device="cuda")
    rc = _lib.lib.rlgan_column_sums_noalloc(x.data_ptr(), None, 4, 257, x.data_ptr(), None)
    assert rc == -6
    assert b"n_cols <= 256" in _lib.lib.rlgan_error_string(rc)
    assert _lib.lib.rlgan_column_sums_noalloc(None, None, 4, 8, x.data_ptr(), None) == -1
    assert _lib.lib.rlgan_column_sums_noalloc(x.data_ptr(), None, 0, 8, x.data_ptr(), None) == -1


def test_recorded_in_a_graph_after_the_init_call():
    """The callers the entry point is for include code under HIP graph capture: after `rlgan_noalloc_init` (the symbol
    lookup, outside capture) the call is recorded, the graph replayed twice on fresh inputs, and every replay gives the
    bits of rlgan_column_sums."""
    _lib.check(_lib.lib.rlgan_noalloc_init(), "rlgan_noalloc_init")
    rows, cols = 30000, 64
    dy = torch.randn(rows, cols, device="cuda")
    y = torch.randn(rows, cols, device="cuda")
    part = torch.empty(512 * cols, device="cuda")
    ref = torch.empty(cols, device="cuda")
    got = torch.full((cols,), float("nan"), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):       # one launched call first, as for any kernel that is going to be recorded
        _lib.check(_lib.lib.rlgan_column_sums_noalloc(dy.data_ptr(), y.data_ptr(), rows, cols, got.data_ptr(), side.cuda_stream), "warm-up")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib.rlgan_column_sums_noalloc(dy.data_ptr(), y.data_ptr(), rows, cols, got.data_ptr(), st), "recorded call")
    for seed in (1, 2):
        g = torch.Generator(device="cuda").manual_seed(seed)
        dy.copy_(torch.randn(rows, cols, device="cuda", generator=g))
        y.copy_(torch.randn(rows, cols, device="cuda", generator=g))
        got.fill_(float("nan"))
        graph.replay()
        _lib.check(_lib.lib.rlgan_column_sums(dy.data_ptr(), y.data_ptr(), rows, cols, part.data_ptr(), ref.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "rlgan_column_sums")
        torch.cuda.synchronize()
        assert np.all(np.isfinite(ref.cpu().numpy()))
        np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())

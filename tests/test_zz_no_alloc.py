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

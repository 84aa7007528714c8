"""CPU checks of the drop-in boundary: the shared library loads without a GPU, exports every symbol
the header declares, the ctypes records have the C sizes, and the host-only entry points (geometry,
index maps, error strings) behave as the header documents.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rlgan_amd.h")

from rl_gan_amd import _lib  # noqa: E402
from oracle import nets  # noqa: E402


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rlgan_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 29
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} is declared in include/rlgan_amd.h but not exported by the library"
    bound = set(_lib.PROTOTYPES) | {"rlgan_error_string", "rlgan_deconv_fused_workspace"}     # the two that do not return int
    assert set(names) == bound, (set(names) ^ bound)


def test_library_exports_nothing_undeclared():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("rlgan_")}
    assert exported == set(_declared())


def test_abi_version_matches_header():
    m = re.search(r"#define\s+RLGAN_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert _lib.lib.rlgan_abi_version() == int(m.group(1))


def test_ctypes_records_have_the_c_sizes(tmp_path):
    """Compile a C program against the header and compare sizeof() of every record with ctypes."""
    recs = {"rlgan_geom_t": _lib.Geom, "rlgan_net_t": _lib.Net, "rlgan_acts_t": _lib.Acts, "rlgan_mat_t": _lib.Mat,
            "rlgan_env_t": _lib.Env, "rlgan_critic_step_t": _lib.CriticStep, "rlgan_actor_step_t": _lib.ActorStep,
            "rlgan_disc_step_t": _lib.DiscStep, "rlgan_gen_step_t": _lib.GenStep, "rlgan_wgrad_job_t": _lib.WgradJob,
            "rlgan_reduce_job_t": _lib.ReduceJob, "rlgan_apply_job_t": _lib.ApplyJob,
            "rlgan_pixel_critic_step_t": _lib.PixelCriticStep, "rlgan_pixel_actor_step_t": _lib.PixelActorStep,
            "rlgan_deconv_fused_geom_t": _lib.DeconvFusedGeom}
    src = tmp_path / "sz.c"
    body = "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in recs)
    src.write_text(f'#include <stdio.h>\n#include "rlgan_amd.h"\nint main(void) {{\n{body}  return 0;\n}}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        name, size = line.split()
        assert C.sizeof(recs[name]) == int(size), name


SPECS = {_lib.NET_ACTOR: nets.actor_spec, _lib.NET_CRITIC: nets.critic_spec,
         _lib.NET_GENERATOR: nets.generator_spec, _lib.NET_DISCRIMINATOR: nets.discriminator_spec}


def _maps(kind, s_dim=7, a_dim=2):
    g = _lib.geometry(kind, s_dim, a_dim)
    mf = np.full(g.n_params, -7, np.int32)
    mb = np.full(g.n_params, -7, np.int32)
    mg = np.full(g.n_params, -7, np.int32)
    _lib.check(_lib.lib.rlgan_build_maps(kind, s_dim, a_dim, mf.ctypes.data, mb.ctypes.data, mg.ctypes.data))
    return g, mf, mb, mg


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_geometry_counts_match_the_reference_networks(kind):
    """n_params = number of TF variables' elements of the reference graph (DDPG.py:78-95,153-172;
    Curious_GAN.py:46-82,112-135) as restated by the oracle's specs."""
    spec = SPECS[kind]()
    g = _lib.geometry(kind, 7, 2)
    flat = nets.flatten(spec, nets.init_params(spec, np.random.RandomState(0)))
    assert g.n_params == flat.size
    assert g.n3 == (50 if kind == _lib.NET_GENERATOR else 10)
    assert g.n_out == {0: 2, 1: 1, 2: 7, 3: 1}[kind]
    assert g.k0 == {0: 7, 1: 9, 2: 9, 3: 16}[kind]


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_maps_are_injective_and_in_range(kind):
    g, mf, mb, mg = _maps(kind)
    assert mf.min() >= 0 and mf.max() < g.wf_floats and np.unique(mf).size == g.n_params
    assert mg.min() >= 0 and mg.max() < g.gpad_floats and np.unique(mg).size == g.n_params
    has_b = mb >= 0
    assert np.all(mb[~has_b] == -1)
    n_bias = 200 + 200 + g.n3 + g.n_out                # biases have no place in the transposed image
    assert (~has_b).sum() == n_bias
    assert mb[has_b].max() < g.wb_floats and np.unique(mb[has_b]).size == has_b.sum()


@pytest.mark.parametrize("kind,a_pad", [(4, 8), (5, 8), (4, 4), (5, 12)])
def test_pixel_network_kinds(kind, a_pad):
    """Actor (4) and critic (5) of the pixel variant (Conv_model.py:621-651,711-744) on 256 features: parameter counts of
    the four layer-wise blocks, a first layer of 17 k-groups, maps that take every parameter once."""
    g, mf, mb, mg = _maps(kind, 256, a_pad)
    k0 = 256 + (a_pad if kind == 5 else 0)
    n_out = a_pad if kind == 4 else 4
    assert (g.k0, g.n3, g.n_out, g.ld3) == (k0, 12, n_out, 16)
    assert g.n_params == (k0 + 1) * 200 + 201 * 200 + 201 * 12 + 13 * n_out
    assert mf.min() >= 0 and mf.max() < g.wf_floats and np.unique(mf).size == g.n_params
    assert mg.min() >= 0 and mg.max() < g.gpad_floats and np.unique(mg).size == g.n_params
    has_b = mb >= 0
    assert (~has_b).sum() == 200 + 200 + 12 + n_out and np.unique(mb[has_b]).size == has_b.sum() and mb[has_b].max() < g.wb_floats
    # the first layer's bias row follows the last input row: k-group 16 of 17, 13 column tiles
    assert mf[k0 * 200] == ((k0 // 16 * 13 + 0) * 64 + 16 * ((k0 % 16) // 4) + 0) * 4 + k0 % 4
    gg = _lib.Geom()
    assert _lib.lib.rlgan_geometry(kind, 128, a_pad, C.byref(gg)) == -2        # 256 features only
    assert _lib.lib.rlgan_geometry(kind, 256, 16, C.byref(gg)) == -2           # 4 .. 12 action columns
    assert _lib.lib.rlgan_geometry(kind, 256, 6, C.byref(gg)) == -2            # a multiple of 4
    assert _lib.lib.rlgan_pixel_critic_step_rows(None, None) == -1 and _lib.lib.rlgan_pixel_actor_step_rows(None, None) == -1


def test_forward_fragment_layout_is_the_documented_one():
    """Layer 1 of the actor: element (k, n) of [W ; b] sits at ((k/16*NT + n/16)*64 + lane)*4 + k%4 with
    lane = 16*((k%16)/4) + n%16 and NT = 13 tiles of 16 columns."""
    g, mf, _, _ = _maps(_lib.NET_ACTOR)
    K, N, NT = 7, 200, 13
    def off(k, n):
        lane = 16 * ((k % 16) // 4) + n % 16
        return ((k // 16 * NT + n // 16) * 64 + lane) * 4 + k % 4
    w = np.array([[off(k, n) for n in range(N)] for k in range(K)]).ravel()
    b = np.array([off(K, n) for n in range(N)])
    assert np.array_equal(mf[:K * N], w)
    assert np.array_equal(mf[K * N:K * N + N], b)


def test_discriminator_first_layer_block_order():
    """TF creates w1_G, w1_s, w1_a (Curious_GAN.py:52-56) while the kernels read the input tile as
    [s, x, a]: the first canonical block must land on operand rows 7..13, the second on rows 0..6."""
    _, mf, _, mg = _maps(_lib.NET_DISCRIMINATOR)
    np_h = 208
    rows = mg[:16 * 200].reshape(16, 200)[:, 0] // np_h
    assert list(rows) == list(range(7, 14)) + list(range(0, 7)) + [14, 15]


def test_argument_errors():
    g = _lib.Geom()
    assert _lib.lib.rlgan_geometry(7, 7, 2, C.byref(g)) == -3
    assert _lib.lib.rlgan_geometry(0, 7, 2, None) == -1
    assert _lib.lib.rlgan_geometry(3, 9, 2, C.byref(g)) == -2          # 2*9+2 > 16 input columns
    assert _lib.lib.rlgan_error_string(-4) == b"Memory has not been fulfilled"   # Curious_GAN.py:167
    with pytest.raises(_lib.RlganError, match="unknown network kind"):
        _lib.check(-3, "x")


def test_wgrad_chunks_cover_the_batch():
    for rows in (16, 32, 4096, 8192, 65536):
        cr, nc = _lib.wgrad_chunks(rows)
        assert cr % 16 == 0 and cr * nc >= rows and cr * (nc - 1) < rows and nc <= 16


def test_missing_library_fails_loudly(tmp_path):
    """Importing the binding from a tree without the built library must raise, not fall back."""
    pkg = tmp_path / "pkg"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "_lib.py").write_text(open(os.path.join(ROOT, "rl-gan_amd", "_lib.py")).read())
    r = subprocess.run([sys.executable, "-c", "import pkg._lib"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_batchnorm_chunks_cover_the_rows():
    """One launch up to 256 rows; above that at most 128 chunks of at least 32 rows whose union is the batch."""
    f = _lib.lib.rlgan_batchnorm_chunks
    assert [f(m) for m in (1, 32, 256)] == [1, 1, 1]
    for m in (257, 1000, 4096, 4101, 1 << 20):
        ch = f(m)
        rows = max(32, -(-m // 128))
        assert 1 < ch <= 128 and ch == -(-m // rows) and rows * ch >= m > rows * (ch - 1)


@pytest.mark.parametrize("shape,kind", [((8, 8, 64, 32, 5, 2), 1), ((19, 19, 32, 1, 6, 2), 2), ((3, 3, 128, 64, 4, 2), 3),
                                        ((1, 1, 200, 128, 3, 2), 0), ((8, 8, 64, 32, 5, 1), 0), ((8, 8, 64, 16, 5, 2), 0)])
def test_fused_decoder_layer_shapes_and_maps(shape, kind):
    """dec_deconv3 and dec_deconv4 (Conv_model.py:468-473) have fused kernels; their two fragment images take every kernel
    element exactly once and leave the bias out; dec_deconv2 (Conv_model.py:463) has a fused forward pass only (kind 3: one
    image, no workspace, no fused backward); any other shape reports kind 0 and needs no workspace."""
    d = _lib.Deconv(*shape)
    g = _lib.DeconvFusedGeom()
    assert _lib.lib.rlgan_deconv_fused_geometry(d, C.byref(g)) == 0 and g.kind == kind
    h, _, ci, co, k, _ = shape
    n = k * k * co * ci
    mf, mb = np.full(n + co, -7, np.int32), np.full(n + co, -7, np.int32)
    if kind == 0:
        assert _lib.lib.rlgan_deconv_fused_workspace(d, 64) == 0
        assert _lib.lib.rlgan_deconv_fused_build_maps(d, mf.ctypes.data, mb.ctypes.data) == -2
        assert _lib.lib.rlgan_deconv_fused_forward(d, 256, 256, 256, 256, 4, 1, None) == -2
        return
    assert g.kernel_floats == n
    if kind == 3:
        _lib.check(_lib.lib.rlgan_deconv_fused_build_maps(d, mf.ctypes.data, None))
        assert np.all(mf[n:] == -1) and mf[:n].min() >= 0 and mf[:n].max() < g.wff_floats and np.unique(mf[:n]).size == n
        assert g.wfb_floats == 0 and _lib.lib.rlgan_deconv_fused_workspace(d, 64) == 0
        assert _lib.lib.rlgan_deconv_fused_backward(d, 256, None, 256, 256, None, 256, 256, 4, None) == -2
        return
    assert _lib.lib.rlgan_deconv_fused_build_maps(d, mf.ctypes.data, None) == -1
    _lib.check(_lib.lib.rlgan_deconv_fused_build_maps(d, mf.ctypes.data, mb.ctypes.data))
    for m, size in ((mf, g.wff_floats), (mb, g.wfb_floats)):
        assert np.all(m[n:] == -1) and m[:n].min() >= 0 and m[:n].max() < size and np.unique(m[:n]).size == n
    chunks = min(g.max_chunks, 100)
    assert _lib.lib.rlgan_deconv_fused_workspace(d, 100) >= chunks * n
    assert _lib.lib.rlgan_deconv_fused_forward(d, None, 256, 256, 256, 4, 1, None) == -1
    assert _lib.lib.rlgan_deconv_fused_forward(d, 256, 256, 260, 256, 4, 1, None) == -5               # 16-byte alignment
    assert _lib.lib.rlgan_deconv_fused_backward(d, 256, None, 256, 256, None, None, 256, 4, None) == -1   # workspace missing


def test_argument_errors_of_the_pixel_entry_points():
    """Checked on the host before anything is launched: missing pointers and sizes that do not fit."""
    L = _lib.lib
    p = 256                                                # stands for a device address; never dereferenced on these paths
    assert L.rlgan_batchnorm_forward(None, 8, p, p, 4, 8, 1e-3, 1, 0, p, 8, p, p, None, None) == -1
    assert L.rlgan_batchnorm_forward(p, 8, p, p, 4, 8, 1e-3, 1, 0, p, 8, None, p, None, None) == -1      # training needs save_mean
    assert L.rlgan_batchnorm_forward(p, 8, p, p, 4, 8, 1e-3, 1, 5, p, 8, p, p, None, None) == -1         # unknown activation
    assert L.rlgan_batchnorm_forward(p, 4, p, p, 4, 8, 1e-3, 1, 0, p, 8, p, p, None, None) < 0           # ld < n
    assert L.rlgan_batchnorm_forward(p, 8, p, p, 1000, 8, 1e-3, 1, 0, p, 8, p, p, None, None) == -1      # chunks > 1 without scratch
    assert L.rlgan_batchnorm_backward(p, 8, None, 8, p, 8, p, p, p, 4, 8, 3, p, 8, p, p, None, None) == -1   # relu6 needs y
    assert L.rlgan_softmax_ce(p, 4, 6, p, 6, 3, p, None, None) < 0                                       # ld < n_classes
    assert L.rlgan_td_error(p, p, 4, None, 0.99, 8, p, p, None) == -1
    assert L.rlgan_act_backward(p, p, p, 10, 1, None) == -1                                              # relu6 / tanh only
    assert L.rlgan_lstm_gates(p, p, 0, 256, 1.0, p, p, None) == -1
    assert L.rlgan_dense_forward(p, 8, p, p, 8, 4, 8, 8, 5, None, 1, None) == -1                         # activation codes 0..4

"""Error behaviour of the C ABI on a live device: every entry point returns a negative code and
leaves a message in vch_last_error() instead of faulting (NULL context, NULL arrays, sizes out of
range, calls out of order); the Python layer maps VCH_ERR_ARG to the reference's ValueError."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    import vch_amd
    vch_amd.build()
    return vch_amd


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_null_context_and_arguments(V):
    lib = V.load()
    z = np.zeros((17, 17))
    assert lib.vch2d_apply_laplacian(None, _dp(z), _dp(z)) < 0 and b"NULL context" in lib.vch_last_error()
    assert lib.vch1d_apply_laplacian(None, _dp(z), _dp(z)) < 0 and b"NULL context" in lib.vch_last_error()
    assert lib.vch2d_batch(None) < 0 and lib.vch2d_uses_fft(None) < 0
    lib.vch2d_destroy(None)                       # no-ops
    lib.vch1d_destroy(None)
    e = V.Engine2D(Nx=16, Ny=16, max_steps=4)
    assert lib.vch2d_apply_laplacian(e.ctx, None, _dp(z)) == -1 and b"NULL" in lib.vch_last_error()
    dts = np.full(4, 1e-2)
    assert lib.vch2d_forward(e.ctx, None, None, 0, _dp(dts), 4, None, None) == -1
    assert lib.vch2d_forward(e.ctx, _dp(z), None, 0, _dp(dts), 5, None, None) == -1          # M > max_steps
    assert b"max_steps" in lib.vch_last_error()
    bad = np.array([1e-2, 0.0, 1e-2, 1e-2])
    assert lib.vch2d_forward(e.ctx, _dp(z), None, 0, _dp(bad), 4, None, None) == -1          # dt <= 0
    assert lib.vch2d_pgd_iterate(e.ctx, 1, None, None, None, None, None) < 0 and b"pgd_init" in lib.vch_last_error()
    assert lib.vch2d_pgd_get(e.ctx, 0, _dp(z)) < 0
    assert lib.vch2d_free_energy(e.ctx, _dp(z), 9, None, 0.1, 0.1, 0.0, _dp(z)) == -1        # rows > max_steps + 1
    out = np.zeros((17, 17))
    assert lib.vch2d_apply_laplacian(e.ctx, _dp(z + 1.0), _dp(out)) == 0 and not out.any()    # still usable afterwards
    e.close()


def test_create_rejects_bad_parameters(V):
    lib = V.load()
    _lib = V.module("_lib")
    p2 = _lib.Params2D(0, 16, 1.0, 1.0, 0.05, 10.0, 0.75, 1.0, 1e-4)
    assert not lib.vch2d_create(C.byref(p2), 1, 4, 0) and lib.vch_last_error()
    p2 = _lib.Params2D(16, 16, 1.0, 1.0, 0.05, 10.0, 0.75, 1.0, 1e-4)
    assert not lib.vch2d_create(C.byref(p2), 0, 4, 0)                                        # batch < 1
    assert not lib.vch2d_create(C.byref(p2), 1, 4, 99)                                       # no such device
    p1 = _lib.Params1D(8192, 1.0, 0.05, 10.0, 0.75, 1.0, 9e-4)
    assert not lib.vch1d_create(C.byref(p1), 1, 4, 0) and b"4096" in lib.vch_last_error()
    with pytest.raises(ValueError):
        V.Engine1D(N=8192)
    with pytest.raises(ValueError):
        V.Engine2D(Nx=16, Ny=16, batch=0)


def test_python_layer_maps_errors(V):
    e = V.Engine2D(Nx=16, Ny=16, max_steps=4)
    with pytest.raises(ValueError):
        e.apply_laplacian(np.zeros((16, 17)))                     # F2:148-149
    with pytest.raises(ValueError):
        e.forward(np.zeros((17, 17)), np.full(5, 1e-2))           # more steps than the context holds
    with pytest.raises(V.VchError):
        e.pgd_iterate(1)                                          # out of order
    e.close()
    e1 = V.Engine1D(N=32, max_steps=4)
    with pytest.raises(IndexError):
        e1.forward(np.zeros(33), np.full(4, 1e-2), u=np.zeros((3, 33)))      # F1:347-353
    with pytest.raises(V.VchError):
        e1.pgd_iterate(1)
    with pytest.raises(ValueError):
        e1.pgd_init(np.zeros(33), np.zeros(33), np.zeros(6), np.full(3, 1e-2), V.make_opt())   # len(dt) != rows - 2
    e1.close()

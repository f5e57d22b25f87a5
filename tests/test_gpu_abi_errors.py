"""Error convention of the C ABI (include/mcpc.h): every entry point returns 0 or a negative MCPC_E* code with a message in
mcpc_last_error(); a bad argument or a violated call order is REPORTED, never a crash, a hang or a silently wrong run.
Called through ctypes directly (no Engine wrapper in between)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EINVAL, EHIP, ENOMEM, ESTATE = -1, -2, -3, -4


def _desc(L, sizes=(8, 16), n_in=8, n_out=24, batch=20, **kw):
    d = L.NetDesc()
    d.abi_version = kw.get("abi", L.ABI_VERSION)
    d.n_latent = kw.get("n_latent", len(sizes)); d.n_in = n_in; d.n_out = n_out; d.batch = batch; d.device = kw.get("device", 0)
    for i, n in enumerate(sizes):
        d.sizes[i] = n; d.acts[i] = kw.get("act", 1); d.ecoef[i] = kw.get("ecoef", 1.0)
    d.tuning = kw.get("tuning", None)
    return d


def _err(lib):
    return lib.mcpc_last_error().decode()


def test_create_rejects_bad_descriptions():
    from montecarlopredictivecoding_amd import _lib as L
    lib = L.load()
    torch.zeros(1, device=DEV)
    h = C.c_void_p()
    for kw, code, word in ((dict(abi=1), EINVAL, "ABI version"), (dict(n_latent=0), EINVAL, "n_latent"), (dict(n_latent=7), EINVAL, "n_latent"),
                           (dict(batch=0), EINVAL, "batch"), (dict(act=3), EINVAL, "acts"), (dict(ecoef=0.0), EINVAL, "ecoef"),
                           (dict(device=99), EINVAL, "device"), (dict(tuning=b"bogus=1"), EINVAL, "unknown tuning key"),
                           (dict(tuning=b"ws=1"), EINVAL, "ws=1")):
        d = _desc(L, **kw)
        assert lib.mcpc_create(C.byref(d), C.byref(h)) == code, kw
        assert word in _err(lib), (kw, _err(lib))
        assert not h.value
    assert lib.mcpc_create(None, C.byref(h)) == EINVAL
    assert lib.mcpc_destroy(None) == 0                       # like free(NULL)


def test_call_order_and_arguments_are_checked():
    from montecarlopredictivecoding_amd import _lib as L
    lib = L.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = C.c_void_p()
    d = _desc(L)
    assert lib.mcpc_create(C.byref(d), C.byref(h)) == 0, _err(lib)
    W = [torch.randn(8, 8, device=DEV) * 0.1, torch.randn(16, 8, device=DEV) * 0.1, torch.randn(24, 16, device=DEV) * 0.1]
    b = [torch.zeros(n, device=DEV) for n in (8, 16, 24)]
    y = (torch.rand(20, 24, device=DEV) < 0.3).float()
    en = torch.zeros(5, 8, dtype=torch.float64, device=DEV)

    def run_desc(**kw):
        r = L.RunDesc()
        r.T = 5; r.t_begin = 0; r.n_steps = 5; r.loss_kind = 2; r.loss_var = 1.0; r.xopt_kind = 0; r.lr = 0.05
        r.beta1 = 0.9; r.beta2 = 0.999; r.eps = 1e-8; r.update_x = 1; r.noise_mode = 0; r.noise_var = 2.0
        r.energy_mode = 2; r.energies_out = en.data_ptr()
        for k, v in kw.items():
            setattr(r, k, v)
        return r

    r = run_desc()
    # nothing bound yet
    assert lib.mcpc_run(h, C.byref(r), stream) == ESTATE and "no bound parameters" in _err(lib)
    assert lib.mcpc_params_changed(h, stream) == ESTATE
    assert lib.mcpc_bind_params(h, 3, W[0].data_ptr(), None) == EINVAL and "out of range" in _err(lib)
    assert lib.mcpc_bind_params(h, 0, None, None) == EINVAL
    for j in range(3):
        assert lib.mcpc_bind_params(h, j, W[j].data_ptr(), b[j].data_ptr()) == 0
    assert lib.mcpc_params_changed(h, stream) == 0
    assert lib.mcpc_bind_inputs(h, None, stream) == 0
    # a loss without a bound target
    assert lib.mcpc_run(h, C.byref(r), stream) == ESTATE and "no target bound" in _err(lib)
    assert lib.mcpc_bind_target(h, None, stream) == EINVAL
    assert lib.mcpc_bind_target(h, y.data_ptr(), stream) == 0
    # argument checks of the run descriptor
    for kw, word in ((dict(n_steps=6), "bad step range"), (dict(t_begin=-1), "bad step range"), (dict(loss_kind=3), "loss_kind"),
                     (dict(loss_kind=1, loss_var=0.0), "loss_var"), (dict(mask_start=24), "mask_start"), (dict(xopt_kind=2), "xopt_kind"),
                     (dict(lr=0.0), "lr must be positive"), (dict(noise_mode=3), "noise_mode"), (dict(noise_mode=1, xopt_kind=1), "SGD on x only"),
                     (dict(noise_mode=2), "ext_noise"), (dict(update_x=0), "xgrad"), (dict(energy_mode=3), "energy_mode"),
                     (dict(energies_out=None), "energies_out"), (dict(rec_count=2, rec_stride=0), "record schedule")):
        rr = run_desc(**kw)
        assert lib.mcpc_run(h, C.byref(rr), stream) == EINVAL, kw
        assert word in _err(lib), (kw, _err(lib))
    assert lib.mcpc_run(None, C.byref(r), stream) == EINVAL
    # state pointers
    xs = [torch.rand(20, n, device=DEV) for n in (8, 16)]
    ptrs = (C.c_void_p * 2)(xs[0].data_ptr(), None)
    assert lib.mcpc_load_state(h, ptrs, stream) == EINVAL and "layer 1" in _err(lib)
    ptrs = (C.c_void_p * 2)(*[x.data_ptr() for x in xs])
    assert lib.mcpc_load_state(h, ptrs, stream) == 0
    # after all that, a good run still works and the device reports no fault
    assert lib.mcpc_run(h, C.byref(r), stream) == 0, _err(lib)
    assert lib.mcpc_sync_check(h, stream) == 0, _err(lib)
    assert torch.isfinite(en).all() and float(en[-1, -1]) < float(en[0, -1])
    # gradient read-out: wrong bucket size, bad index
    flat = torch.empty(lib.mcpc_param_count(h) + 1, device=DEV)
    assert lib.mcpc_read_param_grads_flat(h, flat.data_ptr(), flat.numel(), 1.0, stream) == EINVAL
    assert lib.mcpc_read_param_grads(h, 5, flat.data_ptr(), None, 1.0, 0, stream) == EINVAL
    assert lib.mcpc_read_param_grads(h, 0, None, None, 1.0, 0, stream) == EINVAL
    assert lib.mcpc_destroy(h) == 0


def test_network_without_read_out_rejects_losses_and_targets():
    from montecarlopredictivecoding_amd import _lib as L
    lib = L.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    h = C.c_void_p()
    d = _desc(L, sizes=(4, 12), n_in=4, n_out=0)
    assert lib.mcpc_create(C.byref(d), C.byref(h)) == 0, _err(lib)
    assert lib.mcpc_bind_target(h, torch.zeros(20, 1, device=DEV).data_ptr(), stream) == EINVAL and "no read-out" in _err(lib)
    W = [torch.randn(4, 4, device=DEV), torch.randn(12, 4, device=DEV)]
    for j in range(2):
        assert lib.mcpc_bind_params(h, j, W[j].data_ptr(), None) == 0           # bias=False Linears
    assert lib.mcpc_params_changed(h, stream) == 0
    r = L.RunDesc()
    r.T = 2; r.n_steps = 2; r.loss_kind = 2; r.lr = 0.1; r.update_x = 1
    assert lib.mcpc_run(h, C.byref(r), stream) == EINVAL and "needs a read-out" in _err(lib)
    assert lib.mcpc_destroy(h) == 0

"""devis_evid_loss_fwd_noalloc (include/devis_hip.h): the entry point for callers that cannot allocate.
It runs devis_evid_loss_fwd's kernels with the block partials in a fixed `__device__` array of the library, so its
outputs must be BIT-IDENTICAL to devis_evid_loss_fwd's (same kernels, same reduction order) - on two calls in a
row, the second on a smaller input (the array keeps the first call's contents), and sizes beyond the fixed scratch
must be refused without a launch.  Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(fn_name, z, lab, lam, with_ws):
    from devis_amd import _lib, ops
    lib = _lib.load()
    n, v, k, sn, sv, sk = ops.logits_strides(z, True)
    dev = z.device
    ev = torch.full((n, k) + tuple(z.shape[1:-1]), -7.0, dtype=torch.float32, device=dev)
    pv = torch.full((n * v, 1), -7.0, dtype=torch.float32, device=dev)
    scal = torch.full((_lib.EVID_SCAL_FLOATS,), -7.0, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    common = (z.data_ptr(), sn, sv, sk, ops.dt_code(z.dtype), lab.data_ptr(), ops._label_code(lab), n, v, k, float(lam),
              ev.data_ptr(), pv.data_ptr(), scal.data_ptr())
    if with_ws:
        nb = lib.devis_evid_workspace_bytes(n, v, k)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        rc = getattr(lib, fn_name)(*common, ws.data_ptr(), nb, st)
    else:
        rc = getattr(lib, fn_name)(*common, st)
    torch.cuda.synchronize()
    return rc, ev, pv, scal[:4 + 3 * k].clone()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_noalloc_entry_matches_workspace_entry_twice_second_call_smaller(dtype):
    g = torch.Generator().manual_seed(3)
    # second input smaller (fewer blocks than the first): stale partials of call 1 stay in the scratch
    for shape in ((2, 40, 36, 32, 4), (1, 8, 8, 8, 4)):
        z = (torch.randn(shape, generator=g) * 3).to(dtype).cuda()
        lab = torch.randint(0, 4, shape[:-1], generator=g).cuda()
        rc0, ev0, pv0, sc0 = _call("devis_evid_loss_fwd", z, lab, 0.5, True)
        rc1, ev1, pv1, sc1 = _call("devis_evid_loss_fwd_noalloc", z, lab, 0.5, False)
        assert rc0 == 0 and rc1 == 0
        assert torch.equal(sc0, sc1), (sc0, sc1)
        assert torch.equal(ev0, ev1) and torch.equal(pv0, pv1)
        assert torch.isfinite(sc1).all() and (ev1 != -7.0).all()


def test_noalloc_entry_runs_with_the_allocator_untouched():
    """No device memory is taken by the call: torch's allocator statistics (the only other allocator in the
    process) and the driver's free-memory figure do not move across warm calls."""
    g = torch.Generator().manual_seed(4)
    z = torch.randn((2, 16, 16, 16, 4), generator=g).cuda()
    lab = torch.randint(0, 4, (2, 16, 16, 16), generator=g).cuda()
    _call("devis_evid_loss_fwd_noalloc", z, lab, 1.0, False)          # first call resolves the symbol address
    from devis_amd import _lib, ops
    lib = _lib.load()
    n, v, k, sn, sv, sk = ops.logits_strides(z, True)
    scal = torch.empty((_lib.EVID_SCAL_FLOATS,), dtype=torch.float32, device=z.device)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    a0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(3):
        rc = lib.devis_evid_loss_fwd_noalloc(z.data_ptr(), sn, sv, sk, 0, lab.data_ptr(), 0, n, v, k, 1.0, None, None,
                                             scal.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == a0
    # the driver's figure is device-wide (other processes, lazily loaded code objects move it): a coarse check only
    assert free1 >= free0 - (8 << 20)


def test_noalloc_entry_refuses_more_samples_than_its_scratch_holds():
    from devis_amd import _lib
    lib = _lib.load()
    n = 1025                                              # DEVIS_EVID_NOALLOC_MAX_N = 1024
    z = torch.zeros((n, 2, 4), device="cuda")
    lab = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    scal = torch.full((_lib.EVID_SCAL_FLOATS,), 5.0, device="cuda")
    rc = lib.devis_evid_loss_fwd_noalloc(z.data_ptr(), 8, 4, 1, 0, lab.data_ptr(), 0, n, 2, 4, 1.0, None, None,
                                         scal.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == -5
    assert (scal == 5.0).all()                            # nothing was launched
    with pytest.raises(_lib.DevisError, match="fixed scratch"):
        _lib.check(rc, "evid_loss_fwd_noalloc")

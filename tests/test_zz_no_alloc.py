"""GPU test of pfx_energy_dev_noalloc, the entry point that allocates nothing (static device scratch of the library).

Called through the C ABI and compared with pfx_energy_dev -- the existing entry point it is a variant of -- on two
calls in a row, the second on a smaller mesh (the shared scratch then still holds the first call's element energies
beyond the second mesh's range).  Same kernels and the same fixed summation order: the bar is bitwise equality, and the
energy also matches the CPU oracle to 1e-12 relative.
"""
import numpy as np
import pytest
import torch

from polyfem_amd import engine as E
from polyfem_amd import mesh as M

pytestmark = pytest.mark.gpu


def _both(eng, mesh, lam, mu, amp, with_pattern):
    eng.set_mesh(mesh)
    eng.set_material(E.NEOHOOKEAN, lam, mu)
    x = M.seeded_displacement(mesh, 3, amp)
    d_u = torch.from_numpy(x).cuda()
    d_e = torch.full((2,), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    # the new entry point needs no pattern (nothing of the context's lazily allocated buffers)
    eng.energy_dev_noalloc(d_u.data_ptr(), d_e[0:1].data_ptr())
    eng.synchronize()
    e_new = float(d_e[0].item())
    eng.build_pattern()
    eng.energy_dev(d_u.data_ptr(), d_e[1:2].data_ptr())
    eng.synchronize()
    e_old = float(d_e[1].item())
    return x, e_new, e_old


def test_energy_noalloc_matches_energy_dev_twice(oracle):
    lam, mu = oracle.lame(2e4, 0.3)
    eng = E.Engine(0)
    try:
        big = M.cube_tets(6, 2, jitter=0.1)
        small = M.cube_tets(2, 2, jitter=0.1, seed=5)
        for mesh, amp in ((big, 2e-3), (small, 5e-3)):
            x, e_new, e_old = _both(eng, mesh, lam, mu, amp, True)
            assert e_new == e_old, (e_new, e_old)
            e_ref = oracle.assemble_energy(mesh, oracle.NEOHOOKEAN, lam, mu, x)
            assert abs(e_new - e_ref) <= 1e-12 * abs(e_ref)
        # hexahedra go through the same entry point
        q = M.cube_hexes(2, 2)
        x, e_new, e_old = _both(eng, q, lam, mu, 1e-3, True)
        assert e_new == e_old
    finally:
        eng.close()


def test_energy_noalloc_rejects_oversize_and_missing_mesh():
    eng = E.Engine(0)
    try:
        d = torch.zeros(8, dtype=torch.float64, device="cuda")
        with pytest.raises(E.PfxError):
            eng.energy_dev_noalloc(d.data_ptr(), d.data_ptr())  # no mesh yet
        n = 56  # 6 * 56^3 = 1 053 696 P1 tets > 2^20
        mesh = M.cube_tets(n, 1)
        assert mesh.n_el > (1 << 20)
        eng.set_mesh(mesh)
        eng.set_material(E.NEOHOOKEAN, 1.0, 1.0)
        u = torch.zeros(mesh.n_nodes * 3, dtype=torch.float64, device="cuda")
        with pytest.raises(E.PfxError) as ei:
            eng.energy_dev_noalloc(u.data_ptr(), d.data_ptr())
        assert ei.value.code == E.PFX_ERR_SIZE
    finally:
        eng.close()

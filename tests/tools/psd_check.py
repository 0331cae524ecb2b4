import sys
import numpy as np
sys.path.insert(0, ".")  # run from the repository root
import torch
from polyfem_amd import engine as E, mesh as M
from oracle import pf_oracle as O
eng = E.Engine(0)
lam, mu = O.lame(2e4, 0.3)
for name, mesh in (("p1", M.cube_tets(4, 1, jitter=0.2, seed=1)), ("p2", M.cube_tets(3, 2, numbering="polyfem", jitter=0.2, seed=2)), ("q1", M.cube_hexes(3, 1))):
    for model, mid in (("neo", O.NEOHOOKEAN), ("stvk", O.SAINT_VENANT)):
        eng.set_mesh(mesh); eng.set_material(mid, lam, mu); eng.build_pattern()
        h = 1.0 / 4
        for amp in (1e-2 * h, 0.3 * h):
            x = M.seeded_displacement(mesh, 3, amp)
            try:
                v = eng.hessian(x); vp = eng.hessian(x, project_to_psd=True)
            except E.PfxError as ex:
                print(name, model, amp, "error", ex); eng.take_numeric_flag(); continue
            r = O.assemble_hessian(mesh, mid, lam, mu, x)[2]; rp = O.assemble_hessian_psd(mesh, mid, lam, mu, x)[2]
            print(name, model, amp, "plain err %.2e" % (abs(v - r).max() / abs(r).max()), "psd err %.2e" % (abs(vp - rp).max() / abs(rp).max()), "psd changes %.2e" % (abs(rp - r).max() / abs(r).max()))

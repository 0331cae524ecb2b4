import numpy as np, sys
sys.path.insert(0, '.')
from polyfem_amd import engine as E, mesh as M
from oracle import pf_oracle as O
eng = E.Engine(0)
worst = 0
for mesh, dim in ((M.cube_tets(3, 2, jitter=0.2, seed=1), 3), (M.cube_hexes(2, 2), 3), (M.cube_tets(4, 1, jitter=0.2), 3), (M.square_tris(5, 2, jitter=0.2), 2), (M.cube_hexes(3, 1), 3)):
    lam, mu = O.lame(2e4, 0.4, dim)
    eng.set_mesh(mesh); eng.set_material(E.NEOHOOKEAN, lam, mu); eng.build_pattern()
    n = mesh.n_nodes * dim
    pos = M.node_positions(mesh)
    for amp in (0.05, 0.15, 0.3):
        rng = np.random.default_rng(3)
        # a large smooth deformation (stretch + shear + bending) plus noise
        u = np.zeros((mesh.n_nodes, dim))
        u[:, 0] = amp * pos[:, 0] + 0.5 * amp * pos[:, 1] ** 2
        u[:, 1] = -0.3 * amp * pos[:, 1] + amp * np.sin(2 * pos[:, 0])
        if dim == 3:
            u[:, 2] = 0.7 * amp * pos[:, 2] * pos[:, 0]
        u += rng.standard_normal(u.shape) * amp * 0.02
        x = u.reshape(-1)
        try:
            h = eng.hessian(x)
        except E.PfxError as ex:
            print("inverted", amp); continue
        ho = O.assemble_hessian(mesh, O.NEOHOOKEAN, lam, mu, x)
        err = np.abs(h - ho[2]).max() / np.abs(ho[2]).max()
        hp = eng.hessian(x, project_to_psd=True)
        hpo = O.assemble_hessian_psd(mesh, O.NEOHOOKEAN, lam, mu, x)
        errp = np.abs(hp - hpo[2]).max() / np.abs(hpo[2]).max()
        worst = max(worst, err)
        print(M.KIND_NAME[mesh.kind], amp, "hessian rel err %.2e  psd %.2e" % (err, errp))
print("worst", worst)
assert worst < 1e-12

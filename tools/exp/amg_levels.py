"""Level graphs of the AMG hierarchy rebuilt with scipy from the aggregates the library reports: degrees and the size of a
greedy maximal matching per level (diagnostic for the device-side aggregation)."""
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, '/root/repo')
from polyfem_amd import engine as E, distributed as D

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 20
part, _, _ = D.slab_part(grid, grid, 0, 1, 2, "tet")
mesh = part.mesh
eng = E.Engine(0)
eng.set_mesh(mesh)
eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6)
eng.build_pattern()
fixed_nodes = np.nonzero(np.abs(mesh.node_xyz[:, 0]) < 1e-12)[0]
eng.set_dirichlet(np.sort((fixed_nodes[:, None] * 3 + np.arange(3)[None]).reshape(-1)).astype(np.int32))
eng.assemble_linear()
eng.amg_setup()
info = eng.amg_info()
rp, ci = eng.get_pattern()
n = mesh.n_nodes
A = sp.csr_matrix((np.ones(ci.size), ci, rp), shape=(3 * n, 3 * n))
G = A[::3, ::3].tocsr()
G.data[:] = 1
for l in range(info["n_levels"] - 1):
    agg = eng.amg_aggregates(l)
    nv = G.shape[0]
    deg = np.diff(G.indptr)
    # greedy maximal matching in vertex order, heaviest free neighbour
    matched = np.zeros(nv, bool)
    matched[agg < 0] = True
    pairs = 0
    for v in range(nv):
        if matched[v]:
            continue
        cols = G.indices[G.indptr[v]:G.indptr[v + 1]]
        ws = G.data[G.indptr[v]:G.indptr[v + 1]]
        ok = (~matched[cols]) & (cols != v)
        matched[v] = True
        if ok.any():
            j = cols[ok][np.argmax(ws[ok])]
            matched[j] = True
            pairs += 1
    na = int(agg.max()) + 1
    sz = np.bincount(agg[agg >= 0], minlength=na)
    print(f"level {l}: {nv} vertices, degree min/mean/max {deg.min()}/{deg.mean():.1f}/{deg.max()}, greedy pairs {pairs} of {int((agg >= 0).sum())} free; "
          f"{na} aggregates, sizes min/mean/max {sz.min()}/{sz.mean():.1f}/{sz.max()}")
    keep = agg >= 0
    P = sp.csr_matrix((np.ones(keep.sum()), (np.nonzero(keep)[0], agg[keep])), shape=(nv, na))
    G = (P.T @ G @ P).tocsr()
eng.close()

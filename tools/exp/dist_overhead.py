"""Host overhead of the Python-driven distributed CG loop (single rank, no communication) against the C++ loop."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyfem_amd import engine as E, mesh as M, distributed as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
part, bounds, ng = D.slab_part(n, n, 0, 1, 2)
mesh = part.mesh
eng = E.Engine(0); eng.set_mesh(mesh)
eng.set_material(E.LINEAR_ELASTICITY, 1e5 * 0.3 / (1.3 * 0.4), 1e5 / 2.6); eng.build_pattern()
eng.set_owned(part.n_owned, part.elem_owned)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); eng.set_stream(stream.cuda_stream)
eng.assemble_linear()
nd = mesh.n_nodes * 3
rhs = torch.randn(nd, dtype=torch.float64, device=dev); x = torch.zeros(nd, dtype=torch.float64, device=dev)
comm = D.Comm()
dcg = D.DistributedCG(D.HipOps(eng), part, 3, comm, dev)
for name, fn in (("c++ loop", lambda: eng.cg_dev(rhs.data_ptr(), x.data_ptr(), E.PRECOND_JACOBI, 500, 1e-12)),
                 ("python step loop", lambda: dcg.solve(rhs, x, E.PRECOND_JACOBI, 500, 1e-12, check_every=50))):
    for rep in range(2):
        x.zero_(); torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(name, r, "%.1f ms, %.1f us/iter" % (dt * 1e3, dt * 1e6 / 500))
# pure host cost of the step calls: tiny problem so the GPU is never the bottleneck

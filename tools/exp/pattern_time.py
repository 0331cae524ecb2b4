"""Wall time of pfx_build_pattern for consecutive contexts in one process: python tools/exp/pattern_time.py [grid] [seconds to wait after each close]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from polyfem_amd import engine as E, mesh as M
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 119
wait = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
mesh = M.cube_tets(grid, 2)
for rep in range(4):
    free0 = torch.cuda.mem_get_info()[0]
    eng = E.Engine(0)
    eng.set_mesh(mesh)
    eng.set_material(E.NEOHOOKEAN, 1.0, 1.0)
    eng.synchronize()
    t0 = time.time(); eng.build_pattern(); eng.synchronize(); t1 = time.time()
    free1 = torch.cuda.mem_get_info()[0]
    t2 = time.time(); eng.close(); t3 = time.time()
    print(f"grid {grid}: build_pattern {t1 - t0:.3f} s, close {t3 - t2:.3f} s, free before {free0 / 2**30:.1f} GiB, with pattern {free1 / 2**30:.1f} GiB, after close {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
    time.sleep(wait)

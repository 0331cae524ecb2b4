"""CPU test: the C-ABI library loads and exports every symbol declared in include/polyfem_amd.h
(no compute call is made here: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

from polyfem_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "polyfem_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pfx_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(engine.LIB_PATH), "build with python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(engine.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/polyfem_amd.h but not exported"
    assert sorted(engine.EXPORTED_SYMBOLS) == names


def test_no_cpu_fallback_when_no_device():
    """Without a GPU the engine must refuse to start, never silently compute elsewhere."""
    import pytest
    L = engine.load()
    h = ctypes.c_void_p()
    rc = L.pfx_create(ctypes.byref(h), 0)
    if rc == 0:  # running on a GPU box
        L.pfx_destroy(h)
        pytest.skip("a HIP device is present")
    assert rc < 0 and not h
    with pytest.raises(engine.PfxError):
        engine.Engine(0)


def test_elasticity_tensor_host_helper(oracle):
    """engine.elasticity_tensor follows ElasticityTensor's three setters (utils/ElasticityUtils.cpp:298-437)."""
    import numpy as np
    lam, mu = oracle.lame(1e5, 0.3)
    a = engine.elasticity_tensor(lam=lam, mu=mu)
    b = engine.elasticity_tensor(young=1e5, nu=0.3)
    assert np.array_equal(a, a.T) and np.allclose(a, b, rtol=1e-14, atol=0)
    assert np.array_equal(a, oracle.elasticity_tensor(lam, mu))
    e = np.arange(1.0, 22.0)
    c = engine.elasticity_tensor(entries=e)
    assert np.array_equal(c, c.T) and np.array_equal(c[np.triu_indices(6)], e)
    assert np.array_equal(c, oracle.elasticity_tensor(entries=e))
    assert c[0, 5] == 6.0 and c[1, 1] == 7.0 and c[5, 5] == 21.0  # row by row, as set_from_entries


def test_rccl_is_linked_and_its_symbols_resolve():
    """The partitioned solve talks RCCL from inside the library: librccl is a link-time dependency and the collectives the
    library calls resolve in the loaded image (no call is made: there is no GPU here)."""
    import ctypes
    import subprocess
    from polyfem_amd import engine as E
    out = subprocess.run(["readelf", "-d", E.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl.so" in out, "libpolyfem_amd.so does not link RCCL"
    E.load()
    rccl = ctypes.CDLL("librccl.so.1")  # the copy the loader already mapped for the library
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
                "ncclAllReduce", "ncclGetErrorString"):
        assert getattr(rccl, sym) is not None

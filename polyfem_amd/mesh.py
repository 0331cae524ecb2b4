"""Synthetic, deterministic finite-element meshes in the flattened layout the assembly engine consumes.

The engine's input is what the reference hands to its assemblers once `FEBasis3d::build_bases` has run
(`src/polyfem/basis/FEBasis3d.cpp:1465-1546`): per element the global index of every local basis
(`Local2Global.index`, one entry of weight 1 per basis on conforming meshes, `basis/Basis.hpp:21-38`) in
the reference's local node order (`autogen/auto_p_bases.cpp:1080-1087,1328-1341`,
`autogen/auto_q_bases.cpp:1262-1275,1873-1903`), and the geometry nodes (`Local2Global.node`) of the
linear geometric map (`space/advanced/isoparametric=false`, `State.cpp:522-555`).

Global numbering is an input to the path; two deterministic numberings are offered:
  * "lattice": node id = lexicographic index of the node in the (p*n+1)^3 lattice (x fastest);
  * "polyfem": vertex nodes first (lattice order), then the remaining nodes in first-touch order while
    walking the elements, which is how `mesh/MeshNodes.cpp:157-308` hands out edge/face/cell nodes.
"""
from __future__ import annotations

import dataclasses
import itertools

import numpy as np

P1TET, P2TET, Q1HEX, Q2HEX, P1TRI, P2TRI = 0, 1, 2, 3, 4, 5
N_LOC = {P1TET: 4, P2TET: 10, Q1HEX: 8, Q2HEX: 27, P1TRI: 3, P2TRI: 6}
GEOM_KIND = {P1TET: P1TET, P2TET: P1TET, Q1HEX: Q1HEX, Q2HEX: Q1HEX, P1TRI: P1TRI, P2TRI: P1TRI}
KIND_NAME = {P1TET: "P1TET", P2TET: "P2TET", Q1HEX: "Q1HEX", Q2HEX: "Q2HEX", P1TRI: "P1TRI", P2TRI: "P2TRI"}
KIND_DIM = {P1TET: 3, P2TET: 3, Q1HEX: 3, Q2HEX: 3, P1TRI: 2, P2TRI: 2}
# local P2 triangle nodes 3..5 are the midpoints of these vertex pairs (autogen/auto_p_bases.cpp:275-284)
_P2_TRI_EDGES = ((0, 1), (1, 2), (2, 0))

# local P2 edge nodes 4..9 are the midpoints of these vertex pairs
_P2_EDGES = ((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))
# Q1 corner offsets and Q2 node offsets (units of half a cell)
_Q1_CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])
_Q2_NODES = np.array(
    [(0, 0, 0), (2, 0, 0), (2, 2, 0), (0, 2, 0), (0, 0, 2), (2, 0, 2), (2, 2, 2), (0, 2, 2),
     (1, 0, 0), (2, 1, 0), (1, 2, 0), (0, 1, 0), (0, 0, 1), (2, 0, 1), (2, 2, 1), (0, 2, 1),
     (1, 0, 2), (2, 1, 2), (1, 2, 2), (0, 1, 2), (0, 1, 1), (2, 1, 1), (1, 0, 1), (1, 2, 1),
     (1, 1, 0), (1, 1, 2), (1, 1, 1)])


@dataclasses.dataclass
class Mesh:
    kind: int
    gkind: int
    conn: np.ndarray      # (n_el, n_loc) int32, global basis index per local basis
    gconn: np.ndarray     # (n_el, n_gloc) int32, geometry node per local geometry basis
    xyz: np.ndarray       # (n_gnodes, dim) float64 geometry node positions (dim = 2 for triangles)
    n_nodes: int          # number of global (scalar) bases
    node_xyz: np.ndarray  # (n_nodes, dim) position of every basis node (for boundary conditions only)
    quad_order: int = -1

    @property
    def n_el(self) -> int:
        return int(self.conn.shape[0])

    @property
    def n_loc(self) -> int:
        return int(self.conn.shape[1])


def _kuhn_tets():
    """The six tetrahedra of the Kuhn/Freudenthal split of the unit cube, as corner offsets, each with a
    positive Jacobian determinant."""
    tets = []
    for perm in itertools.permutations(range(3)):
        v = [np.zeros(3, dtype=np.int64)]
        for ax in perm:
            nxt = v[-1].copy()
            nxt[ax] += 1
            v.append(nxt)
        v = np.array(v)
        det = np.linalg.det((v[1:] - v[0]).astype(float))
        if det < 0:
            v[[2, 3]] = v[[3, 2]]
        tets.append(v)
    return np.array(tets)  # (6, 4, 3)


def _renumber_first_touch(conn_lattice: np.ndarray, vertex_lattice_ids: np.ndarray):
    """polyfem-style numbering: vertices first, then other nodes in order of first appearance."""
    flat = conn_lattice.reshape(-1)
    n_lat = int(flat.max()) + 1
    new_id = np.full(n_lat, -1, dtype=np.int64)
    new_id[vertex_lattice_ids] = np.arange(vertex_lattice_ids.size)
    is_other = new_id[flat] < 0
    others = flat[is_other]
    uniq, first = np.unique(others, return_index=True)
    order = np.argsort(first, kind="stable")
    new_id[uniq[order]] = vertex_lattice_ids.size + np.arange(uniq.size)
    return new_id


def _finish(kind, conn_lat, gconn, xyz, lat_dim, numbering, h_lat):
    used = np.unique(conn_lat)
    if numbering == "lattice":
        # compress in case some lattice points are unused (never happens for the generators below)
        new_id = np.full(int(conn_lat.max()) + 1, -1, dtype=np.int64)
        new_id[used] = np.arange(used.size)
    elif numbering == "polyfem":
        L = lat_dim
        step = 2 if kind in (P2TET, Q2HEX) else 1
        ii = np.arange(0, L, step)
        vz, vy, vx = np.meshgrid(ii, ii, ii, indexing="ij")
        vert_ids = (vx + L * (vy + L * vz)).reshape(-1)
        new_id = _renumber_first_touch(conn_lat, vert_ids)
    else:
        raise ValueError("numbering must be 'lattice' or 'polyfem'")
    conn = new_id[conn_lat].astype(np.int32)
    n_nodes = int(used.size)
    node_xyz = np.zeros((n_nodes, 3))
    L = lat_dim
    lx, ly, lz = used % L, (used // L) % L, used // (L * L)
    node_xyz[new_id[used]] = np.stack([lx, ly, lz], axis=1) * h_lat
    return Mesh(kind, GEOM_KIND[kind], np.ascontiguousarray(conn), np.ascontiguousarray(gconn.astype(np.int32)),
                np.ascontiguousarray(xyz), n_nodes, node_xyz)


def cube_tets(n: int, order: int = 1, numbering: str = "lattice", scale=(1.0, 1.0, 1.0), jitter: float = 0.0,
              seed: int = 0) -> Mesh:
    """Unit cube split in n^3 cells, each cell into 6 Kuhn tetrahedra; P1 (order=1) or P2 (order=2) bases on
    the linear geometry.  `jitter` moves interior geometry vertices by a seeded uniform offset
    (fraction of the cell size) so that tests can run on non-axis-aligned elements."""
    assert order in (1, 2)
    kind = P1TET if order == 1 else P2TET
    tets = _kuhn_tets()  # (6,4,3)
    kz, ky, kx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    cell = np.stack([kx, ky, kz], axis=-1).reshape(-1, 1, 1, 3)  # cells in x-fastest order
    corners = cell + tets[None]  # (n^3, 6, 4, 3) integer vertex coords
    corners = corners.reshape(-1, 4, 3)
    nv = n + 1
    gconn = corners[..., 0] + nv * (corners[..., 1] + nv * corners[..., 2])
    gz, gy, gx = np.meshgrid(np.arange(nv), np.arange(nv), np.arange(nv), indexing="ij")
    xyz = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float64) / n
    if jitter > 0:
        rng = np.random.default_rng(seed)
        interior = np.all((xyz > 0) & (xyz < 1), axis=1)
        xyz[interior] += rng.uniform(-jitter / n, jitter / n, size=(int(interior.sum()), 3))
    xyz = xyz * np.asarray(scale, dtype=np.float64)
    if order == 1:
        lat_dim = nv
        conn_lat = gconn.copy()
        h_lat = 1.0 / n
    else:
        lat_dim = 2 * n + 1
        c2 = 2 * corners  # lattice coords of the vertices
        loc = [c2[:, k] for k in range(4)]
        for a, b in _P2_EDGES:
            loc.append((c2[:, a] + c2[:, b]) // 2)
        loc = np.stack(loc, axis=1)  # (n_el, 10, 3)
        conn_lat = loc[..., 0] + lat_dim * (loc[..., 1] + lat_dim * loc[..., 2])
        h_lat = 0.5 / n
    m = _finish(kind, conn_lat, gconn, xyz, lat_dim, numbering, h_lat)
    m.node_xyz = m.node_xyz * np.asarray(scale, dtype=np.float64)
    return m


def cube_hexes(n: int, order: int = 1, numbering: str = "lattice", scale=(1.0, 1.0, 1.0)) -> Mesh:
    """Structured n^3 hexahedra on the unit cube with Q1 (order=1) or Q2 (order=2) bases, trilinear geometry."""
    assert order in (1, 2)
    kind = Q1HEX if order == 1 else Q2HEX
    kz, ky, kx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    cell = np.stack([kx, ky, kz], axis=-1).reshape(-1, 1, 3)
    nv = n + 1
    corners = cell + _Q1_CORNERS[None]
    gconn = corners[..., 0] + nv * (corners[..., 1] + nv * corners[..., 2])
    gz, gy, gx = np.meshgrid(np.arange(nv), np.arange(nv), np.arange(nv), indexing="ij")
    xyz = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float64) / n
    xyz = xyz * np.asarray(scale, dtype=np.float64)
    if order == 1:
        lat_dim, conn_lat, h_lat = nv, gconn.copy(), 1.0 / n
    else:
        lat_dim = 2 * n + 1
        loc = 2 * cell + _Q2_NODES[None]
        conn_lat = loc[..., 0] + lat_dim * (loc[..., 1] + lat_dim * loc[..., 2])
        h_lat = 0.5 / n
    m = _finish(kind, conn_lat, gconn, xyz, lat_dim, numbering, h_lat)
    m.node_xyz = m.node_xyz * np.asarray(scale, dtype=np.float64)
    return m


def square_tris(n: int, order: int = 1, numbering: str = "lattice", scale=(1.0, 1.0), jitter: float = 0.0, seed: int = 0) -> Mesh:
    """Unit square split in n^2 cells, each cell into two counter-clockwise triangles along its diagonal; P1 (order=1) or
    P2 (order=2) bases on the linear geometry, local node order of `autogen/auto_p_bases.cpp:138-144, 275-284`.  The 2-D
    meshes of the reference's own assembler tests (`tests/test_assembler.cpp:12-64`, `tests/test_form_derivatives.cpp:31-81`)
    are triangle meshes of this kind.  `jitter` moves the interior vertices as in cube_tets."""
    assert order in (1, 2)
    kind = P1TRI if order == 1 else P2TRI
    tris = np.array([[(0, 0), (1, 0), (1, 1)], [(0, 0), (1, 1), (0, 1)]])
    ky, kx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    cell = np.stack([kx, ky], axis=-1).reshape(-1, 1, 1, 2)
    corners = (cell + tris[None]).reshape(-1, 3, 2)
    nv = n + 1
    gconn = corners[..., 0] + nv * corners[..., 1]
    gy, gx = np.meshgrid(np.arange(nv), np.arange(nv), indexing="ij")
    xyz = np.stack([gx, gy], axis=-1).reshape(-1, 2).astype(np.float64) / n
    if jitter > 0:
        rng = np.random.default_rng(seed)
        interior = np.all((xyz > 0) & (xyz < 1), axis=1)
        xyz[interior] += rng.uniform(-jitter / n, jitter / n, size=(int(interior.sum()), 2))
    xyz = xyz * np.asarray(scale, dtype=np.float64)
    if order == 1:
        L, conn_lat, h_lat = nv, gconn.copy(), 1.0 / n
    else:
        L = 2 * n + 1
        c2 = 2 * corners
        loc = [c2[:, k] for k in range(3)] + [(c2[:, a] + c2[:, b]) // 2 for a, b in _P2_TRI_EDGES]
        loc = np.stack(loc, axis=1)
        conn_lat = loc[..., 0] + L * loc[..., 1]
        h_lat = 0.5 / n
    used = np.unique(conn_lat)
    if numbering == "lattice":
        new_id = np.full(int(conn_lat.max()) + 1, -1, dtype=np.int64)
        new_id[used] = np.arange(used.size)
    elif numbering == "polyfem":
        ii = np.arange(0, L, order)
        vy, vx = np.meshgrid(ii, ii, indexing="ij")
        new_id = _renumber_first_touch(conn_lat, (vx + L * vy).reshape(-1))
    else:
        raise ValueError("numbering must be 'lattice' or 'polyfem'")
    node_xyz = np.zeros((used.size, 2))
    node_xyz[new_id[used]] = np.stack([used % L, used // L], axis=1) * h_lat * np.asarray(scale, dtype=np.float64)
    return Mesh(kind, P1TRI, np.ascontiguousarray(new_id[conn_lat].astype(np.int32)), np.ascontiguousarray(gconn.astype(np.int32)),
                np.ascontiguousarray(xyz), int(used.size), node_xyz)


def node_positions(mesh: Mesh) -> np.ndarray:
    """Physical position of every basis node (the geometry map at the node's reference position); on tetrahedra with
    moved vertices these differ from the lattice positions kept in mesh.node_xyz."""
    if mesh.kind in (P1TRI, P2TRI):
        v = mesh.xyz[mesh.gconn]
        pos = np.zeros((mesh.n_nodes, 2))
        for k in range(3):
            pos[mesh.conn[:, k]] = v[:, k]
        if mesh.kind == P2TRI:
            for e, (a, b) in enumerate(_P2_TRI_EDGES):
                pos[mesh.conn[:, 3 + e]] = 0.5 * (v[:, a] + v[:, b])
        return pos
    if mesh.kind not in (P1TET, P2TET):
        return np.asarray(mesh.node_xyz, dtype=float)
    v = mesh.xyz[mesh.gconn]
    pos = np.zeros((mesh.n_nodes, 3))
    for k in range(4):
        pos[mesh.conn[:, k]] = v[:, k]
    if mesh.kind == P2TET:
        for e, (a, b) in enumerate(_P2_EDGES):
            pos[mesh.conn[:, 4 + e]] = 0.5 * (v[:, a] + v[:, b])
    return pos


def boundary_dofs(mesh: Mesh, dim: int, face: str = "all") -> np.ndarray:
    """Sorted dof indices (node*dim + c) of the basis nodes on the cube boundary ('all') or on the x=0 face
    ('x0'); the analogue of the reference's `boundary_nodes` list (`state/StateSolveLinear.cpp:27-29`)."""
    p = mesh.node_xyz
    lo, hi = p.min(axis=0), p.max(axis=0)
    tol = 1e-12 * max(1.0, float(np.abs(hi - lo).max()))
    if face == "all":
        on = np.any((np.abs(p - lo) < tol) | (np.abs(p - hi) < tol), axis=1)
    elif face == "x0":
        on = np.abs(p[:, 0] - lo[0]) < tol
    else:
        raise ValueError(face)
    nodes = np.nonzero(on)[0]
    return np.sort((nodes[:, None] * dim + np.arange(dim)[None]).reshape(-1)).astype(np.int32)


def seeded_displacement(mesh: Mesh, dim: int, amplitude: float, seed: int = 12345) -> np.ndarray:
    """Node-major interleaved displacement x[node*dim + c] = N(0,1) * amplitude (PCG64 stream)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(mesh.n_nodes * dim) * amplitude

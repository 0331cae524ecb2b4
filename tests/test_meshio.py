"""Gmsh reader / flattening (input side of the path): round trips through both formats in ASCII and binary, the
first-touch numbering rule of MeshNodes, P2 / Q2 node sets against the generators, body ids, re-orientation, rejected
inputs.  The reference
parses MSH files with the un-vendored MshIO library and holds no mesh fixture: parity unpinned, checked for
self-consistency only."""
import numpy as np
import pytest

from polyfem_amd import mesh as M
from polyfem_amd import meshio as IO


@pytest.mark.parametrize("version", ["2.2", "4.1"])
def test_msh_round_trip_and_p2_numbering(tmp_path, version):
    ref1 = M.cube_tets(2, 1, jitter=0.1)
    body = (np.arange(ref1.n_el) >= ref1.n_el // 3).astype(np.int32) * 7 + 1  # two bodies, contiguous in the file
    path = tmp_path / "cube.msh"
    IO.write_msh(path, ref1.xyz, ref1.gconn, body, version)
    v, c, b = IO.read_msh(path)
    assert np.array_equal(v, ref1.xyz) and np.array_equal(c, ref1.gconn) and np.array_equal(b, body)
    m1 = IO.flatten(v, ref1.gconn, 1)
    assert m1.kind == M.P1TET and np.array_equal(m1.conn, ref1.conn) and m1.n_nodes == ref1.n_nodes
    # P2: vertices first, then edge nodes in first-touch order == the generator's "polyfem" numbering
    ref2 = M.cube_tets(2, 2, numbering="polyfem", jitter=0.1)
    m2 = IO.flatten(v, ref1.gconn, 2)
    assert m2.kind == M.P2TET and m2.n_nodes == ref2.n_nodes
    assert np.array_equal(m2.conn, ref2.conn) and np.array_equal(m2.gconn, ref2.gconn)
    mid = 0.5 * (v[m2.gconn[:, 0]] + v[m2.gconn[:, 1]])
    assert np.allclose(m2.node_xyz[m2.conn[:, 4]], mid)
    mesh, b2 = IO.load(path, 2)
    assert mesh.n_nodes == ref2.n_nodes and b2.size == ref1.n_el


def test_msh_hexes_sparse_tags_and_orientation(tmp_path):
    ref = M.cube_hexes(2, 1)
    path = tmp_path / "hex.msh"
    IO.write_msh(path, ref.xyz, ref.gconn, None, "4.1")
    m, body = IO.load(path, 1)
    assert m.kind == M.Q1HEX and np.array_equal(m.conn, ref.conn) and not body.any()
    with pytest.raises(ValueError):
        IO.flatten(ref.xyz, ref.gconn, 3)
    # node tags with gaps are condensed in file order; lower-dimensional elements are ignored; higher-order
    # tetrahedra contribute their four vertices
    path = tmp_path / "sparse.msh"
    path.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n5\n10 0 0 0\n20 1 0 0\n30 0 1 0\n40 0 0 1\n50 1 1 1\n$EndNodes\n"
                    "$Elements\n3\n1 2 2 5 5 10 20 30\n2 4 2 3 3 10 30 20 40\n3 4 2 4 4 20 30 40 50\n$EndElements\n")
    v, c, b = IO.read_msh(path)
    assert v.shape == (5, 3) and np.array_equal(c, [[0, 2, 1, 3], [1, 2, 3, 4]]) and np.array_equal(b, [3, 4])
    m = IO.flatten(v, c, 1)
    p = v[m.gconn]
    det = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
    assert np.all(det > 0), "inverted tetrahedra are re-oriented"
    bad = tmp_path / "bad.msh"
    bad.write_text("$MeshFormat\n2.2 1 8\n$EndMeshFormat\n")
    with pytest.raises(ValueError):
        IO.read_msh(bad)
    surf = tmp_path / "surf.msh"
    surf.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n3\n1 0 0 0\n2 1 0 0\n3 0 1 0\n$EndNodes\n$Elements\n1\n1 2 2 0 0 1 2 3\n$EndElements\n")
    v, c, b = IO.read_msh(surf)  # a triangle mesh is a 2-D problem
    assert c.shape == (1, 3) and IO.flatten(v, c, 2).kind == M.P2TRI
    quad = tmp_path / "quad.msh"
    quad.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n4\n1 0 0 0\n2 1 0 0\n3 1 1 0\n4 0 1 0\n$EndNodes\n$Elements\n1\n1 3 2 0 0 1 2 3 4\n$EndElements\n")
    with pytest.raises(ValueError):
        IO.read_msh(quad)  # quadrilaterals are not implemented


def test_flattened_file_mesh_assembles_like_the_generated_one(tmp_path, oracle):
    """The oracle's stiffness matrix of the mesh read from a file equals that of the generated mesh (same numbering)."""
    ref = M.cube_tets(2, 2, numbering="polyfem", jitter=0.15)
    base = M.cube_tets(2, 1, jitter=0.15)
    path = tmp_path / "c.msh"
    IO.write_msh(path, base.xyz, base.gconn, None, "4.1")
    mesh, _ = IO.load(path, 2)
    lam, mu = oracle.lame(1e5, 0.3)
    a = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    b = oracle.assemble_linear(ref, oracle.LINEAR_ELASTICITY, lam, mu)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("version", ["2.2", "4.1"])
@pytest.mark.parametrize("hexes", [False, True])
def test_binary_msh_round_trip(tmp_path, version, hexes):
    ref = M.cube_hexes(2, 1) if hexes else M.cube_tets(2, 1, jitter=0.1)
    body = (np.arange(ref.n_el) % 3 == 0).astype(np.int32) * 4 + 2
    body = np.sort(body)  # bodies contiguous in the file (4.1 stores one entity block per body)
    pa, pb = tmp_path / "a.msh", tmp_path / "b.msh"
    IO.write_msh(pa, ref.xyz, ref.gconn, body, version)
    IO.write_msh(pb, ref.xyz, ref.gconn, body, version, binary=True)
    va, ca, ba = IO.read_msh(pa)
    vb, cb, bb = IO.read_msh(pb)
    assert np.array_equal(va, vb) and np.array_equal(ca, cb) and np.array_equal(ba, bb)
    assert np.array_equal(vb, ref.xyz) and np.array_equal(cb, ref.gconn) and np.array_equal(bb, body)
    raw = pb.read_bytes()
    (tmp_path / "cut.msh").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        IO.read_msh(tmp_path / "cut.msh")


def _first_occurrences_ascending(conn):
    flat = conn.reshape(-1)
    _, first = np.unique(flat, return_index=True)
    return bool(np.all(np.diff(first) > 0))  # node k is first touched before node k + 1


@pytest.mark.parametrize("kind", ["p2", "q2"])
def test_first_touch_numbering_and_q2_flattening(kind, oracle):
    """MeshNodes::node_id_from_primitive: ids in the order the element loop first asks for a primitive (vertices, edges,
    faces, cell of each element in local order).  The flattened Q2 / P2 mesh is the generated one up to the node
    renumbering: same node positions, and the oracle's stiffness matrices agree after the permutation."""
    gen = M.cube_hexes(2, 2) if kind == "q2" else M.cube_tets(2, 2)
    base = M.cube_hexes(2, 1) if kind == "q2" else M.cube_tets(2, 1)
    for numbering in ("first_touch", "vertices_first"):
        m = IO.flatten(base.xyz, base.gconn, 2, numbering)
        assert m.kind == gen.kind and m.n_nodes == gen.n_nodes and m.conn.shape == gen.conn.shape
        if numbering == "first_touch":
            assert _first_occurrences_ascending(m.conn) and m.conn[0, 0] == 0 and np.array_equal(m.conn[0], np.arange(m.conn.shape[1]))
        else:
            assert np.array_equal(m.conn[:, : base.gconn.shape[1]], base.gconn)  # vertex id = node id
        # node k of an element sits where the generator's node k of that element sits
        assert np.allclose(m.node_xyz[m.conn], gen.node_xyz[gen.conn], atol=1e-14)
        perm = np.empty(gen.n_nodes, dtype=np.int64)  # generator id -> flattened id
        perm[gen.conn.reshape(-1)] = m.conn.reshape(-1)
        lam, mu = oracle.lame(1e5, 0.3)
        n = gen.n_nodes
        from tests.helpers import csc_to_scipy
        A = csc_to_scipy(oracle.assemble_linear(m, oracle.LAPLACIAN, lam, mu), n).toarray()
        B = csc_to_scipy(oracle.assemble_linear(gen, oracle.LAPLACIAN, lam, mu), n).toarray()
        assert np.abs(A[np.ix_(perm, perm)] - B).max() <= 1e-13 * np.abs(B).max()


@pytest.mark.gpu
@pytest.mark.parametrize("element", ["p2tet", "q2hex"])
@pytest.mark.parametrize("binary", [False, True])
def test_file_mesh_through_the_hip_path(tmp_path, oracle, element, binary):
    """A mesh FILE (jittered tetrahedra / hexahedra, MSH 4.1), loaded with the node numbering MeshNodes hands out while the
    bases are built (src/polyfem/mesh/MeshNodes.cpp:157-308, src/polyfem/basis/FEBasis3d.cpp:201-395), assembled on the HIP
    path: NeoHookean Hessian and LinearElasticity stiffness, pattern bit-exact and values to 1e-12 against the oracle on
    the same flattened arrays.  The reference parses MSH through the un-vendored MshIO and holds no mesh file: the
    flattening itself stays parity unpinned."""
    from polyfem_amd import engine as E
    from tests.helpers import rel_err
    if element == "p2tet":
        src = M.cube_tets(3, 1, jitter=0.15, seed=4)
    else:
        src = M.cube_hexes(2, 1)
        rng = np.random.default_rng(5)
        inner = np.all((src.xyz > 1e-9) & (src.xyz < 1 - 1e-9), axis=1)
        src.xyz[inner] += 0.08 * (rng.random((int(inner.sum()), 3)) - 0.5)  # trilinear, no longer affine
    path = tmp_path / "mesh.msh"
    IO.write_msh(path, src.xyz, src.gconn, None, "4.1", binary=binary)
    mesh, body = IO.load(path, 2, numbering="first_touch")
    assert mesh.kind == (M.P2TET if element == "p2tet" else M.Q2HEX) and body.size == src.n_el
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        lam, mu = oracle.lame(2e4, 0.3)
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.build_pattern()
        x = M.seeded_displacement(mesh, 3, 2e-3)
        v = eng.hessian(x)
        rp, ci = eng.get_pattern()
        outer, inner_idx, vals = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner_idx)
        assert rel_err(v, vals) <= 1e-12
        assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= 1e-12
        lam, mu = oracle.lame(1e5, 0.3)
        eng.set_material(E.LINEAR_ELASTICITY, lam, mu)
        eng.build_pattern()
        eng.assemble_linear()
        rp, ci, v = eng.get_linear_csr()
        outer, inner_idx, vals = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner_idx)
        assert rel_err(v, vals) <= 1e-12
    finally:
        eng.close()


@pytest.mark.parametrize("fmt", ["msh2", "msh4", "msh4bin", "obj"])
def test_triangle_mesh_files(tmp_path, oracle, fmt):
    """2-D meshes: triangles from MSH (2.2 / 4.1, ASCII / binary) and OBJ files, some of them clockwise in the file; the
    flattened P2 mesh assembles to the matrix of the generated mesh up to the node permutation."""
    gen = M.square_tris(4, 2, jitter=0.2, seed=6)
    src = M.square_tris(4, 1, jitter=0.2, seed=6)
    cells = src.gconn.copy()
    cells[::3] = cells[::3][:, [0, 2, 1]]  # clockwise triangles: flatten re-orients them
    path = tmp_path / ("mesh.obj" if fmt == "obj" else "mesh.msh")
    if fmt == "obj":
        with open(path, "w") as fh:
            fh.write("# test mesh\n")
            for p in src.xyz:
                fh.write("v %.17g %.17g 0\n" % (p[0], p[1]))
            for c in cells:
                fh.write("f %d/1 %d/1 %d/1\n" % (c[0] + 1, c[1] + 1, c[2] + 1))
    else:
        IO.write_msh(path, src.xyz, cells, None, "2.2" if fmt == "msh2" else "4.1", binary=fmt == "msh4bin")
    mesh, body = IO.load(path, 2)
    assert mesh.kind == M.P2TRI and mesh.xyz.shape[1] == 2 and body.size == src.n_el and mesh.n_nodes == gen.n_nodes
    assert np.array_equal(mesh.xyz, src.xyz)
    # same node set, different numbering: match the nodes by position
    def key(p):
        return np.round(p * 1e9).astype(np.int64)
    a, b = key(M.node_positions(mesh)), key(M.node_positions(gen))
    ia, ib = np.lexsort(a.T), np.lexsort(b.T)
    assert np.array_equal(a[ia], b[ib])
    perm = np.empty(gen.n_nodes, dtype=np.int64)  # perm[node of gen] = node of mesh
    perm[ib] = ia
    lam, mu = oracle.lame(1e5, 0.3, 2)
    from tests.helpers import csc_to_scipy
    n = 2 * gen.n_nodes
    A = csc_to_scipy(oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu), n).toarray()
    B = csc_to_scipy(oracle.assemble_linear(gen, oracle.LINEAR_ELASTICITY, lam, mu), n).toarray()
    dof = (2 * perm[:, None] + np.arange(2)[None]).reshape(-1)
    assert np.abs(A[np.ix_(dof, dof)] - B).max() <= 1e-12 * np.abs(B).max()


@pytest.mark.gpu
def test_triangle_file_mesh_through_the_hip_path(tmp_path, oracle):
    """An OBJ triangle mesh (the format of the reference's 2-D test meshes), first-touch numbering, P2, on the HIP path."""
    from polyfem_amd import engine as E
    from tests.helpers import rel_err
    src = M.square_tris(7, 1, jitter=0.25, seed=8)
    path = tmp_path / "mesh.obj"
    with open(path, "w") as fh:
        for p in src.xyz:
            fh.write("v %.17g %.17g 0\n" % (p[0], p[1]))
        for c in src.gconn:
            fh.write("f %d %d %d\n" % (c[0] + 1, c[2] + 1, c[1] + 1))  # clockwise in the file
    mesh, _ = IO.load(path, 2, numbering="first_touch")
    eng = E.Engine(0)
    try:
        eng.set_mesh(mesh)
        lam, mu = oracle.lame(2e4, 0.3, 2)
        eng.set_material(E.NEOHOOKEAN, lam, mu)
        eng.build_pattern()
        x = M.seeded_displacement(mesh, 2, 2e-3)
        v = eng.hessian(x)
        rp, ci = eng.get_pattern()
        outer, inner_idx, vals = oracle.assemble_hessian(mesh, oracle.NEOHOOKEAN, lam, mu, x)
        assert np.array_equal(rp, outer.astype(np.int64)) and np.array_equal(ci, inner_idx)
        assert rel_err(v, vals) <= 1e-12
        assert rel_err(eng.gradient(x), oracle.assemble_gradient(mesh, oracle.NEOHOOKEAN, lam, mu, x)) <= 1e-12
    finally:
        eng.close()


@pytest.mark.parametrize("element", ["tet", "hex", "tri"])
def test_medit_mesh_files(tmp_path, oracle, element):
    """ASCII MEDIT `.mesh` files (read by geogram in the reference, un-vendored: parity unpinned): round trip, the same
    flattened mesh as the MSH path -- and therefore the same oracle matrix --, keywords in any case, `Dimension` on its
    own line, comments, lower-dimensional sections ignored, rejected inputs."""
    if element == "tet":
        base = M.cube_tets(2, 1, jitter=0.1)
    elif element == "hex":
        base = M.cube_hexes(2, 1)
    else:
        base = M.square_tris(3, 1, jitter=0.1)
    path = tmp_path / "m.mesh"
    IO.write_medit(path, base.xyz, base.gconn)
    v, c, b = IO.read_medit(path)
    assert np.array_equal(v[:, :base.xyz.shape[1]], base.xyz) and np.array_equal(c, base.gconn) and not b.any()
    mesh, body = IO.load(path, 2)
    msh = tmp_path / "m.msh"
    IO.write_msh(msh, base.xyz, base.gconn, None, "2.2")
    ref, _ = IO.load(msh, 2)
    assert mesh.kind == ref.kind and np.array_equal(mesh.conn, ref.conn) and np.array_equal(mesh.gconn, ref.gconn)
    assert np.array_equal(mesh.xyz, ref.xyz) and body.size == base.n_el
    dim = M.KIND_DIM[mesh.kind]
    lam, mu = oracle.lame(1e5, 0.3, dim)
    a = oracle.assemble_linear(mesh, oracle.LINEAR_ELASTICITY, lam, mu)
    r = oracle.assemble_linear(ref, oracle.LINEAR_ELASTICITY, lam, mu)
    assert all(np.array_equal(x, y) for x, y in zip(a, r))


def test_medit_variants_and_rejected_inputs(tmp_path):
    p = tmp_path / "v.mesh"
    p.write_text("# a comment\nMeshVersionFormatted 2\nDimension\n3\nVertices\n5\n0 0 0 1\n1 0 0 1\n0 1 0 1\n0 0 1 2\n1 1 1 2\n"
                 "Edges\n1\n1 2 5\nTriangles\n1\n1 2 3 9\nTETRAHEDRA\n2\n1 3 2 4 7\n2 3 4 5 8\nCorners\n1\n1\nEnd\n")
    v, c, b = IO.read_medit(p)
    assert v.shape == (5, 3) and np.array_equal(c, [[0, 2, 1, 3], [1, 2, 3, 4]]) and not b.any()
    m = IO.flatten(v, c, 1)  # the first tetrahedron is inverted in the file
    q = v[m.gconn]
    assert np.all(np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0]) > 0)
    two = tmp_path / "t.mesh"
    two.write_text("MeshVersionFormatted 1\nDimension 2\nVertices\n3\n0 0 0\n1 0 0\n0 1 0\nTriangles\n1\n1 2 3 0\nEnd\n")
    v, c, _ = IO.read_medit(two)
    assert c.shape == (1, 3) and IO.flatten(v, c, 2).kind == M.P2TRI
    for text in ("$MeshFormat\n2.2 0 8\n", "MeshVersionFormatted 1\nDimension 3\nVertices\n2\n0 0 0 0\n",
                 "MeshVersionFormatted 1\nDimension 3\nVertices\n1\n0 0 0 0\nTetrahedra\n1\n1 2 3 4 0\nEnd\n",
                 "MeshVersionFormatted 1\nDimension 2\nVertices\n4\n0 0 0\n1 0 0\n1 1 0\n0 1 0\nQuadrilaterals\n1\n1 2 3 4 0\nEnd\n",
                 "MeshVersionFormatted 1\nDimension 3\nVertices\n4\n0 0 0 0\n1 0 0 0\n0 1 0 0\n0 0 1 0\nPrisms\n0\nEnd\n"):
        bad = tmp_path / "bad.mesh"
        bad.write_text(text)
        with pytest.raises(ValueError):
            IO.read_medit(bad)

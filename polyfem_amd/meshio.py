"""Volume meshes from Gmsh files, flattened to the layout the assembly engine consumes.

Input side of the hot path: what the reference does between reading a mesh and calling its assemblers.

  * `read_msh` follows `io/MshReader.cpp:29-160` (which parses with the un-vendored MshIO library): the elements of
    the highest dimension present become `cells` (first 4 nodes of every tetrahedron type, first 8 of every
    hexahedron type), node tags become 0-based vertex ids (tag - 1, or file order when tags are not dense, as the
    reference does with a warning), and the body id of a cell is the first physical tag of its entity.  ASCII files of
    format 2.2 and 4.1 are read; binary files are rejected.
  * `flatten` stands in for `FEBasis3d::build_bases` + `mesh/MeshNodes.cpp:157-308` on conforming meshes: every mesh
    vertex is a node (vertex id = node id), P2 edge nodes follow in the order in which the element loop first touches
    them, local node order is that of the reference's basis functions (`autogen/auto_p_bases.cpp`: vertices, then the
    edges (0,1) (1,2) (2,0) (0,3) (1,3) (2,3)), geometry map linear / trilinear on the cell's own vertices.
    Elements with a negative Jacobian are re-oriented by swapping two vertices.

No reference fixture pins this numbering (the reference's meshes live in the external polyfem-data repository), so
results on real meshes are comparable to a PolyFEM run only up to a node permutation.
"""
from __future__ import annotations

import numpy as np

from . import mesh as M

# Gmsh element type -> (dimension, nodes per element, leading nodes that are the cell's vertices)
_ELEMENT_TYPES = {
    15: (0, 1, 1), 1: (1, 2, 2), 8: (1, 3, 2), 26: (1, 4, 2), 27: (1, 5, 2),
    2: (2, 3, 3), 9: (2, 6, 3), 21: (2, 10, 3), 23: (2, 15, 3), 25: (2, 21, 3), 3: (2, 4, 4), 10: (2, 9, 4), 16: (2, 8, 4),
    4: (3, 4, 4), 11: (3, 10, 4), 29: (3, 20, 4), 30: (3, 35, 4), 31: (3, 56, 4), 5: (3, 8, 8), 12: (3, 27, 8), 17: (3, 20, 8),
    6: (3, 6, 6), 7: (3, 5, 5), 13: (3, 18, 6), 14: (3, 14, 5),
}


def _sections(path):
    with open(path, "rb") as fh:
        raw = fh.read()
    try:
        text = raw.decode("ascii")
    except UnicodeDecodeError as exc:
        raise ValueError(f"{path}: binary MSH files are not supported") from exc
    out, name, buf = {}, None, []
    for line in text.splitlines():
        s = line.strip()
        if not s:
            continue
        if s.startswith("$End"):
            out[name] = buf
            name, buf = None, []
        elif s.startswith("$"):
            name, buf = s[1:], []
        elif name is not None:
            buf.append(s)
    return out


def read_msh(path):
    """Returns (vertices (n, 3) float64, cells (m, 4 | 8) int32, body_ids (m,) int32)."""
    sec = _sections(path)
    if "MeshFormat" not in sec:
        raise ValueError(f"{path}: not an MSH file")
    version, file_type = sec["MeshFormat"][0].split()[:2]
    if int(file_type) != 0:
        raise ValueError(f"{path}: binary MSH files are not supported")
    major = int(float(version))
    tags, xyz, blocks = [], [], []  # blocks: (dim, physical tag, type, [node tags per element])
    if major == 2:
        lines = sec["Nodes"]
        for ln in lines[1:1 + int(lines[0])]:
            t = ln.split()
            tags.append(int(t[0]))
            xyz.append((float(t[1]), float(t[2]), float(t[3])))
        lines = sec["Elements"]
        for ln in lines[1:1 + int(lines[0])]:
            t = ln.split()
            etype, ntags = int(t[1]), int(t[2])
            if etype not in _ELEMENT_TYPES:
                raise ValueError(f"{path}: unsupported element type {etype}")
            phys = int(t[3]) if ntags > 0 else 0
            blocks.append((_ELEMENT_TYPES[etype][0], phys, etype, [[int(v) for v in t[3 + ntags:]]]))
    elif major == 4:
        phys_of = {}  # (dim, entity tag) -> first physical tag
        if "Entities" in sec:
            lines = sec["Entities"]
            counts = [int(v) for v in lines[0].split()]
            pos = 1
            for dim, cnt in enumerate(counts):
                for _ in range(cnt):
                    t = lines[pos].split()
                    pos += 1
                    k = 4 if dim == 0 else 7  # tag + point coordinates | tag + bounding box
                    nphys = int(t[k])
                    phys_of[(dim, int(t[0]))] = int(t[k + 1]) if nphys > 0 else 0
        lines = sec["Nodes"]
        nblocks = int(lines[0].split()[0])
        pos = 1
        for _ in range(nblocks):
            nb = int(lines[pos].split()[3])
            pos += 1
            tags.extend(int(lines[pos + k]) for k in range(nb))
            pos += nb
            for k in range(nb):
                t = lines[pos + k].split()
                xyz.append((float(t[0]), float(t[1]), float(t[2])))
            pos += nb
        lines = sec["Elements"]
        nblocks = int(lines[0].split()[0])
        pos = 1
        for _ in range(nblocks):
            edim, etag, etype, nb = (int(v) for v in lines[pos].split())
            pos += 1
            if etype not in _ELEMENT_TYPES:
                raise ValueError(f"{path}: unsupported element type {etype}")
            blocks.append((edim, phys_of.get((edim, etag), 0), etype, [[int(v) for v in lines[pos + k].split()[1:]] for k in range(nb)]))
            pos += nb
    else:
        raise ValueError(f"{path}: MSH format {version} is not supported (2.2 and 4.1 are)")

    n_vertices = len(tags)
    if n_vertices == 0 or not blocks:
        raise ValueError(f"{path}: no nodes or no elements")
    max_tag = max(tags)
    tag_to_index = np.full(max_tag + 1, -1, dtype=np.int64)
    tag_arr = np.asarray(tags, dtype=np.int64)
    # dense tags keep the input order tag - 1; otherwise the nodes are condensed in file order (io/MshReader.cpp:64-82)
    tag_to_index[tag_arr] = tag_arr - 1 if n_vertices == max_tag else np.arange(n_vertices)
    vertices = np.zeros((n_vertices, 3))
    vertices[tag_to_index[tag_arr]] = np.asarray(xyz, dtype=np.float64)

    dim = max(b[0] for b in blocks)
    if dim != 3:
        raise ValueError(f"{path}: only volume meshes are supported (highest element dimension is {dim})")
    cols = None
    cells, body = [], []
    for edim, phys, etype, elems in blocks:
        if edim != dim:
            continue
        nverts = _ELEMENT_TYPES[etype][2]
        if nverts not in (4, 8):
            raise ValueError(f"{path}: only tetrahedra and hexahedra are supported (element type {etype})")
        if cols is not None and cols != nverts:
            raise ValueError(f"{path}: mixed tetrahedra / hexahedra are not supported")
        cols = nverts
        for nodes in elems:
            if len(nodes) != _ELEMENT_TYPES[etype][1]:
                raise ValueError(f"{path}: element of type {etype} with {len(nodes)} nodes")
            cells.append(nodes[:nverts])
            body.append(phys)
    cells = tag_to_index[np.asarray(cells, dtype=np.int64)]
    if cells.min() < 0:
        raise ValueError(f"{path}: an element refers to a node tag that is not defined")
    return vertices, np.ascontiguousarray(cells.astype(np.int32)), np.asarray(body, dtype=np.int32)


def write_msh(path, vertices, cells, body_ids=None, version: str = "2.2"):
    """ASCII MSH writer for tetrahedra (4 columns) / hexahedra (8 columns); the counterpart of read_msh for tests and
    for exporting generated meshes."""
    vertices = np.asarray(vertices, dtype=np.float64)
    cells = np.asarray(cells)
    etype = {4: 4, 8: 5}[cells.shape[1]]
    body = np.zeros(cells.shape[0], dtype=np.int64) if body_ids is None else np.asarray(body_ids, dtype=np.int64)
    with open(path, "w") as fh:
        if version == "2.2":
            fh.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % vertices.shape[0])
            for k, p in enumerate(vertices):
                fh.write("%d %.17g %.17g %.17g\n" % (k + 1, p[0], p[1], p[2]))
            fh.write("$EndNodes\n$Elements\n%d\n" % cells.shape[0])
            for k, c in enumerate(cells):
                fh.write("%d %d 2 %d %d %s\n" % (k + 1, etype, body[k], body[k], " ".join(str(int(v) + 1) for v in c)))
            fh.write("$EndElements\n")
        elif version == "4.1":
            ids = sorted(set(int(b) for b in body))
            lo, hi = vertices.min(axis=0), vertices.max(axis=0)
            fh.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Entities\n0 0 0 %d\n" % len(ids))
            for t, b in enumerate(ids):
                fh.write("%d %.17g %.17g %.17g %.17g %.17g %.17g %s 0\n" % (t + 1, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2],
                                                                             ("1 %d" % b) if b != 0 else "0"))
            fh.write("$EndEntities\n$Nodes\n1 %d 1 %d\n3 1 0 %d\n" % (vertices.shape[0], vertices.shape[0], vertices.shape[0]))
            for k in range(vertices.shape[0]):
                fh.write("%d\n" % (k + 1))
            for p in vertices:
                fh.write("%.17g %.17g %.17g\n" % (p[0], p[1], p[2]))
            fh.write("$EndNodes\n$Elements\n%d %d 1 %d\n" % (len(ids), cells.shape[0], cells.shape[0]))
            tag = 1
            for t, b in enumerate(ids):
                sel = np.nonzero(body == b)[0]
                fh.write("3 %d %d %d\n" % (t + 1, etype, sel.size))
                for k in sel:
                    fh.write("%d %s\n" % (tag, " ".join(str(int(v) + 1) for v in cells[k])))
                    tag += 1
            fh.write("$EndElements\n")
        else:
            raise ValueError("version must be '2.2' or '4.1'")


def flatten(vertices, cells, order: int = 1) -> M.Mesh:
    """Mesh of P1 / P2 tetrahedra (cells with 4 columns) or Q1 hexahedra (8 columns, Gmsh corner order = the
    reference's) in the engine's layout."""
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    cells = np.array(cells, dtype=np.int64, copy=True)
    nv = vertices.shape[0]
    if cells.min() < 0 or cells.max() >= nv:
        raise ValueError("cell refers to a vertex that does not exist")
    if cells.shape[1] == 4:
        p = vertices[cells]
        det = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
        if np.any(det == 0):
            raise ValueError("degenerate tetrahedron")
        flip = det < 0
        cells[flip, 2], cells[flip, 3] = cells[flip, 3].copy(), cells[flip, 2].copy()
        if order == 1:
            return M.Mesh(M.P1TET, M.P1TET, np.ascontiguousarray(cells.astype(np.int32)), np.ascontiguousarray(cells.astype(np.int32)),
                          vertices, nv, vertices.copy())
        if order != 2:
            raise ValueError("tetrahedra: order 1 or 2")
        a = np.stack([cells[:, e[0]] for e in M._P2_EDGES], axis=1)
        b = np.stack([cells[:, e[1]] for e in M._P2_EDGES], axis=1)
        key = (np.minimum(a, b) * nv + np.maximum(a, b)).reshape(-1)  # element-major, local edge order
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        rank = np.empty(uniq.size, dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(uniq.size)  # first-touch order of the edges
        edge_node = nv + rank[inv].reshape(-1, 6)
        conn = np.concatenate([cells, edge_node], axis=1)
        n_nodes = nv + uniq.size
        if n_nodes >= 2 ** 31:
            raise ValueError("too many nodes for 32-bit indices")
        node_xyz = np.zeros((n_nodes, 3))
        node_xyz[:nv] = vertices
        node_xyz[nv + rank] = 0.5 * (vertices[uniq // nv] + vertices[uniq % nv])
        return M.Mesh(M.P2TET, M.P1TET, np.ascontiguousarray(conn.astype(np.int32)), np.ascontiguousarray(cells.astype(np.int32)),
                      vertices, n_nodes, node_xyz)
    if cells.shape[1] == 8:
        if order != 1:
            raise ValueError("hexahedra: only order 1 is flattened from files (Q2 needs edge, face and cell nodes)")
        p = vertices[cells]
        det = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 3] - p[:, 0]), p[:, 4] - p[:, 0])
        if np.any(det <= 0):
            raise ValueError("hexahedron with non-positive orientation at its first corner")
        c32 = np.ascontiguousarray(cells.astype(np.int32))
        return M.Mesh(M.Q1HEX, M.Q1HEX, c32, c32.copy(), vertices, nv, vertices.copy())
    raise ValueError("cells must have 4 (tetrahedra) or 8 (hexahedra) columns")


def load(path, order: int = 1):
    """read_msh + flatten; returns (Mesh, body_ids)."""
    vertices, cells, body = read_msh(path)
    return flatten(vertices, cells, order), body

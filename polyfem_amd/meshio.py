"""Volume meshes (and 2-D triangle meshes) from Gmsh / MEDIT / OBJ files, flattened to the layout the assembly engine consumes.

Input side of the hot path: what the reference does between reading a mesh and calling its assemblers.

  * `read_msh` follows `io/MshReader.cpp:29-160` (which parses with the un-vendored MshIO library): the elements of
    the highest dimension present become `cells` (first 4 nodes of every tetrahedron type, first 8 of every
    hexahedron type), node tags become 0-based vertex ids (tag - 1, or file order when tags are not dense, as the
    reference does with a warning), and the body id of a cell is the first physical tag of its entity.  ASCII and
    binary files of format 2.2 and 4.1 are read.
  * `flatten` stands in for `FEBasis3d::build_bases` + `mesh/MeshNodes.cpp:157-308` on conforming meshes.  Local node
    order is that of the reference's basis functions: tetrahedra (`basis/FEBasis3d.cpp:201-395`, `autogen/auto_p_bases.cpp`)
    vertices, then the edges (0,1) (1,2) (2,0) (0,3) (1,3) (2,3); hexahedra (`basis/FEBasis3d.cpp:397-520`,
    `autogen/auto_q_bases.cpp:1873-1903`) vertices, the edges (0,1) (1,2) (2,3) (3,0) (0,4) (1,5) (2,6) (3,7) (4,5) (5,6)
    (6,7) (7,4), the faces (0,3,4,7) (1,2,5,6) (0,1,5,4) (3,2,6,7) (0,1,2,3) (4,5,6,7), the cell.  Global numbering:
    `numbering="first_touch"` is MeshNodes' rule -- a primitive (vertex, edge, face, cell) gets the next free node id the
    first time the element loop asks for it, vertices included (`MeshNodes::node_id_from_primitive`, :157-175);
    `numbering="vertices_first"` (default) keeps vertex id = node id and numbers the other primitives in first-touch order
    after them.  Geometry map linear / trilinear on the cell's own vertices.  Tetrahedra with a negative Jacobian are
    re-oriented by swapping two vertices.

  * 2-D: a file whose highest element dimension is 2 and whose elements are triangles gives `cells` with 3 columns
    (`read_msh`, or `read_obj` for the Wavefront files the reference's 2-D tests load, `tests/test_assembler.cpp:24`);
    `flatten` keeps the first two coordinates (`mesh/mesh2D/CMesh2D.cpp`), orients every triangle counter-clockwise and
    numbers the P2 edge nodes in the order (0,1) (1,2) (2,0) of `autogen/auto_p_bases.cpp:275-284`
    (`basis/FEBasis2d.cpp`).

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


def _sections(text):
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


def _read_ascii(sec, path, major):
    """ASCII MSH 2.2 / 4.1: (node tags, coordinates, element blocks (dim, physical tag, type, node tags per element))."""
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

    return tags, xyz, blocks


class _Bin:
    """Cursor over the raw bytes of a binary MSH file."""

    def __init__(self, raw, path):
        self.raw, self.pos, self.path = raw, 0, path

    def section(self, name):
        k = self.raw.find(b"$" + name + b"\n")
        if k < 0:
            return False
        self.pos = k + len(name) + 2
        return True

    def line(self):
        e = self.raw.index(b"\n", self.pos)
        s = self.raw[self.pos:e].decode("ascii")
        self.pos = e + 1
        return s

    def take(self, dtype, count=1):
        dt = np.dtype(dtype)
        n = dt.itemsize * count
        if self.pos + n > len(self.raw):
            raise ValueError(f"{self.path}: truncated binary section")
        a = np.frombuffer(self.raw, dtype=dt, count=count, offset=self.pos)
        self.pos += n
        return a


def _read_binary(raw, path, major):
    """Binary MSH 2.2 / 4.1 (little endian, 8-byte floats): same (tags, xyz, blocks) as the ASCII branch."""
    b = _Bin(raw, path)
    b.section(b"MeshFormat")
    b.line()
    if int(b.take("<i4")[0]) != 1:
        raise ValueError(f"{path}: binary MSH file with the other byte order")
    tags, xyz, blocks = [], [], []
    if major == 2:
        if not b.section(b"Nodes"):
            raise ValueError(f"{path}: no $Nodes section")
        n = int(b.line())
        rec = b.take(np.dtype([("tag", "<i4"), ("p", "<f8", 3)]), n)
        tags, xyz = rec["tag"].astype(np.int64).tolist(), rec["p"].tolist()
        if not b.section(b"Elements"):
            raise ValueError(f"{path}: no $Elements section")
        n = int(b.line())
        done = 0
        while done < n:
            etype, cnt, ntags = (int(v) for v in b.take("<i4", 3))
            if etype not in _ELEMENT_TYPES:
                raise ValueError(f"{path}: unsupported element type {etype}")
            per = _ELEMENT_TYPES[etype][1]
            data = b.take("<i4", cnt * (1 + ntags + per)).reshape(cnt, 1 + ntags + per)
            for row in data:
                blocks.append((_ELEMENT_TYPES[etype][0], int(row[1]) if ntags > 0 else 0, etype, [row[1 + ntags:].astype(np.int64).tolist()]))
            done += cnt
        return tags, xyz, blocks
    phys_of = {}
    if b.section(b"Entities"):
        counts = [int(v) for v in b.take("<u8", 4)]
        for dim, cnt in enumerate(counts):
            for _ in range(cnt):
                tag = int(b.take("<i4")[0])
                b.take("<f8", 3 if dim == 0 else 6)
                nphys = int(b.take("<u8")[0])
                ph = b.take("<i4", nphys)
                phys_of[(dim, tag)] = int(ph[0]) if nphys > 0 else 0
                if dim > 0:
                    nb = int(b.take("<u8")[0])
                    b.take("<i4", nb)
    if not b.section(b"Nodes"):
        raise ValueError(f"{path}: no $Nodes section")
    nblocks = int(b.take("<u8", 4)[0])
    for _ in range(nblocks):
        _, _, parametric = (int(v) for v in b.take("<i4", 3))
        nb = int(b.take("<u8")[0])
        tags.extend(b.take("<u8", nb).astype(np.int64).tolist())
        edim_par = 0
        if parametric:
            raise ValueError(f"{path}: parametric node blocks are not supported")
        xyz.extend(b.take("<f8", nb * (3 + edim_par)).reshape(nb, 3).tolist())
    if not b.section(b"Elements"):
        raise ValueError(f"{path}: no $Elements section")
    nblocks = int(b.take("<u8", 4)[0])
    for _ in range(nblocks):
        edim, etag, etype = (int(v) for v in b.take("<i4", 3))
        nb = int(b.take("<u8")[0])
        if etype not in _ELEMENT_TYPES:
            raise ValueError(f"{path}: unsupported element type {etype}")
        per = _ELEMENT_TYPES[etype][1]
        data = b.take("<u8", nb * (1 + per)).reshape(nb, 1 + per)
        blocks.append((edim, phys_of.get((edim, etag), 0), etype, data[:, 1:].astype(np.int64).tolist()))
    return tags, xyz, blocks


def read_msh(path):
    """Returns (vertices (n, 3) float64, cells (m, 4 | 8, or 3 for a triangle mesh) int32, body_ids (m,) int32)."""
    with open(path, "rb") as fh:
        raw = fh.read()
    head = raw[:256].split(b"\n")
    if len(head) < 2 or head[0].strip() != b"$MeshFormat":
        raise ValueError(f"{path}: not an MSH file")
    fmt = head[1].split()
    if len(fmt) < 3:
        raise ValueError(f"{path}: malformed $MeshFormat")
    version, file_type, data_size = fmt[0].decode("ascii"), int(fmt[1]), int(fmt[2])
    major = int(float(version))
    if major not in (2, 4):
        raise ValueError(f"{path}: MSH format {version} is not supported (2.2 and 4.1 are)")
    if file_type == 1:
        if data_size != 8:
            raise ValueError(f"{path}: binary MSH files with {data_size}-byte floats are not supported")
        if len(raw) < 300 and b"$Nodes" not in raw:
            raise ValueError(f"{path}: binary MSH file without nodes")
        tags, xyz, blocks = _read_binary(raw, path, major)
    else:
        try:
            sec = _sections(raw.decode("ascii"))
        except UnicodeDecodeError as exc:
            raise ValueError(f"{path}: not an ASCII MSH file") from exc
        tags, xyz, blocks = _read_ascii(sec, path, major)
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
    if dim not in (2, 3):
        raise ValueError(f"{path}: only volume and triangle meshes are supported (highest element dimension is {dim})")
    cols = None
    cells, body = [], []
    for edim, phys, etype, elems in blocks:
        if edim != dim:
            continue
        nverts = _ELEMENT_TYPES[etype][2]
        if (dim == 3 and nverts not in (4, 8)) or (dim == 2 and nverts != 3):
            raise ValueError(f"{path}: only tetrahedra, hexahedra and triangles are supported (element type {etype})")
        if cols is not None and cols != nverts:
            raise ValueError(f"{path}: mixed element types are not supported")
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


def _write_msh_binary(path, vertices, cells, body, etype, version):
    nv, ne = vertices.shape[0], cells.shape[0]
    edim = 2 if etype == 2 else 3
    with open(path, "wb") as fh:
        fh.write(("$MeshFormat\n%s 1 8\n" % version).encode())
        fh.write(np.array([1], dtype="<i4").tobytes())
        fh.write(b"\n$EndMeshFormat\n")
        if version == "2.2":
            fh.write(b"$Nodes\n%d\n" % nv)
            rec = np.zeros(nv, dtype=np.dtype([("tag", "<i4"), ("p", "<f8", 3)]))
            rec["tag"], rec["p"] = np.arange(1, nv + 1), vertices
            fh.write(rec.tobytes())
            fh.write(b"\n$EndNodes\n$Elements\n%d\n" % ne)
            fh.write(np.array([etype, ne, 2], dtype="<i4").tobytes())
            data = np.concatenate([np.arange(1, ne + 1)[:, None], body[:, None], body[:, None], cells + 1], axis=1).astype("<i4")
            fh.write(data.tobytes())
            fh.write(b"\n$EndElements\n")
            return
        ids = sorted(set(int(b) for b in body))
        lo, hi = vertices.min(axis=0), vertices.max(axis=0)
        fh.write(b"$Entities\n")
        fh.write(np.array([0, 0, len(ids), 0] if edim == 2 else [0, 0, 0, len(ids)], dtype="<u8").tobytes())
        for t, b in enumerate(ids):
            fh.write(np.array([t + 1], dtype="<i4").tobytes())
            fh.write(np.concatenate([lo, hi]).astype("<f8").tobytes())
            fh.write(np.array([1 if b != 0 else 0], dtype="<u8").tobytes())
            if b != 0:
                fh.write(np.array([b], dtype="<i4").tobytes())
            fh.write(np.array([0], dtype="<u8").tobytes())
        fh.write(b"\n$EndEntities\n$Nodes\n")
        fh.write(np.array([1, nv, 1, nv], dtype="<u8").tobytes())
        fh.write(np.array([edim, 1, 0], dtype="<i4").tobytes())
        fh.write(np.array([nv], dtype="<u8").tobytes())
        fh.write(np.arange(1, nv + 1, dtype="<u8").tobytes())
        fh.write(vertices.astype("<f8").tobytes())
        fh.write(b"\n$EndNodes\n$Elements\n")
        fh.write(np.array([len(ids), ne, 1, ne], dtype="<u8").tobytes())
        tag = 1
        for t, b in enumerate(ids):
            sel = np.nonzero(body == b)[0]
            fh.write(np.array([edim, t + 1, etype], dtype="<i4").tobytes())
            fh.write(np.array([sel.size], dtype="<u8").tobytes())
            data = np.concatenate([np.arange(tag, tag + sel.size)[:, None], cells[sel] + 1], axis=1).astype("<u8")
            fh.write(data.tobytes())
            tag += sel.size
        fh.write(b"\n$EndElements\n")


def write_msh(path, vertices, cells, body_ids=None, version: str = "2.2", binary: bool = False):
    """MSH writer (ASCII or binary) for tetrahedra (4 columns) / hexahedra (8 columns) / triangles (3 columns; vertices
    with two coordinates get z = 0); the counterpart of read_msh for tests and for exporting generated meshes."""
    vertices = np.asarray(vertices, dtype=np.float64)
    if vertices.shape[1] == 2:
        vertices = np.hstack([vertices, np.zeros((vertices.shape[0], 1))])
    cells = np.asarray(cells)
    etype = {4: 4, 8: 5, 3: 2}[cells.shape[1]]
    edim = 2 if etype == 2 else 3
    body = np.zeros(cells.shape[0], dtype=np.int64) if body_ids is None else np.asarray(body_ids, dtype=np.int64)
    if version not in ("2.2", "4.1"):
        raise ValueError("version must be '2.2' or '4.1'")
    if binary:
        return _write_msh_binary(path, vertices, cells.astype(np.int64), body, etype, version)
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
            fh.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Entities\n%s\n" % ("0 0 %d 0" % len(ids) if edim == 2 else "0 0 0 %d" % len(ids)))
            for t, b in enumerate(ids):
                fh.write("%d %.17g %.17g %.17g %.17g %.17g %.17g %s 0\n" % (t + 1, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2],
                                                                             ("1 %d" % b) if b != 0 else "0"))
            fh.write("$EndEntities\n$Nodes\n1 %d 1 %d\n%d 1 0 %d\n" % (vertices.shape[0], vertices.shape[0], edim, vertices.shape[0]))
            for k in range(vertices.shape[0]):
                fh.write("%d\n" % (k + 1))
            for p in vertices:
                fh.write("%.17g %.17g %.17g\n" % (p[0], p[1], p[2]))
            fh.write("$EndNodes\n$Elements\n%d %d 1 %d\n" % (len(ids), cells.shape[0], cells.shape[0]))
            tag = 1
            for t, b in enumerate(ids):
                sel = np.nonzero(body == b)[0]
                fh.write("%d %d %d %d\n" % (edim, t + 1, etype, sel.size))
                for k in sel:
                    fh.write("%d %s\n" % (tag, " ".join(str(int(v) + 1) for v in cells[k])))
                    tag += 1
            fh.write("$EndElements\n")
        else:
            raise ValueError("version must be '2.2' or '4.1'")


_HEX_EDGES = [(0, 1), (1, 2), (2, 3), (3, 0), (0, 4), (1, 5), (2, 6), (3, 7), (4, 5), (5, 6), (6, 7), (7, 4)]
_HEX_FACES = [(0, 3, 4, 7), (1, 2, 5, 6), (0, 1, 5, 4), (3, 2, 6, 7), (0, 1, 2, 3), (4, 5, 6, 7)]


def _number_primitives(cells, nv, groups, numbering):
    """cells: (n_el, n_vert) vertex ids.  groups: list of (n_el, k_g, m_g) arrays: for every further local node the m_g
    vertices of its primitive (edge: 2, face: 4, cell: 8).  Returns conn (n_el, n_loc) and node positions' recipe:
    (n_nodes, list of (node ids, vertex tuples) per group)."""
    n_el = cells.shape[0]
    prim_cols, offset, tables = [cells.astype(np.int64)], nv, []
    for g in groups:
        key = np.sort(g.reshape(-1, g.shape[2]), axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        prim_cols.append(offset + inv.reshape(n_el, g.shape[1]))
        tables.append((offset, uniq))
        offset += uniq.shape[0]
    prims = np.concatenate(prim_cols, axis=1)  # element-major, local node order = the order the reference asks in
    flat = prims.reshape(-1)
    node_of = np.full(offset, -1, dtype=np.int64)
    if numbering == "first_touch":
        uniq, first = np.unique(flat, return_index=True)
        node_of[uniq[np.argsort(first, kind="stable")]] = np.arange(uniq.size)
        n_nodes = int(uniq.size)
    elif numbering == "vertices_first":
        node_of[:nv] = np.arange(nv)
        rest = flat[flat >= nv]
        uniq, first = np.unique(rest, return_index=True)
        node_of[uniq[np.argsort(first, kind="stable")]] = nv + np.arange(uniq.size)
        n_nodes = nv + int(uniq.size)
    else:
        raise ValueError("numbering must be 'vertices_first' or 'first_touch'")
    return node_of[prims], n_nodes, node_of, tables


def flatten(vertices, cells, order: int = 1, numbering: str = "vertices_first") -> M.Mesh:
    """Mesh of P1 / P2 tetrahedra (cells with 4 columns) or Q1 / Q2 hexahedra (8 columns, Gmsh corner order = the
    reference's) in the engine's layout."""
    vertices = np.ascontiguousarray(vertices, dtype=np.float64)
    cells = np.array(cells, dtype=np.int64, copy=True)
    nv = vertices.shape[0]
    if cells.min() < 0 or cells.max() >= nv:
        raise ValueError("cell refers to a vertex that does not exist")
    if order not in (1, 2):
        raise ValueError("order 1 or 2")
    if cells.shape[1] == 4:
        p = vertices[cells]
        det = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
        if np.any(det == 0):
            raise ValueError("degenerate tetrahedron")
        flip = det < 0
        cells[flip, 2], cells[flip, 3] = cells[flip, 3].copy(), cells[flip, 2].copy()
        groups = [] if order == 1 else [np.stack([cells[:, list(e)] for e in M._P2_EDGES], axis=1)]
        kind = M.P1TET if order == 1 else M.P2TET
        gkind = M.P1TET
    elif cells.shape[1] == 8:
        p = vertices[cells]
        det = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 3] - p[:, 0]), p[:, 4] - p[:, 0])
        if np.any(det <= 0):
            raise ValueError("hexahedron with non-positive orientation at its first corner")
        groups = [] if order == 1 else [np.stack([cells[:, list(e)] for e in _HEX_EDGES], axis=1),
                                        np.stack([cells[:, list(f)] for f in _HEX_FACES], axis=1), cells[:, None, :]]
        kind = M.Q1HEX if order == 1 else M.Q2HEX
        gkind = M.Q1HEX
    elif cells.shape[1] == 3:
        vertices = np.ascontiguousarray(vertices[:, :2])
        p = vertices[cells]
        a, b = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
        det = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        if np.any(det == 0):
            raise ValueError("degenerate triangle")
        flip = det < 0
        cells[flip, 1], cells[flip, 2] = cells[flip, 2].copy(), cells[flip, 1].copy()
        groups = [] if order == 1 else [np.stack([cells[:, list(e)] for e in M._P2_TRI_EDGES], axis=1)]
        kind = M.P1TRI if order == 1 else M.P2TRI
        gkind = M.P1TRI
    else:
        raise ValueError("cells must have 4 (tetrahedra), 8 (hexahedra) or 3 (triangles) columns")
    conn, n_nodes, node_of, tables = _number_primitives(cells, nv, groups, numbering)
    if n_nodes >= 2 ** 31:
        raise ValueError("too many nodes for 32-bit indices")
    node_xyz = np.zeros((n_nodes, vertices.shape[1]))
    used_v = node_of[:nv] >= 0
    node_xyz[node_of[:nv][used_v]] = vertices[used_v]
    for offset, uniq in tables:  # barycentre of the primitive's vertices (MeshNodes.cpp:166-173)
        ids = node_of[offset: offset + uniq.shape[0]]
        node_xyz[ids] = vertices[uniq].mean(axis=1)
    c32 = np.ascontiguousarray(cells.astype(np.int32))
    return M.Mesh(kind, gkind, np.ascontiguousarray(conn.astype(np.int32)), c32, vertices, n_nodes, node_xyz)


def read_obj(path):
    """Triangle mesh from a Wavefront OBJ file (`v x y [z]` and `f a b c` records, 1-based or negative indices, the
    `a/b/c` forms accepted): what the reference's 2-D tests load (`tests/test_assembler.cpp:24`, `tests/test_form_derivatives.cpp:45`).
    Returns (vertices (n, 3), cells (m, 3) int32, body_ids (m,) zeros)."""
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                c = [float(v) for v in t[1:4]]
                verts.append(c + [0.0] * (3 - len(c)))
            elif t[0] == "f":
                ids = [int(v.split("/")[0]) for v in t[1:]]
                if len(ids) != 3:
                    raise ValueError(f"{path}: only triangles are supported (face with {len(ids)} vertices)")
                faces.append(ids)
    if not verts or not faces:
        raise ValueError(f"{path}: no vertices or no faces")
    cells = np.asarray(faces, dtype=np.int64)
    cells = np.where(cells < 0, cells + len(verts), cells - 1)
    if cells.min() < 0 or cells.max() >= len(verts):
        raise ValueError(f"{path}: a face refers to a vertex that is not defined")
    return np.asarray(verts, dtype=np.float64), np.ascontiguousarray(cells.astype(np.int32)), np.zeros(len(faces), dtype=np.int32)


_MEDIT_CELLS = {"tetrahedra": (3, 4), "hexahedra": (3, 8), "triangles": (2, 3), "quadrilaterals": (2, 4), "edges": (1, 2),
                "corners": (0, 1), "requiredvertices": (0, 1), "ridges": (0, 1), "requirededges": (0, 1)}


def read_medit(path):
    """ASCII MEDIT / INRIA `.mesh` file (`MeshVersionFormatted`, `Dimension`, `Vertices`, `Tetrahedra` / `Hexahedra` /
    `Triangles` sections, 1-based indices, a reference number after every record): the format of most volume meshes the
    reference is run on.  There it goes through the un-vendored geogram loader (`mesh/Mesh.cpp:172-230` falls through to
    `GEO::mesh_load`, `mesh/mesh3D/CMesh3D.cpp:86-91`); as for every format but MSH, no body ids come with it (:212-226).
    The elements of the highest dimension present become `cells`; hexahedra are in the corner order of Gmsh and of the
    reference.  Returns (vertices (n, 3), cells (m, 4 | 8 | 3) int32, body_ids (m,) zeros)."""
    with open(path) as fh:
        tok = []
        for line in fh:
            line = line.split("#", 1)[0]
            tok.extend(line.split())
    if not tok or tok[0].lower() != "meshversionformatted":
        raise ValueError(f"{path}: not an ASCII MEDIT mesh file")
    pos, dim, verts, found = 2, None, None, {}
    try:
        while pos < len(tok):
            key = tok[pos].lower()
            pos += 1
            if key == "end":
                break
            if key == "dimension":
                dim = int(tok[pos])
                pos += 1
                if dim not in (2, 3):
                    raise ValueError(f"{path}: Dimension {dim}")
            elif key == "vertices":
                if dim is None:
                    raise ValueError(f"{path}: Vertices before Dimension")
                n = int(tok[pos])
                pos += 1
                a = np.array(tok[pos: pos + n * (dim + 1)], dtype=np.float64).reshape(n, dim + 1)
                pos += n * (dim + 1)
                verts = np.zeros((n, 3))
                verts[:, :dim] = a[:, :dim]
            elif key in _MEDIT_CELLS:
                edim, nv = _MEDIT_CELLS[key]
                n = int(tok[pos])
                pos += 1
                width = nv + (0 if key in ("corners", "requiredvertices", "ridges", "requirededges") else 1)
                a = np.array(tok[pos: pos + n * width], dtype=np.int64).reshape(n, width)
                pos += n * width
                if edim >= 2 and n > 0:
                    found[key] = a[:, :nv] - 1
            else:
                raise ValueError(f"{path}: section {tok[pos - 1]} is not supported")
    except (IndexError, ValueError) as exc:
        if isinstance(exc, ValueError) and str(exc).startswith(str(path)):
            raise
        raise ValueError(f"{path}: truncated or malformed MEDIT file") from exc
    if verts is None or not found:
        raise ValueError(f"{path}: no vertices or no elements")
    top = max(_MEDIT_CELLS[k][0] for k in found)
    kinds = [k for k in found if _MEDIT_CELLS[k][0] == top]
    if len(kinds) != 1:
        raise ValueError(f"{path}: mixed element types are not supported")
    if kinds[0] == "quadrilaterals":
        raise ValueError(f"{path}: only tetrahedra, hexahedra and triangles are supported")
    cells = found[kinds[0]]
    if cells.min() < 0 or cells.max() >= verts.shape[0]:
        raise ValueError(f"{path}: an element refers to a vertex that is not defined")
    return verts, np.ascontiguousarray(cells.astype(np.int32)), np.zeros(cells.shape[0], dtype=np.int32)


def write_medit(path, vertices, cells):
    """Writes tetrahedra, hexahedra or triangles as an ASCII MEDIT file (tests and round trips)."""
    vertices = np.asarray(vertices, dtype=np.float64)
    cells = np.asarray(cells)
    name = {4: "Tetrahedra", 8: "Hexahedra", 3: "Triangles"}[cells.shape[1]]
    dim = 2 if cells.shape[1] == 3 and vertices.shape[1] == 2 else 3
    with open(path, "w") as fh:
        fh.write(f"MeshVersionFormatted 1\nDimension {dim}\nVertices\n{vertices.shape[0]}\n")
        for v in vertices:
            fh.write(" ".join(repr(float(c)) for c in v[:dim]) + " 0\n")
        fh.write(f"{name}\n{cells.shape[0]}\n")
        for c in cells:
            fh.write(" ".join(str(int(i) + 1) for i in c) + " 0\n")
        fh.write("End\n")


def load(path, order: int = 1, numbering: str = "vertices_first"):
    """read_msh / read_obj / read_medit (by extension: .obj, .mesh, anything else is read as MSH) + flatten; returns
    (Mesh, body_ids)."""
    low = str(path).lower()
    reader = read_obj if low.endswith(".obj") else read_medit if low.endswith(".mesh") else read_msh
    vertices, cells, body = reader(path)
    return flatten(vertices, cells, order, numbering), body

"""Multi-GPU layer: node partition, halo maps and the distributed CG driver (one process per GPU).

The reference is single-process (no MPI/NCCL anywhere in its tree); this module is the new work the hot path
needs to run on the 8 GPUs of one node:

  * nodes (matrix rows) are split into contiguous owner ranges;
  * every rank takes all elements that touch one of its nodes (its own elements plus one layer of its
    neighbours'), numbers its nodes [owned | halo] and assembles with the single-GPU kernels -- owned rows are then
    complete, so ASSEMBLY NEEDS NO COMMUNICATION (energy: each element is summed by the owner of its lowest node,
    followed by one all-reduce of a double);
  * the CG solve keeps all vectors distributed by rows; per iteration one halo exchange of the search direction
    (grouped point-to-point send/recv between neighbouring ranks, i.e. ncclSend/ncclRecv over xGMI under
    torch.distributed's "nccl" backend = RCCL) and two tiny all-reduces (p.Ap, then r.r and r.z together) -- the
    classic two-reduction CG, so the iterates are those of the serial Eigen-order iteration.

The per-rank compute is delegated to an `ops` object: `HipOps` (the HIP engine, the product path) or any object
with the same methods working on CPU tensors (the CPU tests use one built on the oracle to check the partition
and the communication pattern under the gloo backend).
"""
from __future__ import annotations

import dataclasses

import numpy as np

from . import mesh as M


# ------------------------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------------------------


def owner_ranges(n_nodes: int, world: int) -> np.ndarray:
    """Contiguous, nearly equal node ranges: rank r owns [b[r], b[r+1])."""
    return np.array([(n_nodes * r) // world for r in range(world + 1)], dtype=np.int64)


@dataclasses.dataclass
class Part:
    mesh: M.Mesh               # local mesh: conn in local numbering [owned | halo]
    n_owned: int
    lo: int                    # owned global ids are lo .. lo+n_owned-1
    halo_global: np.ndarray    # global ids of the halo nodes, ascending
    halo_owner: np.ndarray     # owner rank per halo node
    elem_owned: np.ndarray     # uint8 per local element: its energy is summed on this rank
    elem_global: np.ndarray    # index of each local element in the candidate list handed to partition()

    @property
    def n_local(self) -> int:
        return self.n_owned + int(self.halo_global.size)

    def local_to_global(self) -> np.ndarray:
        return np.concatenate([np.arange(self.lo, self.lo + self.n_owned, dtype=np.int64), self.halo_global])


def partition(kind: int, conn: np.ndarray, gconn: np.ndarray, xyz: np.ndarray, bounds: np.ndarray, rank: int,
              node_xyz=None, quad_order: int = -1) -> Part:
    """Local part of rank `rank`.  `conn` (global basis ids) / `gconn` (global geometry ids) may be any superset of
    the elements that touch the rank's nodes; `xyz` is indexed by global geometry id."""
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    touch = np.nonzero(((conn >= lo) & (conn < hi)).any(axis=1))[0]
    c = conn[touch].astype(np.int64)
    used = np.unique(c)
    owned_mask = (used >= lo) & (used < hi)
    halo_global = used[~owned_mask]
    n_owned = hi - lo  # every node of the range is owned, used or not (keeps ranges contiguous)
    # local id: owned -> g - lo ; halo -> n_owned + rank in halo_global
    loc = np.where((c >= lo) & (c < hi), c - lo, n_owned + np.searchsorted(halo_global, c))
    halo_owner = (np.searchsorted(bounds, halo_global, side="right") - 1).astype(np.int32)
    gused, ginv = np.unique(gconn[touch], return_inverse=True)
    lmesh = M.Mesh(kind, M.GEOM_KIND[kind], np.ascontiguousarray(loc.astype(np.int32)),
                   np.ascontiguousarray(ginv.reshape(-1, gconn.shape[1]).astype(np.int32)),
                   np.ascontiguousarray(xyz[gused]), n_owned + int(halo_global.size), None, quad_order)
    if node_xyz is not None:
        l2g = np.concatenate([np.arange(lo, hi, dtype=np.int64), halo_global])
        lmesh.node_xyz = node_xyz(l2g) if callable(node_xyz) else node_xyz[l2g]
    cmin = c.min(axis=1)
    elem_owned = ((cmin >= lo) & (cmin < hi)).astype(np.uint8)
    return Part(lmesh, n_owned, lo, halo_global, halo_owner, elem_owned, touch)


def morton_numbering(node_xyz: np.ndarray, bits: int = 10) -> np.ndarray:
    """new_id[g] for a renumbering of the nodes along a Morton (Z-order) curve through their bounding box: contiguous
    owner ranges of the new numbering are compact blocks instead of slabs of whatever order the mesh came in, which is
    what keeps the halo -- and the x gathers of the matrix-vector product -- local on a general mesh.  Ties keep the
    incoming order (stable)."""
    p = np.asarray(node_xyz, dtype=np.float64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    q = np.minimum(((p - lo) / span * (1 << bits)).astype(np.uint64), np.uint64((1 << bits) - 1))
    key = np.zeros(p.shape[0], dtype=np.uint64)
    dim = p.shape[1]  # 3, or 2 for triangle meshes
    for b in range(bits):
        for c in range(dim):
            key |= ((q[:, c] >> np.uint64(b)) & np.uint64(1)) << np.uint64(dim * b + c)
    order = np.argsort(key, kind="stable")
    new_id = np.empty(p.shape[0], dtype=np.int64)
    new_id[order] = np.arange(p.shape[0])
    return new_id


def renumber_nodes(mesh: M.Mesh, new_id: np.ndarray) -> M.Mesh:
    """The same mesh with basis node g renamed new_id[g] (geometry nodes untouched)."""
    node_xyz = None
    if mesh.node_xyz is not None:
        node_xyz = np.empty_like(mesh.node_xyz)
        node_xyz[new_id] = mesh.node_xyz
    return M.Mesh(mesh.kind, mesh.gkind, np.ascontiguousarray(new_id[mesh.conn].astype(np.int32)), mesh.gconn, mesh.xyz,
                  mesh.n_nodes, node_xyz, mesh.quad_order)


def halo_nodes_total(kind: int, conn: np.ndarray, gconn: np.ndarray, xyz: np.ndarray, bounds: np.ndarray) -> int:
    """Sum over the ranks of their halo node counts (what one halo exchange moves, in nodes)."""
    return int(sum(partition(kind, conn, gconn, xyz, bounds, r).halo_global.size for r in range(len(bounds) - 1)))


def cube_tets_slab(n: int, nz: int, zc0: int, zc1: int, order: int = 2):
    """Cells z in [zc0, zc1) of the n x n x nz Kuhn grid on [0,1]^2 x [0, nz/n], P1/P2 bases, GLOBAL lattice
    numbering (x fastest).  Returns (kind, conn, gconn, xyz_of_all_vertices, n_nodes_global, lattice_dim_xy)."""
    tets = M._kuhn_tets()
    kz, ky, kx = np.meshgrid(np.arange(zc0, zc1), np.arange(n), np.arange(n), indexing="ij")
    cell = np.stack([kx, ky, kz], axis=-1).reshape(-1, 1, 1, 3)
    corners = (cell + tets[None]).reshape(-1, 4, 3)
    nv = n + 1
    gconn = corners[..., 0] + nv * (corners[..., 1] + nv * corners[..., 2])
    gz, gy, gx = np.meshgrid(np.arange(nz + 1), np.arange(nv), np.arange(nv), indexing="ij")
    xyz = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float64) / n
    if order == 1:
        return M.P1TET, gconn.astype(np.int64), gconn, xyz, nv * nv * (nz + 1), nv
    L = 2 * n + 1
    c2 = 2 * corners
    loc = [c2[:, k] for k in range(4)]
    for a, b in M._P2_EDGES:
        loc.append((c2[:, a] + c2[:, b]) // 2)
    loc = np.stack(loc, axis=1)
    conn = loc[..., 0] + L * (loc[..., 1] + L * loc[..., 2])
    return M.P2TET, conn.astype(np.int64), gconn, xyz, L * L * (2 * nz + 1), L


def cube_hexes_slab(n: int, nz: int, zc0: int, zc1: int, order: int = 2):
    """Cells z in [zc0, zc1) of the n x n x nz grid of hexahedra on [0,1]^2 x [0, nz/n], Q1/Q2 bases on trilinear
    geometry, GLOBAL lattice numbering (x fastest); same return values as cube_tets_slab."""
    kz, ky, kx = np.meshgrid(np.arange(zc0, zc1), np.arange(n), np.arange(n), indexing="ij")
    cell = np.stack([kx, ky, kz], axis=-1).reshape(-1, 1, 3)
    nv = n + 1
    corners = cell + M._Q1_CORNERS[None]
    gconn = corners[..., 0] + nv * (corners[..., 1] + nv * corners[..., 2])
    gz, gy, gx = np.meshgrid(np.arange(nz + 1), np.arange(nv), np.arange(nv), indexing="ij")
    xyz = np.stack([gx, gy, gz], axis=-1).reshape(-1, 3).astype(np.float64) / n
    if order == 1:
        return M.Q1HEX, gconn.astype(np.int64), gconn, xyz, nv * nv * (nz + 1), nv
    L = 2 * n + 1
    loc = 2 * cell + M._Q2_NODES[None]
    conn = loc[..., 0] + L * (loc[..., 1] + L * loc[..., 2])
    return M.Q2HEX, conn.astype(np.int64), gconn, xyz, L * L * (2 * nz + 1), L


def slab_part(n: int, cells_per_rank: int, rank: int, world: int, order: int = 2, element: str = "tet") -> tuple[Part, np.ndarray, int]:
    """Weak-scaling decomposition used by the benchmark: the global grid is n x n x (cells_per_rank*world) cells
    (Kuhn-split into tetrahedra, or hexahedra), rank r owns the lattice planes of its cell layers (the last rank also
    the top plane)."""
    nz = cells_per_rank * world
    step = 2 if order == 2 else 1
    z0, z1 = rank * cells_per_rank, (rank + 1) * cells_per_rank
    slab = {"tet": cube_tets_slab, "hex": cube_hexes_slab}[element]
    kind, conn, gconn, xyz, n_nodes, L = slab(n, nz, max(z0 - 1, 0), z1, order)
    bounds = np.array([step * r * cells_per_rank * L * L for r in range(world)] + [n_nodes], dtype=np.int64)
    h = (0.5 if order == 2 else 1.0) / n

    def node_xyz(g):
        return np.stack([g % L, (g // L) % L, g // (L * L)], axis=1) * h

    return partition(kind, conn, gconn, xyz, bounds, rank, node_xyz), bounds, n_nodes


# ------------------------------------------------------------------------------------------------------------
# communication
# ------------------------------------------------------------------------------------------------------------


class Comm:
    """Thin wrapper over torch.distributed (backend "nccl" = RCCL on the GPUs; "gloo" in the CPU tests).  With
    world size 1 (or no process group) every call is a no-op."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.stage_cpu = self.on and dist.get_backend() == "gloo"  # gloo moves device tensors through the host

    def all_reduce_sum(self, t):
        if not self.on:
            return
        if self.stage_cpu and t.is_cuda:
            c = t.cpu()
            self.dist.all_reduce(c)
            t.copy_(c)
        else:
            self.dist.all_reduce(t)

    def all_gather_object(self, obj):
        if not self.on:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class HaloExchange:
    """Fills the halo part of a node-major vector [owned | halo] from the owners: one grouped batch of
    point-to-point sends/receives per call.  Receives land directly in the vector (halo nodes are sorted by global
    id, so each neighbour's nodes are one contiguous slice); sends are packed with an index gather."""

    def __init__(self, part: Part, comm: Comm, device):
        import torch
        self.torch, self.comm, self.part = torch, comm, part
        self.n_owned = part.n_owned
        want = {}
        for s in np.unique(part.halo_owner):
            sel = np.nonzero(part.halo_owner == s)[0]
            want[int(s)] = (int(sel[0]), int(sel[-1]) + 1, part.halo_global[sel])
        self.recv = {s: (a, b) for s, (a, b, _) in want.items()}
        requests = comm.all_gather_object({s: ids for s, (_, _, ids) in want.items()})
        self.send_idx = {}
        for s, req in enumerate(requests):
            if s != comm.rank and comm.rank in req:
                idx = np.asarray(req[comm.rank], dtype=np.int64) - part.lo
                assert idx.min() >= 0 and idx.max() < part.n_owned
                self.send_idx[s] = torch.from_numpy(idx).to(device)
        self.send_buf = {}

    def neighbours(self):
        return sorted(set(self.recv) | set(self.send_idx))

    def bytes_per_exchange(self, bs: int) -> int:
        return 8 * bs * (sum(int(v.numel()) for v in self.send_idx.values()) + sum(b - a for a, b in self.recv.values()))

    def start(self, vec_nodes):
        """Packs the send buffers and posts the grouped sends/receives; returns a handle for finish().  Between the
        two calls work that does not read the halo part of the vector may be enqueued: it runs concurrently with the
        transfers (RCCL works on its own stream)."""
        if not self.comm.on:
            return None
        torch, dist = self.torch, self.comm.dist
        ops, staged = [], []
        for s, idx in self.send_idx.items():
            buf = self.send_buf.get(s)
            if buf is None or buf.shape[1] != vec_nodes.shape[1]:
                buf = torch.empty((idx.numel(), vec_nodes.shape[1]), dtype=vec_nodes.dtype, device=vec_nodes.device)
                self.send_buf[s] = buf
            torch.index_select(vec_nodes, 0, idx, out=buf)
            ops.append(dist.P2POp(dist.isend, buf.cpu() if self.comm.stage_cpu and buf.is_cuda else buf, s))
        for s, (a, b) in self.recv.items():
            dst = vec_nodes[self.n_owned + a: self.n_owned + b]
            if self.comm.stage_cpu and dst.is_cuda:
                tmp = torch.empty(dst.shape, dtype=dst.dtype)
                staged.append((dst, tmp))
                ops.append(dist.P2POp(dist.irecv, tmp, s))
            else:
                ops.append(dist.P2POp(dist.irecv, dst, s))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, staged

    def finish(self, handle):
        if handle is None:
            return
        reqs, staged = handle
        for r in reqs:
            r.wait()
        for dst, tmp in staged:
            dst.copy_(tmp)

    def __call__(self, vec_nodes):
        """vec_nodes: (n_local, bs) tensor view of the vector."""
        self.finish(self.start(vec_nodes))


# ------------------------------------------------------------------------------------------------------------
# per-rank compute back end (product path: the HIP engine)
# ------------------------------------------------------------------------------------------------------------


class HipOps:
    """Adapter from tensors to the engine's device-pointer step functions (include/polyfem_amd.h, pfx_cgs_*).

    Stream contract: the drivers below interleave torch operations (index gathers, axpys, reductions) with the
    engine's kernels on the same buffers, so both must be enqueued on ONE stream.  The constructor therefore puts the
    engine on torch's current stream of the engine's device; that stream has to be a non-default one (make it current
    with torch.cuda.set_stream / torch.cuda.stream before constructing HipOps), because the C ABI uses the null handle
    for "the context's own stream".  check_stream() -- called at the start of every distributed solve -- raises if
    torch's current stream is no longer the engine's."""

    def __init__(self, engine, stream=None):
        import torch
        self.e = engine
        self.torch = torch
        s = stream if stream is not None else torch.cuda.current_stream()
        if int(s.cuda_stream) == 0:
            raise RuntimeError("HipOps needs a non-default torch stream: call torch.cuda.set_stream(torch.cuda.Stream()) "
                               "first (the engine would otherwise run on a stream of its own, unordered with torch)")
        self.stream_handle = int(s.cuda_stream)
        engine.set_stream(self.stream_handle)

    def check_stream(self):
        cur = int(self.torch.cuda.current_stream().cuda_stream)
        if cur != self.stream_handle:
            raise RuntimeError("torch's current stream is not the stream the engine was bound to by HipOps")

    def set_workspace(self, work):
        if work is None:
            self.e.cgs_set_workspace(0, 0)
        else:
            self.e.cgs_set_workspace(work.data_ptr(), work.numel())

    def release_workspace(self, work):
        """Takes `work` out of the engine if it is still the installed workspace: a newer DistributedCG on the same
        engine may have installed its own since, and that one must stay."""
        if work is not None and getattr(self.e, "cgs_workspace", 0) == work.data_ptr():
            self.set_workspace(None)

    def prepare(self, x, precond):
        self.e.cgs_prepare(x.data_ptr(), precond)

    def residual(self, b, x, scal):
        self.e.cgs_residual(b.data_ptr(), x.data_ptr(), scal.data_ptr())

    def start(self, x, tol, scal):
        self.e.cgs_start(x.data_ptr(), tol, scal.data_ptr())

    def start2(self, scal):
        self.e.cgs_start2(scal.data_ptr())

    def spmv(self, scal):
        self.e.cgs_spmv(scal.data_ptr())

    def spmv_interior(self):
        self.e.cgs_spmv_interior()

    def spmv_boundary(self, scal):
        self.e.cgs_spmv_boundary(scal.data_ptr())

    def fetch_product(self):
        """A p in row order in the workspace's third quarter (the engine may have left it in its matrix copy's order)."""
        self.e.cgs_fetch_product()

    def update(self, it, x, scal):
        self.e.cgs_update(it, x.data_ptr(), scal.data_ptr())

    def direction(self, it, scal):
        self.e.cgs_direction(it, scal.data_ptr())

    def status(self):
        return self.e.cgs_status()

    # ---- block-Jacobi AMG preconditioner (one hierarchy per rank, on the rows and columns the rank owns) ----
    def amg_setup(self):
        self.e.amg_setup()

    def amg_apply(self, r, z):
        """z[owned rows] = M^-1 r, M = this rank's AMG cycle for its owned-owned block; r, z: local vectors."""
        self.e.amg_apply_dev(r.data_ptr(), z.data_ptr())

    def apply_matrix(self, v, y):
        """y[owned rows] = A v with the (masked) matrix the engine holds; v, y: local vectors, v with its halo entries."""
        self.e.spmv_dev(v.data_ptr(), y.data_ptr())

    # ---- nonlinear pieces (DistributedNewton) ----
    def energy(self, x, out):
        """out[0] = energy of the elements this rank sums (pfx_set_owned's flags)."""
        self.e.energy_dev(x.data_ptr(), out.data_ptr())

    def gradient(self, x, g):
        self.e.gradient_dev(x.data_ptr(), g.data_ptr())

    def hessian(self, x, project_to_psd):
        self.e.hessian_dev(x.data_ptr(), project_to_psd)

    def apply(self, v, y):
        """y[owned rows] = (A + shift I) v with the Dirichlet rows replaced by identity; v needs a current halo."""
        self.e.spmv_dev(v.data_ptr(), y.data_ptr())

    def set_diag_shift(self, shift):
        self.e.set_diag_shift(shift)

    def clear_numeric_flag(self):
        self.e.take_numeric_flag()

    def mass_apply(self, v, y):
        """y[owned rows] = M v (mass matrix kept by Engine.inertia_set_mass); v needs a current halo."""
        self.e.mass_apply_dev(v.data_ptr(), y.data_ptr())

    def hessian_add_mass(self, weight):
        self.e.hessian_add_mass_dev(weight)


class LibComm:
    """pfx_comm: the library's communicator.  Under torch.distributed's "nccl" backend the library talks RCCL itself
    (its own ncclComm, the id made on rank 0 and handed round through the process group); under "gloo" (CPU tests, the
    single-card rehearsal) it gets a host transport whose two callbacks move host buffers through the process group."""

    def __init__(self, comm: Comm, device_index: int = 0):
        import ctypes as C

        import torch

        from . import engine as E
        self.L = E.load()
        self.comm = comm
        self.h = C.c_void_p()
        self.kind = "single"
        dist = comm.dist
        if comm.on and dist.get_backend() == "nccl":
            ident = [bytes(128)]
            if comm.rank == 0:
                buf = C.create_string_buffer(128)
                if self.L.pfx_comm_rccl_unique_id(buf) != 0:
                    raise RuntimeError("pfx_comm_rccl_unique_id failed")
                ident = [buf.raw]
            dist.broadcast_object_list(ident, src=0)
            buf = C.create_string_buffer(ident[0], 128)
            rc = self.L.pfx_comm_create_rccl(C.byref(self.h), buf, comm.rank, comm.world, device_index)
            if rc != 0:
                raise RuntimeError(f"pfx_comm_create_rccl failed ({rc})")
            self.kind = "rccl"
            return

        def exchange(user, n, peer, send, send_bytes, recv, recv_bytes):
            try:
                ops, keep = [], []
                for k in range(n):
                    if send_bytes[k] > 0:
                        t = torch.frombuffer((C.c_char * send_bytes[k]).from_address(send[k]), dtype=torch.uint8)
                        keep.append(t)
                        ops.append(dist.P2POp(dist.isend, t, int(peer[k])))
                    if recv_bytes[k] > 0:
                        t = torch.frombuffer((C.c_char * recv_bytes[k]).from_address(recv[k]), dtype=torch.uint8)
                        keep.append(t)
                        ops.append(dist.P2POp(dist.irecv, t, int(peer[k])))
                for r in (dist.batch_isend_irecv(ops) if ops else []):
                    r.wait()
                return 0
            except Exception as exc:  # an exception must not cross the C frame
                print(f"[polyfem_amd] host transport exchange failed: {exc!r}", flush=True)
                return 1

        def all_reduce_sum(user, buf, count):
            try:
                t = torch.frombuffer((C.c_double * count).from_address(C.addressof(buf.contents)), dtype=torch.float64)
                dist.all_reduce(t)
                return 0
            except Exception as exc:
                print(f"[polyfem_amd] host transport all_reduce failed: {exc!r}", flush=True)
                return 1

        if comm.on:
            self._cb = (E.EXCHANGE_FN(exchange), E.ALL_REDUCE_FN(all_reduce_sum))  # kept alive with the object
        else:
            self._cb = (E.EXCHANGE_FN(lambda *a: 0), E.ALL_REDUCE_FN(lambda *a: 0))
        self._t = E.HostTransport(None, self._cb[0], self._cb[1])
        rc = self.L.pfx_comm_create_host(C.byref(self.h), C.byref(self._t), comm.rank, comm.world)
        if rc != 0:
            raise RuntimeError(f"pfx_comm_create_host failed ({rc})")
        self.kind = "host" if comm.on else "single"

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.pfx_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LibDist:
    """pfx_dist: an engine holding one part of the partition, bound to a LibComm and the halo lists of `part`.  The
    distributed iteration then runs inside the library (pfx_dist_cg_dev): no Python between the kernels, no host read
    except the convergence flag every `check_every` iterations."""

    def __init__(self, engine, part: Part, comm: Comm, device, libcomm: LibComm | None = None):
        import ctypes as C
        self.L, self.engine, self.part = engine.L, engine, part
        self.libcomm = libcomm or LibComm(comm, device.index if getattr(device, "index", None) is not None else 0)
        self._own_comm = libcomm is None
        hx = HaloExchange(part, comm, "cpu")  # the same lists the Python driver uses: who needs which of my nodes
        peers = hx.neighbours()
        send_ptr, recv_ptr, send_nodes = [0], [0], []
        for s in peers:
            idx = hx.send_idx.get(s)
            if idx is not None:
                send_nodes.append(idx.numpy().astype(np.int32))
            send_ptr.append(send_ptr[-1] + (0 if idx is None else int(idx.numel())))
            a, b = hx.recv.get(s, (recv_ptr[-1], recv_ptr[-1]))
            assert a == recv_ptr[-1], "halo nodes are numbered owner by owner"
            recv_ptr.append(b)
        peer = np.asarray(peers, dtype=np.int32)
        sp, rp = np.asarray(send_ptr, dtype=np.int64), np.asarray(recv_ptr, dtype=np.int64)
        sn = np.concatenate(send_nodes).astype(np.int32) if send_nodes else np.zeros(0, dtype=np.int32)
        self.h = C.c_void_p()
        i32, i64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        rc = self.L.pfx_dist_create(C.byref(self.h), engine.h, self.libcomm.h, int(peer.size), peer.ctypes.data_as(i32), sp.ctypes.data_as(i64),
                                    sn.ctypes.data_as(i32), rp.ctypes.data_as(i64))
        if rc != 0:
            raise RuntimeError(f"pfx_dist_create failed ({rc}): {self.L.pfx_last_error(engine.h).decode()}")

    def _ck(self, rc):
        if rc != 0:
            from .engine import PfxError
            raise PfxError(rc, self.L.pfx_dist_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.pfx_dist_destroy(self.h)
            self.h = None
        if self._own_comm and self.libcomm is not None:
            self.libcomm.close()
            self.libcomm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def halo_exchange(self, vec):
        self._ck(self.L.pfx_dist_halo_exchange_dev(self.h, vec.data_ptr()))

    def all_reduce(self, buf):
        self._ck(self.L.pfx_dist_all_reduce_dev(self.h, buf.data_ptr(), int(buf.numel())))

    def energy(self, u, out):
        self._ck(self.L.pfx_dist_energy_dev(self.h, u.data_ptr(), out.data_ptr()))

    def halo_bytes(self) -> int:
        import ctypes as C
        b = C.c_int64()
        self._ck(self.L.pfx_dist_halo_bytes(self.h, C.byref(b)))
        return b.value

    def amg_apply(self, r, z):
        """z = M^-1 r with the hierarchy that spans the ranks (collective; r, z local vectors)."""
        self._ck(self.L.pfx_dist_amg_apply_dev(self.h, r.data_ptr(), z.data_ptr()))

    def newton(self, x, rhs=None, **params):
        """pfx_dist_newton_dev: the whole Newton solve over the partition inside the library.  Returns (rc, message, info
        dict); rc != 0 with info['status'] FAILED / ITERATION_LIMIT where the reference throws."""
        import ctypes as C

        from . import engine as E
        p = E.NewtonParams()
        self.L.pfx_newton_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        info = E.NewtonInfo()
        rc = self.L.pfx_dist_newton_dev(self.h, C.c_void_p(rhs.data_ptr()) if rhs is not None else None, C.c_void_p(x.data_ptr()), C.byref(p), C.byref(info))
        out = {f[0]: getattr(info, f[0]) for f in E.NewtonInfo._fields_}
        return rc, (self.L.pfx_dist_last_error(self.h).decode() if rc else ""), out

    def implicit_euler_step(self, dt, x, v, a=None, rhs=None, **params):
        """pfx_dist_implicit_euler_step_dev: one implicit Euler step over the partition inside the library; same return
        convention as newton()."""
        import ctypes as C

        from . import engine as E
        p = E.NewtonParams()
        self.L.pfx_newton_default_params(C.byref(p))
        for k, val in params.items():
            setattr(p, k, val)
        info = E.NewtonInfo()
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = self.L.pfx_dist_implicit_euler_step_dev(self.h, float(dt), ptr(rhs), ptr(x), ptr(v), ptr(a), C.byref(p), C.byref(info))
        out = {f[0]: getattr(info, f[0]) for f in E.NewtonInfo._fields_}
        return rc, (self.L.pfx_dist_last_error(self.h).decode() if rc else ""), out

    def cg(self, b, x, precond: int = 1, max_iter: int = 1000, tol: float = 1e-12, check_every: int = 50):
        import ctypes as C
        it, err = C.c_int(), C.c_double()
        self._ck(self.L.pfx_dist_cg_dev(self.h, b.data_ptr(), x.data_ptr(), precond, max_iter, tol, check_every, C.byref(it), C.byref(err)))
        return it.value, err.value


class DistributedCG:
    """Jacobi-CG in Eigen's operation order (see polyfem_amd/csrc/pfx_solve.hip) over a node partition.

    With the HIP engine the iteration runs inside the library (pfx_dist_cg_dev); precond = 2 (PFX_PRECOND_AMG) then
    preconditions with ONE hierarchy that spans the ranks (aggregates local to a rank, Galerkin products across the
    partition boundaries, row-distributed coarse levels, gathered coarsest solve).

    solve(..., use_library=False) keeps the loop driven from Python over the engine's step functions; there precond = 2
    uses one hierarchy per rank, built on the block of the matrix whose rows and columns the rank owns (block-Jacobi
    over the ranks, no communication inside the hierarchy), and enable_coarse_space() adds a small coarse problem
    that spans the ranks: the rigid-body modes of every rank's block (six per rank for bs = 3, the constant for
    bs = 1), z = M_bj^-1 r + Z (Z^T A Z)^-1 Z^T r -- additive, hence symmetric positive definite; one more
    all-reduce of 6 N doubles per iteration, the 6 N x 6 N coarse matrix is inverted redundantly on every rank."""

    def __init__(self, ops, part: Part, bs: int, comm: Comm, device):
        import torch
        self.torch, self.ops, self.part, self.bs, self.comm = torch, ops, part, bs, comm
        self.n_local = part.n_local
        self.n = self.n_local * bs
        self.device = device
        self.halo = HaloExchange(part, comm, device)
        self.work = None  # workspace of the Python-driven loops, made on first use (_ensure_work)
        self.scal = torch.zeros(4, dtype=torch.float64, device=device)
        self.cs = None  # coarse space over the ranks (enable_coarse_space)
        # The product path: the whole iteration inside the library (pfx_dist_cg_dev) over the library's own communicator.
        # Available when the per-rank compute is the HIP engine; the CPU stand-ins of the tests keep the Python loop.
        self.lib = None
        if isinstance(ops, HipOps):
            self._open_library()

    def _open_library(self):
        """Binds the engine to the library's own communicator and checks it against torch.distributed before it is
        trusted: one halo exchange of a vector of global node ids and one all-reduce, compared on every rank with what
        the process group delivers.  The decision is taken collectively, so either every rank runs the library loop or
        every rank keeps the Python-driven one (a transport that errs on one rank must not split the ranks)."""
        t, ok = self.torch, 1.0
        try:
            self.lib = LibDist(self.ops.e, self.part, self.comm, self.device)
        except Exception as exc:
            print(f"[polyfem_amd] rank {self.comm.rank}: the library communicator could not be created ({exc!r})", flush=True)
            self.lib, ok = None, 0.0
        flag = t.tensor([ok], dtype=t.float64, device=self.device)
        if self.comm.on:
            self._all_min(flag)
        if float(flag.item()) >= 1.0 and self.comm.on:
            try:
                ids = t.from_numpy(self.part.local_to_global().astype(np.float64)).to(self.device)
                v = (ids[:, None] * self.bs + t.arange(self.bs, dtype=t.float64, device=self.device)[None]).contiguous()
                w = v.clone()
                w[self.part.n_owned:] = -1.0
                self.lib.halo_exchange(w.view(-1))
                s = t.tensor([float(self.comm.rank + 1), 1.0], dtype=t.float64, device=self.device)
                self.lib.all_reduce(s)
                self.torch.cuda.synchronize() if w.is_cuda else None
                want = self.comm.world * (self.comm.world + 1) / 2.0
                good = bool(t.equal(w, v)) and float(s[0].item()) == want and float(s[1].item()) == float(self.comm.world)
            except Exception as exc:
                print(f"[polyfem_amd] rank {self.comm.rank}: the library transport failed its check ({exc!r})", flush=True)
                good = False
            flag[0] = 1.0 if good else 0.0
            self._all_min(flag)
        if float(flag.item()) < 1.0 and self.lib is not None:
            if self.comm.rank == 0:
                print("[polyfem_amd] the library's communicator is not used: the distributed loops are driven from Python over torch.distributed", flush=True)
            self.lib.close()
            self.lib = None

    def _all_min(self, flag):
        dist = self.comm.dist
        if self.comm.stage_cpu and flag.is_cuda:
            c = flag.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.MIN)
            flag.copy_(c)
        else:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)

    def _ensure_work(self):
        if self.work is None:
            self.work = self.torch.zeros(4 * self.n, dtype=self.torch.float64, device=self.device)
            self.p_nodes = self.work[self.n: 2 * self.n].view(self.n_local, self.bs)
        self.ops.set_workspace(self.work)  # another solver object on the same engine may have installed its own since

    # ---- coarse space over the ranks for the AMG-preconditioned solve ------------------------------------------
    def enable_coarse_space(self, node_pos, fixed_dofs=None):
        """node_pos: (n_local, 3) positions of this rank's nodes (owned, then halo); fixed_dofs: local indices of the
        constrained dofs (the list given to the engine).  The coarse matrix is rebuilt by every AMG-preconditioned
        solve (6 (1 + neighbours) products with the current matrix)."""
        t, dev = self.torch, self.device
        bs, nl, no = self.bs, self.n_local, self.part.n_owned
        free = np.ones(nl * bs, dtype=np.float64)
        if fixed_dofs is not None and len(fixed_dofs):
            free[np.asarray(fixed_dofs, dtype=np.int64)] = 0.0
        owner = np.full(nl, self.comm.rank, dtype=np.int64)
        owner[no:] = np.asarray(self.part.halo_owner, dtype=np.int64)
        node_pos = np.asarray(node_pos, dtype=np.float64)
        if node_pos.shape[1] == 2:  # triangle meshes: the plane z = 0
            node_pos = np.hstack([node_pos, np.zeros((node_pos.shape[0], 1))])
        pos = t.from_numpy(np.ascontiguousarray(node_pos)).to(dev)
        # centre of every rank's owned nodes (any point would do; the centre keeps the rotations well scaled)
        cen = t.zeros((self.comm.world, 3), dtype=t.float64, device=dev)
        cen[self.comm.rank] = pos[:no].mean(dim=0)
        self.comm.all_reduce_sum(cen)
        self.cs = dict(pos=pos, free=t.from_numpy(free).to(dev).view(nl, bs), owner=t.from_numpy(owner).to(dev), centres=cen,
                       modes={3: 6, 2: 3, 1: 1}[bs], einv=None, sources=sorted(set(int(o) for o in owner)))

    def _cs_modes(self, s, k):
        """Local part (n_local, bs) of mode k of rank s: non-zero on the free dofs of the nodes rank s owns."""
        t, c = self.torch, self.cs
        sel = (c["owner"] == s).to(t.float64)[:, None]
        if self.bs == 1:
            return sel * c["free"]
        if self.bs == 2:  # the two translations and the in-plane rotation
            v = t.zeros((self.n_local, 2), dtype=t.float64, device=c["pos"].device)
            if k < 2:
                v[:, k] = 1.0
            else:
                d = c["pos"] - c["centres"][s]
                v[:, 0], v[:, 1] = -d[:, 1], d[:, 0]
            return v * sel * c["free"]
        if k < 3:
            v = t.zeros((self.n_local, 3), dtype=t.float64, device=c["pos"].device)
            v[:, k] = 1.0
        else:
            axis = t.zeros(3, dtype=t.float64, device=c["pos"].device)
            axis[k - 3] = 1.0
            v = t.cross(axis.expand(self.n_local, 3), c["pos"] - c["centres"][s], dim=1)
        return v * sel * c["free"]

    def _cs_restrict(self, y_owned):
        """Z_r^T y for this rank's own modes: y_owned (n_owned * bs)."""
        t, c, no = self.torch, self.cs, self.part.n_owned
        yn = y_owned.view(no, self.bs) * c["free"][:no]
        if self.bs == 1:
            return yn.sum().reshape(1)
        d = c["pos"][:no] - c["centres"][self.comm.rank]
        if self.bs == 2:
            return t.cat([yn.sum(dim=0), (d[:, 0] * yn[:, 1] - d[:, 1] * yn[:, 0]).sum().reshape(1)])
        return t.cat([yn.sum(dim=0), t.cross(d, yn, dim=1).sum(dim=0)])

    def _cs_setup(self):
        """E = Z^T A Z with the matrix the engine holds now; rank r computes its block row (products of A with the modes
        of itself and of the ranks that own its halo nodes), the rows are summed over the ranks."""
        t, c, no, bs = self.torch, self.cs, self.part.n_owned, self.bs
        m, N, r = c["modes"], self.comm.world, self.comm.rank
        E = t.zeros((m * N, m * N), dtype=t.float64, device=c["pos"].device)
        y = t.zeros(self.n, dtype=t.float64, device=c["pos"].device)
        for s in c["sources"]:
            for k in range(m):
                v = self._cs_modes(s, k).reshape(-1).contiguous()
                self.ops.apply_matrix(v, y)
                E[m * r: m * r + m, m * s + k] = self._cs_restrict(y[:no * bs])
        self.comm.all_reduce_sum(E)
        E = 0.5 * (E + E.T)
        # a rank without free dofs, or a mode its constraints remove, leaves a zero row: decouple it
        dg = t.diagonal(E)
        dmax = float(dg.abs().max().item())
        dead = dg.abs() <= 1e-14 * (dmax if dmax > 0 else 1.0)
        E[dead, :] = 0.0
        E[:, dead] = 0.0
        E[dead, dead] = 1.0
        c["einv"] = t.linalg.inv(E)
        c["einv"][dead, :] = 0.0
        c["einv"][:, dead] = 0.0

    def _cs_apply(self, r_owned, z_owned):
        """z += Z E^-1 Z^T r"""
        t, c, no = self.torch, self.cs, self.part.n_owned
        m, N, r = c["modes"], self.comm.world, self.comm.rank
        w = t.zeros(m * N, dtype=t.float64, device=c["pos"].device)
        w[m * r: m * r + m] = self._cs_restrict(r_owned)
        self.comm.all_reduce_sum(w)
        a = (c["einv"] @ w)[m * r: m * r + m]
        zn = z_owned.view(no, self.bs)
        if self.bs == 1:
            zn += a[0] * c["free"][:no]
            return
        d = c["pos"][:no] - c["centres"][r]
        if self.bs == 2:
            zn += t.stack([a[0] - a[2] * d[:, 1], a[1] + a[2] * d[:, 0]], dim=1) * c["free"][:no]
            return
        zn += (a[0:3] + t.cross(a[3:6].expand(no, 3), d, dim=1)) * c["free"][:no]

    def close(self):
        """Hands the solver workspace back to the engine (the engine must not keep a pointer into a freed tensor)."""
        if getattr(self, "lib", None) is not None:
            self.lib.close()
            self.lib = None
        if self.work is not None:
            try:
                release = getattr(self.ops, "release_workspace", None)
                if release is not None:
                    release(self.work)  # only if the engine still points at THIS object's buffer
                else:
                    self.ops.set_workspace(None)
            except Exception:
                pass
            self.work = None

    def __del__(self):
        self.close()

    def solve(self, b, x, precond: int = 1, max_iter: int = 1000, tol: float = 1e-12, check_every: int = 25, use_library: bool = True):
        """b, x: local vectors of n_local*bs doubles (halo entries of b are ignored).  Returns (iters, rel_res).
        With the HIP engine the Jacobi / unpreconditioned iteration runs inside the library (use_library = False keeps
        the Python-driven loop over the step functions: same kernels, same scalars, for comparison)."""
        ops, scal, ar = self.ops, self.scal, self.comm.all_reduce_sum
        if hasattr(ops, "check_stream"):
            ops.check_stream()
        if self.lib is not None and use_library:
            # precond = 2: ONE hierarchy that spans the ranks (use_library = False: the per-rank hierarchies of _solve_amg)
            ops.set_workspace(None)  # the library loop works in the context's own vectors
            return self.lib.cg(b, x, precond, max_iter, tol, check_every)
        self._ensure_work()
        if precond == 2:
            return self._solve_amg(b, x, max_iter, tol)
        ops.prepare(x, precond)
        self.halo(x.view(self.n_local, self.bs))
        ops.residual(b, x, scal)
        ar(scal[0:2])
        ops.start(x, tol, scal)
        ar(scal[2:3])
        ops.start2(scal)
        it, done, iters, rr, bb = 0, False, 0, 0.0, 0.0
        while it < max_iter and not done:
            stop = min(it + check_every, max_iter)
            while it < stop:
                # the rows that need no halo entry of p are multiplied while the exchange is in flight
                pending = self.halo.start(self.p_nodes)
                ops.spmv_interior()
                self.halo.finish(pending)
                ops.spmv_boundary(scal)
                ar(scal[0:1])
                ops.update(it, x, scal)
                ar(scal[1:3])
                ops.direction(it, scal)
                it += 1
            done, iters, rr, bb = ops.status()
        if max_iter <= 0:
            done, iters, rr, bb = ops.status()
        return iters, (0.0 if bb == 0.0 else float(np.sqrt(rr / bb)))


    def _solve_amg(self, b, x, max_iter, tol):
        """Preconditioned CG with z = M^-1 r from ops.amg_apply; same stopping rule and iteration count convention as
        the Jacobi loop (|r|^2 < tol^2 |b|^2 tested before the direction update).  The scalars of an iteration are read
        on the host (one synchronisation per iteration, small against the cost of a cycle)."""
        t, ops, ar = self.torch, self.ops, self.comm.all_reduce_sum
        if not hasattr(ops, "amg_apply"):
            raise NotImplementedError("this back end has no AMG preconditioner")
        n, no = self.n, self.part.n_owned * self.bs
        scal = self.scal
        ops.prepare(x, 1)  # constrained dofs of the guess zeroed, solver copy of the matrix current
        ops.amg_setup()    # may use the direction vector as scratch: before p is first written
        self.halo(x.view(self.n_local, self.bs))
        ops.residual(b, x, scal)
        ar(scal[0:2])
        bb, rr = (float(v) for v in scal[0:2].tolist())
        if bb == 0.0:
            x.zero_()
            return 0, 0.0
        thr = max(tol * tol * bb, np.finfo(np.float64).tiny)
        if rr < thr:
            return 0, float(np.sqrt(rr / bb))
        r_full = self.work[0:n]
        r, p, ap = self.work[0:no], self.work[n:n + no], self.work[2 * n:2 * n + no]
        if getattr(self, "z", None) is None:
            self.z = t.zeros(n, dtype=t.float64, device=self.work.device)
        z_full, z = self.z, self.z[0:no]
        xo = x[0:no]
        if self.cs is not None:
            self._cs_setup()
        ops.amg_apply(r_full, z_full)
        if self.cs is not None:
            self._cs_apply(r, z)
        p.copy_(z)
        scal[2] = t.dot(r, z)
        ar(scal[2:3])
        rz = float(scal[2].item())
        iters = 0
        for it in range(max_iter):
            pending = self.halo.start(self.p_nodes)
            ops.spmv_interior()
            self.halo.finish(pending)
            ops.spmv_boundary(scal)  # scal[0] = local p.Ap
            ops.fetch_product()      # this loop reads A p in the workspace itself
            ar(scal[0:1])
            alpha = rz / scal[0]     # device scalar
            xo.addcmul_(p, alpha)
            r.addcmul_(ap, alpha, value=-1.0)
            ops.amg_apply(r_full, z_full)
            if self.cs is not None:
                self._cs_apply(r, z)
            scal[1] = t.dot(r, r)
            scal[2] = t.dot(r, z)
            ar(scal[1:3])
            rr, rz_new = (float(v) for v in scal[1:3].tolist())
            if rr < thr:
                iters = it
                break
            p.mul_(rz_new / rz).add_(z)
            rz = rz_new
            iters = it + 1
        return iters, float(np.sqrt(rr / bb))


# ------------------------------------------------------------------------------------------------------------
# Newton's method over the partition
# ------------------------------------------------------------------------------------------------------------


class NewtonFailure(RuntimeError):
    """Raised where the reference's nonlinear solver throws; `info` holds the state reached."""

    def __init__(self, msg, info):
        super().__init__(msg)
        self.info = info


NEWTON_DEFAULTS = dict(f_delta=1e-10, grad_norm=1e-8, first_grad_norm_tol=1e-10, use_grad_norm_tol=1e-4,
                       max_iterations=1000, use_grad_norm=1, relative_gradient=0, line_search=1, cg_precond=1,
                       cg_max_iter=1000, cg_tol=1e-12, cg_check_every=25)

CONTINUE, ITERATION_LIMIT, F_DELTA, GRAD_NORM, NOT_STARTED, FAILED = range(6)


class DistributedNewton:
    """cppoptlib::NonlinearSolver::minimize with SparseNewtonDescentSolver and BacktrackingLineSearch
    (src/polyfem/solver/NonlinearSolver.tpp:33-275, SparseNewtonDescentSolver.tpp:40-193,
    line_search/BacktrackingLineSearch.hpp:29-255, LineSearch.tpp:90-145) on E(x) - rhs.x, every vector distributed
    by node rows.  Same control flow as the single-GPU pfx_newton_solve_dev (polyfem_amd/csrc/pfx_newton.hip); the
    differences are where the numbers come from:

      * energy: every rank sums its own elements, one all-reduce of (energy, rhs.x) per evaluation;
      * gradient / Hessian: owned rows are complete after the local assembly -- no communication;
      * linear solve: DistributedCG; afterwards ONE halo exchange of the direction, so that every trial point
        x + a dx of the line search is consistent on the halo without further exchanges;
      * norms and dot products: local over the owned dofs, all-reduced (batched into one small tensor per step).

    All ranks take the same branches because every decision is made on all-reduced scalars.
    """

    def __init__(self, ops, part: Part, bs: int, comm: Comm, device, fixed_dofs_local):
        import torch
        self.torch, self.ops, self.part, self.bs, self.comm, self.device = torch, ops, part, bs, comm, device
        self.cg = DistributedCG(ops, part, bs, comm, device)
        if comm.on and hasattr(ops, "apply_matrix"):
            # the AMG-preconditioned solves (cg_precond = 2) get the coarse space over the ranks
            from .mesh import node_positions
            self.cg.enable_coarse_space(node_positions(part.mesh), fixed_dofs_local)
        self.n = part.n_local * bs
        self.no = part.n_owned * bs
        fixed = np.zeros(self.n, dtype=bool)
        fixed[np.asarray(fixed_dofs_local, dtype=np.int64)] = True
        self.fixed = torch.from_numpy(fixed[:self.no].copy()).to(device)
        z = lambda k: torch.zeros(k, dtype=torch.float64, device=device)  # noqa: E731
        self.g, self.dx, self.xn, self.w, self.gls = z(self.n), z(self.n), z(self.n), z(self.n), z(self.n)
        self.dd, self.md = z(self.n), z(self.n)  # x - x_tilde and M (x - x_tilde) of a transient step
        self.e1 = z(1)
        self.red = z(6)
        self.x_tilde, self.weight = None, 1.0

    # -- reductions over the ranks: sums in red[0:2], maximum in red[2]
    def _sums(self, *vals):
        t, red = self.torch, self.red
        k = len(vals)
        t.stack(vals, out=red[:k]) if k > 1 else red[:1].copy_(vals[0].reshape(1))
        self.comm.all_reduce_sum(red[:k])
        return [float(v) for v in red[:k].tolist()]

    def _max(self, val):
        """Maximum over the ranks of a non-negative local value; nan if it is nan anywhere (MAX reductions may drop it)."""
        r = self.red[4:6]
        r[0] = self.torch.nan_to_num(val, nan=0.0)
        r[1] = self.torch.isnan(val).to(r.dtype)
        if self.comm.on:
            if self.comm.stage_cpu and r.is_cuda:
                c = r.cpu()
                self.comm.dist.all_reduce(c, op=self.comm.dist.ReduceOp.MAX)
                r.copy_(c)
            else:
                self.comm.dist.all_reduce(r, op=self.comm.dist.ReduceOp.MAX)
        m, bad = r.tolist()
        return float("nan") if bad else float(m)

    # -- NLProblem::value (solver/NLProblem.cpp:79-88): nan/inf when an element is inverted anywhere
    def _value(self, x):
        no = self.no
        self.ops.energy(x, self.e1)
        rx = (self.rhs[:no] @ x[:no]) if self.rhs is not None else self.torch.zeros((), dtype=x.dtype, device=x.device)
        if self.x_tilde is None:
            e, d = self._sums(self.e1[0], rx)
            self.info["objective_evaluations"] += 1
            return e - d if np.isfinite(e) else float("nan")
        # transient step: weight * (E - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde)   (InertiaForm.cpp:14-21)
        self.torch.sub(x, self.x_tilde, out=self.dd)
        self.ops.mass_apply(self.dd, self.md)
        e, d, q = self._sums(self.e1[0], rx, self.dd[:no] @ self.md[:no])
        self.info["objective_evaluations"] += 1
        return self.weight * (e - d) + 0.5 * q if np.isfinite(e) else float("nan")

    # -- NLProblem::gradient (:90-99) restricted to the free dofs; returns its squared norm
    def _gradient(self, x, out):
        no = self.no
        self.ops.gradient(x, out)
        if self.rhs is not None:
            out[:no] -= self.rhs[:no]
        if self.x_tilde is not None:
            self.torch.sub(x, self.x_tilde, out=self.dd)
            self.ops.mass_apply(self.dd, self.md)
            out[:no].mul_(self.weight).add_(self.md[:no])
        out[:no].masked_fill_(self.fixed, 0.0)
        out[no:] = 0.0
        (n2,) = self._sums(out[:no] @ out[:no])
        return n2

    def _line_search(self, x, par):
        """BacktrackingLineSearch::line_search without contact; nan on failure."""
        info, t = self.info, self.torch
        max_it, cur = 100, 0
        old_in = self._value(x)
        info["line_search_iterations"] += 1
        if np.isnan(old_in):
            return float("nan")
        step = 1.0
        t.add(x, self.dx, alpha=step, out=self.xn)
        while step > 0 and cur < max_it:
            e = self._value(self.xn)
            info["line_search_iterations"] += 1
            if np.isfinite(e):
                break
            step *= 0.5
            t.add(x, self.dx, alpha=step, out=self.xn)
            cur += 1
        if cur >= max_it or step <= 0:
            return float("nan")
        gn2 = self._gradient(x, self.gls)
        use_gn = np.sqrt(gn2) < par["use_grad_norm_tol"]
        old = gn2 if use_gn else old_in
        while step > 0 and cur < max_it:
            t.add(x, self.dx, alpha=step, out=self.xn)
            c = self._gradient(self.xn, self.gls) if use_gn else self._value(self.xn)
            info["line_search_iterations"] += 1
            if np.isfinite(c) and not c > old:
                break
            step /= 2.0
            cur += 1
        if cur >= max_it or step <= 0:
            return float("nan")
        return step

    def _library_result(self, rc, msg, out):
        out["objective_evaluations"] = None  # not counted by the library
        self.info = out
        if rc != 0:
            from .engine import PFX_ERR_NUMERIC, PfxError
            if rc == PFX_ERR_NUMERIC:
                raise NewtonFailure(msg, out)
            raise PfxError(rc, msg)
        return out

    @staticmethod
    def _check(par, iterations, f_delta, grad_norm, grad_tol):
        if par["max_iterations"] > 0 and iterations > par["max_iterations"]:
            return ITERATION_LIMIT
        if par["f_delta"] > 0 and f_delta < par["f_delta"]:
            return F_DELTA
        if grad_tol > 0 and grad_norm < grad_tol:
            return GRAD_NORM
        return CONTINUE

    def solve(self, x, rhs=None, x_tilde=None, weight: float = 1.0, use_library: bool = True, **params):
        """x: local vector [owned | halo] of n_local*bs doubles, starting point in, minimiser out (the halo part is
        kept current); rhs: local vector or None.  With `x_tilde` (local vector with a current halo) the objective is
        weight * (E(x) - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde), M being the mass matrix the back end holds
        (Engine.inertia_set_mass on every rank).  Returns the info dict; raises NewtonFailure where the reference
        throws.

        With the HIP engine the solve runs inside the library (pfx_dist_newton_dev, the inertia term through
        pfx_inertia_enable); use_library = False keeps the loop below, driven from Python over the engine's step
        functions."""
        self.x_tilde, self.weight = x_tilde, (float(weight) if x_tilde is not None else 1.0)
        par = dict(NEWTON_DEFAULTS)
        for k in params:
            if k not in par:
                raise TypeError(f"unknown Newton parameter {k!r}")
        par.update(params)
        if use_library and self.cg.lib is not None:
            if hasattr(self.ops, "check_stream"):
                self.ops.check_stream()
            self.ops.set_workspace(None)  # the library's loops work in the context's own vectors
            if x_tilde is not None:
                self.ops.e.inertia_enable(self.weight, x_tilde.data_ptr())
            try:
                rc, msg, out = self.cg.lib.newton(x, rhs, **{k: v for k, v in par.items() if k != "cg_check_every"})
            finally:
                if x_tilde is not None:
                    self.ops.e.inertia_disable()
            return self._library_result(rc, msg, out)
        ops, t, no, bs = self.ops, self.torch, self.no, self.bs
        if hasattr(ops, "check_stream"):
            ops.check_stream()
        nl = self.part.n_local
        self.rhs = rhs
        info = self.info = dict(status=CONTINUE, iterations=0, line_search_iterations=0, descent_strategy=0,
                                cg_iterations=0, energy=float("nan"), grad_norm=0.0, objective_evaluations=0)
        self.dx.zero_()
        self.cg.halo(x.view(nl, bs))
        ops.clear_numeric_flag()
        strategy, reg = 0, 0.0
        reg_min, reg_max, reg_inc, reg_dec = 1e-8, 1e8, 10.0, 2.0

        def increase():
            nonlocal strategy, reg
            if strategy == 0 or reg > reg_max:
                strategy += 1
            else:
                reg = max(reg_inc * reg, reg_min)

        def fail(msg, status=FAILED):
            ops.set_diag_shift(0.0)
            ops.clear_numeric_flag()
            info["status"] = status
            raise NewtonFailure(msg, info)

        first = float(np.sqrt(self._gradient(x, self.g)))
        grad_norm = first
        if np.isnan(first):
            fail("Gradient is nan; stopping")
        info["grad_norm"] = first
        if self._check(par, 0, float("nan"), first / (first if par["relative_gradient"] else 1.0),
                       par["first_grad_norm_tol"]) != CONTINUE:
            info["energy"] = self._value(x)
            info["status"] = NOT_STARTED
            ops.clear_numeric_flag()
            return info

        old_energy, status, iterations = float("nan"), CONTINUE, 0
        while True:
            energy = info["energy"] = self._value(x)
            if not np.isfinite(energy):
                fail("f(x) is nan or inf; stopping")
            grad_norm = info["grad_norm"] = float(np.sqrt(self._gradient(x, self.g)))
            if np.isnan(grad_norm):
                fail("Gradient is nan; stopping")
            # ---- compute_update_direction ----
            while True:
                if strategy > 2:
                    fail("no descent direction found")
                if strategy == 2:
                    t.neg(self.g, out=self.dx)
                    self.cg.halo(self.dx.view(nl, bs))
                    (dg,) = self._sums(self.dx[:no] @ self.g[:no])
                    dx_max = self._max(self.dx[:no].abs().max())
                    break
                ops.hessian(x, strategy == 1)
                if self.x_tilde is not None:
                    ops.hessian_add_mass(self.weight)
                ops.set_diag_shift(reg)
                t.neg(self.g, out=self.w)
                its, _ = self.cg.solve(self.w, self.dx, par["cg_precond"], par["cg_max_iter"], par["cg_tol"],
                                       par["cg_check_every"])
                info["cg_iterations"] += its
                self.cg.halo(self.dx.view(nl, bs))
                # check_direction: || H dx + g || and g.dx
                ops.apply(self.dx, self.w)
                rres = self.w[:no] + self.g[:no]
                r2, dg = self._sums(rres @ rres, self.dx[:no] @ self.g[:no])
                dx_max = self._max(self.dx[:no].abs().max())
                ops.set_diag_shift(0.0)
                residual = np.sqrt(r2)
                if np.isnan(residual) or residual > max(1e-8 * grad_norm, 1e-5) or dg >= 0:
                    increase()
                    continue
                reg /= reg_dec
                if reg < reg_min:
                    reg = 0.0
                break
            if (grad_norm != 0 and dg >= 0) or np.isnan(dx_max):
                increase()
                status = CONTINUE
                continue
            cur = grad_norm / (first if par["relative_gradient"] else 1.0) if par["use_grad_norm"] else dx_max
            status = self._check(par, iterations, abs(old_energy - energy), cur, par["grad_norm"])
            old_energy = energy
            rate = 1.0
            if par["line_search"]:
                rate = self._line_search(x, par)
                if np.isnan(rate):
                    if strategy < 2:
                        increase()
                        status = CONTINUE
                        continue
                    fail("Line search failed on gradient descent")
            x.add_(self.dx, alpha=rate)  # halo included: dx carries a current halo
            info["descent_strategy"] = strategy
            strategy = 0
            iterations += 1
            info["iterations"] = iterations
            if status != CONTINUE:
                break
        info["status"] = status
        ops.clear_numeric_flag()
        if status == ITERATION_LIMIT:
            fail("Reached iteration limit", ITERATION_LIMIT)
        return info


class DistributedImplicitEuler:
    """The reference's transient nonlinear solve with the ImplicitEuler integrator over the partition
    (time_integrator/ImplicitEuler.cpp:5-33, state/StateSolveNonlinear.cpp:78-84): per step x_tilde = x + dt v, a Newton
    solve of dt^2 (E(x) - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde) starting at x, then v = (x_new - x)/dt,
    a = (v_new - v)/dt.  x, v, a are local vectors [owned | halo]; their halos stay consistent without extra exchanges
    because the Newton driver returns x with a current halo."""

    def __init__(self, newton: DistributedNewton, dt: float):
        import torch
        assert dt > 0
        self.torch, self.newton, self.dt = torch, newton, float(dt)
        n, dev = newton.n, newton.device
        self.x_prev = torch.zeros(n, dtype=torch.float64, device=dev)
        self.x_tilde = torch.zeros(n, dtype=torch.float64, device=dev)

    def step(self, x, v, a=None, rhs=None, use_library: bool = True, **params):
        """Advances (x, v) in place; writes the acceleration into `a` if given.  Returns the Newton info dict.  With the
        HIP engine the whole step runs inside the library (pfx_dist_implicit_euler_step_dev)."""
        t, dt = self.torch, self.dt
        nw = self.newton
        if use_library and nw.cg.lib is not None:
            par = dict(NEWTON_DEFAULTS)
            for k in params:
                if k not in par:
                    raise TypeError(f"unknown Newton parameter {k!r}")
            par.update(params)
            if hasattr(nw.ops, "check_stream"):
                nw.ops.check_stream()
            nw.ops.set_workspace(None)
            rc, msg, out = nw.cg.lib.implicit_euler_step(dt, x, v, a, rhs, **{k: val for k, val in par.items() if k != "cg_check_every"})
            return nw._library_result(rc, msg, out)
        self.x_prev.copy_(x)
        t.add(x, v, alpha=dt, out=self.x_tilde)
        info = self.newton.solve(x, rhs, x_tilde=self.x_tilde, weight=dt * dt, use_library=False, **params)
        v_new = (x - self.x_prev) / dt
        if a is not None:
            t.sub(v_new, v, out=a)
            a.div_(dt)
        v.copy_(v_new)
        return info


class DistributedTransientSolver:
    """The transient nonlinear solve with any of the implicit integrators of polyfem_amd.time_integrator (ImplicitEuler,
    ImplicitNewmark, BDF; time_integrator/ImplicitTimeIntegrator.hpp, state/StateSolveNonlinear.cpp:78-84) over the
    partition: per step x_tilde from the integrator, a Newton solve of
    acceleration_scaling * (E(x) - rhs.x) + 1/2 (x - x_tilde)^T M (x - x_tilde) starting at the previous solution, then the
    integrator's update.  Every vector of the integrator is a local vector [owned | halo]: its updates are pointwise, so
    halos that are consistent at init stay consistent (the Newton driver returns x with a current halo) and the time
    stepping itself needs no communication."""

    def __init__(self, newton: DistributedNewton, integrator, rhs=None, **newton_params):
        self.newton, self.integrator, self.rhs, self.newton_params = newton, integrator, rhs, newton_params

    def step(self, **params):
        """One time step; returns (x, Newton info).  x is a new local tensor, the integrator's history is updated."""
        ti = self.integrator
        x_tilde = ti.x_tilde().contiguous()
        x = ti.x_prev().clone()
        par = dict(self.newton_params)
        par.update(params)
        info = self.newton.solve(x, self.rhs, x_tilde=x_tilde, weight=ti.acceleration_scaling(), **par)
        ti.update_quantities(x)
        return x, info


"""Test-only CPU stand-in for the per-rank compute back end of polyfem_amd.distributed.DistributedCG.

It mirrors, step by step, what the HIP step functions (pfx_cgs_*) do, on CPU tensors with the oracle's matrix, so
that the partition, the halo maps and the communication pattern can be exercised under the gloo backend without a
GPU.  It is NOT a fallback of the product: nothing in polyfem_amd/ imports it."""
import sys

import numpy as np
import scipy.sparse as sp


class OracleOps:
    def __init__(self, A_local_csr, n_owned_dofs, fixed_local):
        self.A = A_local_csr[:n_owned_dofs]          # owned rows, all local columns
        self.no = n_owned_dofs
        self.fixed = np.asarray(fixed_local[:n_owned_dofs], dtype=bool)
        self.state = dict(abs_new=[0.0, 0.0], rr=0.0, thr=0.0, bb=0.0, done=False, iters=0)
        self.shift = 0.0

    def set_workspace(self, work):
        if work is None:
            return
        n = work.numel() // 4
        w = work.numpy()
        self.r, self.p, self.tmp, self.invdiag = w[0:n], w[n:2 * n], w[2 * n:3 * n], w[3 * n:4 * n]

    def _apply(self, v):
        y = self.A @ v + self.shift * v[:self.no]
        y[self.fixed] = v[:self.no][self.fixed]
        return y

    def prepare(self, x, precond):
        xv = x.numpy()
        xv[:self.no][self.fixed] = 0.0
        d = self.A.diagonal() + self.shift if precond else np.ones(self.no)
        inv = np.ones(self.no)
        nz = (d != 0) & ~self.fixed
        inv[nz] = 1.0 / d[nz]
        self.invdiag[:self.no] = inv
        self.state = dict(abs_new=[0.0, 0.0], rr=0.0, thr=0.0, bb=0.0, done=False, iters=0)

    def residual(self, b, x, scal):
        no = self.no
        self.tmp[:no] = self._apply(x.numpy())
        bk = np.where(self.fixed, 0.0, b.numpy()[:no])
        self.r[:no] = np.where(self.fixed, 0.0, bk - self.tmp[:no])
        s = scal.numpy()
        s[0], s[1] = bk @ bk, self.r[:no] @ self.r[:no]

    def start(self, x, tol, scal):
        no, s, st = self.no, scal.numpy(), self.state
        bb, rr = s[0], s[1]
        st["thr"] = max(tol * tol * bb, sys.float_info.min)
        st["bb"], st["rr"] = bb, (0.0 if bb == 0 else rr)
        st["done"] = bool(bb == 0 or rr < st["thr"])
        if bb == 0:
            x.numpy()[:no] = 0.0
        self.p[:no] = self.invdiag[:no] * self.r[:no]
        s[2] = self.r[:no] @ self.p[:no]

    def start2(self, scal):
        self.state["abs_new"][0] = float(scal.numpy()[2])

    def spmv_interior(self):
        pass  # the CPU stand-in multiplies all rows once the halo is there

    def spmv_boundary(self, scal):
        self.spmv(scal)

    def fetch_product(self):
        pass  # the stand-in's product is always in the workspace

    def spmv(self, scal):
        if self.state["done"]:
            return
        no = self.no
        self.tmp[:no] = self._apply(self.p)
        scal.numpy()[0] = self.p[:no] @ self.tmp[:no]

    def update(self, it, x, scal):
        if self.state["done"]:
            return
        no, s = self.no, scal.numpy()
        alpha = self.state["abs_new"][it & 1] / s[0]
        x.numpy()[:no] += alpha * self.p[:no]
        self.r[:no] -= alpha * self.tmp[:no]
        s[1] = self.r[:no] @ self.r[:no]
        s[2] = self.r[:no] @ (self.invdiag[:no] * self.r[:no])

    def direction(self, it, scal):
        st = self.state
        if st["done"]:
            return
        no, s = self.no, scal.numpy()
        rr, rz = float(s[1]), float(s[2])
        st["rr"] = rr
        if rr < st["thr"]:
            st["done"], st["iters"] = True, it
            return
        beta = rz / st["abs_new"][it & 1]
        self.p[:no] = self.invdiag[:no] * self.r[:no] + beta * self.p[:no]
        st["abs_new"][(it + 1) & 1] = rz
        st["iters"] = it + 1

    def status(self):
        st = self.state
        return st["done"], st["iters"], st["rr"], st["bb"]

    # ---- what DistributedCG's preconditioned loop (precond = 2) asks for.  The device back end applies an AMG cycle for
    # the block of the matrix whose rows and columns the rank owns; the stand-in solves that block exactly (block-Jacobi
    # with exact subdomain solves): the same structure, so the loop, the halo exchange and the coarse space over the
    # ranks are exercised.
    def amg_setup(self):
        import scipy.sparse.linalg as spla
        no = self.no
        free = ~self.fixed
        B = self.A[:, :no].tocsr()
        Dm = sp.diags(free.astype(float))
        self._lu = spla.splu((Dm @ B @ Dm + sp.diags(self.fixed.astype(float))).tocsc())

    def amg_apply(self, r, z):
        no = self.no
        rv = np.where(self.fixed, 0.0, r.numpy()[:no])
        zv = self._lu.solve(rv)
        zv[self.fixed] = 0.0
        z.numpy()[:no] = zv
        z.numpy()[no:] = 0.0

    def apply_matrix(self, v, y):
        y.numpy()[:self.no] = self._apply(v.numpy())


class OracleNLOps(OracleOps):
    """OracleOps plus the nonlinear pieces polyfem_amd.distributed.DistributedNewton asks of its back end, computed by
    the oracle on the rank's local mesh (energy: on the sub-mesh of the elements the rank sums)."""

    def __init__(self, part, model, lam, mu, fixed_local):
        import dataclasses
        from oracle import pf_oracle as O
        self.O, self.part, self.model, self.lam, self.mu = O, part, model, lam, mu
        self.nloc = part.n_local * 3
        keep = part.elem_owned.astype(bool)
        m = part.mesh
        self.own_mesh = dataclasses.replace(m, conn=np.ascontiguousarray(m.conn[keep]), gconn=np.ascontiguousarray(m.gconn[keep]))
        empty = sp.csr_matrix((self.nloc, self.nloc))
        super().__init__(empty, part.n_owned * 3, fixed_local)
        self.hessians = 0

    def energy(self, x, out):
        out.numpy()[0] = self.O.assemble_energy(self.own_mesh, self.model, self.lam, self.mu, x.numpy())

    def gradient(self, x, g):
        g.numpy()[:] = self.O.assemble_gradient(self.part.mesh, self.model, self.lam, self.mu, x.numpy())

    def hessian(self, x, project_to_psd):
        f = self.O.assemble_hessian_psd if project_to_psd else self.O.assemble_hessian
        self.A = csc_tuple_to_csr(f(self.part.mesh, self.model, self.lam, self.mu, x.numpy()), self.nloc)[:self.no]
        self.hessians += 1

    def apply(self, v, y):
        y.numpy()[:self.no] = self._apply(v.numpy())

    def set_diag_shift(self, shift):
        self.shift = float(shift)

    def clear_numeric_flag(self):
        pass

    def set_mass(self, density):
        """Counterpart of Engine.inertia_set_mass: owned rows of the mass matrix of the local mesh."""
        self.M = csc_tuple_to_csr(self.O.assemble_mass(self.part.mesh, 3, density), self.nloc)[:self.no]

    def mass_apply(self, v, y):
        y.numpy()[:self.no] = self.M @ v.numpy()

    def hessian_add_mass(self, weight):
        self.A = (weight * self.A + self.M).tocsr()


def csc_tuple_to_csr(csc, n):
    outer, inner, values = csc
    return sp.csc_matrix((values, inner, outer), shape=(n, n)).tocsr()

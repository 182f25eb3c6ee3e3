"""Column-sharded single-signal Homotopy: ONE signal, the dictionary split by columns across
the ranks of a `torch.distributed` group — rank r owns A[:, lo_r:hi_r] — for dictionaries that do not fit one GPU or
whose per-iteration sweep dominates (engine-0 regime: n >> 10^6 columns).

It is the reference's iteration, statement for statement (`/root/reference/src/solvers/homotopy-cpu.cpp:186-275`:
residual_vector :87-98, find_max_gamma :100-164, inverse_add_or_remove :166-183, direction :257-267, inf_norm
:32-44; online_column_inverse `src/linalg/online_inverse.h:183-293`), with the O(m n) work local to the shard and
everything that is a reduction over all columns turned into a collective:

    p = A d, A x            local A_loc @ d_loc, A_loc @ x_loc      -> all_reduce(sum) of an m-vector
    q = A^T p, c = A^T r    local sweeps over the shard's columns    (no communication)
    ||c||_inf, first index  local (max, index)                       -> all_gather of one pair per rank
    step length, index      local scan with the reference's rules    -> all_gather of one pair per rank
    entering column a_idx   owner's column                           -> broadcast of an m-vector
    c on the support        owners' entries                          -> all_reduce(sum) of a K-vector

The active set — sorted support, the active columns, the explicit (A_S^T A_S)^-1, x_S, d_S — is replicated: every
rank performs the same O(K m + K^2) update on the same inputs.  Per iteration that is a handful of latency-bound
collectives of at most m floats; over xGMI (RCCL) they cost tens of microseconds, which is why this form only pays
where a sweep of the local shard costs more than that.  Bug-for-bug like the single-GPU path: the
first-step sign quirk (:223-227), strict `t > 0`, left-most tie-break, the residue on a leaving column.

The local sweeps are `torch.matmul` on whatever device the shard lives on (gloo + CPU tensors in the tests; with
`nccl` = RCCL on MI355X the shard is a device tensor).  `sweep_t` / `sweep_n` can be replaced, e.g. by the
HBM-bound kernels of libss_hip.so through `sship.Homotopy(A_loc).gemv_t`.
"""
import numpy as np


def _first_max_abs(v):
    """(max |v|, first index) like blas::ixamax; (-1, 0) for an empty shard"""
    import torch
    if v.numel() == 0:
        return -1.0, 0
    a = v.abs()
    m = torch.max(a)
    idx = int(torch.nonzero(a == m)[0].item()) if not torch.isnan(m) else 0
    return float(m), idx


class ColumnShardedHomotopy:
    def __init__(self, A_local, col_lo, n_total, group=None, sweep_t=None, sweep_n=None):
        """A_local: (m, n_local) torch tensor, the columns [col_lo, col_lo + n_local) of the m x n_total dictionary"""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.A = A_local
        self.m, self.nl = int(A_local.shape[0]), int(A_local.shape[1])
        self.lo, self.n = int(col_lo), int(n_total)
        self.dtype, self.dev = A_local.dtype, A_local.device
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # gloo moves host buffers, nccl (= RCCL) device buffers: collectives run on the matching side
        self.comm_dev = self.dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.sweep_t = sweep_t or (lambda v: self.A.t() @ v)        # c_loc = A_loc^T v
        self.sweep_n = sweep_n or (lambda v: self.A @ v)            # A_loc v_loc
        # who owns which column: gather the shard boundaries once
        lohi = [None] * self.world
        dist.all_gather_object(lohi, (self.lo, self.lo + self.nl), group=group)
        self.bounds = lohi

    def _owner(self, col):
        for r, (lo, hi) in enumerate(self.bounds):
            if lo <= col < hi:
                return r
        raise ValueError("column %d is owned by no rank" % col)

    def _gather_pairs(self, value, index):
        """every rank contributes (value, global index); returns the list of all pairs"""
        t = self.torch.tensor([value, float(index)], dtype=self.torch.float64, device=self.comm_dev)
        out = [self.torch.empty(2, dtype=self.torch.float64, device=self.comm_dev) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [(float(o[0]), int(o[1])) for o in out]

    def _allreduce(self, v):
        w = v.detach().to(self.comm_dev, copy=True)
        self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM, group=self.group)
        return w.to(v.device)

    def _inf_norm(self, c):
        """inf_norm (:32-44): max |c| over ALL columns and its first (left-most) index"""
        v, i = _first_max_abs(c)
        pairs = self._gather_pairs(v, self.lo + i)
        best = max(p[0] for p in pairs)
        idx = min(p[1] for p in pairs if p[0] == best)
        return self.torch.tensor(best, dtype=self.dtype).item(), idx

    def solve(self, y, tolerance, max_iterations, trace=False):
        """-> (x_local (n_local,), iter, solution_error[, trace]); y: (m,) tensor, identical on every rank"""
        torch, dist = self.torch, self.dist
        T = self.dtype
        eps = float(torch.finfo(T).eps)
        if not (max_iterations > 0 and eps <= tolerance < 1):
            raise ValueError("eps <= tolerance < 1 and max_iterations > 0 (homotopy-cpu.cpp:193-199)")
        tol = torch.tensor(tolerance, dtype=T).item()
        tmax = float(torch.finfo(T).max)
        y = y.to(self.dev)
        x = torch.zeros(self.nl, dtype=T, device=self.dev)
        d = torch.zeros(self.nl, dtype=T, device=self.dev)
        gam = []                      # sorted support (global column indices), replicated
        cols = {}                     # global index -> the column (m,), replicated, for columns ever inserted
        inv = torch.zeros((0, 0), dtype=T)
        path = {"idx": [], "added": [], "gamma": [], "c_inf": []}

        def residual():
            # residual_vector (:87-98): c = A^T (y - A x); A x over all columns = sum of the shards' parts
            r = y - self._allreduce(self.sweep_n(x))
            return self.sweep_t(r)

        def column(idx):
            # the entering column travels from its owner to every rank
            owner = self._owner(idx)
            col = (self.A[:, idx - self.lo].to(self.comm_dev, copy=True) if owner == self.rank
                   else torch.empty(self.m, dtype=T, device=self.comm_dev))
            dist.broadcast(col, src=owner if self.group is None else dist.get_global_rank(self.group, owner), group=self.group)
            return col.cpu()

        def toggle(idx):
            # inverse_add_or_remove (:166-183) with online_column_inverse insert / remove, replicated
            nonlocal inv
            rank = int(np.searchsorted(gam, idx))
            if rank < len(gam) and gam[rank] == idx:
                K = len(gam)
                gam.pop(rank)
                if K > 1:                                       # remove (online_inverse.h:253-293)
                    keep = [i for i in range(K) if i != rank]
                    dd = inv[rank, rank]
                    u = inv[keep, rank] * (-(1.0 / dd))
                    inv = inv[keep][:, keep] + (-dd * u)[:, None] * u[None, :]
                else:
                    inv = torch.zeros((0, 0), dtype=T)
                return False
            col = column(idx)
            cols[idx] = col
            K = len(gam)
            if K == 0:                                          # :193-201, through the norm
                nrm = torch.sqrt(torch.dot(col, col))
                inv = (1.0 / (nrm * nrm)).reshape(1, 1).to(T)
            else:                                               # bordering (:209-248), written in sorted order
                AS = torch.stack([cols[g] for g in gam])        # K x m
                u1 = AS @ col
                u2 = inv @ u1
                dv = 1.0 / (torch.dot(col, col) - torch.dot(u1, u2))
                grown = torch.empty((K + 1, K + 1), dtype=T)
                old = [i for i in range(K + 1) if i != rank]
                grown[np.ix_(old, old)] = inv + (dv * u2)[:, None] * u2[None, :]
                grown[rank, old] = -dv * u2
                grown[old, rank] = -dv * u2
                grown[rank, rank] = dv
                inv = grown
            gam.insert(rank, idx)
            return True

        def owned():
            """positions in the sorted support / local column numbers of the support columns this rank owns"""
            js = [j for j, g in enumerate(gam) if self.lo <= g < self.lo + self.nl]
            return js, [gam[j] - self.lo for j in js]

        def support_values(v):
            # entries of a sharded n-vector on the support, replicated (owners contribute, the others add zeros)
            out = torch.zeros(len(gam), dtype=T)
            js, ls = owned()
            if js:
                out[js] = v[ls].cpu()
            return self._allreduce(out)

        def set_direction(dS):
            d.zero_()
            js, ls = owned()
            if js:
                d[ls] = dS[js].to(self.dev)

        # ---- :215-229 ----------------------------------------------------------------------------------------
        c = residual()
        c_inf, idx = self._inf_norm(c)
        toggle(idx)
        seed = 0.0 if c_inf <= tol else 1.0                     # sign(c_inf, tol): the quirk — c_inf = |c[idx]| >= 0
        set_direction(torch.tensor([seed], dtype=T) * inv[0, 0])
        path["idx"].append(idx); path["added"].append(1); path["gamma"].append(0.0); path["c_inf"].append(c_inf)

        it = 0
        while True:                                             # do { ... } while (iter < max_iter && c_inf > tol)
            it += 1
            # find_max_gamma (:100-164): q = A^T (A d); the scan over this shard's columns
            p = self._allreduce(self.sweep_n(d))
            q = self.sweep_t(p)
            best, best_i = tmax, None
            if self.nl:
                insup = torch.zeros(self.nl, dtype=torch.bool, device=self.dev)
                loc = [g - self.lo for g in gam if self.lo <= g < self.lo + self.nl]
                if loc:
                    insup[loc] = True
                t_on = -x / d
                dl, dr = 1 - q, 1 + q
                t_l = (c_inf - c) / dl
                t_r = (c_inf + c) / dr
                big = torch.full_like(c, tmax)
                t_l = torch.where((dl != 0) & (t_l > 0), t_l, big)
                t_r = torch.where((dr != 0) & (t_r > 0), t_r, big)
                t_off = torch.minimum(t_l, t_r)
                t = torch.where(insup, torch.where(t_on > 0, t_on, big), t_off)
                mv = torch.min(t)
                if float(mv) < tmax:
                    best, best_i = float(mv), self.lo + int(torch.nonzero(t == mv)[0].item())
            pairs = self._gather_pairs(best, -1 if best_i is None else best_i)
            gmin = min(pv for pv, _ in pairs)
            if gmin < tmax:
                gamma, idx = gmin, min(pi for pv, pi in pairs if pv == gmin and pi >= 0)
            else:
                gamma, idx = tmax, 0                            # no positive candidate: (T_MAX, 0) like the reference
            gamma = torch.tensor(gamma, dtype=T).item()
            added = toggle(idx)                                 # :246
            if not gam:                                         # :248-249
                path["idx"].append(idx); path["added"].append(int(added)); path["gamma"].append(gamma); path["c_inf"].append(c_inf)
                break
            x += gamma * d                                      # :252 (d is the OLD support's direction)
            c = residual()                                      # :255
            sg = support_values(c)                              # :259-260
            sg = torch.where(sg > tol, torch.ones_like(sg), torch.where(sg < -tol, -torch.ones_like(sg), torch.zeros_like(sg)))
            set_direction(inv @ sg)                             # :263-266
            c_inf, _ = self._inf_norm(c)                        # :270
            path["idx"].append(idx); path["added"].append(int(added)); path["gamma"].append(gamma); path["c_inf"].append(c_inf)
            if not (it < max_iterations and c_inf > tol):
                break
        if trace:
            return x, it, float(c_inf), {k: np.asarray(v) for k, v in path.items()}
        return x, it, float(c_inf)

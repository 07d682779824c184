"""Multi-GPU driver/worker protocol of the patched mean-field object.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI).  Rank 0 owns the PySCF object: every
patched call (``get_jk``, the grid path's ``rho_fun`` / ``vxc_fun``, the two-electron gradient ``_jqc_jk_energy_per_atom``) first broadcasts a small header and its matrix
argument (the density matrix or the weighted potential), then every rank evaluates ITS share of the work -- shell-quartet
task rows for J/K, ranges of 256-point grid blocks for the grid path -- and the partial results meet in ONE all-reduce
(raw ``[vj; vk]``, ``rho`` or ``vxcmat``).  Ranks > 0 sit in ``serve()`` and mirror the calls:

    mf = joltqc_amd.pyscf.apply(mf, {"parallel": True})       # every rank, same molecule / basis
    if rank == 0:  e = mf.kernel(); joltqc_amd.pyscf.parallel.stop()
    else:          joltqc_amd.pyscf.parallel.serve(mf)

The reference has no multi-GPU path (one CuPy device); this replaces nothing there and follows the north star's
"single RCCL all-reduce of the Fock matrix per SCF iteration".
"""
import numpy as np

OP_STOP, OP_JK, OP_RHO, OP_VXC, OP_VV10, OP_GRADJK = 0, 1, 2, 3, 4, 5
_HEADER = 12


def world():
    """(rank, world_size) of the default process group; (0, 1) when torch.distributed is not initialised."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def _device():
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _bcast_header(vals):
    import torch
    import torch.distributed as dist
    h = torch.zeros(_HEADER, dtype=torch.float64, device=_device())
    if vals is not None:
        h[:len(vals)] = torch.tensor([float(v) for v in vals], dtype=torch.float64)
    dist.broadcast(h, src=0)
    return [float(x) for x in h.cpu()]


def _bcast_matrix(mat, shape):
    """Rank 0 passes ``mat``; the others pass None and receive a tensor of ``shape`` (kept on the collective's device)."""
    import torch
    import torch.distributed as dist
    dev = _device()
    if mat is not None:
        t = (mat if torch.is_tensor(mat) else torch.as_tensor(np.asarray(mat))).to(device=dev, dtype=torch.float64).contiguous()
    else:
        t = torch.empty(shape, dtype=torch.float64, device=dev)
    dist.broadcast(t, src=0)
    return t


def drive_jk(get_jk_sharded):
    """Rank-0 wrapper of a sharded ``get_jk`` closure: announce the call, broadcast D, then take part in it."""
    def get_jk(mol_ref=None, dm=None, hermi=0, vhfopt=None, with_j=True, with_k=True, omega=None, verbose=None):
        import torch
        d = dm if torch.is_tensor(dm) else np.asarray(dm)
        shape = tuple(d.shape)
        _bcast_header([OP_JK, len(shape), shape[0] if len(shape) == 3 else 1, shape[-1], hermi, int(with_j), int(with_k),
                       float(omega) if omega else 0.0])
        t = _bcast_matrix(d, shape)
        out = get_jk_sharded(mol_ref, t, hermi, vhfopt, with_j, with_k, omega, verbose)
        if isinstance(dm, np.ndarray) and getattr(get_jk, "return_numpy", False):
            out = tuple(x.cpu().numpy() if torch.is_tensor(x) else x for x in out)
        return out
    for k in ("quartet_counts", "stats", "layout", "set_probe", "set_streams", "set_increment_of"):
        if hasattr(get_jk_sharded, k):
            setattr(get_jk, k, getattr(get_jk_sharded, k))
    get_jk.return_numpy = False
    return get_jk


def drive_grad_jk(fn_sharded):
    """Rank-0 wrapper of a sharded ``jk_energy_per_atom`` (pyscf/grad.py): announce, broadcast the densities, take part; the
    partial per-atom gradients meet in the closure's all-reduce of natm x 3 doubles."""
    def jk_energy_per_atom(mol=None, dm=None, j_factor=1.0, k_factor=1.0, omega=None, hermi=1, verbose=None):
        import torch
        d = dm if torch.is_tensor(dm) else np.asarray(dm)
        shape = tuple(d.shape)
        _bcast_header([OP_GRADJK, len(shape), shape[0] if len(shape) == 3 else 1, shape[-1], float(j_factor), float(k_factor),
                       float(omega) if omega else 0.0])
        out = fn_sharded(mol, _bcast_matrix(d, shape), j_factor, k_factor, omega, hermi, verbose)
        if isinstance(dm, np.ndarray) and torch.is_tensor(out):
            out = out.cpu().numpy()
        return out
    for k in ("stats", "quartet_count", "layout"):
        if hasattr(fn_sharded, k):
            setattr(jk_energy_per_atom, k, getattr(fn_sharded, k))
    return jk_energy_per_atom


def drive_grid(fn, op, which=0):
    """Rank-0 wrapper of the sharded ``rho_fun`` / ``vxc_fun``: broadcast (xc type, which grid, matrix), then take part.
    ``which``: 0 = the object's ``grids``, 1 = its ``nlcgrids``."""
    XC = {"LDA": 1, "GGA": 4, "MGGA": 5}

    def wrapped(mol, grids, xctype, mat, ref=None):
        import torch
        m = mat if torch.is_tensor(mat) else np.asarray(mat)
        shape = tuple(m.shape)
        if len(shape) not in (1, 2):      # checked BEFORE anything is announced: a shape the workers cannot rebuild would
            raise ValueError(f"rho_fun / vxc_fun take one matrix [n0, n1] or one vector [n1], got shape {shape}")   # desynchronise the ranks
        # (ref: magnitude of the full matrix an increment belongs to -- rks._windows; 0 = none)
        _bcast_header([op, len(shape), shape[0], shape[-1], XC[xctype.upper()], which, float(ref) if ref else 0.0])
        return fn(mol, grids, xctype, _bcast_matrix(m, shape), ref)
    if hasattr(fn, "stats"):
        wrapped.stats = fn.stats
    return wrapped


def drive_vv10(sums):
    """Rank-0 wrapper of ``rks.vv10_sums``: broadcast the outer and inner point tables, then take part."""
    def wrapped(outer, inner, fp32):
        _bcast_header([OP_VV10, outer.shape[0], outer.shape[1], inner.shape[0], inner.shape[1], int(bool(fp32))])
        return sums(_bcast_matrix(outer, tuple(outer.shape)), _bcast_matrix(inner, tuple(inner.shape)), fp32)
    return wrapped


def stop():
    """Rank 0: release the workers from ``serve()``."""
    if world()[1] > 1:
        _bcast_header([OP_STOP])


class driving:
    """``with parallel.driving(): mf.kernel()`` on rank 0: the workers are released from ``serve()`` whatever happens inside
    the block (an exception between a call's announcement and its all-reduce still leaves them inside that collective: give
    the process group a timeout, ``init_process_group(..., timeout=...)``, so that they fail instead of waiting forever)."""

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            stop()
        except Exception:            # noqa: BLE001  (the group may already be broken: do not mask the original error)
            if exc_type is None:
                raise
        return False


def serve(handlers):
    """Ranks > 0: mirror rank 0's calls until ``stop()``.  ``handlers`` = {OP_JK: sharded get_jk, OP_RHO / OP_VXC: {grid
    index: (rho_fun / vxc_fun, mol, grids or a callable returning them)}, OP_VV10: vv10_sums} or a patched mean-field
    object (its ``_jqc_parallel`` attribute)."""
    if not isinstance(handlers, dict):
        handlers = handlers._jqc_parallel
    XC = {1: "LDA", 4: "GGA", 5: "MGGA"}
    ncalls = 0
    while True:
        h = _bcast_header(None)
        op = int(h[0])
        if op == OP_STOP:
            return ncalls
        ncalls += 1
        if op in (OP_RHO, OP_VXC):
            assert int(h[1]) in (1, 2), f"grid call announced with a {int(h[1])}-dimensional argument"
        if op == OP_JK:
            ndim, n_dm, nao = int(h[1]), int(h[2]), int(h[3])
            shape = (n_dm, nao, nao) if ndim == 3 else (nao, nao)
            dm = _bcast_matrix(None, shape)
            handlers[OP_JK](None, dm, int(h[4]), None, bool(h[5]), bool(h[6]), h[7] if h[7] > 0 else None, None)
        elif op in (OP_RHO, OP_VXC):
            ndim, n0, n1 = int(h[1]), int(h[2]), int(h[3])
            mat = _bcast_matrix(None, (n0, n1) if ndim == 2 else (n1,))
            fn, mol, grids = handlers[op][int(h[5])]
            fn(mol, grids() if callable(grids) else grids, XC[int(h[4])], mat, h[6] if h[6] > 0 else None)
        elif op == OP_GRADJK:
            ndim, n_dm, nao = int(h[1]), int(h[2]), int(h[3])
            dm = _bcast_matrix(None, (n_dm, nao, nao) if ndim == 3 else (nao, nao))
            handlers[OP_GRADJK](None, dm, h[4], h[5], h[6] if h[6] > 0 else None)
        elif op == OP_VV10:
            outer = _bcast_matrix(None, (int(h[1]), int(h[2])))
            inner = _bcast_matrix(None, (int(h[3]), int(h[4])))
            handlers[OP_VV10](outer, inner, bool(h[5]))
        else:
            raise RuntimeError(f"unknown operation code {op} in the worker loop")


def split_blocks(cost, rank, nranks):
    """Contiguous range [b0, b1) of grid blocks for ``rank``: prefix sums of the per-block cost cut into ``nranks``
    equal parts (every rank computes the same cuts)."""
    cost = np.asarray(cost, dtype=np.float64)
    n = len(cost)
    if nranks <= 1 or n == 0:
        return 0, n
    cum = np.concatenate([[0.0], np.cumsum(np.maximum(cost, 1e-30))])
    cuts = np.searchsorted(cum, cum[-1] * np.arange(nranks + 1) / nranks, side="left")
    cuts[0], cuts[-1] = 0, n
    cuts = np.maximum.accumulate(np.minimum(cuts, n))
    return int(cuts[rank]), int(cuts[rank + 1])

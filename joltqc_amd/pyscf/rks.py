"""DFT numerical integration on MI355X behind the reference's ``_numint`` interface.

Mirrors ``/root/reference/jqc/pyscf/rks.py``: ``generate_rks_kernel`` (:285) -> ``rks_fun`` (:308, the
incremental ``nr_rks``), ``rho_fun`` (:366), ``vxc_fun`` (:515); ``generate_get_rho`` (:263),
``generate_nr_rks`` (:270), ``generate_nr_nlc_vxc`` (:661); ``build_grids`` (:100) and the RKS
``get_veff`` (:180-260); VV10 driver ``vv10nlc`` (``jqc/backend/rks.py:542-715``).

MI355X-first organisation (see joltqc_amd/csrc/dft_kernels.inc): per 256-point block the significant
shells are found once (cached per grid and cutoff bucket), the block's AOs are evaluated in batches
into an HBM workspace and contracted with FP64 MFMA (``C = D_bb AO_b`` for the density,
``V_bb = AO_b X_b^T`` for the potential).  XC functional evaluation itself stays with the caller's
``ni.eval_xc_eff`` (libxc, third party), exactly as in the reference (:341).

Precision windows as in the reference (rks.py:446-493): the block's AO rows are sorted by their log estimate, so the
AO pairs above ``cutoff_fp64`` form a corner of the block's pair matrix that goes through the FP64 MFMA, the band down to
``cutoff_fp32`` through the FP32 MFMA (twice the rate), the rest is skipped.
"""
import math
import os

import numpy as np

from ..backend import lib as _lib

__all__ = ["generate_rks_kernel", "generate_get_rho", "generate_nr_rks", "generate_nr_nlc_vxc", "build_grids",
           "vv10nlc", "patch"]

ao_cutoff = 1e-13                    # reference rks.py:55
DIM_BY_XC = {"LDA": 1, "GGA": 4, "MGGA": 5}
NG = 256
WORKSPACE_BYTES = 6 << 30            # AO workspace per batch: floor; raised to WORKSPACE_FRACTION of the device memory
WORKSPACE_FRACTION = 0.10            # (288 GB of HBM: 28 GB, i.e. a 112-atom / def2-TZVPP GGA grid of 3.7e5 points in ONE batch)


def _t(x, dev):
    import torch
    if torch.is_tensor(x):
        return x.to(device=dev, dtype=torch.float64)
    return torch.as_tensor(np.asarray(x), dtype=torch.float64, device=dev)


class _GridCache:
    """Padded SoA coordinates + per-block shell lists of one grid (identity + cutoff bucket)."""

    def __init__(self):
        self.key = None
        self.sparsity = {}

    def coords(self, grids, dev):
        import torch
        c = grids.coords
        # the keyed array itself is kept (an id() can be reused by a new array once the old one is collected); a grid
        # whose coords were replaced -- or whose generation counter (bumped by build_grids) changed -- is re-staged
        key = (c, tuple(c.shape), getattr(grids, "_jqc_generation", 0))
        if self.key is None or self.key[0] is not c or self.key[1:] != key[1:]:
            ct = _t(c, dev)
            n = ct.shape[0]
            npad = (-n) % NG
            if npad:
                ct = torch.cat([ct, ct[-1:].expand(npad, 3)], dim=0)
            self.key = key
            self.generation = getattr(self, "generation", 0) + 1     # incremental caches keyed on a grid check this
            self.ngrids = n
            self.ngrids_pad = n + npad
            self.soa = ct.T.contiguous()
            self.sparsity = {}
        return self.soa

    def shells(self, layout, soa, log_cutoff):
        """Per-block shell lists for a cutoff; bucketed downwards (= more shells, safe) and cached."""
        import torch
        bucket = math.floor(log_cutoff * 2.0) / 2.0
        if bucket not in self.sparsity:
            dev = soa.device
            L = _lib.lib()
            nbas = layout.nbasis
            nblk = self.ngrids_pad // NG
            shell_list = torch.empty((nblk, nbas), dtype=torch.int16, device=dev)
            row_of = torch.empty((nblk, nbas), dtype=torch.int32, device=dev)
            nshl = torch.empty(nblk, dtype=torch.int32, device=dev)
            nrow = torch.empty(nblk, dtype=torch.int32, device=dev)
            shell_la = torch.empty((nblk, nbas), dtype=torch.float32, device=dev)
            _lib.check(L.jqc_dft_ao_screen(soa.data_ptr(), self.ngrids_pad, layout.basis_data_fp64["packed"].data_ptr(),
                                           layout.device_ao_loc().data_ptr(), nbas, float(bucket), shell_list.data_ptr(),
                                           row_of.data_ptr(), nshl.data_ptr(), nrow.data_ptr(), shell_la.data_ptr(),
                                           _lib.stream_ptr()))
            nrow_h = nrow.cpu().numpy().astype(np.int64)
            if len(self.sparsity) > 16:
                self.sparsity.clear()
            self.sparsity[bucket] = (shell_list, row_of, nshl, nrow, nrow_h, shell_la)
        return self.sparsity[bucket]


def _batches(nrow_h, ncomp, workspace_bytes=WORKSPACE_BYTES):
    """Split the blocks into batches whose padded AO rows fit the workspace; yields (blk0, nblk, row_base, rows)."""
    pad = (nrow_h + 15) // 16 * 16
    max_rows = max(int(workspace_bytes // (ncomp * NG * 8)), int(pad.max()) if pad.size else 16)
    b0, n = 0, len(pad)
    while b0 < n:
        acc, b1 = 0, b0
        while b1 < n and acc + pad[b1] <= max_rows:
            acc += pad[b1]
            b1 += 1
        base = np.concatenate([[0], np.cumsum(pad[b0:b1])[:-1]]).astype(np.int64)
        yield b0, b1 - b0, base, max(int(acc), 16)
        b0 = b1


class IncrementPolicy:
    """When does an incremental build (potential of D = potential of D_last + potential of D - D_last) start over from the full
    matrix?  Every increment drops the terms below the absolute cutoffs and, in the mixed-precision windows, rounds to FP32;
    neither error shrinks with the increment, so over the 20-30 iterations of an SCF run they pile up (measured on 112 atoms /
    B3LYP / def2-SVP, profiles/r04_config3_scf_noise.txt: E_xc 3-5e-6 Eh off the from-scratch value, jittering by 1e-6 per
    iteration).  Rule: a full build when the largest element of the increment has fallen below ``SHRINK`` x its value at the last
    full build, and after ``MAX_STEPS`` increments in a row -- 3-4 full builds per SCF run instead of one.  An increment that has
    GROWN by ``GROW`` since the last full build is a restart (a second ``kernel()`` on the same closures, a new ``dm0``, a
    stability follow-up: the density jumps while the reference value is that of a converged run) and is a full build as well."""
    SHRINK = 1e-3
    GROW = 10.0
    MAX_STEPS = 12

    def __init__(self):
        self.reset()

    def reset(self):
        self.ref, self.steps = None, 0

    def full_build(self, ddmax):
        """``ddmax``: largest |element| of this call's increment.  True: build from the full matrix (and restart the count)."""
        if self.ref is None or ddmax < self.SHRINK * self.ref or ddmax > self.GROW * self.ref or self.steps >= self.MAX_STEPS:
            self.ref, self.steps = max(float(ddmax), 1e-300), 0
            return True
        self.steps += 1
        return False


def generate_rks_kernel(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, shard=None):
    """``shard=(rank, world_size)``: this process evaluates a contiguous range of the 256-point grid blocks (cut by the
    blocks' AO-pair counts) and the partial ``rho`` / ``vxcmat`` are summed over ranks with one all-reduce each."""
    import torch
    layout = basis_layout
    nao = layout.nao
    rank, nranks = shard if shard is not None else (0, 1)
    cache = {"dm_prev": 0, "rho_prev": 0, "wv_prev": 0, "vxcmat_prev": 0, "grid": None}
    policy = IncrementPolicy()
    gcache = _GridCache()
    state = {"ws": None, "stats": {}}
    log_ao_cutoff = math.log(min(ao_cutoff, cutoff_fp32))
    # precision windows of the AO-pair contributions (reference rks.py:446-493): >= cutoff_fp64 in FP64, [cutoff_fp32,
    # cutoff_fp64) in FP32, below cutoff_fp32 dropped
    log_cut32 = math.log(cutoff_fp32)
    log_cut64 = math.log(max(cutoff_fp64, cutoff_fp32))

    def _windows(log_mag, ref):
        """Thresholds on log AO_a + log AO_b of one call whose matrix has largest element exp(log_mag): pairs above ``thr64``
        go through the FP64 MFMA, pairs in (thr32, thr64] through the FP32 one, the rest is skipped (cutoff_fp32 on the pair's
        contribution, as in the reference).  For an INCREMENT of a matrix whose largest element is ``ref`` the FP64 threshold is
        the one of a build of the full matrix: the reference's absolute window (pair contribution < cutoff_fp64 -> FP32)
        hands a shrinking increment to FP32 almost entirely, and the 6e-8 relative error of ~1e5 pairs per point of mixed sign,
        each up to cutoff_fp64, integrates to 1e-6 Eh of jitter in E_xc per SCF iteration
        (profiles/r04_config3_scf_noise.txt); with the full matrix's threshold every increment carries the relative precision
        of a full build."""
        log_ref = log_mag if ref is None else max(log_mag, math.log(float(ref) + 1e-300))
        thr32 = log_cut32 - log_mag
        return max(log_cut64 - log_ref, thr32), thr32

    def _workspace(dev, rows, ncomp):
        need = rows * NG * ncomp
        if state["ws"] is None or state["ws"].numel() < need:
            state["ws"] = torch.empty(need, dtype=torch.float64, device=dev)
        return state["ws"]

    def _run(grids, ncomp_ao, log_cutoff, body, eval_ao=None):
        dev = _lib.require_gpu()
        L = _lib.lib()
        soa = gcache.coords(grids, dev)
        shell_list, row_of, nshl, nrow, nrow_h, shell_la = gcache.shells(layout, soa, log_cutoff)
        basis = layout.basis_data_fp64["packed"]
        stream = _lib.stream_ptr()
        rows_total = 0
        b_lo, b_hi = 0, len(nrow_h)
        if nranks > 1:
            from .parallel import split_blocks
            b_lo, b_hi = split_blocks(nrow_h.astype(np.float64) ** 2 + 64.0 * nrow_h + 1.0, rank, nranks)
        state["stats"]["block_range"] = (b_lo, b_hi)
        # the batch plan (and its row-base table on the device) only depends on the shell lists: kept with them, so that a
        # call issues no blocking host-to-device copy
        pkey = (id(nrow_h), ncomp_ao, b_lo, b_hi)
        plans = state.setdefault("plans", {})
        if pkey not in plans or plans[pkey][0] is not nrow_h:          # (the array is kept: an id() can be reused)
            cap = max(WORKSPACE_BYTES, int(WORKSPACE_FRACTION * torch.cuda.get_device_properties(dev).total_memory))
            plan = []
            for blk0, nblk, base, rows in _batches(nrow_h[b_lo:b_hi], ncomp_ao, cap):
                # launch order of the contraction kernels: blocks with the most AO rows (work ~ rows^2) first, so that the
                # longest blocks do not start last (1 452 blocks of 200..1 100 rows on 256 CUs)
                order = np.argsort(-nrow_h[b_lo + blk0:b_lo + blk0 + nblk], kind="stable").astype(np.int32)
                plan.append((blk0 + b_lo, nblk, torch.from_numpy(base).to(dev), rows,
                             torch.empty(rows, dtype=torch.int32, device=dev), torch.empty(rows, dtype=torch.float32, device=dev),
                             torch.from_numpy(order).to(dev)))
            if len(plans) > 8:
                plans.clear()
            plans[pkey] = (nrow_h, plan)
        for blk0, nblk, base_d, rows, ao_idx, row_la, order_d in plans[pkey][1]:
            state["order"] = order_d
            ws = _workspace(dev, rows, ncomp_ao)
            comp_stride = rows * NG
            if eval_ao is None:
                _lib.check(L.jqc_dft_eval_ao(soa.data_ptr(), gcache.ngrids_pad, basis.data_ptr(), layout.nbasis, blk0, nblk,
                                             shell_list.data_ptr(), row_of.data_ptr(), nshl.data_ptr(), nrow.data_ptr(),
                                             base_d.data_ptr(), ncomp_ao, comp_stride, ws.data_ptr(), ao_idx.data_ptr(),
                                             shell_la.data_ptr(), row_la.data_ptr(), stream))
            else:
                eval_ao(L, soa, basis, blk0, nblk, shell_list, row_of, nshl, nrow, base_d, comp_stride, ws, ao_idx, shell_la,
                        row_la, stream)
            body(L, blk0, nblk, nrow, base_d, comp_stride, ws, ao_idx, row_la, stream)
            rows_total += int(rows)
        state["stats"]["ao_rows"] = rows_total
        state["stats"]["blocks"] = len(nrow_h)
        state["stats"]["nrow_h"] = nrow_h

    def rho_fun(mol, grids, xctype, dm, ref=None):
        """rho[ndim, ngrids] (ndim 1/4/5) for a density matrix in the molecule's AO basis
        (reference rho_fun, rks.py:366-513).  ``ref``: when ``dm`` is an INCREMENT of a density matrix whose largest element
        is ``ref``, the FP64 / FP32 split treats every AO pair as a build of that full matrix would (see ``_windows``)."""
        dev = _lib.require_gpu()
        xctype = xctype.upper()
        ndim = DIM_BY_XC[xctype]
        d = layout.dm_from_mol(_t(dm, dev).reshape(layout.nao_mol, layout.nao_mol))
        d = (0.5 * (d + d.T)).contiguous()
        log_dm = math.log(float(d.abs().max().item()) + 1e-200)          # reference :402-405
        thr64, thr32 = _windows(log_dm, ref)
        soa = gcache.coords(grids, dev)
        rho = torch.zeros((ndim, gcache.ngrids_pad), dtype=torch.float64, device=dev)

        def body(L, blk0, nblk, nrow, base_d, comp_stride, ws, ao_idx, row_la, stream):
            # AO pairs with log_ao_a + log_ao_b + log|D|max in [log cutoff_fp32, log cutoff_fp64) go through the FP32 MFMA,
            # above through the FP64 one, below nowhere (reference rks.py:446-493, eval_rho.cu:93-106)
            _lib.check(L.jqc_dft_rho(blk0, nblk, gcache.ngrids_pad, nrow.data_ptr(), base_d.data_ptr(), comp_stride,
                                     ws.data_ptr(), ao_idx.data_ptr(), d.data_ptr(), nao, ndim, rho.data_ptr(),
                                     row_la.data_ptr(), thr64, thr32, state["order"].data_ptr(), stream))
        _run(grids, 1 if ndim == 1 else 4, log_ao_cutoff - log_dm, body)
        if nranks > 1:
            import torch.distributed as dist
            dist.all_reduce(rho)                  # every rank filled its own block range, zeros elsewhere
        return rho[:, :gcache.ngrids]

    def vxc_fun(mol, grids, xctype, wv, ref=None):
        """V_xc matrix in the molecule's AO basis from weighted potential wv[ndim, ngrids]
        (reference vxc_fun, rks.py:515-656).  ``ref``: largest |wv| of the full potential ``wv`` is an increment of."""
        dev = _lib.require_gpu()
        xctype = xctype.upper()
        ndim = DIM_BY_XC[xctype]
        soa = gcache.coords(grids, dev)
        w = _t(wv, dev).reshape(ndim, -1)
        if w.shape[1] != gcache.ngrids_pad:
            wp = torch.zeros((ndim, gcache.ngrids_pad), dtype=torch.float64, device=dev)
            wp[:, :w.shape[1]] = w
            w = wp
        w = w.contiguous()
        ngrids_per_atom = gcache.ngrids / max(getattr(mol, "natm", 1), 1)
        log_wv_max = math.log((float(w.abs().max().item()) + 1e-300) * ngrids_per_atom)   # reference :548-551
        thr64, thr32 = _windows(log_wv_max, None if ref is None else ref * ngrids_per_atom)
        vmat = torch.zeros((nao, nao), dtype=torch.float64, device=dev)

        def body(L, blk0, nblk, nrow, base_d, comp_stride, ws, ao_idx, row_la, stream):
            _lib.check(L.jqc_dft_vxc(blk0, nblk, gcache.ngrids_pad, nrow.data_ptr(), base_d.data_ptr(), comp_stride,
                                     ws.data_ptr(), ao_idx.data_ptr(), w.data_ptr(), ndim, nao, vmat.data_ptr(),
                                     row_la.data_ptr(), thr64, thr32, state["order"].data_ptr(), stream))
        _run(grids, 1 if ndim == 1 else 4, log_ao_cutoff - log_wv_max, body)
        if nranks > 1:
            import torch.distributed as dist
            dist.all_reduce(vmat)                 # the one collective of nr_rks: sum of the ranks' block ranges
        return layout.dm_to_mol(vmat + vmat.T)          # the kernel accumulates T = phi X^T only (reference epilogue A + A^T, :654-655)

    def xcgrad_fun(mol, grids, xctype, dm, wv):
        """Nuclear gradient [natm, 3] of E_xc at fixed density matrix and fixed grid (no grid response), LDA / GGA, for the
        weighted potential ``wv[ndim, ngrids]`` = weights x vxc that ``vxc_fun`` takes, ndim 1 / 4 / 5 (the reference has
        no gradient code and defers to GPU4PySCF).  FP64 throughout; AO pairs below cutoff_fp32 are skipped."""
        dev = _lib.require_gpu()
        xctype = xctype.upper()
        ndim = DIM_BY_XC[xctype]
        d = layout.dm_from_mol(_t(dm, dev).reshape(layout.nao_mol, layout.nao_mol))
        d = (0.5 * (d + d.T)).contiguous()
        log_dm = math.log(float(d.abs().max().item()) + 1e-200)
        gcache.coords(grids, dev)
        w = _t(wv, dev).reshape(ndim, -1)
        if w.shape[1] != gcache.ngrids_pad:
            wp = torch.zeros((ndim, gcache.ngrids_pad), dtype=torch.float64, device=dev)
            wp[:, :w.shape[1]] = w
            w = wp
        w = w.contiguous()
        log_wv = math.log(float(w.abs().max().item()) + 1e-300)
        gao = torch.zeros((nao, 3), dtype=torch.float64, device=dev)

        def eval_ao(L, soa, basis, blk0, nblk, shell_list, row_of, nshl, nrow, base_d, comp_stride, ws, ao_idx, shell_la, row_la,
                    stream):
            _lib.check(L.jqc_dft_xcgrad_ao(soa.data_ptr(), gcache.ngrids_pad, basis.data_ptr(), layout.nbasis, blk0, nblk,
                                           shell_list.data_ptr(), row_of.data_ptr(), nshl.data_ptr(), nrow.data_ptr(),
                                           base_d.data_ptr(), w.data_ptr(), ndim, comp_stride, ws.data_ptr(), ao_idx.data_ptr(),
                                           shell_la.data_ptr(), row_la.data_ptr(), stream))

        def body(L, blk0, nblk, nrow, base_d, comp_stride, ws, ao_idx, row_la, stream):
            _lib.check(L.jqc_dft_xcgrad(blk0, nblk, nrow.data_ptr(), base_d.data_ptr(), comp_stride, ws.data_ptr(),
                                        ao_idx.data_ptr(), d.data_ptr(), nao, gao.data_ptr(), row_la.data_ptr(),
                                        log_cut32 - log_dm - log_wv, state["order"].data_ptr(), ndim, stream))
        _run(grids, 14 if ndim > 4 else 8, log_ao_cutoff - max(log_dm, 0.0) - max(log_wv, 0.0), body, eval_ao)
        if "ao_atom" not in state:
            ao_atom = np.repeat(layout.atom_of, np.diff(layout.ao_loc))
            state["ao_atom"] = torch.from_numpy(ao_atom.astype(np.int64)).to(dev)
            state["natm"] = int(getattr(mol, "natm", int(layout.atom_of.max()) + 1))
        out = torch.zeros((state["natm"], 3), dtype=torch.float64, device=dev)
        out.index_add_(0, state["ao_atom"], gao)
        if nranks > 1:
            import torch.distributed as dist
            dist.all_reduce(out)
        if isinstance(dm, np.ndarray):
            return out.cpu().numpy()
        return out

    def _eval_xc(ni, xc_code, rho, xctype, dev):
        """``ni.eval_xc_eff`` (libxc, third party) on the caller's side of the boundary: a plain PySCF NumInt takes and
        returns NumPy arrays, a device-resident one (GPU4PySCF-like, ``_jqc_numpy_boundary`` False) device arrays."""
        if getattr(ni, "_jqc_numpy_boundary", False):      # (unmarked NumInt: device-side, like the reference's generate_* closures)
            exc, vxc = ni.eval_xc_eff(xc_code, rho.cpu().numpy(), deriv=1, xctype=xctype)[:2]
        else:
            exc, vxc = ni.eval_xc_eff(xc_code, rho, deriv=1, xctype=xctype)[:2]
        return _t(exc, dev).reshape(-1), _t(vxc, dev).reshape(rho.shape[0], -1)

    def rks_fun(ni, mol, grids, xc_code, dm):
        """Incremental nr_rks (reference rks.py:308-364): returns (nelec, excsum, vxcmat).  Increments as in the reference,
        with two refinements (``IncrementPolicy``): the density and the potential are rebuilt from the full matrices whenever
        the increment has shrunk by 1e3 since the last full build (and every 12 calls), so that dropped sub-cutoff terms and FP32
        rounding do not accumulate over an SCF run, and the FP32 window of an increment is the one of the full matrix."""
        dev = _lib.require_gpu()
        xctype = ni._xc_type(xc_code) if hasattr(ni, "_xc_type") else _xc_type(xc_code)
        gcache.coords(grids, dev)
        if cache["grid"] != gcache.generation:      # another grid (new geometry, rebuilt grid): the increments restart
            cache.update(dm_prev=0, rho_prev=0, wv_prev=0, vxcmat_prev=0, grid=gcache.generation)
            policy.reset()
        weights = _t(grids.weights, dev)
        dm_t = _t(dm, dev)
        ddm = dm_t - cache["dm_prev"]
        dmax, ddmax = (float(x) for x in torch.stack([dm_t.abs().max(), ddm.abs().max()]).tolist())
        full = policy.full_build(ddmax) or not torch.is_tensor(cache["dm_prev"])
        if full:
            rho = rho_k(mol, grids, xctype, dm_t)
        else:
            rho = cache["rho_prev"] + rho_k(mol, grids, xctype, ddm, dmax)
        exc, vxc = _eval_xc(ni, xc_code, rho, xctype, dev)
        den = rho[0] * weights
        nelec = float(den.sum())
        excsum = float((den * exc).sum())
        wv = vxc * weights
        if full:
            vxcmat = vxc_k(mol, grids, xctype, wv)
        else:
            vxcmat = cache["vxcmat_prev"] + vxc_k(mol, grids, xctype, wv - cache["wv_prev"], float(wv.abs().max()))
        cache.update(dm_prev=dm_t.clone(), rho_prev=rho, wv_prev=wv, vxcmat_prev=vxcmat.clone())
        state["stats"]["full_build"] = full
        if getattr(ni, "_jqc_numpy_boundary", False):
            return nelec, excsum, vxcmat.cpu().numpy()
        return nelec, excsum, vxcmat

    # rank 0 of a multi-GPU run announces every grid call to the workers (parallel.drive_grid); set by ``patch``
    rho_k, vxc_k = rho_fun, vxc_fun

    def set_drivers(rho_d, vxc_d):
        nonlocal rho_k, vxc_k
        rho_k, vxc_k = rho_d, vxc_d
    rks_fun.set_drivers = set_drivers
    rks_fun.gcache = gcache
    rks_fun.reset_cache = lambda: (cache.update(dm_prev=0, rho_prev=0, wv_prev=0, vxcmat_prev=0), policy.reset())

    rho_fun.stats = state["stats"]
    vxc_fun.stats = state["stats"]
    rks_fun.xcgrad_fun = xcgrad_fun
    return rks_fun, rho_fun, vxc_fun


def _xc_type(xc_code):
    from pyscf.dft import libxc           # only reachable when PySCF exists
    return libxc.xc_type(xc_code)


def generate_get_rho(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, kernels=None):
    _, rho_fun, _ = kernels or generate_rks_kernel(basis_layout, cutoff_fp64, cutoff_fp32)

    def get_rho(mol, dm, grids, *args, **kwargs):
        return rho_fun(mol, grids, "LDA", dm)[0]
    return get_rho


def generate_nr_rks(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, kernels=None):
    rks_fun, _, _ = kernels or generate_rks_kernel(basis_layout, cutoff_fp64, cutoff_fp32)
    return rks_fun


# --------------------------------------------------------------------------------------------- VV10
def vv10_sums(outer, inner, fp32=True, shard=None):
    """The O(N_outer x N_inner) part of VV10: F, U, W sums of ``vv10_kernel`` (reference dft/vv10.cu:29-118).
    ``outer`` = [x, y, z, W0, K][n_o] and ``inner`` = [x, y, z, W0p, Kp, RpW][n_i] device matrices, both padded to 256.
    ``shard=(rank, world)``: this process takes a contiguous range of the outer points; ONE all-reduce sums the stacked
    [F; U; W] (zeros outside the own range)."""
    import torch
    dev = outer.device
    L = _lib.lib()
    no_pad, ni_pad = int(outer.shape[1]), int(inner.shape[1])
    out = torch.zeros((3, no_pad), dtype=torch.float64, device=dev)
    rank, nranks = shard if shard is not None else (0, 1)
    nblk = no_pad // NG
    b0, b1 = (nblk * rank) // nranks, (nblk * (rank + 1)) // nranks
    if b1 > b0 and ni_pad:
        o = outer[:, b0 * NG:b1 * NG].contiguous()
        # FP32 inner loop: with K, Kp >= 1e-3 everywhere the denominator g' (g gt)^2 >= 4e-15 cannot fail the reference's
        # `> 1e-30` test (vv10.cu:104), which the kernel then skips (mode 3, include/jqc_hip.h); one device reduction decides
        mode = 0 if not fp32 else (3 if float(torch.minimum(o[4].min(), inner[4].min())) >= 1e-3 else 1)
        res = torch.empty((3, (b1 - b0) * NG), dtype=torch.float64, device=dev)
        _lib.check(L.jqc_vv10(res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), inner[:3].contiguous().data_ptr(),
                              o[:3].contiguous().data_ptr(), inner[3].data_ptr(), o[3].contiguous().data_ptr(),
                              o[4].contiguous().data_ptr(), inner[4].data_ptr(), inner[5].data_ptr(), ni_pad,
                              (b1 - b0) * NG, mode, _lib.stream_ptr()))
        out[:, b0 * NG:b1 * NG] = res
    if nranks > 1:
        import torch.distributed as dist
        dist.all_reduce(out)
    return out


def vv10nlc(rho, coords, vvrho, vvweight, vvcoords, nlc_pars, dtype=np.float32, sums=vv10_sums):
    """VV10 non-local correlation: exc[ngrids], vxc[2, ngrids] (reference jqc/backend/rks.py:542-715).
    ``sums``: the kernel call (rank 0 of a multi-GPU run passes the broadcasting form, parallel.drive_vv10)."""
    import torch
    dev = _lib.require_gpu()
    rho, vvrho = _t(rho, dev), _t(vvrho, dev)
    coords, vvcoords, vvweight = _t(coords, dev), _t(vvcoords, dev), _t(vvweight, dev)
    thresh = 1e-10
    m = rho[0] >= thresh
    mi = vvrho[0] >= thresh
    dens, g2 = rho[0][m], (rho[1:4][:, m] ** 2).sum(0)
    idens, ig2 = vvrho[0][mi], (vvrho[1:4][:, mi] ** 2).sum(0)
    Bvv, Cvv = nlc_pars
    Pi43 = 4.0 * math.pi / 3.0
    Kvv = Bvv * 1.5 * math.pi * ((9.0 * math.pi) ** (-1.0 / 6.0))
    Beta = ((3.0 / (Bvv * Bvv)) ** 0.75) / 32.0
    W0p = torch.sqrt(Cvv * (ig2 / idens ** 2) ** 2 + Pi43 * idens)
    Kp = Kvv * idens ** (1.0 / 6.0)
    W0tmp = Cvv * (g2 / dens ** 2) ** 2
    W0 = torch.sqrt(W0tmp + Pi43 * dens)
    K = Kvv * dens ** (1.0 / 6.0)
    dKdR = K / 6.0
    RpW = idens * vvweight[mi]

    def pad(a, n, fill):
        out = torch.full((n,) + tuple(a.shape[1:]), fill, dtype=torch.float64, device=dev)
        out[: a.shape[0]] = a
        return out
    n_o, n_i = int(dens.numel()), int(idens.numel())
    no_pad, ni_pad = n_o + (-n_o) % NG, n_i + (-n_i) % NG
    outer = torch.cat([pad(coords[m], no_pad, 0.0).T, pad(W0, no_pad, 1.0)[None], pad(K, no_pad, 1.0)[None]]).contiguous()
    # padding points of the inner grid: far away, zero weight
    inner = torch.cat([pad(vvcoords[mi], ni_pad, 1.0e4).T, pad(W0p, ni_pad, 1.0)[None], pad(Kp, ni_pad, 1.0)[None],
                       pad(RpW, ni_pad, 0.0)[None]]).contiguous()
    if no_pad and ni_pad:
        F, U, W = sums(outer, inner, np.dtype(dtype) == np.float32)[:, :n_o]
    else:
        F = U = W = torch.zeros(n_o, dtype=torch.float64, device=dev)
    dW0dR = (0.5 * Pi43 * dens - 2.0 * W0tmp) / W0
    dW0dG = W0tmp * dens / (g2 * W0)
    n = rho.shape[1]
    exc = torch.zeros(n, dtype=torch.float64, device=dev)
    vxc = torch.zeros((2, n), dtype=torch.float64, device=dev)
    exc[m] = Beta + 0.5 * F
    vxc[0, m] = Beta + F + 1.5 * (U * dKdR + W * dW0dR)
    vxc[1, m] = 1.5 * W * dW0dG
    return exc, vxc


def transform_vxc_gga(rho, vxc):
    """(d e/d rho, d e/d sigma) -> weights on (rho, grad rho) for a spin-restricted GGA
    (what gpu4pyscf's xc_deriv.transform_vxc(rho, vxc, 'GGA', spin=0) returns; reference rks.py:700)."""
    import torch
    out = torch.empty_like(rho[:4])
    out[0] = vxc[0]
    out[1:4] = 2.0 * vxc[1] * rho[1:4]
    return out


def generate_nr_nlc_vxc(basis_layout, cutoff_fp64=1e-13, cutoff_fp32=1e-13, shard=None):
    """Incremental nr_nlc_vxc (reference rks.py:661-714)."""
    _, rho_fun, vxc_fun = generate_rks_kernel(basis_layout, cutoff_fp64, cutoff_fp32, shard)
    cache = {"dm_prev": 0, "rho_prev": 0, "wv_prev": 0, "vmat_prev": 0, "grid": None}
    policy = IncrementPolicy()
    calls = {"rho": rho_fun, "vxc": vxc_fun, "sums": lambda o, i, f: vv10_sums(o, i, f, shard)}
    gc = _GridCache()

    def nr_nlc_vxc(ni, mol, grids, xc_code, dms):
        import torch
        dev = _lib.require_gpu()
        gc.coords(grids, dev)
        if cache["grid"] != gc.generation:
            cache.update(dm_prev=0, rho_prev=0, wv_prev=0, vmat_prev=0, grid=gc.generation)
            policy.reset()
        dm_t = _t(dms, dev)
        ddm = dm_t - cache["dm_prev"]
        dmax, ddmax = (float(x) for x in torch.stack([dm_t.abs().max(), ddm.abs().max()]).tolist())
        full = policy.full_build(ddmax) or not torch.is_tensor(cache["dm_prev"])          # (same rule as rks_fun)
        rho = calls["rho"](mol, grids, "GGA", dm_t) if full else cache["rho_prev"] + calls["rho"](mol, grids, "GGA", ddm, dmax)
        weights, coords = _t(grids.weights, dev), _t(grids.coords, dev)
        exc, vxc = 0, 0
        for nlc_pars, fac in ni.nlc_coeff(xc_code):
            e, v = vv10nlc(rho, coords, rho, weights, coords, nlc_pars, sums=calls["sums"])
            exc = exc + e * fac
            vxc = vxc + v * fac
        den = rho[0] * weights
        nelec = float(den.sum())
        excsum = float((den * exc).sum())
        wv = transform_vxc_gga(rho, vxc) * weights
        if full:
            vmat = calls["vxc"](mol, grids, "GGA", wv)
        else:
            vmat = cache["vmat_prev"] + calls["vxc"](mol, grids, "GGA", wv - cache["wv_prev"], float(wv.abs().max()))
        cache.update(dm_prev=dm_t.clone(), rho_prev=rho, wv_prev=wv, vmat_prev=vmat.clone())
        if getattr(ni, "_jqc_numpy_boundary", False):
            return nelec, excsum, vmat.cpu().numpy()
        return nelec, excsum, vmat
    nr_nlc_vxc.calls = calls
    nr_nlc_vxc.kernels = (rho_fun, vxc_fun)
    return nr_nlc_vxc


# --------------------------------------------------------------------------------------------- grids
GROUP_BOX_SIZE = 3.0      # reference rks.py:56


def arg_group_grids(coords, box_size=GROUP_BOX_SIZE):
    """Order that groups grid points by cubic boxes so that 256-point blocks are spatially compact
    (reference arg_group_grids, rks.py:71-97).  (Tried on the 112-atom Becke grid and rejected: boxes along a Morton curve -- sum of
    squared AO counts per block +3 %, rho 9.0 -> 10.2 ms; balanced k-d bisection into 256-point leaves -- +5.7 %, 9.2 -> 10.3 ms.
    Atom-centred shells of points are already compact in the row order.)"""
    c = np.asarray(coords)
    lo = c.min(axis=0)
    box = np.floor((c - lo) / box_size).astype(np.int64)
    nb = box.max(axis=0) + 1
    key = (box[:, 0] * nb[1] + box[:, 1]) * nb[2] + box[:, 2]
    return np.argsort(key, kind="stable")


def build_grids(grids, mol=None, with_non0tab=False, sort_grids=True, **kwargs):
    """``grids.build`` replacement (reference rks.py:100-177): generate the grid with the object's own
    (PySCF/GPU4PySCF) machinery, then pad to a multiple of 256 with zero-weight points and sort by boxes."""
    orig = getattr(grids, "_jqc_original_build", None)
    if orig is None:
        raise RuntimeError("build_grids needs the object's original build method (installed by apply())")
    orig(mol, with_non0tab=with_non0tab, sort_grids=sort_grids, **kwargs)
    coords, weights = np.asarray(grids.coords), np.asarray(grids.weights)
    order = arg_group_grids(coords)
    coords, weights = coords[order], weights[order]
    npad = (-coords.shape[0]) % NG
    if npad:
        coords = np.vstack([coords, np.repeat(coords[-1:], npad, axis=0)])
        weights = np.concatenate([weights, np.zeros(npad)])
    grids.coords, grids.weights = coords, weights
    grids._jqc_generation = getattr(grids, "_jqc_generation", 0) + 1
    return grids


def tag_array(x, **attrs):
    """Attach ecoul/exc/vj/vk to the returned potential (role of gpu4pyscf's tag_array, reference rks.py:181,259)."""
    if isinstance(x, np.ndarray):
        try:
            from pyscf.lib import tag_array as _tag
            return _tag(x, **attrs)
        except ImportError:
            class _Tagged(np.ndarray):
                pass
            x = x.view(_Tagged)
    for k, v in attrs.items():
        setattr(x, k, v)
    return x


def generate_get_veff():
    """RKS get_veff: XC from the grid path + J (and scaled K for hybrids / range-separated hybrids) from the
    patched get_j / get_jk / get_k, incremental in the density (same logic as reference rks.py:180-260)."""
    import torch
    policy = IncrementPolicy()

    def get_veff(ks, mol=None, dm=None, dm_last=0, vhf_last=0, hermi=1):
        if mol is None:
            mol = ks.mol
        if dm is None:
            dm = ks.make_rdm1()
        dev = _lib.require_gpu()
        if hasattr(ks, "initialize_grids"):
            ks.initialize_grids(mol, dm)
        else:                                       # plain PySCF: rks.get_veff builds the grids on first use
            if getattr(ks.grids, "coords", None) is None:
                ks.grids.build(with_non0tab=False)
            if (hasattr(ks, "do_nlc") and ks.do_nlc() and getattr(ks, "nlcgrids", None) is not None
                    and getattr(ks.nlcgrids, "coords", None) is None):
                ks.nlcgrids.build(with_non0tab=False)
        dm_t = _t(dm, dev)
        ground_state = dm_t.ndim == 2
        ni = ks._numint
        if hermi == 2:
            n, exc, vxc = 0, 0, 0
        else:
            n, exc, vxc = ni.nr_rks(mol, ks.grids, ks.xc, dm_t)
            vxc = _t(vxc, dev)                    # (NumPy when the object is a plain CPU one)
            if hasattr(ks, "do_nlc") and ks.do_nlc():
                xc = ks.xc if ni.libxc.is_nlc(ks.xc) else ks.nlc
                n, enlc, vnlc = ni.nr_nlc_vxc(mol, ks.nlcgrids, xc, dm_t)
                exc += enlc
                vxc = vxc + _t(vnlc, dev)
        is_hybrid = ni.libxc.is_hybrid_xc(ks.xc) if hasattr(ni, "libxc") else False
        incremental = getattr(ks, "_eri", None) is None and getattr(ks, "direct_scf", True)
        if incremental and getattr(vhf_last, "vj", None) is not None:
            # J/K increments start over from the full density by the rule of the grid path (IncrementPolicy)
            ddm = dm_t - _t(dm_last, dev)
            incremental = not policy.full_build(float(ddm.abs().max()))
        else:
            incremental = False
            policy.reset()
            policy.full_build(float(dm_t.abs().max()))
        if not is_hybrid:
            vk = None
            if incremental:
                announce = getattr(ks.get_jk, "set_increment_of", None)       # (see the hybrid branch)
                if announce:
                    announce(float(dm_t.abs().max()))
                try:
                    vj = _t(ks.get_j(mol, ddm, hermi), dev) + _t(vhf_last.vj, dev)
                finally:
                    if announce:
                        announce(None)
            else:
                vj = _t(ks.get_j(mol, dm_t, hermi), dev)
            vxc = vxc + vj
        else:
            omega, alpha, hyb = ni.rsh_and_hybrid_coeff(ks.xc, spin=getattr(mol, "spin", 0))
            last = incremental
            d = ddm if last else dm_t
            # (mixed-precision J/K of an increment is split where a build of the full matrix would be split: pyscf/jk.py)
            announce = getattr(ks.get_jk, "set_increment_of", None) if last else None
            if announce:
                announce(float(dm_t.abs().max()))
            try:
                vj, vk = ks.get_jk(mol, d, hermi)
                vj, vk = _t(vj, dev), _t(vk, dev) * hyb
                if abs(omega) > 1e-10:                    # long-range exchange of range-separated hybrids
                    vk = vk + _t(ks.get_k(mol, d, hermi, omega=omega), dev) * (alpha - hyb)
            finally:
                if announce:
                    announce(None)
            if last:
                vj = vj + _t(vhf_last.vj, dev)
                vk = vk + _t(vhf_last.vk, dev)
            vxc = vxc + vj - 0.5 * vk
            if ground_state:
                exc -= float((dm_t * vk.T).sum()) * 0.25
        ecoul = float((dm_t * vj.T).sum()) * 0.5 if ground_state else None
        if getattr(ks, "_jqc_numpy_boundary", False):
            vxc, vj = vxc.cpu().numpy(), vj.cpu().numpy()
            vk = vk.cpu().numpy() if vk is not None else None
        return tag_array(vxc, ecoul=ecoul, exc=exc, vj=vj, vk=vk)
    return get_veff


def patch(obj, basis_layout, cutoff_fp32, cutoff_fp64, numpy_boundary, shard=None):
    """Install the grid-path closures on an RKS object (reference __init__.py:191-206).  ``shard=(rank, world)``: the grid
    blocks (and the VV10 outer points) are shared by the ranks; rank 0's closures announce every call to the workers
    (parallel.serve on the other ranks, handlers in ``obj._jqc_parallel``)."""
    from types import MethodType
    from . import parallel as par
    ni = obj._numint
    kernels = generate_rks_kernel(basis_layout, cutoff_fp32=cutoff_fp32, cutoff_fp64=cutoff_fp64, shard=shard)
    rks_fun, rho_fun, vxc_fun = kernels
    nlc = generate_nr_nlc_vxc(basis_layout, cutoff_fp32=cutoff_fp32, cutoff_fp64=cutoff_fp64, shard=shard)
    if shard is not None and shard[1] > 1:
        handlers = getattr(obj, "_jqc_parallel", None)
        handlers = {} if handlers is None else handlers
        mol = obj.mol

        def main_grid():
            if getattr(obj.grids, "coords", None) is None:
                obj.grids.build(with_non0tab=False)
            return obj.grids

        def nlc_grid():
            if getattr(obj.nlcgrids, "coords", None) is None:
                obj.nlcgrids.build(with_non0tab=False)
            return obj.nlcgrids
        handlers[par.OP_RHO] = {0: (rho_fun, mol, main_grid), 1: (nlc.kernels[0], mol, nlc_grid)}
        handlers[par.OP_VXC] = {0: (vxc_fun, mol, main_grid), 1: (nlc.kernels[1], mol, nlc_grid)}
        handlers[par.OP_VV10] = lambda o, i, f: vv10_sums(o, i, f, shard)
        obj._jqc_parallel = handlers
        if shard[0] == 0:
            rho_d, vxc_d = par.drive_grid(rho_fun, par.OP_RHO, 0), par.drive_grid(vxc_fun, par.OP_VXC, 0)
            rks_fun.set_drivers(rho_d, vxc_d)
            kernels = (rks_fun, rho_d, vxc_d)
            nlc.calls.update(rho=par.drive_grid(nlc.kernels[0], par.OP_RHO, 1), vxc=par.drive_grid(nlc.kernels[1], par.OP_VXC, 1),
                             sums=par.drive_vv10(lambda o, i, f: vv10_sums(o, i, f, shard)))
    ni.get_rho = generate_get_rho(basis_layout, kernels=kernels)
    # XC part of the forces (LDA / GGA, no grid response): ``obj._jqc_xc_energy_per_atom(mol, grids, xctype, dm, wv)``.  Unsharded
    # in a multi-GPU run as well (the worker protocol announces rho / vxc / VV10 calls only)
    obj._jqc_xc_energy_per_atom = (rks_fun.xcgrad_fun if shard is None or shard[1] == 1 else
                                   generate_rks_kernel(basis_layout, cutoff_fp32=cutoff_fp32, cutoff_fp64=cutoff_fp64)[0].xcgrad_fun)
    ni.nr_rks = MethodType(rks_fun, ni)
    ni.nr_nlc_vxc = MethodType(nlc, ni)
    ni._jqc_numpy_boundary = numpy_boundary
    for g in {id(x): x for x in (getattr(obj, "grids", None), getattr(obj, "nlcgrids", None)) if x is not None}.values():
        if hasattr(g, "build") and not hasattr(g, "_jqc_original_build"):
            g._jqc_original_build = g.build
            g.build = MethodType(build_grids, g)
    obj._jqc_numpy_boundary = numpy_boundary
    obj.get_veff = MethodType(generate_get_veff(), obj)
    return obj

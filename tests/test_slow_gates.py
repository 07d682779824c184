"""Gates that do not fit the 600 s budget of the driver-run GPU suite (advisor finding of round 5): selected with ``-m slow`` on the round's
final GPU pass (tools/gpu_runs/), skipped without a GPU, NOT marked ``gpu`` so that the driver's ``-m gpu`` run leaves them out.  Their last
green run is recorded under profiles/ and in the evidence string of joltqc_amd/data/verified_kernels.json."""
import pytest


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return True


pytestmark = [pytest.mark.slow, pytest.mark.skipif(_no_gpu(), reason="needs an MI355X")]


def test_every_angular_class_with_three_density_matrices_s_to_g(monkeypatch):
    """The three-matrix call (a pair contracted against one integral evaluation + the odd tail) over ALL 140 classes s..g -- the driver-run
    suite covers s..d (25 classes); this is the part that includes the quad / chunked-quad, h-form and k-chunk builds of the f and g classes
    (reference: every density matrix against one evaluation, jk/1q1t.cu:423-638)."""
    import test_jk_gpu
    monkeypatch.setenv("JQC_TEST_FULL_3DM", "1")
    test_jk_gpu._CLASS_ORACLE.clear()
    test_jk_gpu.test_every_angular_class_against_the_oracle("jk_3dm", monkeypatch)


def test_112_atoms_tzvpp_long_range_tiled_vs_queue_kernels():
    """The long-range (omega = 0.3) J/K build at the north-star size, tiled kernels against the independent queue kernels -- the leg the
    driver-run suite keeps only at config-4 size."""
    import os
    import sys
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import big_check
    import test_configs_gpu as tc
    from joltqc_amd.pyscf import jk as jkmod
    mol, lay, dm = big_check.setup("0112-elongated-nitrogenous", "def2-tzvpp")
    g = jkmod.generate_jk_kernel(lay, 1e-13, 1e-13)
    vj, vk = (x.clone() for x in g(mol, dm, hermi=1, omega=0.3))
    sc = float(max(vj.abs().max(), vk.abs().max()))
    gq = tc._queue_kernels(lay)
    try:
        qj, qk = gq(mol, dm, hermi=1, omega=0.3)
        assert float((qj - vj).abs().max()) < 1e-11 * sc and float((qk - vk).abs().max()) < 1e-11 * sc
    finally:
        tc._restore_router()

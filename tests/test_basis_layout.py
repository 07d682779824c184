"""Host logic of BasisLayout (modelled on the reference's tests/test_basis_layout.py)."""
import numpy as np
import pytest

from conftest import H2O, benzene_atoms
from joltqc_amd.constants import NPRIM_MAX
from joltqc_amd.gto import mole
from joltqc_amd.gto.c2s import cart2sph_l
from joltqc_amd.pyscf.basis import BasisLayout, split_basis


@pytest.fixture(scope="module")
def mol():
    return mole.Mole(atom=H2O, basis="def2-tzvpp")


def test_split_respects_nprim_max(mol):
    shells, parent = split_basis(mol)
    assert max(len(s.exps) for s in shells) <= NPRIM_MAX
    # O 6-primitive s -> two pieces, O 4-primitive p -> 3+1, sharing their parent
    assert len(shells) == 25 and parent.max() + 1 == 23
    assert np.all(np.diff(parent) >= 0)


def test_groups_sorted_and_padded(mol):
    for align in (1, 4):
        lay = BasisLayout.from_mol(mol, alignment=align)
        key = lay.group_key
        order = [(int(l), -int(n)) for l, n in key]
        assert order == sorted(order)
        sizes = np.diff(lay.group_offset)
        assert np.all(sizes % align == 0)
        for g in range(lay.ngroups):
            sl = slice(lay.group_offset[g], lay.group_offset[g + 1])
            assert np.all(lay.angs[sl] == key[g, 0]) and np.all(lay.nprims[sl] == key[g, 1])
        # pads have zero AO width and never change nao
        w = np.diff(lay.ao_loc)
        assert np.all(w[lay.pad_id] == 0)
        assert lay.nao == 70
        assert np.all(lay.packed[:, 3] == lay.ao_loc[:-1])


def test_packed_rows(mol):
    lay = BasisLayout.from_mol(mol)
    for n in range(lay.nbasis):
        np_ = int(lay.packed[n, 10])
        assert np_ == lay.nprims[n] and int(lay.packed[n, 11]) == lay.angs[n]
        assert np.all(lay.packed[n, 5:5 + 2 * np_:2] > 0)
        assert np.all(lay.packed[n, 4 + 2 * np_:10:2] == 0)


@pytest.mark.parametrize("cart", [False, True])
def test_dm_roundtrip_dimensions_and_trace(cart):
    mol = mole.Mole(atom=H2O, basis="def2-tzvpp", cart=cart)
    lay = BasisLayout.from_mol(mol)
    T = lay.transform_matrix()
    assert T.shape == (lay.nao, mol.nao)
    rng = np.random.default_rng(0)
    d = rng.random((mol.nao, mol.nao))
    d = d + d.T
    v = rng.random((lay.nao, lay.nao))
    # <D_int, V_int> == <D_mol, V_mol>: the two transforms are adjoint
    lhs = np.sum((T @ d @ T.T) * v)
    rhs = np.sum(d * (T.T @ v @ T))
    assert abs(lhs - rhs) < 1e-9 * abs(lhs)


def test_spatial_sort_is_a_permutation():
    mol = mole.Mole(atom=benzene_atoms(), basis="def2-svp")
    a = BasisLayout.from_mol(mol, spatial_sort=True)
    b = BasisLayout.from_mol(mol, spatial_sort=False)
    assert sorted(a.to_split_map.tolist()) == sorted(b.to_split_map.tolist())
    assert np.array_equal(a.group_key, b.group_key) and np.array_equal(a.group_offset, b.group_offset)


def test_cart2sph_known_values():
    d = cart2sph_l(2)
    assert abs(d[1, 0] - 1.092548430592079070) < 1e-15        # d_xy
    assert abs(d[5, 2] - 0.630783130505040012) < 1e-15        # d_z2: zz
    assert abs(d[0, 2] + 0.315391565252520002) < 1e-15
    assert abs(d[0, 4] - 0.546274215296039535) < 1e-15        # d_x2-y2
    f = cart2sph_l(3)
    assert abs(f[4, 1] - 2.890611442640554055) < 1e-14        # f_xyz
    assert abs(f[9, 3] - 0.746352665180230782) < 1e-14        # f_z3: zzz
    for l in range(2, 5):
        c = cart2sph_l(l)
        assert c.shape == ((l + 1) * (l + 2) // 2, 2 * l + 1)


@pytest.mark.parametrize("cart", [False, True])
def test_general_contractions_are_decontracted_and_split(cart):
    """nctr > 1 shells and contractions of 4-5 primitives (reference split_basis, basis.py:678-837; the patterns its
    tests/test_basis_sets_jk.py runs through with 6-31G / cc-pVDZ / cc-pVTZ): the layout of the generally contracted basis gives
    the same overlap and the same J/K (CPU oracle) as the same functions written as segmented shells."""
    from conftest import GENERAL_BASIS, SEGMENTED_BASIS
    from oracle import dense
    m1 = mole.Mole(atom=H2O, basis=GENERAL_BASIS, cart=cart)
    m2 = mole.Mole(atom=H2O, basis=SEGMENTED_BASIS, cart=cart)
    assert m1.nao == m2.nao and (np.asarray(m1._bas)[:, 3] > 1).any()                 # general contractions present
    l1, l2 = BasisLayout.from_mol(m1), BasisLayout.from_mol(m2)
    assert l1.nprims.max() <= NPRIM_MAX and l2.nprims.max() <= NPRIM_MAX
    s1, s2 = dense.int1e_mol(l1, m1)[0], dense.int1e_mol(l2, m2)[0]
    assert np.abs(s1 - s2).max() < 1e-13
    rng = np.random.default_rng(1)
    d = rng.random((m1.nao, m1.nao))
    d = d + d.T
    (j1, k1), (j2, k2) = dense.get_jk(l1, d, 1), dense.get_jk(l2, d, 1)
    assert np.abs(j1 - j2).max() < 1e-12 * np.abs(j1).max() and np.abs(k1 - k2).max() < 1e-12 * np.abs(k1).max()


def test_no_splitting_needed_and_alignments():
    """Reference test_split_basis_no_splitting_needed / test_different_alignments / test_padding_logic: a basis whose
    contractions have at most NPRIM_MAX primitives is only sorted; every alignment pads each (l, nprim) group to a multiple of
    itself with zero-width copies and leaves the AO count alone."""
    basis = {"O": [[0, [8.0, 0.3], [1.5, 0.5], [0.4, 0.4]], [0, [0.2, 1.0]], [1, [1.2, 0.6], [0.3, 0.5]], [2, [0.8, 1.0]]],
             "H": [[0, [3.0, 0.2], [0.5, 0.8]], [1, [0.7, 1.0]]]}
    m = mole.Mole(atom=H2O, basis=basis)
    shells, parent = split_basis(m)
    assert len(shells) == np.asarray(m._bas).shape[0] and np.array_equal(parent, np.arange(len(shells)))
    base = BasisLayout.from_mol(m, alignment=1)
    assert not base.pad_id.any() and base.nbasis == len(shells)
    for align in (2, 4, 8):
        lay = BasisLayout.from_mol(m, alignment=align)
        assert np.all(np.diff(lay.group_offset) % align == 0) and lay.nao == base.nao
        assert (~lay.pad_id).sum() == base.nbasis and np.all(np.diff(lay.ao_loc)[lay.pad_id] == 0)
        # a pad is a copy of a real shell of its own group (finite exponents, harmless in every kernel)
        for g in range(lay.ngroups):
            sl = slice(lay.group_offset[g], lay.group_offset[g + 1])
            real = lay.packed[sl][~lay.pad_id[sl]]
            for row in lay.packed[sl][lay.pad_id[sl]]:
                assert any(np.array_equal(row[[0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11]], r[[0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11]]) for r in real)


@pytest.mark.parametrize("cart", [False, True])
def test_maps_and_transformation_consistency(cart):
    """Reference test_basis_mapping_consistency / test_to_decontracted_map_consistency / test_matrix_transformation_stability:
    every internal shell points at a decontracted parent of the same angular momentum and atom; the overlap of the internal
    (split, Cartesian) functions folded back with T is the overlap of the molecule's normalised AOs.  (Stacked matrices through
    the device transforms: tests/test_jk_gpu.py::test_dm_transforms_of_stacked_matrices.)"""
    from conftest import GENERAL_BASIS
    from oracle import md_eri
    m = mole.Mole(atom=H2O, basis=GENERAL_BASIS, cart=cart)
    lay = BasisLayout.from_mol(m, alignment=4)
    shells, parent = split_basis(m)
    assert len(parent) == len(shells) and parent.max() + 1 == int(np.asarray(m._bas)[:, 3].sum())
    for n in range(lay.nbasis):
        s = shells[lay.to_split_map[n]]
        assert s.l == lay.angs[n] and s.atom == lay.atom_of[n] and len(s.exps) == lay.nprims[n]
    # overlap of the internal (split, Cartesian) functions folded back with T: symmetric, positive definite, and -- for a
    # spherical basis, whose AOs PySCF normalises to one -- with a unit diagonal (checks T, the s/p factors and the split
    # coefficients together; libcint's Cartesian d..g components are not all unit-normalised)
    T = lay.transform_matrix()
    s_int = md_eri.int1e(lay.packed, lay.ao_loc, m.atom_coords(), m.atom_charges())[0]
    s_mol = T.T @ s_int @ T
    assert np.abs(s_mol - s_mol.T).max() < 1e-13 and np.linalg.eigvalsh(s_mol).min() > 1e-4
    if not cart:
        assert np.abs(np.diag(s_mol) - 1.0).max() < 1e-10


def test_tiles_are_compact_clusters_and_the_order_is_a_deterministic_permutation():
    """Tile composition (pyscf/basis.py:_cluster_tiles / _refine_tiles): whatever the mode, the layout is a permutation of the
    split shells grouped by (l, nprim); the default clusters have a markedly smaller radius than runs cut from a Morton curve
    (that is what raises the survival of the quartet screening); two builds give the same order."""
    import joltqc_amd.pyscf.basis as B
    from joltqc_amd.constants import tile_width
    mol = mole.Mole(atom=benzene_atoms() + [("C", (6.0 + 1.4 * k, 0.3 * k, 0.2 * k * k)) for k in range(6)], basis="def2-tzvpp")

    def radii(lay):
        out = {}
        for g in range(lay.ngroups):
            w = tile_width(int(lay.group_key[g, 0]))
            c = lay.packed[lay.group_offset[g]:lay.group_offset[g + 1], :3]
            n = len(c) // w
            if w > 1 and n > 1:
                t = c[:n * w].reshape(n, w, 3)
                out[tuple(lay.group_key[g])] = float(np.sqrt(((t - t.mean(1, keepdims=True)) ** 2).sum(2)).max(1).mean())
        return out

    saved = B.SPATIAL_MODE, B.EXP_WEIGHT
    try:
        lays = {}
        for mode, wgt in (("morton", 0.0), ("cluster2", 0.0), (saved[0], saved[1])):
            B.SPATIAL_MODE, B.EXP_WEIGHT = mode, wgt
            lay = BasisLayout.from_mol(mol, alignment=tile_width)
            real = lay.to_split_map[~lay.pad_id]
            assert sorted(real.tolist()) == list(range(len(real)))
            for g in range(lay.ngroups):
                sl = slice(lay.group_offset[g], lay.group_offset[g + 1])
                assert np.all(lay.angs[sl] == lay.group_key[g, 0]) and np.all(lay.nprims[sl] == lay.group_key[g, 1])
            lays[mode] = lay
        again = BasisLayout.from_mol(mol, alignment=tile_width)
        assert np.array_equal(again.to_split_map, lays[saved[0]].to_split_map)
        r_m, r_c = radii(lays["morton"]), radii(lays["cluster2"])
        # (the 8-wide s tiles of so small a molecule span most of it either way: the p and d groups show the effect)
        pd = [k for k in r_m if k[0] in (1, 2) and k[1] == 1]
        assert pd and all(r_c[k] < 0.8 * r_m[k] for k in pd), (r_m, r_c)
        assert sum(r_c.values()) < sum(r_m.values())
    finally:
        B.SPATIAL_MODE, B.EXP_WEIGHT = saved

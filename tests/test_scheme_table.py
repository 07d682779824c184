"""Host logic around the gfx950 scheme table (joltqc_amd/data/gfx950_scheme.json, role of the reference's
jqc/backend/data/optimal_scheme_<GPU>_fp64.json + backend/jk.py:37-53): table shape, the small-launch override, what the
tuner may pick, and the generator's degrade path for variants that do not fit LDS.  No GPU needed (compile-only)."""
import glob
import json
import os
import subprocess

import pytest

from joltqc_amd.backend import jk as router
from joltqc_amd.backend import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _classes(lmax):
    return [(a, b, c, d) for a in range(lmax + 1) for b in range(a + 1) for c in range(a + 1) for d in range(c + 1)]


def test_every_class_up_to_g_has_an_entry_and_obeys_the_build_rules():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import autotune
    with open(os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")) as f:
        sch = json.load(f)
    for prec in ("fp64", "fp32"):
        for ang in _classes(4):
            v = sch[prec].get(router.class_key(ang))
            assert v is not None, (prec, ang)
            assert (v & 0xf) in (L.ALGO_TILE, L.ALGO_TILE1Q) or ((v & 0xf) == L.ALGO_TILE512 and v & router.VARIANT_KW), (prec, ang, hex(v))
            if v & router.VARIANT_KW:      # k chunks on wave groups: exactly two chunks, a quartet inside one wave
                assert router.k_chunks(ang, v) == 2 and router.forced_variant(ang, v) == v, (prec, ang, hex(v))
            assert autotune.allowed(v), (prec, ang, hex(v))         # no banned register allocation in the table
            if (v & 0xf) == L.ALGO_TILE1Q:
                # one lane per quartet up to 200 integrals; the quad form a third of up to 330 per lane, in chunks above 180
                assert router.nint(ang) <= (router.QUAD_FORCE_MAX if v & router.VARIANT_QUAD else router.TILE1Q_FORCE_MAX)
                assert router.forced_variant(ang, v) == v, (prec, ang, hex(v))      # (the table holds what the class supports)
                if v & router.VARIANT_QUAD and router.nint(ang) > 180:
                    assert (v >> 25) & 3, (prec, ang, hex(v))
            else:
                assert not (v & 0x3000)                              # several ket pairs: lane-per-quartet mode only
    for key, v in sch["fp64_small"].items():
        assert autotune.allowed(v) and key in sch["fp64"]


def test_small_launches_use_the_override_table(monkeypatch):
    monkeypatch.delenv("JQC_JK_ALGO", raising=False)
    with open(os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")) as f:
        sch = json.load(f)
    changed = 0
    for ang in _classes(3):
        key = router.class_key(ang)
        big, small = router.select_algo(ang), router.select_algo(ang, small=True)
        assert big == sch["fp64"][key]
        assert small == sch["fp64_small"].get(key, big)
        changed += big != small
        assert router.select_algo(ang, True, small=True) == sch["fp32"][key]      # fp32 has one table
    assert changed == len(sch["fp64_small"]) > 0


def test_forced_variant_respects_what_a_class_supports():
    assert router.forced_variant((3, 3, 3, 3), 0x2022) == L.ALGO_TILE          # integral block too large for one lane
    assert router.forced_variant((2, 1, 1, 0), 0x2022) == 0x2022
    assert router.forced_variant((2, 1, 1, 0), 0x1021) == 0x21                # row lanes: one ket pair per iteration
    assert router.forced_variant((2, 1, 1, 0), 0x421) & 0x400                 # 18 row lanes fit one wave
    assert not router.forced_variant((3, 3, 1, 0), 0x421) & 0x400             # 100 row lanes do not


def test_quad_variants_only_where_the_class_supports_them():
    """JQC_VARIANT_QUAD (bit 24) needs a lane-per-quartet build of a class with a p shell and at most four Rys roots; the chunk bits
    (25-28, JQC_VARIANT_QCHUNK) a shell other than the split p shell whose component count the chunk count divides."""
    Q = router.VARIANT_QUAD
    assert router.forced_variant((2, 1, 1, 1), 0x1122 | Q) == 0x1122 | Q
    assert router.forced_variant((2, 0, 2, 0), 0x1122 | Q) == 0x1122                  # no p shell
    assert router.forced_variant((3, 3, 1, 1), 0x1122 | Q) == L.ALGO_TILE             # 900 integrals: row lanes
    assert router.forced_variant((2, 2, 1, 1), 0x1122 | Q) & Q                        # 324 integrals, 4 roots: allowed (spills; chunks help)
    assert router.forced_variant((2, 1, 1, 1), 0x21 | Q) == 0x21                      # row-lane algorithm: bit dropped
    two_over_i, three_over_k = router.VARIANT_QCHUNK(1, 0), router.VARIANT_QCHUNK(2, 2)
    assert router.forced_variant((2, 1, 2, 1), 0x1122 | Q | two_over_i) == 0x1122 | Q | two_over_i      # d shell: 6 components, 2 chunks
    assert router.forced_variant((1, 1, 1, 1), 0x1122 | Q | two_over_i) == 0x1122 | Q                   # p shell: 3 components, not by 2
    assert router.forced_variant((2, 1, 2, 1), 0x1122 | Q | three_over_k) == 0x1122 | Q | three_over_k
    assert router.forced_variant((2, 2, 1, 0), 0x1122 | Q | three_over_k) == 0x1122 | Q                 # k IS the split p shell
    assert router.forced_variant((2, 1, 1, 1), 0x1122 | two_over_i) == 0x1122                            # chunks without the quad bit
    assert not router.supports_mixed((2, 1, 1, 1), 0x1122 | Q) and router.supports_ndm2((2, 1, 1, 1), 0x1122 | Q)
    # the table's quad entries: (dp|dp) in three chunks over its bra d shell, (fp|pp) in two over its f shell
    assert router.select_algo((2, 1, 2, 1)) == 0x1001122 | router.VARIANT_QCHUNK(2, 0)
    assert router.select_algo((3, 1, 1, 1)) == 0x1001122 | router.VARIANT_QCHUNK(1, 0)


def test_h_form_variants_only_where_a_quartet_fits_one_wave():
    """JQC_VARIANT_HB (bit 29; round 6): row-lane builds with the owner reduction, lane = (bra component i, group of j components).  The
    j-components-per-lane code (bits 25-26: at most 1 / 2 / 3 / 6, rounded down to a divisor of nf_j) is widened until nf_i * nf_j / EJ
    fits the 64 lanes of a wave; the bit is dropped where that is impossible or the class is forced to the lane-per-quartet mode."""
    HB, HEJ, ORED = router.VARIANT_HB, router.VARIANT_HEJ, router.VARIANT_ORED
    assert router.hb_ej((3, 2, 2, 1), 0x521 | ORED | HB | HEJ(2)) == 3 and router.hb_ej((3, 3, 2, 1), 0x521 | ORED | HB | HEJ(2)) == 2
    assert router.hb_ej((3, 3, 2, 1), 0x521 | ORED | HB | HEJ(3)) == 5 and router.hb_ej((4, 4, 0, 0), 0x521 | ORED | HB | HEJ(1)) == 1
    assert router.lanes_per_quartet((3, 2, 2, 1), 0x521 | ORED | HB | HEJ(2)) == 20
    v = router.forced_variant((3, 3, 2, 2), 0x521 | ORED | HB)              # 100 lanes with one component per lane: widened to two
    assert v & HB and router.lanes_per_quartet((3, 3, 2, 2), v) == 50
    v = router.forced_variant((4, 4, 2, 2), 0x521 | ORED | HB)              # (gg|: 225 lanes -> five components per lane, 45 lanes
    assert v & HB and router.lanes_per_quartet((4, 4, 2, 2), v) == 45
    assert not router.forced_variant((2, 1, 1, 0), 0x1022 | HB) & HB        # lane-per-quartet request: no h form
    assert router.forced_variant((3, 2, 2, 1), 0x121 | HB) & ORED           # the form needs the owner reduction
    with open(os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")) as f:
        sch = json.load(f)
    n = 0
    for key, v in sch["fp64"].items():
        if v & HB:
            ang = tuple(int(c) for c in key.zfill(4))
            assert (v & 0xf) == L.ALGO_TILE and v & ORED and not v & 0x800 and router.lanes_per_quartet(ang, v) <= 64, (key, hex(v))
            assert router.forced_variant(ang, v) == v, (key, hex(v))
            n += 1
    assert n >= 13


def test_k_chunks_on_wave_groups_only_for_two_chunk_classes():
    """JQC_VARIANT_KW (bit 30; round 6): 512-thread row-lane builds with the owner reduction whose integral block needs exactly TWO chunks
    over the ket components k (jk_tile.hip pick_nch, mirrored by router.k_chunks); elsewhere the bit is dropped, and it excludes the
    wave-local steps."""
    KW, ORED = router.VARIANT_KW, router.VARIANT_ORED
    assert router.k_chunks((3, 2, 2, 1), 0x923 | ORED) == 2 and router.k_chunks((3, 1, 2, 1), 0x923 | ORED) == 1     # 108 / 54 integrals per lane
    assert router.k_chunks((3, 2, 2, 1), 0x10923 | ORED) == 6                                                       # capped at 32 per chunk
    assert router.k_chunks((3, 2, 2, 1), 0x123 | ORED) == 1                                                          # lane = (ci, cj): 18 per lane
    v = router.forced_variant((3, 2, 2, 1), 0xd23 | ORED | KW)
    assert v & KW and not v & 0x400
    assert not router.forced_variant((3, 1, 2, 1), 0x923 | ORED | KW) & KW          # one chunk
    assert not router.forced_variant((3, 2, 2, 1), 0x921 | ORED | KW) & KW          # 256-thread workgroups
    assert not router.forced_variant((3, 3, 3, 3), 0x923 | ORED | KW) & KW          # ten chunks
    with open(os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")) as f:
        sch = json.load(f)
    kw = [k for k, v in sch["fp64"].items() if v & KW]
    assert sorted(kw) == ["2122", "2221", "3122", "3221"], kw


def test_too_many_ket_pairs_degrade_to_fewer(tmp_path, monkeypatch):
    """(dp|ps) with 8 ket pairs per iteration needs > 160 KB of LDS: the router retries with 4 (still the HIP path)."""
    with pytest.raises(RuntimeError):
        L.gen_jk_kernel((2, 1, 1, 0), True, True, False, False, 0x3022, compile_only=True)
    router.gen_jk_kernel.cache_clear()
    h = router.gen_jk_kernel((2, 1, 1, 0), True, True, False, False, 0x3022, True)
    assert h >= 0
    router.gen_jk_kernel.cache_clear()
    # the fallback is on record next to the code objects (source tag, build key, resolved variant) ...
    tag = L.lib().jqc_source_tag().decode()
    key = router._fallback_key((2, 1, 1, 0), True, True, False, False, 0x3022)
    lines = [l.split() for l in open(router._fallback_file())]
    assert [tag, key, str(0x2022)] in lines
    # ... and the next process asks for the resolved variant at once: no compile of the variant that cannot fit
    asked = []
    real = L.gen_jk_kernel
    monkeypatch.setattr(L, "gen_jk_kernel", lambda ang, dj, dk, lr, f32, algo, co=False: asked.append(algo) or real(ang, dj, dk, lr, f32, algo, co))
    monkeypatch.setattr(router, "_fallbacks", None)            # as a fresh process: read the file again
    router.gen_jk_kernel((2, 1, 1, 0), True, True, False, False, 0x3022, True)
    assert asked == [0x2022]
    assert router.resolved_algo((2, 1, 1, 0), True, True, False, False, 0x3022) == 0x2022
    router.gen_jk_kernel.cache_clear()


def test_builds_with_scratch_reread_their_arguments():
    """Generator policy (jqc_hip.cpp:jqc_gen_jk_kernel): a class kernel whose code object spills VGPRs to scratch is the
    KARG_RELOAD build, which keeps (almost) no SGPR spilled; tools/spill_survey.py reads the same metadata."""
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    import re
    L.gen_jk_kernel((2, 1, 1, 0), True, True, False, False, 0x22, compile_only=True)      # scratch build
    L.gen_jk_kernel((1, 0, 1, 0), True, True, False, False, 0x22, compile_only=True)      # no scratch
    def meta(pattern):
        f = sorted(glob.glob(os.path.join(L.KERNEL_CACHE, pattern)), key=os.path.getmtime)[-1]
        out = subprocess.run([readelf, "--notes", f], capture_output=True, text=True).stdout
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, out).group(1))
        return g("private_segment_fixed_size"), g("sgpr_spill_count")
    scratch, sgpr_spills = meta("jk34_2110_j1k1_lr0_f64_*.hsaco")
    assert scratch > 0 and sgpr_spills <= 8
    scratch, sgpr_spills = meta("jk34_1010_j1k1_lr0_f64_*.hsaco")
    assert scratch == 0


def test_gradient_kernel_form_is_chosen_per_class():
    """jqc_gen_jk_grad_kernel: the cooperative form (T lanes per quartet, 1-D arrays in LDS) where it measured faster, the
    one-quartet-per-lane form for small ket blocks (profiles/r03_grad_forms_per_class_112atoms_tzvpp.txt); g classes by ket size."""
    if os.environ.get("JQC_GRAD_COOP") is not None:
        pytest.skip("JQC_GRAD_COOP forces one form")
    lib = L.lib()
    tag = lib.jqc_grad_source_tag().decode()
    for cls, coop in (((2, 1, 2, 1), True), ((3, 2, 2, 2), True), ((2, 1, 1, 0), False), ((1, 1, 0, 0), False),
                      ((4, 2, 2, 2), True), ((4, 1, 1, 0), False)):
        assert L.check(lib.jqc_gen_jk_grad_kernel(*cls, 0, 1)) >= 0
        name = "jkgrad_%d%d%d%d_lr0%s_%s.hsaco" % (*cls, "_coop" if coop else "", tag)
        assert os.path.exists(os.path.join(L.KERNEL_CACHE, name)), name


def test_risky_builds_are_gated_twice_or_quarantined():
    """Quarantine rule of the verified manifest (tools/make_manifest.py, tools/risky_builds_gate.py): a build with
    more than 256 registers per lane AND SGPRs spilled to VGPR lanes -- the family of the wrong-result builds of rounds 1-3 -- may
    only be listed as verified with a record of the forced-ket-chunk gate run TWICE on an MI355X for the same source tag: run-to-run
    agreement <= 1e-12, agreement with the plain reference variant <= 1e-10 (2e-4 for an FP32 build)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import risky_builds_gate as RG
    assert RG.is_risky(290, 65) and RG.is_risky(512, 1)
    assert not RG.is_risky(256, 70) and not RG.is_risky(400, 0)
    assert RG.parse_key("jk789793_2121_j1k1_lr0_f64_t8.4.4.2.1") == (789793, (2, 1, 2, 1), 1, 1, 0, False)
    data = os.path.join(ROOT, "joltqc_amd", "data")
    man = json.load(open(os.path.join(data, "verified_kernels.json")))
    if "risky" not in man:
        pytest.skip("manifest written before the quarantine rule")
    gate = json.load(open(os.path.join(data, "risky_builds_gate.json")))
    assert gate["src_tag"] == man["src_tag"]
    listed = set(man["keys"])
    assert not listed & set(man["quarantined"])
    for key in man["risky"]:
        if key in listed:
            r = gate["results"][key]
            loose = key.split("_")[4] == "f32" or r.get("fp32_phase")
            assert r["ok"] and r["run_to_run"] <= 1e-12 and r["vs_ref"] <= (2e-4 if loose else 1e-10), (key, r)
        else:
            assert key in man["quarantined"], key


def test_increment_policy_schedules_full_rebuilds():
    """rks.IncrementPolicy: a full build on the first call, whenever the increment has shrunk by 1e3 since the last full
    build, and after 12 increments in a row -- the schedule that keeps dropped sub-cutoff terms and FP32 rounding from piling up over
    an SCF run (3-4 full builds per run instead of one)."""
    from joltqc_amd.pyscf.rks import IncrementPolicy
    p = IncrementPolicy()
    # |dD|max of the 112-atom B3LYP run (profiles/r04_config3_scf_noise.txt)
    seq = [2.2, 7.1e-1, 6.0e-1, 4.9e-1, 1.2e-1, 3.6e-2, 1.5e-2, 5.4e-3, 1.6e-3, 4.7e-4, 1.2e-4, 6.3e-5, 2.4e-5, 6.4e-6, 3.3e-6, 9.4e-7,
           5.9e-7, 3.1e-7, 2.3e-7, 4.3e-8, 1.8e-8]
    full = [i for i, d in enumerate(seq) if p.full_build(d)]
    assert full == [0, 8, 15]
    # a stagnating run is rebuilt every 12 increments
    p.reset()
    full = [i for i in range(30) if p.full_build(1e-3)]
    assert full == [0, 13, 26]
    # a converged density evaluated again (increment exactly zero) is a full build
    assert p.full_build(0.0)
    p.reset()
    assert p.full_build(5.0) and not p.full_build(4.0)
    # a restart on the same closures (second kernel(), new dm0): the increment GROWS against the reference of a converged run
    p.reset()
    assert [p.full_build(d) for d in (1.0, 1e-2, 1e-4, 2e-7, 1.5e-7, 0.8, 0.3)] == [True, False, True, False, False, True, False]

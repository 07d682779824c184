"""Algorithm router for the J/K kernels (role of ``gen_jk_kernel``, /root/reference/jqc/backend/jk.py:57-115).

The reference looks a class key ``1000 li + 100 lj + 10 lk + ll`` up in a per-GPU tuning JSON and
chooses between its 1q1t and 1qnt kernels.  Here the table is for gfx950: ``ALGO_1Q1T`` (one quartet
per lane) or ``ALGO_TILE`` (lane group per quartet with LDS Fock tiles); see
``joltqc_amd/data/gfx950_scheme.json`` when present, otherwise the rule below.
"""
import json
import os
from functools import lru_cache

from . import lib as _lib

# (JQC_SCHEME_JSON: another table for A/B runs, e.g. the previous round's)
_SCHEME = os.environ.get("JQC_SCHEME_JSON") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "gfx950_scheme.json")


@lru_cache(maxsize=1)
def _table():
    if os.path.exists(_SCHEME):
        with open(_SCHEME) as f:
            return json.load(f)
    return {}


def fp32_pays(ang):
    """Mixed precision: does the fp32 kernel of class ``ang`` beat the fp64 one per quartet on this chip?  Measured table
    (gfx950_scheme.json "fp32_pays", tools/autotune.py with JQC_TUNE_FP32=1); JQC_FP32_WINDOW=1/0 forces the answer."""
    force = os.environ.get("JQC_FP32_WINDOW")
    if force is not None:
        return force == "1"
    return bool(_table().get("fp32_pays", {}).get(class_key(ang), False))    # unmeasured classes (g): one fp64 launch


def fp32_tile_split(ang):
    """Mixed precision of class ``ang`` by TILE PAIRS: do the tile pairs whose bound is at or below cutoff_fp64 go to the class's
    FP32 kernel (pyscf/jk.py build_tile_plan; each tile pair is staged by ONE of the two launches)?  Measured table
    (gfx950_scheme.json "fp32_tile_split", tools/mixed_bench.py per class); JQC_FP32_TILE_SPLIT=1/0 forces the answer."""
    force = os.environ.get("JQC_FP32_TILE_SPLIT")
    if force is not None:
        return force == "1"
    return bool(_table().get("fp32_tile_split", {}).get(class_key(ang), False))


def class_cost_table():
    """{class key: measured ns per dispatched quartet} (gfx950_scheme.json, tools/class_profile.py)."""
    return _table().get("ns_per_quartet", {})


TILE1Q_MAX_NINT = int(os.environ.get("JQC_TILE1Q_MAX", "108"))   # above this the lane-per-quartet body spills heavily
QUAD_FORCE_MAX = int(os.environ.get("JQC_QUAD_MAX", "330"))      # largest integral block the quad form (JQC_VARIANT_QUAD) is tried on (with chunks)
TILE1Q_FORCE_MAX = 200    # largest integral block the lane-per-quartet mode is ever tried on (512 VGPRs at 1 wave/SIMD)


def nint(ang):
    n = 1
    for l in ang:
        n *= (l + 1) * (l + 2) // 2
    return n


def class_key(ang):
    li, lj, lk, ll = ang
    return str(1000 * li + 100 * lj + 10 * lk + ll)


VARIANT_ORED, VARIANT_PAROOT, VARIANT_NDM2 = 1 << 18, 1 << 19, 1 << 20     # include/jqc_hip.h
VARIANT_RSPLIT = lambda code: code << 22      # row-lane mode: Rys roots in code + 1 groups through phase A / B (half the TRR array)
VARIANT_QUAD = 1 << 24         # lane-per-quartet mode with one quartet per DPP quad of lanes (classes with a p shell, <= 4 Rys roots)
VARIANT_QCHUNK = lambda code, index: (code << 25) | (index << 27)     # quad builds: 2 / 3 / 5 chunks (code 1 / 2 / 3) over the components of shell `index`
QCHUNKS = (1, 2, 3, 5)
VARIANT_MIXED = 1 << 21        # both precision windows in one launch of an FP64 lane-per-quartet build (FP32 phase packed)
VARIANT_HB = 1 << 29           # row-lane mode, h form: bra horizontal recurrence in phase A, lane = (bra component i, group of j components)
VARIANT_HEJ = lambda code: code << 25      # h form: j components per lane capped at HEJ_CAPS[code] (bits shared with the quad chunk code)
HEJ_CAPS = (1, 2, 3, 6)
VARIANT_KW = 1 << 30           # 512-thread row-lane builds: the two k chunks of a class on different waves, sharing the recurrence arrays
ECAPS = (64, 32, 16, 48)       # integrals per lane and chunk by JQC_VARIANT_ECAP code (jqc_hip.cpp)


def k_chunks(ang, v):
    """Chunks over the ket components k a row-lane build of variant ``v`` walks (jk_tile.hip pick_nch)."""
    nf = lambda l: (l + 1) * (l + 2) // 2
    ej = hb_ej(ang, v) if v & VARIANT_HB else nf(ang[1]) if v & 0x800 else 1
    cap = ECAPS[(v >> 16) & 3]
    for n in range(1, nf(ang[2]) + 1):
        if nf(ang[2]) % n == 0 and (nf(ang[2]) // n) * nf(ang[3]) * ej <= cap:
            return n
    return nf(ang[2])


def hb_ej(ang, v):
    """j components per lane of the h-form variant ``v``: the largest divisor of nf_j that is <= the cap (jk_tile.hip pick_hej)."""
    nfj = (ang[1] + 1) * (ang[1] + 2) // 2
    cap = HEJ_CAPS[(v >> 25) & 3]
    return max(n for n in range(1, nfj + 1) if nfj % n == 0 and n <= cap)


def supports_mixed(ang, v):
    """May variant ``v`` of class ``ang`` be built with the fused FP64 + packed-FP32 compute phases (JQC_VARIANT_MIXED)?"""
    return (v & 0xf) == _lib.ALGO_TILE1Q and not (v & VARIANT_NDM2) and not (v & VARIANT_QUAD)


def mixed_variant(ang, v):
    """Variant code of the fused build that goes with the fp64 variant ``v`` of class ``ang``.  JQC_MIXED_NKS_SHIFT=n (tuning):
    2^n times the ket tile pairs per iteration of the fp64 variant (the packed phase needs twice the survivors to fill a wave)."""
    v |= VARIANT_MIXED
    up = int(os.environ.get("JQC_MIXED_NKS_SHIFT", "0"))
    if up:
        nks = min(3, ((v >> 12) & 3) + up)
        v = (v & ~0x3000) | (nks << 12)
    return v


def mixed_fused(ang, v):
    """Mixed precision of class ``ang``: ONE launch of the MIXED build of variant ``v`` (FP64 phase + packed-FP32 phase behind one
    staging of every tile pair) instead of the fp64 kernel alone?  Measured table (gfx950_scheme.json "mixed_fused",
    tools/mixed_class_bench.py); JQC_MIXED_FUSED=1/0 forces the answer for every class that supports the build."""
    if not supports_mixed(ang, v):
        return False
    force = os.environ.get("JQC_MIXED_FUSED")
    if force is not None:
        return force == "1"
    return bool(_table().get("mixed_fused", {}).get(class_key(ang), False))


def lanes_per_quartet(ang, v):
    """Row lanes T of one quartet under variant ``v`` (j components in registers: nf_i, otherwise nf_i * nf_j)."""
    nf = lambda l: (l + 1) * (l + 2) // 2
    if v & VARIANT_HB:
        return nf(ang[0]) * (nf(ang[1]) // hb_ej(ang, v))
    return nf(ang[0]) if (v & 0x800) else nf(ang[0]) * nf(ang[1])


def supports_ndm2(ang, v):
    """May variant ``v`` of class ``ang`` be built for two density matrices per integral evaluation (NDM = 2)?
    Lane-per-quartet builds (plain contraction order) and row-lane builds with the owner reduction (a quartet inside one wave)."""
    a = v & 0xf
    if a == _lib.ALGO_TILE1Q:
        return not (v & 0xc000)
    if a == _lib.ALGO_TILE:
        return bool(v & VARIANT_ORED) and lanes_per_quartet(ang, v) <= 64
    return False


def forced_variant(ang, v):
    """Variant code ``v`` adjusted to what class ``ang`` supports: lane-per-quartet only where the integral block fits,
    the wave-local variant only where a quartet fits one wave."""
    # (the quad form holds a third of the block per lane: up to 3 x 90 integrals)
    if (v & 0xf) == _lib.ALGO_TILE1Q and nint(ang) > (QUAD_FORCE_MAX if (v & VARIANT_QUAD) and 1 in ang else TILE1Q_FORCE_MAX):
        return _lib.ALGO_TILE
    if (v & 0xf) == _lib.ALGO_TILE1Q:
        v &= ~(0x30000 | VARIANT_ORED | VARIANT_PAROOT | VARIANT_RSPLIT(3))   # integral chunks, owner reduction, per-root phase A,
                                                                              # root groups: row-lane mode only
    nf = lambda l: (l + 1) * (l + 2) // 2
    if (v & VARIANT_QUAD) and ((v & 0xf) != _lib.ALGO_TILE1Q or 1 not in ang or sum(ang) // 2 + 1 > 4):
        v &= ~VARIANT_QUAD                            # quad form: lane-per-quartet builds of classes with a p shell, <= 4 roots
    if v & VARIANT_QUAD:
        v &= ~0xc000                                  # (strided queue / row-ordered contraction belong to the one-lane form)
        code, qy = (v >> 25) & 3, (v >> 27) & 3
        xs = 3 if ang[3] == 1 else 2 if ang[2] == 1 else 1 if ang[1] == 1 else 0          # the split p shell (jk_tile.hip XS)
        if code and (qy == xs or nf(ang[qy]) % QCHUNKS[code]):
            v &= ~(0xf << 25)                         # chunks must divide the component count of another shell
    elif v & VARIANT_HB:
        # h form: row-lane builds with the owner reduction and a quartet inside one wave; where nf_i * nf_j / EJ exceeds 64 lanes the
        # j-group is widened, and the form is dropped where even that does not fit (or the class is not a row-lane class here)
        v &= ~(0x3 << 27)
        if (v & 0xf) not in (_lib.ALGO_TILE, _lib.ALGO_TILE512):
            v &= ~(VARIANT_HB | (0xf << 25))
        else:
            v = (v | VARIANT_ORED) & ~0x800
            while lanes_per_quartet(ang, v) > 64 and ((v >> 25) & 3) < 3:
                v += 1 << 25
            if lanes_per_quartet(ang, v) > 64:
                v &= ~(VARIANT_HB | (0xf << 25))
    else:
        v &= ~(0xf << 25)
    if (v & 0xf) != _lib.ALGO_TILE1Q:
        v &= ~(0xf000 | VARIANT_MIXED)                # several ket pairs per iteration, strided queue, row-ordered contraction,
                                                      # fused precision phases: lane-per-quartet mode only
    nf = lambda l: (l + 1) * (l + 2) // 2
    if v & VARIANT_KW:
        # k chunks on wave groups: 512-thread builds with the owner reduction (a quartet inside one wave), workgroup-wide steps, exactly two chunks
        if (v & 0xf) != _lib.ALGO_TILE512 or not (v & VARIANT_ORED) or lanes_per_quartet(ang, v) > 64 or k_chunks(ang, v) != 2:
            v &= ~VARIANT_KW
        else:
            v &= ~0x400
    if (v & 0x400) and ((v & 0xf) == _lib.ALGO_TILE1Q or (lanes_per_quartet(ang, v) if v & VARIANT_HB else nf(ang[0]) * nf(ang[1])) > 64):
        v &= ~0x400
    # a fused mixed-precision build reads its two cutoffs differently from every other build (FP64 above cut_hi, FP32 below): it is
    # only ever reached through mixed_variant() on the fused launch path of pyscf/jk.py, never selected as a class's kernel
    return v & ~VARIANT_MIXED


def select_algo(ang, fp32=False, small=False):
    """Algorithm + tuning variant of class ``ang`` (low 4 bits: lib.ALGO_*, bits 4-7: waves per SIMD, bit 8: Rys table
    through L2, bit 9: single TRR buffer, ...; include/jqc_hip.h JQC_VARIANT_*).  ``small``: the launch has fewer
    workgroups than the chunking target (a benzene-size molecule): the "fp64_small" table, tuned on benzene, overrides
    the main one, which is tuned on a 112-atom molecule (one-workgroup-per-CU builds lose when the grid cannot fill
    the chip anyway)."""
    t = _table().get("fp32" if fp32 else "fp64", {})
    if small and not fp32:
        t = {**t, **_table().get("fp64_small", {})}
    forced = os.environ.get("JQC_JK_ALGO")
    if forced:
        f = forced.lower()
        if f.startswith("v"):                         # raw variant code, e.g. v289
            return forced_variant(ang, int(f[1:]))
        if f in ("0", "1q1t"):
            return _lib.ALGO_1Q1T
        if f in ("2", "tile1q"):
            return _lib.ALGO_TILE1Q if nint(ang) <= TILE1Q_MAX_NINT else _lib.ALGO_TILE
        if f in ("3", "tile512"):
            return _lib.ALGO_TILE512
        return _lib.ALGO_TILE
    v = t.get(class_key(ang))
    if v is None:
        return _lib.ALGO_TILE
    return int(v)


def _lds_overflow(err):
    """hiprtc's message when a build does not fit the 160 KiB of LDS of a gfx950 CU."""
    return "local memory" in str(err) and "exceeds limit" in str(err)


def kernel_key(ang, do_j, do_k, rys_lr, fp32, algo):
    """Name of one class-kernel BUILD: the string the library keys its code objects on (tile widths included: other widths are
    other code objects with other register allocations), minus the source tag."""
    from ..constants import TILE_WIDTHS
    return "jk%d_%d%d%d%d_j%dk%d_lr%d_%s_t%s" % (int(algo), *ang, int(do_j), int(do_k), int(rys_lr), "f32" if fp32 else "f64",
                                                ".".join(str(int(w)) for w in TILE_WIDTHS))


_MANIFEST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "verified_kernels.json")


@lru_cache(maxsize=1)
def _manifest():
    """Builds that passed the gates of the scheme table (tests/test_jk_gpu.py all-class / all-variant tests,
    tests/test_jk_fullsize_gpu.py) for one version of the kernel sources and one compiler: written by
    tools/make_manifest.py after a green GPU run.  A build that is not listed -- another variant forced through
    JQC_JK_ALGO, JQC_EXTRA_DEFS, edited sources, another hiprtc -- is cross-checked against the independent
    one-quartet-per-lane kernel on its first use (joltqc_amd/pyscf/jk.py)."""
    if os.path.exists(_MANIFEST):
        with open(_MANIFEST) as f:
            return json.load(f)
    return {}


def is_verified(ang, do_j, do_k, rys_lr, fp32, algo):
    if os.environ.get("JQC_TRUST_KERNELS") == "1":          # kernel development / tuning runs
        return True
    m = _manifest()
    if m.get("src_tag") != _lib.lib().jqc_source_tag().decode():
        return False
    return kernel_key(ang, do_j, do_k, rys_lr, fp32, algo) in _manifest_keys()


@lru_cache(maxsize=1)
def _manifest_keys():
    return frozenset(_manifest().get("keys", ()))


_resolved = {}


def resolved_algo(ang, do_j, do_k, rys_lr, fp32, algo):
    """The variant ``gen_jk_kernel`` actually built for a request (differs when a several-ket-pairs build did not fit LDS)."""
    return _resolved.get((tuple(ang), bool(do_j), bool(do_k), bool(rys_lr), bool(fp32), int(algo)), int(algo))


_fallbacks = None      # {build key: variant the build resolved to}: the LDS-overflow fallbacks already found for these sources


def _fallback_file():
    return os.path.join(_lib.KERNEL_CACHE, "lds_fallbacks.txt")


def _known_fallbacks():
    """Fallbacks recorded next to the code objects (one line per build: source tag, key, resolved variant).  A variant that does
    not fit LDS leaves no code object behind, so without this record every process would compile it again just to see it fail
    (seconds per build, on the GPU box inside the first J/K call); the ahead-of-time build writes the record."""
    global _fallbacks
    if _fallbacks is None:
        _fallbacks = {}
        tag = _lib.lib().jqc_source_tag().decode()
        try:
            with open(_fallback_file()) as f:
                for line in f:
                    w = line.split()
                    if len(w) == 3 and w[0] == tag and w[2].isdigit():        # (a torn or foreign line is ignored)
                        _fallbacks[w[1]] = int(w[2])
        except OSError:
            pass
    return _fallbacks


def _fallback_key(ang, do_j, do_k, rys_lr, fp32, want):
    from ..constants import TILE_WIDTHS          # (other tile widths = other LDS footprints: part of the key, as in kernel_key)
    return "%d%d%d%d:%d%d%d%d:%d:t%s" % (*ang, bool(do_j), bool(do_k), bool(rys_lr), bool(fp32), want,
                                          ".".join(str(int(w)) for w in TILE_WIDTHS))


@lru_cache(maxsize=None)
def gen_jk_kernel(ang, do_j=True, do_k=True, rys_lr=False, fp32=False, algo=None, compile_only=False):
    ang = tuple(int(x) for x in ang)
    if algo is None:
        algo = select_algo(ang, fp32)
    want = int(algo)
    fkey = _fallback_key(ang, do_j, do_k, rys_lr, fp32, want)
    algo = _known_fallbacks().get(fkey, want)
    while True:
        try:
            h = _lib.gen_jk_kernel(ang, do_j, do_k, rys_lr, fp32, algo, compile_only)
            _resolved[(ang, bool(do_j), bool(do_k), bool(rys_lr), bool(fp32), want)] = int(algo)
            if int(algo) != want and _known_fallbacks().get(fkey) != int(algo):
                _fallbacks[fkey] = int(algo)
                try:        # (one short O_APPEND write per entry: safe from the parallel workers of the ahead-of-time build)
                    fd = os.open(_fallback_file(), os.O_WRONLY | os.O_APPEND | os.O_CREAT, 0o644)
                    os.write(fd, ("%s %s %d\n" % (_lib.lib().jqc_source_tag().decode(), fkey, int(algo))).encode())
                    os.close(fd)
                except OSError:
                    pass
            return h
        except RuntimeError as e:
            # the ONE expected failure: several ket pairs per iteration (or the j-in-registers form) do not fit LDS for this
            # build of the class (e.g. its long-range form with the larger Rys table) -> same kernel with fewer ket
            # pairs per iteration, then the plain row-lane kernel.  Anything else (compiler error, missing source) is raised.
            if not _lds_overflow(e):
                raise
            if algo & VARIANT_MIXED:
                algo &= ~VARIANT_MIXED                # the FP32 copy of the Rys table does not fit: the caller sees the resolved
                                                      # variant and keeps the class on its fp64 launch (pyscf/jk.py)
            elif (algo & VARIANT_NDM2) and (algo & 0xf) != _lib.ALGO_TILE1Q:
                algo &= ~VARIANT_NDM2                 # the tiles of two density matrices do not fit: one matrix per pass
            elif algo & 0x3000:
                nks = (algo >> 12) & 3
                algo = (algo & ~0x3000) | ((nks - 1) << 12)
            elif algo & VARIANT_NDM2:
                # two matrices do not fit even with one ket pair per iteration: one matrix per pass, with the ket pairs per
                # iteration the variant was tuned for (not the 1 the steps above ended on)
                algo = (algo & ~VARIANT_NDM2 & ~0x3000) | (want & 0x3000)
            elif (algo & 0xf) != _lib.ALGO_TILE or (algo & 0xc00) or (algo & VARIANT_HB) or (algo & VARIANT_KW):
                algo = _lib.ALGO_TILE | (algo & 0x1f0)
            else:
                raise

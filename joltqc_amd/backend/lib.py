"""ctypes binding of libjqc_hip.so (C ABI: include/jqc_hip.h).

The product path has no CPU fallback: if the shared library (or a GPU) is missing every entry point
raises.  Device buffers are torch tensors; only their raw pointers cross the boundary.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
# JQC_KERNEL_SRC / JQC_KERNEL_CACHE: kernel development against a scratch copy of the sources (own code-object cache), so
# that the verified AOT set of the shipped sources stays usable until a change is adopted
KERNEL_SRC = os.environ.get("JQC_KERNEL_SRC", os.path.join(CSRC, "kernels"))
KERNEL_CACHE = os.environ.get("JQC_KERNEL_CACHE", os.path.join(CSRC, "kcache" if "JQC_KERNEL_SRC" not in os.environ else "kcache_dev"))
LIB_PATH = os.environ.get("JQC_LIB_PATH", os.path.join(CSRC, "libjqc_hip.so"))      # (override: a development build)

ALGO_1Q1T = 0
ALGO_TILE = 1
ALGO_TILE1Q = 2
ALGO_TILE512 = 3

_lib = None
_lock = threading.Lock()
_rys_uploaded = False


def build_library(force=False):
    """hipcc-compile the C-ABI library in-tree (cross-compiles for gfx950 without a GPU)."""
    import glob
    src = os.path.join(CSRC, "jqc_hip.cpp")
    hdr = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "jqc_hip.h")
    deps = [src, hdr] + glob.glob(os.path.join(CSRC, "*.inc"))          # (dft_kernels.inc, ecp_kernels.inc, ecp_kernel_body.inc)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(f) for f in deps):
        return LIB_PATH
    cmd = ["hipcc", "-O2", "-shared", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", src,
           "-o", LIB_PATH, "-lhiprtc"]
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the J/K and grid kernels)")
        import torch  # noqa: F401  -- load torch's HIP runtime first so both share ONE libamdhip64 in the process
        L = ctypes.CDLL(LIB_PATH)
        c = ctypes
        vp, i32, i64, f32, f64 = c.c_void_p, c.c_int, c.c_int64, c.c_float, c.c_double
        L.jqc_last_error.restype = c.c_char_p
        L.jqc_version.restype = c.c_char_p
        L.jqc_source_tag.restype = c.c_char_p
        L.jqc_set_kernel_dirs.argtypes = [c.c_char_p, c.c_char_p]
        L.jqc_set_rys_tables.argtypes = [vp, c.c_size_t]
        L.jqc_gen_jk_kernel.argtypes = [i32] * 10
        L.jqc_jk_launch.argtypes = [i32, i32, vp, vp, vp, vp, f64, vp, vp, i64, i32, i32, vp]
        L.jqc_jk_tile_launch.argtypes = [i32, i32, vp, vp, vp, vp, f64, vp, i32, i32, vp, vp, vp, vp, i32, f32, f32, f32,
                                         i32, vp, vp, vp, vp, vp, vp, vp]
        L.jqc_pair_table.argtypes = [vp, vp, vp, vp, i32, vp, vp]
        L.jqc_screen_jk_tasks.argtypes = [vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, f32, f32, vp, vp, vp, vp]
        L.jqc_shell_block_max.argtypes = [vp, i32, i32, vp, i32, vp, vp]
        L.jqc_schwarz.argtypes = [i32, i32, vp, vp, i32, f64, vp, vp]
        L.jqc_int1e.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]
        L.jqc_ecp_scalar.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp]
        L.jqc_gen_pair_vj_kernel.argtypes = [i32] * 6
        L.jqc_pair_ntrip.argtypes = [i32, i32]
        L.jqc_gen_jk_grad_kernel.argtypes = [i32] * 6
        L.jqc_jk_grad_launch.argtypes = [i32, i32, vp, vp, i32, vp, vp, i32, i32, f64, f64, f64, vp, vp, i64, i32, vp]
        L.jqc_grad_source_tag.restype = c.c_char_p
        L.jqc_pair_source_tag.restype = c.c_char_p
        L.jqc_pair_ket_density.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp, vp]
        L.jqc_pair_vj_launch.argtypes = [i32, i32, vp, vp, vp, f64, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, i32, i32,
                                         i32, vp, vp]
        L.jqc_dft_ao_screen.argtypes = [vp, i32, vp, vp, i32, f32, vp, vp, vp, vp, vp, vp]
        L.jqc_dft_eval_ao.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp]
        L.jqc_dft_rho.argtypes = [i32, i32, i32, vp, vp, i64, vp, vp, vp, i32, i32, vp, vp, f32, f32, vp, vp]
        L.jqc_dft_vxc.argtypes = [i32, i32, i32, vp, vp, i64, vp, vp, vp, i32, i32, vp, vp, f32, f32, vp, vp]
        L.jqc_dft_xcgrad_ao.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp]
        L.jqc_dft_xcgrad.argtypes = [i32, i32, vp, vp, i64, vp, vp, vp, i32, vp, vp, f32, vp, i32, vp]
        L.jqc_vv10.argtypes = [vp] * 10 + [i32, i32, i32, vp]
        from ..constants import TILE_WIDTHS
        L.jqc_set_tile_widths.argtypes = [c.POINTER(c.c_int)]
        if L.jqc_set_tile_widths((c.c_int * 5)(*TILE_WIDTHS)) < 0:
            raise RuntimeError("libjqc_hip: " + L.jqc_last_error().decode())
        os.makedirs(KERNEL_CACHE, exist_ok=True)
        L.jqc_set_kernel_dirs(KERNEL_SRC.encode(), KERNEL_CACHE.encode())
        _lib = L
        return _lib


def purge_stale_cache():
    """Remove cached code objects that were built from other versions of the kernel sources."""
    tag = lib().jqc_source_tag().decode()
    gtag = lib().jqc_grad_source_tag().decode()
    ptag = lib().jqc_pair_source_tag().decode()
    n = 0
    for f in os.listdir(KERNEL_CACHE):
        stem = f[:-len(".scratch")] if f.endswith(".hsaco.scratch") else f       # (markers of pair kernels that would spill)
        if stem.endswith(".hsaco") and not any(stem.endswith("_" + t + ".hsaco") for t in (tag, gtag, ptag)):
            os.remove(os.path.join(KERNEL_CACHE, f))
            n += 1
    # the record of LDS-overflow fallbacks next to the code objects (backend/jk.py) is append-only: drop the lines of other source tags
    fb = os.path.join(KERNEL_CACHE, "lds_fallbacks.txt")
    try:
        with open(fb) as f:
            lines = f.readlines()
        keep = [ln for ln in lines if ln.split()[:1] == [tag]]
        if len(keep) != len(lines):
            with open(fb, "w") as f:
                f.writelines(keep)
    except OSError:
        pass
    return n


def check(rc):
    if rc < 0:
        raise RuntimeError("libjqc_hip: " + lib().jqc_last_error().decode())
    return rc


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("joltqc_amd needs an AMD GPU (gfx950); no CPU fallback exists for this path")
    return torch.device("cuda", torch.cuda.current_device())


def ensure_rys():
    """Upload the Rys tables once per process (after the device is chosen)."""
    global _rys_uploaded
    if _rys_uploaded:
        return
    require_gpu()
    from .rys import pack_tables
    blob = np.ascontiguousarray(pack_tables())
    check(lib().jqc_set_rys_tables(blob.ctypes.data, blob.size))
    _rys_uploaded = True


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def gen_jk_kernel(ang, do_j=True, do_k=True, rys_lr=False, fp32=False, algo=ALGO_1Q1T, compile_only=False):
    li, lj, lk, ll = (int(x) for x in ang)
    return check(lib().jqc_gen_jk_kernel(li, lj, lk, ll, int(do_j), int(do_k), int(rys_lr), int(fp32), int(algo),
                                         int(compile_only)))

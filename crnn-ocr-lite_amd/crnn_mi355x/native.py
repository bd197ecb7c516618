"""ctypes binding of libcrnn_mi355x.so (the C ABI declared in include/crnn_mi355x.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, importing
`lib()` raises.  Build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)                      # crnn-ocr-lite_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
HEADER = os.path.join(REPO_ROOT, "include", "crnn_mi355x.h")
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.path.join(PKG_ROOT, "libcrnn_mi355x.so")
HOOKS_HEADER = os.path.join(REPO_ROOT, "include", "crnn_testhooks.h")
HOOKS_PATH = os.path.join(PKG_ROOT, "libcrnn_testhooks.so")     # measurement / test hooks: never loaded by the product path
SOURCES = ["gemm.hip", "gemm_nt.hip", "gemm_wres.hip", "gemm_wres3.hip", "gemm_wgrad.hip", "gemm_wgrad3.hip", "gemm_pres.hip", "conv.hip", "conv_bwd_fused.hip", "dwconv_stream.hip", "dwconv_bwd_stream.hip", "stn.hip", "rnn.hip", "rnn_persist.hip", "gru_persist.hip", "ctc.hip", "dense.hip", "beam.hip", "optim.hip", "model.hip"]


class crnn_config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("batch", "imgh", "imgw", "num_classes", "max_len", "tds", "units", "gru", "stn", "dropout", "mfma_bf16", "flags")]


FLAG_RNN_STEP_KERNELS = 1      # CRNN_FLAG_RNN_STEP_KERNELS
FLAG_GEMM_TILE_KERNELS = 2     # CRNN_FLAG_GEMM_TILE_KERNELS
FLAG_NO_DW_BN_FUSION = 8       # CRNN_FLAG_NO_DW_BN_FUSION
FLAG_NO_DW_BWD_FUSION = 16     # CRNN_FLAG_NO_DW_BWD_FUSION
FLAG_DW_TILE_KERNEL = 32       # CRNN_FLAG_DW_TILE_KERNEL
FLAG_RNN_LINEAR_CLUSTERS = 64  # CRNN_FLAG_RNN_LINEAR_CLUSTERS
FLAG_NO_BN_STATS_FUSION = 128  # CRNN_FLAG_NO_BN_STATS_FUSION
FLAG_F32_MFMA_GEMMS = 256      # CRNN_FLAG_F32_MFMA_GEMMS
FLAG_DEFERRED_SUMS = 512       # CRNN_FLAG_DEFERRED_SUMS
FLAG_BN2_DW_FUSION = 1024      # CRNN_FLAG_BN2_DW_FUSION (opt-in)
FLAG_BN2_STATS_FUSION = 2048   # CRNN_FLAG_BN2_STATS_FUSION (opt-in)
FLAG_BLOCK1_KERNELS = 16384    # CRNN_FLAG_BLOCK1_KERNELS
FLAG_LOC_NET_KERNELS = 8192    # CRNN_FLAG_LOC_NET_KERNELS
FLAG_THREE_PLANE_BACKWARD = 65536   # CRNN_FLAG_THREE_PLANE_BACKWARD (parity mode: strict backward GEMMs)
FLAG_TWO_PLANE_FORWARD = 131072     # CRNN_FLAG_TWO_PLANE_FORWARD (parity mode, opt-in)
FLAG_NO_GRADIENT_PLANES = 262144   # CRNN_FLAG_NO_GRADIENT_PLANES (parity mode: BatchNorm-2's input gradients stay fp32 tensors)
FLAG_NO_POOL_ARGMAX_Q = 524288     # CRNN_FLAG_NO_POOL_ARGMAX_Q (pooled blocks: the backward's statistics pass scans the windows again)
FLAG_WEIGHT_PLANES = 32768     # CRNN_FLAG_WEIGHT_PLANES (parity mode, opt-in)
FLAG_NO_BN2_DW_FUSION = 4096   # CRNN_FLAG_NO_BN2_DW_FUSION (fp32 tensors: the two fusions above are the default)
RNN_XCD_LOCAL = 0x100          # CRNN_RNN_XCD_LOCAL (or-ed into the uw argument of crnn_lstm_*_persist)


_CTYPE = [("crnn_stream_t", ctypes.c_void_p), ("size_t", ctypes.c_size_t), ("uint64_t", ctypes.c_uint64),
          ("uint32_t", ctypes.c_uint32), ("long", ctypes.c_long), ("int", ctypes.c_int), ("float", ctypes.c_float),
          ("double", ctypes.c_double)]


def _ctype_of(decl):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p
    for key, ct in _CTYPE:
        if re.search(r"\b%s\b" % key, decl):
            return ct
    raise ValueError("cannot map C type: %r" % decl)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|long|size_t)\s+(crnn_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = [] if args.strip() in ("", "void") else [_ctype_of(a) for a in args.split(",")]
        out[name] = ({"int": ctypes.c_int, "long": ctypes.c_long, "size_t": ctypes.c_size_t}[ret], argtypes)
    return out


def build(verbose=False):
    """Compile every HIP source for gfx950 into crnn-ocr-lite_amd/libcrnn_mi355x.so (in-tree)."""
    objs, jobs = [], []
    inc = os.path.join(REPO_ROOT, "include")
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        deps = [src, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "rnn_cell.h"), os.path.join(CSRC, "rnn_exchange.h"), os.path.join(CSRC, "gemm_bf16.inc"), HEADER]
        if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            jobs.append(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", inc, "-c", src, "-o", obj])
        objs.append(obj)
    if jobs:   # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH])
    hsrc = os.path.join(CSRC, "testhooks.hip")
    hdeps = [hsrc, os.path.join(CSRC, "common.h"), HEADER, HOOKS_HEADER]
    if not os.path.exists(HOOKS_PATH) or any(os.path.getmtime(d) > os.path.getmtime(HOOKS_PATH) for d in hdeps):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", inc, hsrc, "-o", HOOKS_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


_LIB = None


def lib():
    """Load the library and attach argtypes/restypes from the header.  Raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcrnn_mi355x.so not built (%s); run `python __graft_entry__.py` "
                               "-- there is no CPU fallback for the CRNN hot path" % LIB_PATH)
        # The library must bind to the SAME HIP runtime as PyTorch (streams and device pointers cross the boundary):
        # torch ships its own libamdhip64 under torch/lib with the same SONAME as /opt/rocm's, and whichever is loaded
        # first serves both -- so torch goes first.  Loading this .so before torch would pull in a second runtime and
        # every launch on a torch stream would fail with hipErrorNoDevice.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in parse_header().items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = ret, args
        _LIB = L
    return _LIB


_HOOKS = None


def hooks():
    """libcrnn_testhooks.so (include/crnn_testhooks.h): crnn_debug_copy / crnn_debug_occupy for bench.py's copy reference, scripts/ and
    tests/.  Separate from the product library on purpose; nothing under crnn_mi355x/ calls this."""
    global _HOOKS
    if _HOOKS is None:
        if not os.path.exists(HOOKS_PATH):
            raise RuntimeError("libcrnn_testhooks.so not built (%s); run `python __graft_entry__.py`" % HOOKS_PATH)
        import torch  # noqa: F401  (same HIP runtime as PyTorch, see lib())
        L = ctypes.CDLL(HOOKS_PATH)
        for name, (ret, args) in parse_header(HOOKS_HEADER).items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = ret, args
        _HOOKS = L
    return _HOOKS


class CrnnError(RuntimeError):
    pass


def check(code, what=""):
    if code != 0:
        kind = {-2: "bad argument", -3: "unsupported configuration"}.get(code, "hipError_t" if code > 0 else "error")
        raise CrnnError("libcrnn_mi355x %s failed: %d (%s)" % (what, code, kind))

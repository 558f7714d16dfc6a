"""ctypes binding of libcpg_hip.so, generated from include/cpg_api.h so header and binding cannot drift apart."""
import ctypes
import os
import re
import subprocess

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(PKG_DIR)
HEADER = os.path.join(ROOT, "include", "cpg_api.h")
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("CPG_LIB_PATH", os.path.join(PKG_DIR, "libcpg_hip.so"))  # override: diagnostic builds only
SOURCES = ["api.hip", "gemm.hip", "gru.hip", "gru_persist.hip", "lstm.hip", "lstm_persist.hip", "decode.hip", "decode_fused.hip", "losses.hip", "optim.hip", "rng.hip", "class.hip", "classifier.hip", "comm.hip", "planes.hip"]


class LibraryMissing(RuntimeError):
    pass


_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
    "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "void": None,
}


def _ctype(decl):
    decl = decl.strip()
    if decl.endswith("*") or "*" in decl:
        if decl.replace(" ", "") == "constchar*":
            return ctypes.c_char_p
        return ctypes.c_void_p
    base = decl.replace("const", "").strip()
    return _CT[base]


def parse_header(path=HEADER):
    """-> {name: (restype, [(argtype, argname), ...])} for every CPG_API declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)  # preprocessor lines (incl. `#define CPG_API`)
    out = {}
    for m in re.finditer(r"CPG_API\s+([^;(]+?)\s*(\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        alist = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)
                alist.append((_ctype(mm.group(1)), mm.group(2)))
        out[name] = (_ctype(ret), alist)
    return out


def abi_version():
    """CPG_ABI_VERSION of csrc/cpg_internal.h: what cpg_version() of a library built from these sources returns."""
    m = re.search(r"#define\s+CPG_ABI_VERSION\s+(\d+)", open(os.path.join(CSRC, "cpg_internal.h")).read())
    return int(m.group(1))


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into the in-tree libcpg_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I", CSRC]
    objdir = os.path.join(os.path.dirname(LIB_PATH), "_build")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in os.listdir(CSRC) if h.endswith(".h"))
    jobs = []
    for src in srcs:  # one translation unit per hipcc process, all at once (the step kernels dominate: ~40 s each)
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = [hipcc, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    return LIB_PATH


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the cpg hot path.")
        self.dll = ctypes.CDLL(LIB_PATH)
        self.sigs = parse_header()
        for name, (ret, args) in self.sigs.items():
            fn = getattr(self.dll, name)  # AttributeError here = header declares a symbol the library lacks
            fn.restype = ret
            fn.argtypes = [a for a, _ in args]
        self.dll.cpg_last_error.restype = ctypes.c_char_p
        want = abi_version()
        got = self.dll.cpg_version()
        if got != want:   # a stale build (or a CPG_LIB_PATH diagnostic library of another revision): its signatures may differ
            raise LibraryMissing(f"{LIB_PATH} reports ABI version {got}, the sources declare {want}: rebuild it "
                                 "(`python __graft_entry__.py build`)")

    def last_error(self):
        return (self.dll.cpg_last_error() or b"").decode()


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib

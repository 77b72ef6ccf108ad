"""ctypes binding of libmi355rn.so — the C-ABI hot path (include/mi355rn.h).

The prototypes are read from the header itself, so the binding cannot drift from the declared ABI.
There is deliberately NO fallback: if the shared library is missing, or a call fails, this raises.
(The reference reaches native code only through torch's dispatcher — train.py:64,81,92 build the three
plugin objects; this module is what sits underneath their replacements.)
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355RN_LIB lets a benchmark A/B two builds of the library in one process-per-variant run (never a fallback)
LIB_PATH = os.environ.get("MI355RN_LIB") or os.path.join(_HERE, "lib", "libmi355rn.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mi355rn.h")

F32, BF16, FP8 = 0, 1, 2

_BASE = {
    "void": None,
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "uint8_t": ctypes.c_uint8,
    "unsigned": ctypes.c_uint,
    "unsigned long long": ctypes.c_ulonglong,
    "unsigned char": ctypes.c_ubyte,
    "char": ctypes.c_char,
}


def _ctype(decl):
    """C parameter / return type text -> ctypes type."""
    d = decl.replace("const", " ").strip()
    arr = re.search(r"\[(\d+)\]\s*$", d)
    if arr:  # `int shape[4]` decays to a pointer
        d = d[: arr.start()].strip()
        nptr = 1
    else:
        nptr = 0
    nptr += d.count("*")
    d = d.replace("*", " ").split()
    base = " ".join(d) if d[0] == "unsigned" else d[0]
    if base in ("mi355_ctx", "mi355_bctx", "mi355_comm", "mi355_crop", "mi355_augment"):
        return ctypes.c_void_p  # opaque handles / descriptor tables (any pointer depth)
    if base == "void" and nptr:
        return ctypes.c_void_p
    if base == "char" and nptr == 1:
        return ctypes.c_char_p
    t = _BASE[base]
    if nptr:
        # typed pointers are passed as raw addresses (torch data_ptr) or byref() objects
        return ctypes.c_void_p
    return t


def parse_header(path=HEADER_PATH):
    """Returns {function name: (restype text, [param decl text, ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    src = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", " ", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mi355_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        protos[name] = (ret, plist)
    return protos


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the MI355X hot path)"
        )
    # torch ships its own copy of the HIP runtime: it must be in the process BEFORE libmi355rn.so is loaded so that
    # both bind to the same runtime instance (loading ours first leaves the library with a second, device-less one)
    import torch  # noqa: F401

    L = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, params) in _protos.items():
        fn = getattr(L, name)  # AttributeError here == header/library mismatch: fail loudly
        rt = ret.replace("const", "").strip()
        if rt == "char *" or rt == "char*":
            fn.restype = ctypes.c_char_p
        else:
            fn.restype = _ctype(rt)
        args = []
        for p in params:
            # drop the parameter name: last identifier (possibly followed by [n])
            m = re.match(r"(.*?)(\b\w+\b)(\s*\[\d+\])?\s*$", p)
            tdecl = (m.group(1) + (m.group(3) or "")).strip()
            args.append(_ctype(tdecl))
        fn.argtypes = args
    _lib = L
    return L


def prototypes():
    lib()
    return dict(_protos)


def last_error():
    return lib().mi355_last_error().decode()


def check(rc):
    if rc != 0:
        raise RuntimeError(f"mi355rn: {last_error()} (status {rc})")


def ptr(t):
    """torch tensor (or None) -> raw address for a pointer argument."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def cur_stream():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(torch_dtype):
    import torch

    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise ValueError(f"unsupported dtype {torch_dtype}")

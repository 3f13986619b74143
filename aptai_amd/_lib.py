"""ctypes loader for libaptai_hip.so (the C ABI in include/aptai_hip.h).

There is NO CPU fallback: every product op goes through this library and raises if it is missing
or if the call fails.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C aptai_amd/csrc`` (hipcc cross-compiles gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libaptai_hip.so")

_lib = None
_lock = threading.Lock()


class AptaiHipError(RuntimeError):
    pass


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 and link libaptai_hip.so in-tree."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], check=True)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise AptaiHipError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C {CSRC}` — aptai_amd has no CPU fallback")
                L = ctypes.CDLL(LIB_PATH)
                L.aptai_last_error.restype = ctypes.c_char_p
                for fn in declared_symbols():
                    if fn.endswith('_workspace_bytes'):
                        getattr(L, fn).restype = ctypes.c_int64
                _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().aptai_last_error().decode(errors="replace")
        raise AptaiHipError(f"{what or 'aptai_hip call'} failed (status {rc}): {msg}")


def declared_symbols() -> list:
    """Every function name declared in include/aptai_hip.h (used by the CPU export test)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "aptai_hip.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aptai_[a-z0-9_]+)\s*\(", text)))

"""Python face of the table-function shells (libplinking_duck_amd.so).

``query("plink_freq", path, counts=True, samples=[0, 2], columns=[...])`` drives
the C++ shell through DuckDB's bind -> init_global -> init_local -> scan protocol
(``pdk_query`` in csrc/shell/extension.cpp) and returns the rows; the named
parameters are the reference's SQL named parameters.  Errors surface as
:class:`InvalidInputException` / :class:`IOException`, the reference's types.
"""

from __future__ import annotations

import ctypes as C
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libplinking_duck_amd.so")


class InvalidInputException(ValueError):
    pass


class IOException(IOError):
    pass


class BinderException(ValueError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run plinking_duck_amd/csrc/build.sh")
    # libpgenhip.so is found through the shell library's $ORIGIN rpath; lib.py loads it first, after torch where
    # torch is installed (two HIP runtimes in one process leave the second without a device, see lib._load)
    from . import lib as _pgenhip  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.pdk_query.restype = C.c_void_p
    lib.pdk_query.argtypes = [C.c_char_p]
    lib.pdk_free.argtypes = [C.c_void_p]
    lib.pdk_functions.restype = C.c_char_p
    return lib


_lib = _load()


def functions():
    return _lib.pdk_functions().decode().split(",")


class Result:
    def __init__(self, doc):
        self.names = doc["names"]
        self.types = doc["types"]
        self.all_names = doc["all_names"]
        self.all_types = doc.get("all_types", [])
        self.threads = doc["threads"]
        self.timing_ms = {k: doc.get(k + "_ms") for k in ("bind", "init", "scan")}
        self.rows = [tuple(r) for r in doc["rows"]]
        self.row_count = doc.get("row_count", len(self.rows))
        self.checksum = doc.get("checksum")  # drain=True: order-independent sum over every projected cell

    def __len__(self):
        return self.row_count

    def column(self, name):
        i = self.names.index(name)
        return [r[i] for r in self.rows]

    def dicts(self):
        return [dict(zip(self.names, r)) for r in self.rows]

    def sorted(self, *keys):
        idx = [self.names.index(k) for k in keys]
        return sorted(self.rows, key=lambda r: tuple((r[i] is None, r[i]) for i in idx))


def query(function: str, *args, columns=None, threads: int = 4, settings=None, drain: bool = False, **named) -> Result:
    """drain=True: the chunks are consumed inside the harness (row count + checksum) instead of coming back as rows."""
    req = {"function": function, "args": list(args), "named": named, "threads": threads}
    if drain:
        req["drain"] = True
    if columns is not None:
        req["columns"] = list(columns)
    if settings:
        req["settings"] = settings
    ptr = _lib.pdk_query(json.dumps(req).encode())
    try:
        doc = json.loads(C.string_at(ptr).decode())
    finally:
        _lib.pdk_free(ptr)
    err = doc.get("error")
    if err:
        kind, msg = err["kind"], err["message"]
        if kind == "Invalid Input Error":
            raise InvalidInputException(msg)
        if kind == "IO Error":
            raise IOException(msg)
        raise BinderException(f"{kind}: {msg}")
    return Result(doc)

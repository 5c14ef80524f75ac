"""Parse include/cruse_hip.h and derive the ctypes signature string of every declaration.

Used by tests (CPU) to check that cruse_amd/_lib.py, the header and the built
library agree symbol by symbol."""
from __future__ import annotations

import os
import re

HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "cruse_hip.h")


def _code(tp: str) -> str:
    tp = tp.strip()
    if "*" in tp:
        return "c" if tp.replace("const", "").replace(" ", "") == "char*" else "p"
    base = tp.replace("const", "").strip()
    return {"int": "i", "unsigned": "i", "float": "f", "long long": "q", "unsigned long long": "Q", "size_t": "z", "double": "d", "void": ""}[base]


def parse_header(path: str = HEADER):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"^([\w\s\*]+?)\b(cruse_\w+)\s*\(([^)]*)\)\s*;", src, flags=re.M):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        codes = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                tp = a if a.endswith("*") else a.rsplit(" ", 1)[0] + ("*" if "*" in a.rsplit(" ", 1)[1] else "")
                if "*" in a:
                    tp = "char*" if a.replace("const", "").strip().startswith("char") else "void*"
                codes += _code(tp)
        r = "s" if ("char" in ret and "*" in ret) else _code(ret)
        out[name] = (codes, r)
    return out


if __name__ == "__main__":
    for k, v in parse_header().items():
        print(k, v)

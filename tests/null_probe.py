"""Calls every entry point declared in include/*.h with NULL for every pointer and 0 for every scalar.  Run as a script (tests/test_abi.py starts it
as a child so that a crash is a finding, not the end of the test run): prints one line per function, `name status` or `name value`.
--from NAME resumes after a crash; no GPU is needed -- an entry point must refuse NULL handles before it touches the device.
--ctx (needs the device): a real context goes wherever a `dfgpu_ctx *` is taken, every other handle stays NULL."""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declarations():
    out = []
    for h in ("dfgpu.h", "dfgpu_exec.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        for m in re.finditer(r"DFGPU_API\s+([^;{]+?)\s*\b(dfgpu_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
            ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
            params = []
            if args and args != "void":
                depth, cur = 0, ""
                for ch in args:                                   # split on top-level commas (function-pointer parameters hold commas of their own)
                    if ch == "(":
                        depth += 1
                    elif ch == ")":
                        depth -= 1
                    if ch == "," and depth == 0:
                        params.append(cur.strip()); cur = ""
                    else:
                        cur += ch
                params.append(cur.strip())
            out.append((name, ret, params))
    return out


def ctype_of(p):
    if "*" in p or "(" in p or "[" in p:
        return C.c_void_p
    if "double" in p:
        return C.c_double
    if "float" in p:
        return C.c_float
    if "int64_t" in p or "size_t" in p:
        return C.c_int64
    return C.c_int32


def main():
    lib = C.CDLL(os.path.join(ROOT, "datafusion-upstream_amd", "libdfgpu.so"))
    start = sys.argv[sys.argv.index("--from") + 1] if "--from" in sys.argv else None
    ctx = C.c_void_p()
    if "--ctx" in sys.argv:
        lib.dfgpu_ctx_create.argtypes = [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
        assert lib.dfgpu_ctx_create(0, None, C.byref(ctx)) == 0
    for name, ret, params in declarations():
        if start is not None:
            if name == start:
                start = None
            continue
        fn = getattr(lib, name)
        fn.argtypes = [ctype_of(p) for p in params]
        fn.restype = None if ret == "void" else C.c_void_p if "*" in ret else C.c_int64 if "int64_t" in ret else C.c_int32
        sys.stdout.write(name + " "); sys.stdout.flush()
        if ctx and name in ("dfgpu_ctx_destroy", "dfgpu_ctx_create"):
            print("value skipped"); continue
        r = fn(*[(ctx if ctx and re.match(r"(const\s+)?dfgpu_ctx\s*\*\s*\w*$", p) else None) if t is C.c_void_p else t(0) for t, p in zip(fn.argtypes, params)])
        print(("status" if "dfgpu_status" in ret else "value"), r, flush=True)


if __name__ == "__main__":
    main()

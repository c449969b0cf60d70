"""Checks a compiled kernel's assembly (hipcc --save-temps) for the one thing the compiler does not
know about loads issued from inline asm: their destination registers must not be touched between
the load and the SECOND `s_waitcnt vmcnt` after it (the first is the ring hand-over, the second the
counted wait that precedes the use).

    python tools/check_async_loads.py file.s [kernel-name-substring]
"""
import re
import sys


def regs(text):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(r) for r in re.findall(r"(?<![\w\[:])v(\d+)\b", text))
    return out


def main():
    lines = open(sys.argv[1]).read().splitlines()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    kernel, in_asm, pending, bad, n_loads = None, False, [], 0, 0
    for no, raw in enumerate(lines, 1):
        t = raw.strip()
        if re.match(r"^[A-Za-z_][\w$.]*:", t) and not t.startswith(".L"):
            kernel, pending = t.split(":")[0], []
            continue
        if kernel is None or want not in kernel or not t or t.startswith((";", ".")) and "ASM" not in t:
            continue
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if in_asm and t.startswith(("global_load_dwordx4", "global_load_ushort")):
            dst = regs(t.split(",")[0])
            pending.append({"dst": dst, "waits": 0, "line": no})
            n_loads += 1
            continue
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            for p in pending:
                p["waits"] += 1
            pending = [p for p in pending if p["waits"] < 2]
            continue
        used = regs(t)
        for p in pending:
            if used & p["dst"]:
                print(f"{kernel}: line {no} touches v{sorted(used & p['dst'])} of the async load at line {p['line']}: {t}")
                bad += 1
    print(f"{n_loads} async loads checked, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

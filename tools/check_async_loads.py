"""Checks compiled kernels' assembly (hipcc -save-temps) for the two things the compiler does not know
about operands requested from inline asm: nothing may touch their destination registers before
the instruction that is allowed to use them.

  tape     the backward chain's tape loads (global_load_* inside an asm statement): from the load to
           the counted wait that carries the comment `async tape loads landed`
  carry    the M16 forward's run blocks that end by requesting the first A operands of the block
           that follows (ds_read_b128 at the end of an asm statement, no wait behind them): from the
           end of that statement to the start of the next

    python tools/check_async_loads.py tape  file.s [kernel-name-substring]
    python tools/check_async_loads.py carry file.s [kernel-name-substring]

Exit status 1 on a violation, 2 if nothing was found to check (has the kernel changed?).
"""
import re
import sys

MARK = "async tape loads landed"


def regs(text):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(r) for r in re.findall(r"(?<![\w\[:])v(\d+)\b", text))
    return out


def kernels(lines, want):
    """-> (kernel name, line number, text, inside an asm statement) for the kernels whose name contains `want`"""
    kernel, in_asm = None, False
    for no, raw in enumerate(lines, 1):
        t = raw.strip()
        if re.match(r"^[A-Za-z_][\w$.]*:", t) and not t.startswith(".L"):
            kernel, in_asm = t.split(":")[0], False
            continue
        if kernel is None or want not in kernel or not t:
            continue
        if "#ASMSTART" in t:
            in_asm = True
            yield kernel, no, "#ASMSTART", True
            continue
        if "#ASMEND" in t:
            in_asm = False
            yield kernel, no, "#ASMEND", False
            continue
        if t.startswith((";", ".")):
            continue
        yield kernel, no, t, in_asm


def check_tape(lines, want):
    pending, bad, n_loads, current = [], 0, 0, None
    for kernel, no, t, in_asm in kernels(lines, want):
        if kernel != current:
            current, pending = kernel, []
        if in_asm and t.startswith(("global_load_dwordx4", "global_load_ushort")):
            pending.append({"dst": regs(t.split(",")[0]), "line": no})
            n_loads += 1
            continue
        if t.startswith("s_waitcnt") and MARK in t:
            pending = []
            continue
        used = regs(t)
        for p in pending:
            if used & p["dst"]:
                print(f"{kernel}: line {no} touches v{sorted(used & p['dst'])} of the async load at line {p['line']}: {t}")
                bad += 1
    return n_loads, bad


def check_carry(lines, want):
    bad, n_carried, current = 0, 0, None
    block, carried, outside = [], None, False
    for kernel, no, t, in_asm in kernels(lines, want):
        if kernel != current:
            current, block, carried, outside = kernel, [], None, False
        if t == "#ASMSTART":
            block, carried, outside = [], None, False
            continue
        if t == "#ASMEND":
            tail = []                                   # the reads behind the statement's last wait or MFMA
            for ins in reversed(block):
                if ins.startswith("ds_read_b128"):
                    tail.append(ins)
                else:
                    break
            if tail:
                carried = {"dst": set().union(*(regs(i.split(",")[0]) for i in tail)), "line": no}
                n_carried += len(tail)
            outside = True
            continue
        if in_asm:
            block.append(t)
        elif outside and carried:
            used = regs(t)
            if used & carried["dst"]:
                print(f"{kernel}: line {no} touches v{sorted(used & carried['dst'])}, requested by the asm statement that ends at line {carried['line']}: {t}")
                bad += 1
    return n_carried, bad


def main():
    mode, path = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else ""
    lines = open(path).read().splitlines()
    found, bad = (check_tape if mode == "tape" else check_carry)(lines, want)
    print(f"{found} async operands checked, {bad} violations")
    return 1 if bad else (2 if found == 0 else 0)


if __name__ == "__main__":
    sys.exit(main())

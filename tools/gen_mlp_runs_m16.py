"""Generates nerf_amd/csrc/mlp_runs_m16.inc: hand-scheduled inner blocks of the bf16 MLP kernel
built on v_mfma_f32_16x16x32_bf16 (mlp_fwd16_impl.hpp).

A wavefront owns 32 samples as two blocks of 16 columns; a weight unit (1 KiB, one ds_read_b128:
16 output rows x 32 input features) therefore feeds TWO MFMAs, one per sample block.  A block of
the file = one acquired run of N units over four 16-row output tiles: unit i is tile i % 4 of
k-unit i / 4, in piece i / 8 at byte offset (i % 8) * 1024.  A operands are read LOOKAHEAD units
ahead of their MFMAs with one counted s_waitcnt lgkmcnt per pair of units.  The blocks are single
asm statements because hipcc degrades every LDS wait to lgkmcnt(0) while an LDS-DMA is in flight.

A 32-unit run is two statements (inline asm takes at most 30 operands): the first ends by
requesting the A operands of the second's first units, which the second receives already in
flight (`carry`).

    python tools/gen_mlp_runs_m16.py > nerf_amd/csrc/mlp_runs_m16.inc
"""
LOOKAHEAD = 4
TILES = 4


class Lgkm:
    """LDS returns in order: the wait before a consumer counts the reads issued after its producer."""

    def __init__(self, preloaded=()):
        self.issued = list(preloaded)

    def read(self, tag):
        self.issued.append(tag)

    def wait_for(self, *tags):
        last = max(len(self.issued) - 1 - self.issued[::-1].index(t) for t in tags)
        later = len(self.issued) - 1 - last
        self.issued = self.issued[last + 1:]            # everything up to `last` has landed
        return f"s_waitcnt lgkmcnt({later})"


def mfmas(i, a, b0, b1):
    t = i % TILES
    return [f"v_mfma_f32_16x16x32_bf16 %[c{t}0], {a}, {b0}, %[c{t}0]",
            f"v_mfma_f32_16x16x32_bf16 %[c{t}1], {a}, {b1}, %[c{t}1]"]


ACC = [f"c{t}{b}" for t in range(TILES) for b in (0, 1)]
ACC_PARAMS = ", ".join(f"f32x4& {c}" for c in ACC)
ACC_OUTS = ", ".join(f'[{c}] "+v"({c})' for c in ACC)


def emit(name, params, decl, lines, outs, ins):
    body = "\n".join(f'        "{l}\\n\\t"' for l in lines)
    return f"""__device__ __forceinline__ void {name}({params}) {{
{decl}    asm volatile(
{body}
        : {outs}
        : {ins}
        : "memory");
}}
"""


def gen_chain(n, carry_in, carry_out, last):
    """n units whose B operands are registers (k-unit k, sample block s = b{k}{s})."""
    pieces, kus = n // 8, n // TILES
    lg = Lgkm([("a", i) for i in range(LOOKAHEAD)] if carry_in else [])
    rd = lambda i: f"ds_read_b128 %[a{i % LOOKAHEAD}], %[p{i // 8}] offset:{(i % 8) * 1024}"
    lines = []
    if not carry_in:
        for i in range(min(LOOKAHEAD, n)):
            lines.append(rd(i)); lg.read(("a", i))
    for i in range(0, n, 2):
        lines.append(lg.wait_for(("a", i), ("a", i + 1)))
        for j in (i, i + 1):
            lines += mfmas(j, f"%[a{j % LOOKAHEAD}]", f"%[b{j // TILES}0]", f"%[b{j // TILES}1]")
        for j in (i, i + 1):
            if j + LOOKAHEAD < n:
                lines.append(rd(j + LOOKAHEAD)); lg.read(("a", j + LOOKAHEAD))
            elif carry_out:                                  # first units of the next statement
                k = j + LOOKAHEAD - n
                lines.append(f"ds_read_b128 %[a{k % LOOKAHEAD}], %[pn] offset:{k * 1024}"); lg.read(("n", k))
    if last:
        lines.append("s_nop 11")                             # MFMA result -> first VALU reader after the statement
    params = ", ".join([ACC_PARAMS] + [f"bf16x8& a{j}" for j in range(LOOKAHEAD)] + [f"uint32_t p{j}" for j in range(pieces)] +
                       (["uint32_t pn"] if carry_out else []) + [f"const bf16x8& b{k}{s}" for k in range(kus) for s in (0, 1)])
    a_con = '"+v"' if carry_in else '"=&v"'
    outs = ", ".join([ACC_OUTS] + [f'[a{j}] {a_con}(a{j})' for j in range(LOOKAHEAD)])
    ins = ", ".join([f'[p{j}] "v"(p{j})' for j in range(pieces)] + (['[pn] "v"(pn)'] if carry_out else []) +
                    [f'[b{k}{s}] "v"(b{k}{s})' for k in range(kus) for s in (0, 1)])
    name = f"m16_run{n}" + ("_in" if carry_in else "") + ("_out" if carry_out else "")
    return emit(name, params, "", lines, outs, ins)


def gen_stash():
    """8 units (2 encoder k-units x 4 tiles) whose four B operands (k-unit k, block s at
    stash + (2 k + s) KiB) are read from the wave's LDS stash."""
    L = LOOKAHEAD
    lg = Lgkm()
    lines = ["v_add_u32 %[q], %[stash], %[lane16]"]
    for k in (0, 1):
        for s in (0, 1):
            lines.append(f"ds_read_b128 %[b{k}{s}], %[q] offset:{(2 * k + s) * 1024}"); lg.read(("b", k, s))
    rd = lambda i: f"ds_read_b128 %[a{i % L}], %[p0] offset:{i * 1024}"
    for i in range(L):
        lines.append(rd(i)); lg.read(("a", i))
    for i in range(0, 8, 2):
        k = i // TILES
        lines.append(lg.wait_for(("a", i), ("a", i + 1), ("b", k, 0), ("b", k, 1)) if any(t[0] == "b" and t[1] == k for t in lg.issued)
                     else lg.wait_for(("a", i), ("a", i + 1)))
        for j in (i, i + 1):
            lines += mfmas(j, f"%[a{j % L}]", f"%[b{k}0]", f"%[b{k}1]")
        for j in (i, i + 1):
            if j + L < 8:
                lines.append(rd(j + L)); lg.read(("a", j + L))
    lines.append("s_nop 11")
    temps = [f"a{j}" for j in range(L)] + [f"b{k}{s}" for k in (0, 1) for s in (0, 1)]
    decl = f"    bf16x8 {', '.join(temps)};\n    uint32_t q;\n"
    outs = ", ".join([ACC_OUTS] + [f'[{t}] "=&v"({t})' for t in temps] + ['[q] "=&v"(q)'])
    return emit("m16_run8_stash", f"{ACC_PARAMS}, uint32_t p0, uint32_t stash, uint32_t lane16", decl, lines, outs,
                '[p0] "v"(p0), [stash] "s"(stash), [lane16] "v"(lane16)')


print("// GENERATED by tools/gen_mlp_runs_m16.py -- do not edit.")
print("// c<t><s>: accumulator of output tile t (16 rows), sample block s; a*: scratch A operands;")
print("// p*: LDS byte address of each piece for this lane; b<k><s>: B operand of k-unit k, block s.")
print(gen_chain(8, False, False, True))
print(gen_chain(16, False, True, False))
print(gen_chain(16, True, False, True))
print(gen_stash())

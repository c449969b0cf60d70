"""Generates nerf_amd/csrc/mlp_runs_m16.inc: hand-scheduled inner blocks of the bf16 MLP kernel
built on v_mfma_f32_16x16x32_bf16 (mlp_fwd16_impl.hpp).

A wavefront owns 32 samples as two blocks of 16 columns; a weight unit (1 KiB, one ds_read_b128:
16 output rows x 32 input features) therefore feeds TWO MFMAs, one per sample block.  A run block
covers N units over four 16-row output tiles: unit i is tile i % 4 of k-unit i / 4, in piece i / 8
at byte offset (i % 8) * 1024.  A operands are read LOOKAHEAD units ahead of their MFMAs with one
counted s_waitcnt lgkmcnt per pair of units.  The blocks are single asm statements because hipcc
degrades every LDS wait to lgkmcnt(0) while an LDS-DMA is in flight.

The two wavefronts of a SIMD take turns: one runs MFMAs while its partner converts results,
refills the ring and prepares its next run; a workgroup barrier separates the phases.  A `_first`
block opens a phase: it requests its first operands, waits for the LDS writes issued before them
(the wave's share of the ring refill), meets the barrier (so the read latency is paid while the
partner's run drains), then runs.  A block either requests the first A operands of
the block that follows (`_out`) or closes the phase (`_last`: s_barrier right behind the last
MFMA, so the partner's run starts while this one's results drain, then the wait states an MFMA
result needs before a VALU instruction may read it).  A 32-unit run is two statements: inline asm
takes at most 30 operands.

    python tools/gen_mlp_runs_m16.py > nerf_amd/csrc/mlp_runs_m16.inc
"""
LOOKAHEAD = 4          # A operands in flight ahead of their MFMAs (units); 8 measured the same
TILES = 4


class Lgkm:
    """LDS returns in order: the wait before a consumer counts the reads issued after its producer."""

    def __init__(self, preloaded=()):
        self.issued = list(preloaded)

    def read(self, tag):
        self.issued.append(tag)

    def wait_for(self, *tags):
        tags = [t for t in tags if t in self.issued]
        if not tags:
            return None
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
A_PARAMS = ", ".join(f"bf16x8& a{j}" for j in range(LOOKAHEAD))


def emit(name, params, decl, parts, outs, ins):
    """`parts` = (opening, body, closing) instruction lists.  The product build issues them as one
    statement.  The stamps build (-DNERFX_STAMPS) cuts the statement at the two barriers and reads
    the clock there, so that a run's own duration and the time spent waiting at each barrier are
    booked apart; the clock read returns through lgkmcnt, hence the full wait in front of the body
    (the counted waits inside stay valid: they only ever ask for less)."""
    quote = lambda ls: "\n".join(f'        "{l}\\n\\t"' for l in ls)
    opening, body, closing = parts
    stmt = lambda ls, o=outs: f"    asm volatile(\n{quote(ls)}\n        : {o}\n        : {ins}\n        : \"memory\");\n"
    later = outs.replace('"=&v"', '"+v"')          # what the opening has requested is an input of the rest
    cut = ""
    if opening:
        cut += stmt(opening) + "    st.mark_wait(ST_ACQUIRE, 0);\n"
    cut += stmt((["s_waitcnt lgkmcnt(0)"] if opening else []) + body, later if opening else outs) + "    st.mark(ST_RUN);\n"
    if closing:
        cut += stmt(closing, later) + "    st.mark_wait(ST_BARRIER, 1);\n    ++st.phase;\n"
    return f"""__device__ __forceinline__ void {name}({params}, Stamps& st) {{
{decl}#if defined(NERFX_STAMPS) && NERFX_STAMPS != 2
{cut}#else
{stmt(opening + body + closing)}#endif
}}
"""


def gen_run(n, first, carry_out, stash_b=False):
    """n units.  `first`: the block opens a phase -- it requests its first LOOKAHEAD A operands,
    then meets the phase barrier, then runs; otherwise the A operands are in flight on entry,
    requested by the block before it.  With `stash_b` (first blocks only) the block reads its B
    operands from the wave's LDS stash at stash + (2 k + s) KiB into b<k><s>, which the caller may
    hand to the blocks that follow; the B operands of the other blocks are registers b<k><s>."""
    assert first or not stash_b
    pieces, kus = n // 8, n // TILES
    lg = Lgkm() if first else Lgkm([("a", i) for i in range(LOOKAHEAD)])
    rd = lambda i: f"ds_read_b128 %[a{i % LOOKAHEAD}], %[p{i // 8}] offset:{(i % 8) * 1024}"
    lines, opening = [], []
    if stash_b:
        lines.append("v_add_u32 %[q], %[stash], %[lane16]")
        for k in range(kus):
            for s in (0, 1):
                lines.append(f"ds_read_b128 %[b{k}{s}], %[q] offset:{(2 * k + s) * 1024}")
                lg.read(("b", k, s))
    if first:
        for i in range(LOOKAHEAD):
            lines.append(rd(i))
            lg.read(("a", i))
        # the ring refill this wave has just written to LDS (older than the reads above, and LDS
        # completes in order) must be in place before the barrier publishes it
        lines.append(f"s_waitcnt lgkmcnt({len(lg.issued)})")
        lines.append("s_barrier")
        opening, lines = lines, []
    for i in range(0, n, 2):
        w = lg.wait_for(("a", i), ("a", i + 1), ("b", i // TILES, 0), ("b", i // TILES, 1))
        if w:
            lines.append(w)
        for j in (i, i + 1):
            lines += mfmas(j, f"%[a{j % LOOKAHEAD}]", f"%[b{j // TILES}0]", f"%[b{j // TILES}1]")
        for j in (i, i + 1):
            if j + LOOKAHEAD < n:
                lines.append(rd(j + LOOKAHEAD))
                lg.read(("a", j + LOOKAHEAD))
            elif carry_out:                                  # first units of the next statement
                k = j + LOOKAHEAD - n
                lines.append(f"ds_read_b128 %[a{k % LOOKAHEAD}], %[pn] offset:{k * 1024}")
                lg.read(("n", k))
    closing = [] if carry_out else ["s_barrier", "s_nop 11"]
    bnames = [f"b{k}{s}" for k in range(kus) for s in (0, 1)]
    a_con = '"=&v"' if first else '"+v"'
    params = [ACC_PARAMS, A_PARAMS] + [f"uint32_t p{j}" for j in range(pieces)] + (["uint32_t pn"] if carry_out else [])
    outs = [ACC_OUTS] + [f'[a{j}] {a_con}(a{j})' for j in range(LOOKAHEAD)]
    ins = [f'[p{j}] "v"(p{j})' for j in range(pieces)] + (['[pn] "v"(pn)'] if carry_out else [])
    decl = ""
    if stash_b:
        params += [f"bf16x8& {b}" for b in bnames] + ["uint32_t stash", "uint32_t lane16"]
        decl = "    uint32_t q;\n"
        outs += [f'[{b}] "=&v"({b})' for b in bnames] + ['[q] "=&v"(q)']
        ins += ['[stash] "s"(stash)', '[lane16] "v"(lane16)']
    else:
        params += [f"const bf16x8& {b}" for b in bnames]
        ins += [f'[{b}] "v"({b})' for b in bnames]
    name = f"m16_run{n}" + ("_stash" if stash_b else "") + ("_first" if first else "") + ("_out" if carry_out else "_last")
    return emit(name, ", ".join(params), decl, (opening, lines, closing), ", ".join(outs), ", ".join(ins))


print("// GENERATED by tools/gen_mlp_runs_m16.py -- do not edit.")
print("// c<t><s>: accumulator of output tile t (16 rows), sample block s; a*: A operands in flight;")
print("// p*: LDS byte address of each piece for this lane (pn: first piece of the next block);")
print("// b<k><s>: B operand of k-unit k, block s.")
print(f"#define M16_LOOKAHEAD {LOOKAHEAD}")
print("#define M16_A_ARGS(a) " + ", ".join(f"a[{j}]" for j in range(LOOKAHEAD)))
print(gen_run(8, True, False, stash_b=True))     # an encoder segment alone
print(gen_run(8, False, True))                   # layer 0 (all sixteen tiles in one phase): middle blocks
print(gen_run(8, False, False))                  # layer 0: last block
print(gen_run(8, True, True, stash_b=True))      # skip layer: encoder segment, chained segment follows
print(gen_run(8, True, True))                    # rgb.0: encoder operands in registers, chained segment follows
print(gen_run(16, True, True))                   # first half of a chained segment opening a phase
print(gen_run(16, False, True))                  # first half of a chained segment behind an encoder segment
print(gen_run(16, False, False))                 # second half of a chained segment

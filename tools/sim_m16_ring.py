"""Replays the weight-ring protocol of the M16 inference kernel (mlp_fwd16_impl.hpp, PhasedRing) on
the host: which stream unit every ring unit holds, from which phase it is visible, and whether every
read of every run (and the read-ahead of the run that follows) finds its unit in place and no refill
lands on a unit that is read in the same phase.  Run for the four stream shapes (depth 4 / 8, with
and without the colour layers), two ring lengths and two places for the wait on a wave's own loads:
before the barrier that opens its next run (`defer False`, what the kernel does) or only before the
barrier that closes that run (`defer True`: safe with a ring of 112 units, not with 80).

    python tools/sim_m16_ring.py
"""
import sys
def simulate(segs, R, defer, blocks=6, chunk=4, ahead=4):
    """segs: units per segment of one block.  Returns list of violations."""
    U = sum(segs)
    n_units = U
    ring = [(u, -10) for u in range(R)]          # (stream unit held, phase from which visible)
    segments = []                                  # global list (stream start, units)
    s = 0
    for b in range(blocks):
        for u in segs:
            segments.append((s, u)); s += u
    bad = []
    # events: for phase k: reads, then (in in-between) issues
    reads_now = set()
    def read(seg_idx, phase, lo, hi):
        s0, u = segments[seg_idx]
        for off in range(lo, min(hi, u)):
            su = s0 + off
            reads_now.add(su % R)
            held, vis = ring[su % R]
            if held != su or vis > phase:
                bad.append((phase, seg_idx, off, held, su, vis))
    def refill(slot_idx, c_lo, c_hi, issue_phase):
        s0, u = segments[slot_idx]
        for c in range(c_lo, c_hi):
            for off in range(c * chunk, c * chunk + chunk):
                su = s0 + off
                if su % R in reads_now: bad.append(('overwrite', issue_phase, slot_idx, off))
                vis = issue_phase + (2 if defer else 1)
                ring[su % R] = (su + R, vis)
    nseg = len(segments)
    for k in range(2 * nseg + 2):
        reads_now.clear()
        # reads in phase k
        if k % 2 == 0 and k // 2 < nseg: read(k // 2, k, 0, 10**9)
        if k % 2 == 1 and (k - 1) // 2 < nseg: read((k - 1) // 2, k, 0, 10**9)
        # read-ahead of the NEXT phase's runner (first `ahead` units) happens during phase k
        nk = k + 1
        if nk % 2 == 0 and nk // 2 < nseg: read(nk // 2, k, 0, ahead)
        if nk % 2 == 1 and (nk - 1) // 2 < nseg: read((nk - 1) // 2, k, 0, ahead)
        # issues during phase k: group 0 ran segment n at phase 2n = k-1 -> k odd; duty: second half of slot n-1
        if k % 2 == 1:
            n = (k - 1) // 2
            if 1 <= n < nseg + 0:
                u = segments[n - 1][1]; chunks = u // chunk; first = (chunks + 1) // 2
                refill(n - 1, first, chunks, k)
        # group 1 ran segment n at phase 2n+1 = k-1 -> k even, k>=2; duty: first half of slot n
        if k % 2 == 0 and k >= 2:
            n = (k - 2) // 2
            if n < nseg:
                u = segments[n][1]; chunks = u // chunk; first = (chunks + 1) // 2
                refill(n, 0, first, k)
    return bad
dens = [32] + [32]*4 + [40]*4 + [32]*4
col = dens + [40]*2 + [32]                      # rgb.0 with `features` folded in, rgb.2 + 24 padding units
dens4 = [32] + [32]*4*3 + [40]*4 + [32]*4*3
col4 = dens4 + [40]*2 + [8]
for name, segs in (("dens", dens), ("col", col), ("dens4", dens4), ("col4", col4)):
    for R in (80, 112):
        for defer in (False, True):
            if sum(segs) % R and name in ("dens", "col") and R == 112: print("not whole", name)
            b = simulate(segs, R, defer)
            print(name, sum(segs), "R", R, "defer", defer, "violations", len(b), b[:3])

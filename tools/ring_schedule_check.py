"""Happens-before check of the LDS-DMA schedule of gemm_ring_kernel (v-express_amd/csrc/vx_gemm_ring.hip).

The kernel orders its own global->LDS copies with counted `s_waitcnt vmcnt(N)` + `s_barrier` (the compiler does not
know about them: they are inline asm), so a wrong count is a silent race.  This script replays the schedule exactly as
the kernel issues it - prologue, the one-slot stagger between the two wave rows, four (L slot, barrier, M slot, barrier)
phases per K-tile, the issue groups g0 = {B0,B1,B2}, g1 = {B3,B4,A0}, g2 = {A1,A2}, g3 = {A3} and every vmcnt immediate -
for both wave rows as two programs that only synchronise at barriers, and proves for every block length S (K-tiles in the
block's sequence) and tile shape (K-tiles per output tile):

  RAW  every fragment read of piece P of K-tile u happens after EVERY wave's copy of (u, P) was retired by that wave's
       own vmcnt wait AND a barrier both sides crossed lies between that wait and the read;
  WAR  every copy into a buffer slot is issued after EVERY wave finished (lgkmcnt(0) + barrier) its reads of the K-tile
       that occupied the slot before (u - 2).

A copy is assumed to land as late as the waits allow (the adversarial case).  Waves of one wave row run the same program,
so two agents stand for the eight waves.  Run: python tools/ring_schedule_check.py   (also a CPU test:
tests/test_host_logic.py::test_ring_dma_schedule_is_race_free)."""
import itertools

GROUPS = {0: ["B0", "B1", "B2"], 1: ["B3", "B4", "A0"], 2: ["A1", "A2"], 3: ["A3"]}
B_PIECES = ["B0", "B1", "B2", "B3", "B4"]


def program(grp, S, nk, m_issue=0):
    """m_issue: the experimental placements of vx_gemm_ring.hip's VX_RING_MISSUE (0 = product schedule).
    Event list of one wave of wave row `grp`: ("issue", u, piece) | ("wait", n) | ("read", u, piece) | ("lgkm0",) |
    ("bar",).  Mirrors the kernel: prologue (vx_gemm_ring.hip 'prologue'), then per output tile the stagger barrier
    of row 1, nk x ktile(e1, e2), the re-align barrier of row 0."""
    ev = []
    iss = 0                                    # K-tile sequence number at the issue pointer

    def issue_group(g):
        for piece in GROUPS[g]:
            ev.append(("issue", iss, piece))
    for g in range(4):
        issue_group(g)
    iss += 1
    if S > 1:
        for g in range(2 if m_issue == 3 else 3):
            issue_group(g)
        ev.append(("wait", 9 if m_issue == 3 else 11))
    else:
        ev.append(("wait", 3))
    ev.append(("bar",))
    u = 0
    tiles = S // nk
    for _ in range(tiles):
        if grp == 1:
            ev.append(("bar",))
        for _ in range(nk):
            e1, e2 = u + 1 < S, u + 2 < S
            for ph in range(4):
                if ph == 0:
                    for b in B_PIECES:
                        ev.append(("read", u, b))
                ev.append(("read", u, f"A{ph}"))
                if ph == 0:
                    if e1:
                        if m_issue == 3:
                            issue_group(2)     # A1, A2 of u+1
                        issue_group(3)         # A3 of u+1 (iss == u+1 here)
                        iss += 1
                        ev.append(("wait", 11))
                    else:
                        ev.append(("wait", 2))
                elif ph == 1:
                    if e2:
                        if m_issue == 2:
                            ev.append(("issue", iss, "B0")); ev.append(("issue", iss, "B1"))
                            ev.append(("wait", 12))
                        else:
                            issue_group(0)
                            ev.append(("wait", 13))
                    else:
                        ev.append(("wait", 10 if e1 else 1))
                elif ph == 2:
                    if e2:
                        if m_issue in (1, 2):
                            ev.append(("issue", iss, "B3")); ev.append(("issue", iss, "B4"))
                            ev.append(("wait", 14))
                        else:
                            issue_group(1)
                            ev.append(("wait", 15))
                    else:
                        ev.append(("wait", 9 if e1 else 0))
                else:
                    if e2:
                        if m_issue in (1, 2):
                            ev.append(("issue", iss, "A1"))
                            ev.append(("wait", 10))
                        elif m_issue == 3:
                            ev.append(("wait", 9))
                        else:
                            issue_group(2)
                            ev.append(("wait", 11))
                    else:
                        ev.append(("wait", 3 if e1 else 0))
                ev.append(("lgkm0",))
                ev.append(("bar",))
                # M slot between the two barriers: MFMAs, and under m_issue the copies that left the L slots
                if e2 and m_issue == 2 and ph == 1:
                    ev.append(("issue", iss, "B2"))
                if e2 and m_issue in (1, 2) and ph == 2:
                    ev.append(("issue", iss, "A0"))
                if e2 and m_issue in (1, 2) and ph == 3:
                    ev.append(("issue", iss, "A2"))
                ev.append(("bar",))
            u += 1
        if grp == 0:
            ev.append(("bar",))
    return ev


def annotate(ev):
    """Per agent: retire position of every copy (index of the wait that guarantees it), barrier count before each
    event, read-completion positions."""
    bars = 0
    out = []                                   # (kind, payload, barriers crossed before this event)
    pending = []                               # issued, not yet retired copies in order: (u, piece)
    retired_at = {}                            # (u, piece) -> barriers crossed when its wait executed
    issued_at = {}
    reads = []                                 # (u, piece, barriers crossed when the read is COMPLETE = at the lgkm0)
    open_reads = []
    for e in ev:
        if e[0] == "bar":
            bars += 1
        elif e[0] == "issue":
            pending.append((e[1], e[2]))
            issued_at[(e[1], e[2])] = bars
        elif e[0] == "wait":
            while len(pending) > e[1]:
                retired_at[pending.pop(0)] = bars
        elif e[0] == "read":
            open_reads.append((e[1], e[2], bars))
        elif e[0] == "lgkm0":
            for (u, piece, at) in open_reads:
                reads.append((u, piece, at, bars))
            open_reads = []
    return dict(bars=bars, retired_at=retired_at, issued_at=issued_at, reads=reads, unretired=pending)


def check(S, nk, m_issue=0):
    agents = [annotate(program(g, S, nk, m_issue)) for g in (0, 1)]
    if agents[0]["bars"] != agents[1]["bars"]:
        return f"barrier counts differ: {agents[0]['bars']} vs {agents[1]['bars']}"
    for a in agents:
        if a["unretired"]:
            return f"copies never retired: {a['unretired'][:3]}"
    for r, reader in enumerate(agents):
        for (u, piece, start_bars, done_bars) in reader["reads"]:
            for w, writer in enumerate(agents):
                # RAW: the writer's wait for (u, piece) ran after `rb` barriers; the read starts after `start_bars`.
                # Same agent: program order suffices for its own slab, but it also reads the other waves' slabs of its
                # own row, which run the same program: a barrier is needed in every case.
                rb = writer["retired_at"].get((u, piece))
                if rb is None:
                    return f"S={S} nk={nk}: ({u},{piece}) is read but never copied"
                if not rb < start_bars:
                    return (f"S={S} nk={nk}: RAW row{r} reads ({u},{piece}) after {start_bars} barriers, row{w} "
                            f"retires its copy only after {rb}")
                # WAR: the next occupant of the slot is (u + 2, piece)
                ib = writer["issued_at"].get((u + 2, piece))
                if ib is not None and not done_bars < ib:
                    return (f"S={S} nk={nk}: WAR row{w} refills ({u + 2},{piece}) after {ib} barriers, row{r} finishes "
                            f"reading ({u},{piece}) only after {done_bars}")
    return None


def main():
    import sys
    m_issue = int(sys.argv[sys.argv.index("--m-issue") + 1]) if "--m-issue" in sys.argv else 0
    bad = 0
    for nk, tiles in itertools.product((1, 2, 3, 5, 10, 45), (1, 2, 3)):
        err = check(nk * tiles, nk, m_issue)
        print(f"nk={nk:3d} tiles={tiles}: {'ok' if err is None else err}")
        bad += err is not None
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()

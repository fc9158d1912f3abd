"""Happens-before check of the weight-stream protocol of csrc/vx_tblock.hip (the fused temporal attention block): the
counted `s_waitcnt vmcnt(N)` waits, the one barrier per chunk and the three-slot LDS ring, restated as the per-wave
sequence of vector-memory operations of two consecutive tiles and checked against what the hardware guarantees:

  * vmcnt counts loads (incl. LDS-DMA copies) and stores of a wave; LOADS retire in issue order among themselves, stores
    need not retire in order relative to loads.  After `s_waitcnt vmcnt(N)` at most N operations are outstanding, so the
    outstanding loads are at most the N youngest loads: a load is GUARANTEED complete iff at least N loads were issued
    after it.  (Extra stores in flight only make a wait stricter.)
  * a wave's copies of chunk c have landed when it passes the wait at the top of iteration c; the barrier behind that wait
    makes that true for every wave before any wave reads the chunk;
  * the slot a copy overwrites held the chunk read in the PREVIOUS iteration: every wave left it at this iteration's barrier.

Checked: (1) every chunk read is preceded by a wait that guarantees the reader's own copies of it, (2) no copy is issued
into a slot before the barrier that ends the reads of its previous content, (3) the O^T parking area and the statistics
scratch area are never written while a reader of the previous content may still be active.  The constants are read from
the kernel source, so a change of the protocol there fails this check (tests/test_host_logic.py runs it)."""
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "v-express_amd", "csrc", "vx_tblock.hip")


def constants():
    s = open(SRC).read()
    nw = int(re.search(r"#define VX_TB_WAVES (\d+)", s).group(1))
    assert "constexpr int CPW = (NBLK + NW - 1) / NW;" in s and "constexpr int NX = NPX * TB_KS;" in s
    assert "static constexpr int NBLK = 20 + 2 * HB;" in s          # 22 copy blocks per QKV chunk at F = 16, 24 at F = 24
    assert "TB_QKV_CHUNKS = 32, TB_TILE_CHUNKS = 44" in s and "constexpr int JB = 3;" in s
    # the three waits of the protocol, as written in the kernel
    assert s.count("tb_wait_vm<CPW>();") == 2 and "tb_wait_vm<CPW + NX>();" in s
    assert "if (hp2 == 0 && part > 0 && more) tb_wait_vm<CPW + NX>();" in s
    assert "load_x(0);       // BEFORE the first copies" in s
    cpw = (22 + nw - 1) // nw
    assert cpw == (24 + nw - 1) // nw                                 # the same number of copies per wave for both window lengths
    nx = (16 // nw) * 10
    return dict(NW=nw, CPW=cpw, NX=nx, QKV=32, CHUNKS=44, SNJ=20 // (nw // 2), JB=3)


def wave_program(K, tiles):
    """Per-wave event list of `tiles` consecutive tiles: ('L', tag) load, ('S', tag) store, ('W', n) wait, ('B', id) barrier,
    ('R', chunk_id) read of a ring slot, with chunk_id = (tile, chunk)."""
    ev = [("L", ("x", 0))] * K["NX"]
    for c in (0, 1):
        ev += [("L", ("copy", (0, c)))] * K["CPW"]
    for t in range(tiles):
        more = t + 1 < tiles
        for c in range(K["CHUNKS"]):
            j = c - K["QKV"]
            n = K["CPW"] + K["NX"] if (more and j in (1, 2)) else K["CPW"]
            ev.append(("W", n))
            ev.append(("B", (t, c)))
            nxt = (t, c + 2) if c + 2 < K["CHUNKS"] else (t + 1, c + 2 - K["CHUNKS"])
            ev += [("L", ("copy", nxt))] * K["CPW"]          # (the last tile's run-ahead copies are issued as well)
            if j == 0 and more:
                ev += [("L", ("x", t + 1))] * K["NX"]
            ev.append(("R", (t, c)))
        done = 0
        while done < K["SNJ"]:                               # epilogue: residual loads of a batch, then its stores
            nb = min(K["JB"], K["SNJ"] - done)
            ev += [("L", ("res", t))] * (4 * nb)
            ev += [("S", ("out", t))] * (4 * nb)
            done += nb
    ev.append(("W", 0))
    return ev


def check_waits(K, tiles=3):
    ev = wave_program(K, tiles)
    loads = [i for i, e in enumerate(ev) if e[0] == "L"]
    pos_in_loads = {i: n for n, i in enumerate(loads)}
    checked = 0
    for i, e in enumerate(ev):
        if e[0] != "R":
            continue
        chunk = e[1]
        mine = [k for k in loads if k < i and ev[k][1] == ("copy", chunk)]
        assert len(mine) == K["CPW"], (chunk, len(mine))
        last_copy = max(mine)
        # the wait at the top of this iteration: the nearest ('W', n) before the read
        w = max(k for k in range(i) if ev[k][0] == "W")
        assert w > last_copy
        n = ev[w][1]
        younger = sum(1 for k in loads if last_copy < k < w)
        assert younger >= n, f"chunk {chunk}: {younger} loads younger than its copies at a vmcnt({n}) wait"
        # and the wait is not needlessly strict in the steady state: the copies of the NEXT chunk may stay in flight
        checked += 1
    assert checked == tiles * K["CHUNKS"]
    return checked


def check_slots(K, tiles=3):
    """Ring slot of the g-th chunk of the stream = g % 3.  A copy of stream chunk g + 2 is issued in iteration g, behind
    barrier g; the previous content of its slot is stream chunk g - 1, whose reads every wave finished before barrier g."""
    g = 0
    for t in range(tiles):
        for c in range(K["CHUNKS"]):
            target = g + 2
            assert target % 3 == (g - 1) % 3                 # the slot of the chunk read in the previous iteration
            # readers of stream chunk g - 1 are in iteration g - 1, i.e. strictly before barrier g; the copy is after it
            g += 1
    return g


def check_parking_areas(K):
    """O^T of head pair hp is written after chunk 8 hp + 7 of phase 1 and read in phase-2 iterations 3 hp .. 3 hp + 2 (chunks
    32 + 3 hp ..); the next tile writes it after ITS chunk 8 hp + 7.  Barriers are numbered by (tile, chunk) = the barrier
    at the top of that iteration.  A write after barrier (t, cw) and a read before barrier (t', cr) cannot overlap iff
    (t', cr) <= (t, cw) [reader done first] or the write precedes a barrier that precedes the read."""
    q = K["QKV"]
    for hp in range(4):
        w_after, w_before = (0, 8 * hp + 7), (0, 8 * hp + 8)            # written between these two barriers
        reads = [((0, q + 3 * hp + p), (0, q + 3 * hp + p + 1)) for p in range(3)]      # read between these barriers
        for r_after, r_before in reads:
            assert w_before <= r_after                                   # the barrier ending the write precedes the read
        nxt_w_after = (1, 8 * hp + 7)
        for r_after, r_before in reads:
            rb = r_before if r_before[1] < K["CHUNKS"] else (1, 0)      # the last part's reads end at the next tile's first barrier
            assert rb <= nxt_w_after
    # statistics scratch (stats_out) = the O^T area of head pair 0: written in the epilogue (after the last barrier of the
    # tile, (0, 43)), read behind one extra barrier; last read of O^T[0] ended at barrier (0, 35) <= (0, 43); the next writer
    # of O^T[0] is behind barrier (1, 7), and every wave reaches barrier (1, 0) only after its own scratch reads
    assert (0, q + 3) <= (0, K["CHUNKS"] - 1) and (1, 0) <= (1, 7)
    # (rstd, -mean rstd) table: written at tile start (before barrier (t, 0)), read by the V blocks (chunks 8 hp + 5 ..)
    # behind barriers >= (t, 5); the previous tile's last reader was in iteration (t - 1, 31), before barrier (t - 1, 32)
    assert (0, 0) <= (0, 5) and (0, 32) <= (1, 0)
    return True


def main():
    K = constants()
    n = check_waits(K)
    g = check_slots(K)
    check_parking_areas(K)
    return K, n, g


if __name__ == "__main__":
    K, n, g = main()
    print(f"vx_tblock.hip weight-stream protocol: NW={K['NW']} CPW={K['CPW']} NX={K['NX']}: {n} chunk reads each behind a "
          f"sufficient counted wait, {g} slot reuses behind the barrier that ends the previous reads, parking areas ok")

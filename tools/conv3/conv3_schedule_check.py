"""Happens-before check of the copy / wait / barrier protocol of tools/conv3/vx_conv3.hip (timing only; the address arithmetic is
tools/conv3_emulate.py's job).

The kernel's workgroup = two wave rows that execute the SAME slot sequence, row 1 one slot behind row 0, with one
s_barrier after every slot.  "Interval" k = the time between barrier k and barrier k + 1: row 0 executes its slot k there,
row 1 its slot k - 1.  Every wave issues its own share of every LDS-DMA copy; copies of a wave retire in order, so at
`s_waitcnt vmcnt(N)` all but the wave's N youngest copies have landed.  Rules proven for every LDS access of the replayed
sequence (prologue, two tiles of two periods each, the run-ahead copies past the end):

  R1  a fragment / table read of a region happens in an interval strictly AFTER the interval in which each wave row's
      wait retired that row's share of the copy that filled the region (wait -> barrier -> read);
  R2  a wave normalises a plane unit in place only after ITS OWN wait retired ITS copy of that unit (same wave: no
      barrier needed), and every fragment read of the plane happens in a later interval than both rows' normalisation;
  R3  a copy into a region is issued in an interval strictly after the last read (and the last in-place write) of the
      region's previous contents by either row;
  R4  the table a normalisation reads was copied for the tile the plane belongs to, and R1 holds for it.

The wait immediates are read from the kernel source, so editing them there without re-proving fails this check
(tests/test_host_logic.py runs it; `--break N` perturbs immediate N by +1 and must fail)."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "vx_conv3.hip")


def kernel_immediates():
    """(prologue waits, {role: N}) as written in the kernel: c3_wait_vm<..> in source order."""
    text = open(SRC).read()
    body = text[text.index("// ------------------------------------------------------------------ prologue"):]
    waits = [int(x) for x in re.findall(r"c3_wait_vm<(\d+)>\(\)", body)]
    # prologue: <10>, <5>; K-tile: copy_odd <9>, copy_even <8>, v in (1, 6) <6>, else <5>; kernel end <0>
    assert len(waits) == 7 and waits[-1] == 0, waits
    return waits[:2], dict(odd=waits[2], even=waits[3], after=waits[4], plain=waits[5])


class Row:
    """One wave row: its copies in issue order and what its waits have retired."""

    def __init__(self, lag):
        self.lag = lag              # 0 / 1: slot s of this row runs in interval s + lag
        self.issued = []            # (op name, region, content tag)
        self.retired = 0            # copies [0, retired) have landed
        self.retire_at = {}         # copy index -> interval of the wait that retired it


class Check:
    def __init__(self, prologue_waits, imm):
        self.pw, self.imm = prologue_waits, imm
        self.rows = [Row(0), Row(1)]
        self.content = {}           # region -> {row: (copy index, tag)} current / in-flight contents
        self.last_touch = {}        # region -> interval of the last read / in-place write of the CURRENT contents
        self.transformed = {}       # (region, row) -> interval of that row's in-place normalisation of the current contents
        self.errors = []
        self.n_checked = 0

    def err(self, msg):
        self.errors.append(msg)

    # ---- actions of one row in one interval
    def issue(self, row, interval, op, region, tag):
        r = self.rows[row]
        lt = self.last_touch.get(region)
        if lt is not None and not lt < interval:
            self.err(f"R3: row {row} copies {op} into {region} in interval {interval}, last access of the old contents in {lt}")
        r.issued.append((op, region, tag))
        self.content.setdefault(region, {})[row] = (len(r.issued) - 1, tag)
        self.transformed.pop((region, row), None)
        self.n_checked += 1

    def wait(self, row, interval, n):
        r = self.rows[row]
        upto = max(r.retired, len(r.issued) - n)
        for i in range(r.retired, upto):
            r.retire_at[i] = interval
        r.retired = upto

    def landed_before(self, region, interval, want_tag, what):
        """R1 for both rows' shares of `region`."""
        for row in (0, 1):
            c = self.content.get(region, {}).get(row)
            if c is None:
                self.err(f"R1: {what} in interval {interval}: row {row} never copied {region}")
                continue
            idx, tag = c
            if want_tag is not None and tag != want_tag:
                self.err(f"{what} in interval {interval}: {region} holds {tag}, wanted {want_tag} (row {row})")
            at = self.rows[row].retire_at.get(idx)
            if at is None or not at < interval:
                self.err(f"R1: {what} in interval {interval}: row {row}'s copy of {region} retired in {at}")
        self.n_checked += 1

    def read(self, row, interval, region, want_tag, what, need_transform=False):
        self.landed_before(region, interval, want_tag, what)
        if need_transform:
            for r2 in (0, 1):
                t = self.transformed.get((region, r2))
                if t is None or not t < interval:
                    self.err(f"R2: {what} in interval {interval}: row {r2} normalised {region} in {t}")
        self.last_touch[region] = max(self.last_touch.get(region, -1), interval)

    def transform(self, row, interval, region, tab_region, want_tag, tab_tag):
        c = self.content.get(region, {}).get(row)
        if c is None or c[1] != want_tag:
            self.err(f"R2: row {row} normalises {region} in interval {interval}: holds {c}, wanted {want_tag}")
        else:
            at = self.rows[row].retire_at.get(c[0])
            if at is None or not at <= interval:      # own copy, own wait: same interval is fine if the wait came first
                self.err(f"R2: row {row} normalises {region} in interval {interval}, its copy retired in {at}")
            elif at == interval:
                self.err(f"R2: row {row} normalises {region} in the interval of the retiring wait ({interval}): the wait is at the END of L3")
        self.landed_before(tab_region, interval, tab_tag, f"table read for {region}")     # R4 (read across waves)
        self.last_touch[tab_region] = max(self.last_touch.get(tab_region, -1), interval)
        self.transformed[(region, row)] = interval
        self.last_touch[region] = max(self.last_touch.get(region, -1), interval)
        self.n_checked += 1


def program(tiles, NP):
    """The per-row slot sequence as (slot kind, [actions]) - a transcription of the kernel's control flow.  Actions:
    ("issue", op, region, tag) / ("wait", N key) / ("read", region, tag, what, need_transform) /
    ("transform", plane region piece, table region, tag, table tag)."""
    slots = []
    nk = 9 * NP
    S = tiles * nk

    def ktag(u):                # weights of K-tile u (clamped past the end exactly like advance_b: the last tile again)
        t, kt = divmod(u, nk)
        if t >= tiles:
            t, kt = tiles - 1, (u - tiles * nk) % nk
        return ("B", t, kt)

    # prologue (one slot, then the barrier that starts the first K-tile)
    pro = [("issue", "T", "tab0", ("T", 0))]
    pro += [("issue", f"H{q}", f"pl0.{q}", ("P", 0, 0)) for q in range(3)]
    pro += [("issue", f"B{q}", f"b0.{q}", ktag(0)) for q in range(5)]
    pro += [("issue", f"B{q}", f"b1.{q}", ktag(1)) for q in range(5)]
    pro += [("wait", "pro0")]
    slots.append(("P0", pro))                                   # c3_wait_vm<10>; barrier
    slots.append(("P1", [("transform", f"pl0.{q}", "tab0", ("P", 0, 0), ("T", 0)) for q in range(3)] + [("wait", "pro1")]))
    u = 0
    for t in range(tiles):
        tab_cur = t & 1
        nxt = t + 1 if t + 1 < tiles else t
        for pp in range(NP):
            for v in range(9):
                b = u & 1
                inflight = None
                for ph in range(4):
                    kk, hf = ph >> 1, ph & 1
                    ts = 2 * v + kk
                    pl = 1 if ts >= 9 else 0
                    chunk = 2 * pp + pl
                    acts = []
                    if hf == 0:
                        acts += [("read", f"b{b}.{q}", ktag(u), f"B fragments of K-tile {u}", False) for q in range(5)]
                    acts += [("read", f"pl{pl}.{q}", ("P", t, chunk), f"A fragments of K-tile {u} kk {kk}", True) for q in range(3)]
                    if ph == 0 and v == 0:
                        acts.append(("issue", "T", f"tab{tab_cur ^ 1}", ("T", nxt)))
                        acts += [("issue", f"H{q}", f"pl1.{q}", ("P", t, 2 * pp + 1)) for q in (0, 1)]
                    if ph == 0 and v == 5:
                        same = pp + 1 < NP
                        tag = ("P", t, 2 * pp + 2) if same else ("P", nxt, 0)
                        acts += [("issue", f"H{q}", f"pl0.{q}", tag) for q in (0, 1)]
                    if ph == 3:
                        acts += [("issue", f"B{q}", f"b{b}.{q}", ktag(u + 2)) for q in range(5)]
                        if v == 0:
                            acts.append(("issue", "H2", "pl1.2", ("P", t, 2 * pp + 1)))
                            acts.append(("wait", "odd"))
                        elif v == 5:
                            same = pp + 1 < NP
                            acts.append(("issue", "H2", "pl0.2", ("P", t, 2 * pp + 2) if same else ("P", nxt, 0)))
                            acts.append(("wait", "even"))
                        elif v in (1, 6):
                            acts.append(("wait", "after"))
                        else:
                            acts.append(("wait", "plain"))
                    # normalisation of the plane in flight: odd chunk of this period at v = 2, 2, 3; even chunk of the next
                    # period (of this tile, or chunk 0 of the next tile with the NEXT tile's table) at v = 7, 7, 8
                    odd_tag, odd_tab = ("P", t, 2 * pp + 1), (f"tab{tab_cur}", ("T", t))
                    same = pp + 1 < NP
                    ev_tag = ("P", t, 2 * pp + 2) if same else ("P", nxt, 0)
                    ev_tab = (f"tab{tab_cur}", ("T", t)) if same else (f"tab{tab_cur ^ 1}", ("T", nxt))
                    if ph == 1 and v == 2:
                        acts.append(("transform", "pl1.0", odd_tab[0], odd_tag, odd_tab[1]))
                    if ph == 2 and v == 2:
                        acts.append(("transform", "pl1.1", odd_tab[0], odd_tag, odd_tab[1]))
                    if ph == 1 and v == 3:
                        acts.append(("transform", "pl1.2", odd_tab[0], odd_tag, odd_tab[1]))
                    if ph == 1 and v == 7:
                        acts.append(("transform", "pl0.0", ev_tab[0], ev_tag, ev_tab[1]))
                    if ph == 2 and v == 7:
                        acts.append(("transform", "pl0.1", ev_tab[0], ev_tag, ev_tab[1]))
                    if ph == 1 and v == 8:
                        acts.append(("transform", "pl0.2", ev_tab[0], ev_tag, ev_tab[1]))
                    slots.append((f"L({u},{ph})", acts))
                    slots.append((f"M({u},{ph})", []))
                u += 1
    assert u == S
    return slots


def run(prologue_waits, imm, tiles=2, NP=2, verbose=False):
    chk = Check(prologue_waits, imm)
    slots = program(tiles, NP)
    nimm = dict(imm, pro0=prologue_waits[0], pro1=prologue_waits[1])
    def execute(row, interval, name, acts):
        # (the wait of an L slot sits at its END, after the slot's reads, copies and normalisation: list order)
        for a in acts:
            if a[0] == "issue":
                chk.issue(row, interval, a[1], a[2], a[3])
            elif a[0] == "read":
                chk.read(row, interval, a[1], a[2], f"row {row} {name}: {a[3]}", a[4])
            elif a[0] == "transform":
                chk.transform(row, interval, a[1], a[2], a[3], a[4])
            elif a[0] == "wait":
                chk.wait(row, interval, nimm[a[1]])
    # the prologue runs un-staggered (both rows in the same two intervals); then row 1 takes one extra barrier
    for s in (0, 1):
        for row in (0, 1):
            execute(row, s, *slots[s])
    main_slots = slots[2:]
    for k in range(len(main_slots) + 1):
        for row in (0, 1):
            s = k - chk.rows[row].lag
            if 0 <= s < len(main_slots):
                execute(row, 2 + k, *main_slots[s])
    if verbose:
        print(f"{len(slots)} slots per row, {chk.n_checked} accesses checked, {len(chk.errors)} violations")
    return chk.errors


def main():
    pw, imm = kernel_immediates()
    brk = None
    if "--break" in sys.argv:
        brk = sys.argv[sys.argv.index("--break") + 1]
        if brk in imm:
            imm[brk] += 1
        else:
            pw[int(brk)] += 1
    errs = run(pw, imm, verbose=True)
    for e in errs[:12]:
        print("  ", e)
    print("conv3 schedule:", "OK" if not errs else f"{len(errs)} VIOLATIONS", f"(immediates {pw} {imm}" + (f", {brk} perturbed)" if brk else ")"))
    sys.exit(1 if errs else 0)


if __name__ == "__main__":
    main()

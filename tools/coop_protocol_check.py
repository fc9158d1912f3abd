"""Exhaustive interleaving check of the rendezvous protocol of the persistent kernel's cooperative two-way K split
(gemm_ring_kernel<..., SK> in v-express_amd/csrc/vx_gemm_ring.hip, vx_gemm_params.ring_hint = 2) - CPU only.

Protocol of ABI 14 (epochs; ADVICE r05: with flag words that every launch had to leave at zero, ONE stale word - a reader that
gave up on its bounded poll, an aborted launch - made every later launch add accumulators that were not written yet).
One (tile, wave) slot of the workspace = an exchange area `ws` and two words `owner`, `ready`; a launch carries an epoch e
(the caller's per-workspace launch counter, >= 1; the workspace starts zeroed).  Two partner waves, one per K half, on
different CUs; each runs, in program order:

    a = exchange(owner, e)                                 agent-scope atomic
    if a != e:   (first of THIS launch to arrive)          if a == e:   (second)
        store ws[0..N)   (write-through, asynchronous)         poll ready until == e      (bounded in the kernel)
        s_waitcnt vmcnt(0)   (all stores performed)            load ws[0..N)   -> must be the partner's values, complete
        ready = e                                              -> epilogue with own + partner
        -> next work item, never waits

Nothing is ever reset.  Stores are modelled as a per-wave buffer that drains in ANY order at ANY time until the wait; the
exchange, the flag store and the polling load act on memory directly (agent scope).  Every interleaving of the two waves (and
of the drain events) is explored for THREE launches in a row on the same slot (stream-ordered launches reuse the
workspace), from EVERY start state an earlier aborted launch can leave behind (owner, ready each zero or a stale epoch, the
exchange area holding stale data):
  * the second wave only ever reads complete partner data of its own launch,
  * exactly one of the two runs the epilogue, the other never blocks.
`--break N` removes one ingredient (1: no wait before the flag store, 2: the flag is a constant instead of the epoch - the
ABI 13 protocol without its reset, 3: flag stored before the data) and must make the check fail - the check checks something.
    python tools/coop_protocol_check.py [--break N]"""
import itertools
import sys

N = 3            # data words per wave (stands for the 40 x 16-byte stores of a wave's accumulators)
LAUNCHES = 3
FIRST_EPOCH = 5  # epochs of the modelled launches: 5, 6, 7; stale start values: 0 (zeroed) or 3 (an older launch)


def programs(brk):
    """Instruction lists per role; a wave's role is decided by its exchange result at run time."""
    first = [("store", i) for i in range(N)] + ([] if brk == 1 else [("wait",)]) + [("set_ready",)]
    if brk == 3:
        first = [("set_ready",)] + [("store", i) for i in range(N)] + [("wait",)]
    second = [("poll",)] + [("load", i) for i in range(N)] + [("epilogue",)]
    return first, second


def explore(brk=0):
    first_prog, second_prog = programs(brk)
    seen, violations = set(), []

    def flag_of(epoch):
        return 1 if brk == 2 else epoch

    # state: (launch, mem_ws tuple, owner, ready, waves) ; wave = (stage, role, pc, buffer frozenset, loaded tuple)
    def start_launch(launch, ws, owner, ready):
        return (launch, ws, owner, ready, (("arrive", None, 0, frozenset(), ()), ("arrive", None, 0, frozenset(), ())))

    def step(state):
        launch, ws, owner, ready, waves = state
        epoch = FIRST_EPOCH + launch
        nxt = []
        if all(w[0] == "done" for w in waves):
            epis = sum(1 for w in waves if w[1] == "second")
            if epis != 1:
                violations.append(("epilogue count", launch, epis))
            if launch + 1 < LAUNCHES:
                nxt.append(start_launch(launch + 1, ws, owner, ready))
            return nxt
        for wi, w in enumerate(waves):
            stage, role, pc, buf, loaded = w
            tag = (launch, wi)                                  # the value this wave writes: identifies (launch, wave)

            def put(new_w, new_ws=ws, new_owner=owner, new_ready=ready):
                ws2 = list(waves)
                ws2[wi] = new_w
                nxt.append((launch, new_ws, new_owner, new_ready, tuple(ws2)))

            for item in buf:                                    # asynchronous drain of one buffered store
                mem = list(ws)
                mem[item] = tag
                put((stage, role, pc, buf - {item}, loaded), new_ws=tuple(mem))
            if stage == "done":
                continue
            if stage == "arrive":
                r = "second" if owner == epoch else "first"
                put(("run", r, 0, buf, loaded), new_owner=epoch)
                continue
            prog = first_prog if role == "first" else second_prog
            if pc >= len(prog):
                if not buf:
                    put(("done", role, pc, buf, loaded))
                continue
            ins = prog[pc]
            if ins[0] == "store":
                put((stage, role, pc + 1, buf | {ins[1]}, loaded))
            elif ins[0] == "wait":
                if not buf:
                    put((stage, role, pc + 1, buf, loaded))
            elif ins[0] == "set_ready":
                put((stage, role, pc + 1, buf, loaded), new_ready=flag_of(epoch))
            elif ins[0] == "poll":
                if ready == flag_of(epoch):
                    put((stage, role, pc + 1, buf, loaded))
                # else: spins (no state change) - progress comes from the partner, which never waits
            elif ins[0] == "load":
                put((stage, role, pc + 1, buf, loaded + (ws[ins[1]],)))
            elif ins[0] == "epilogue":
                partner = (launch, 1 - wi)
                if any(v != partner for v in loaded) or len(loaded) != N:
                    violations.append(("second wave read incomplete / stale partner data", launch, loaded))
                put((stage, role, pc + 1, buf, loaded))
        return nxt

    stale = FIRST_EPOCH - 2
    # what an aborted earlier launch can leave: any mix of zero / stale words, the exchange area holding its data
    starts = [start_launch(0, tuple([("stale", 0)] * N), o, flag_of(r) if r else 0)
              for o, r in itertools.product((0, stale), (0, stale))]
    stack = list(starts)
    while stack:
        st = stack.pop()
        if st in seen:
            continue
        seen.add(st)
        succ = step(st)
        if not succ and not (all(w[0] == "done" for w in st[4]) and st[0] == LAUNCHES - 1):
            violations.append(("no progress possible", st[0]))
        stack.extend(succ)
    return len(seen), violations


def main():
    brk = int(sys.argv[sys.argv.index("--break") + 1]) if "--break" in sys.argv else 0
    states, violations = explore(brk)
    kinds = sorted({v[0] for v in violations})
    print(f"states explored: {states}; violations: {len(violations)} {kinds}")
    print("OK" if not violations else "FAILED")
    sys.exit(0 if not violations else 1)


if __name__ == "__main__":
    main()

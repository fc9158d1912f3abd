"""Exhaustive interleaving check of the rendezvous protocol of the persistent kernel's cooperative two-way K split
(gemm_ring_kernel<..., SK> in v-express_amd/csrc/vx_gemm_ring.hip, vx_gemm_params.ring_hint = 2) - CPU only.

Model (one (tile, wave) slot of the workspace = an exchange area `ws`, two flag words `cnt`, `ready`; two partner waves, one
per K half, on different CUs).  Each wave runs, in program order:

    a = fetch_add(cnt, 1)                                  agent-scope atomic
    if a == 0:   (first to arrive)                         if a == 1:   (second)
        store ws[0..N)   (write-through, asynchronous)         poll ready until != 0
        s_waitcnt vmcnt(0)   (all stores performed)            cnt = 0; ready = 0    (slot clean for the next launch)
        ready = 1 + xcc                                        load ws[0..N)         -> must be the partner's values, complete
        -> next work item, never waits                         -> epilogue with own + partner

Stores are modelled as a per-wave buffer that drains in ANY order at ANY time until the wait; flag operations and the
polling load act on memory directly (agent scope).  Every interleaving of the two waves (and of the drain events) is
explored, for both arrival orders, twice in a row on the same slot (stream-ordered launches reuse the workspace):
  * the second wave only ever reads complete partner data,
  * exactly one of the two runs the epilogue, the other never blocks,
  * the slot ends with cnt == ready == 0.
`--break N` removes one ingredient (1: no wait before the flag store, 2: no flag reset, 3: flag stored before the data) and
must make the check fail - the check checks something.
    python tools/coop_protocol_check.py [--break N]"""
import sys

N = 3            # data words per wave (stands for the 40 x 16-byte stores of a wave's accumulators)


def programs(brk):
    """Instruction lists per role; a wave's role is decided by its fetch_add result at run time."""
    first = [("store", i) for i in range(N)] + ([] if brk == 1 else [("wait",)]) + [("set_ready",)]
    if brk == 3:
        first = [("set_ready",)] + [("store", i) for i in range(N)] + [("wait",)]
    # (the kernel zeroes the flag words right after the poll and loads the partner's values later, in its epilogue walk)
    second = [("poll",)] + ([] if brk == 2 else [("reset",)]) + [("load", i) for i in range(N)] + [("epilogue",)]
    return first, second


def explore(brk=0):
    first_prog, second_prog = programs(brk)
    sys.setrecursionlimit(100000)
    seen, violations, finals = set(), [], set()

    # state: (launch, mem_ws tuple, cnt, ready, waves) ; wave = (pc_stage, role, pc, buffer frozenset, loaded tuple)
    def start_launch(launch, ws, cnt, ready):
        return (launch, ws, cnt, ready, (("arrive", None, 0, frozenset(), ()), ("arrive", None, 0, frozenset(), ())))

    def step(state):
        launch, ws, cnt, ready, waves = state
        nxt = []
        done = all(w[0] == "done" for w in waves)
        if done:
            if cnt != 0 or ready != 0:
                violations.append(("slot not clean after launch", launch, cnt, ready))
            epis = sum(1 for w in waves if w[1] == "second")
            if epis != 1:
                violations.append(("epilogue count", launch, epis))
            if launch == 0:
                nxt.append(start_launch(1, ws, cnt, ready))
            else:
                finals.add((cnt, ready))
            return nxt
        for wi, w in enumerate(waves):
            stage, role, pc, buf, loaded = w
            tag = (launch, wi)                                  # the value this wave writes: identifies (launch, wave)

            def put(new_w, new_ws=ws, new_cnt=cnt, new_ready=ready):
                ws2 = list(waves)
                ws2[wi] = new_w
                nxt.append((launch, new_ws, new_cnt, new_ready, tuple(ws2)))

            # asynchronous drain of one buffered store (any order, any time)
            for item in buf:
                idx = item
                mem = list(ws)
                mem[idx] = tag
                put((stage, role, pc, buf - {item}, loaded), new_ws=tuple(mem))
            if stage == "done":
                continue
            if stage == "arrive":
                a = cnt
                r = "first" if a == 0 else "second"
                if a > 1:
                    violations.append(("third arrival", launch))
                put(("run", r, 0, buf, loaded), new_cnt=cnt + 1)
                continue
            prog = first_prog if role == "first" else second_prog
            if pc >= len(prog):
                # (a first-arrival wave leaves with stores possibly still draining only if the wait was removed)
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
                put((stage, role, pc + 1, buf, loaded), new_ready=1)
            elif ins[0] == "poll":
                if ready != 0:
                    put((stage, role, pc + 1, buf, loaded))
                # else: spins (no state change) - progress comes from the partner, which never waits
            elif ins[0] == "load":
                put((stage, role, pc + 1, buf, loaded + (ws[ins[1]],)))
            elif ins[0] == "reset":
                put((stage, role, pc + 1, buf, loaded), new_cnt=0, new_ready=0)
            elif ins[0] == "epilogue":
                partner = (launch, 1 - wi)
                if any(v != partner for v in loaded) or len(loaded) != N:
                    violations.append(("second wave read incomplete / stale partner data", launch, loaded))
                put((stage, role, pc + 1, buf, loaded))
        return nxt

    init = start_launch(0, tuple([None] * N), 0, 0)
    stack = [init]
    deadlocks = 0
    while stack:
        st = stack.pop()
        if st in seen:
            continue
        seen.add(st)
        succ = step(st)
        if not succ and not all(w[0] == "done" for w in st[4]):
            deadlocks += 1
            violations.append(("no progress possible", st[0]))
        elif not succ and st[0] == 0:
            pass
        stack.extend(succ)
    return len(seen), violations, finals, deadlocks


def main():
    brk = int(sys.argv[sys.argv.index("--break") + 1]) if "--break" in sys.argv else 0
    states, violations, finals, deadlocks = explore(brk)
    kinds = sorted({v[0] for v in violations})
    print(f"states explored: {states}; end states of the slot (cnt, ready): {sorted(finals)}; violations: {len(violations)} {kinds}")
    ok = not violations and finals == {(0, 0)}
    print("OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

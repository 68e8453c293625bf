"""Slot-level model of the resident panel kernel on a tall (sub-)panel: which ticket order / task split keeps 512 workgroup slots busy?

A CPU-only what-if tool (no GPU): 256 compute units x 2 slots; a workgroup takes the next ticket when a slot frees and holds the slot
until its task is done (waiting included).  Task (r, k), r > k: k block products (K = 128 each; a compute unit delivers one per
PROD us, shared between the product-phase workgroups on it), each one runnable once block column j is published on rows r and k; then
the solve behind leaf k (SOLVE us alone, SOLVE2 beside a co-resident product stream).  Diagonal task k: rank-16 updates behind its
row's solves, then the leaf (LEAF us alone, LEAF2 when it shares the unit with a product stream that does not yield).
Calibration targets (profiles/r05 + DESIGN section 12): N=30000 first sub-panel (235 x 16 blocks) 2.7 ms with a leaf every ~170 us;
N=50000 (391 x 16) 4.97 ms.  The model is OPTIMISTIC (2.07 ms for the 235 x 16 case: it has no slot turnover, no per-call overheads and a
solve that never competes for the MFMA pipe) but it got the one thing it was built for right before the GPU did: the ticket order
does not matter for a tall panel (col / sq / crit orders within 2 %), the launch is bound by throughput and by the drain behind the
last column.  The measured counterpart is tools/chain_occupancy.py (profiles/r06_chain_occupancy_n30000.txt).

  python tools/chain_sim.py R n [policy]      policy: col (column-major, shipped), crit<c> (near-diagonal blocks dealt ahead), sq (square first)
"""
import sys
import heapq

PROD, SOLVE, SOLVE2, LEAF, LEAF2, SYRK = 15.5, 12.0, 45.0, 24.5, 42.0, 3.0


def simulate(R, n, policy="col", slots_per_cu=2, cus=256, crit=2, dt=1.0, verbose=False):
    # ---- task lists
    diag = [("d", k, k) for k in range(n)]
    def blocks(k):
        return [("b", r, k) for r in range(k + 1, R)]
    if policy == "col":
        order = []
        for k in range(n):
            order.append(diag[k]); order += blocks(k)
        queues = [order]
    elif policy == "sq":
        order = []
        for k in range(n):
            order.append(diag[k]); order += [("b", r, k) for r in range(k + 1, n)]
        for k in range(n):
            order += [("b", r, k) for r in range(n, R)]
        queues = [order]
    elif policy.startswith("crit"):
        c = int(policy[4:] or crit)
        cq, bq = [], []
        for k in range(n):
            cq.append(diag[k]); cq += [("b", r, k) for r in range(k + 1, min(k + 1 + c, R))]
            bq += [("b", r, k) for r in range(k + 1 + c, R)]
        queues = [cq, bq]
    else:
        raise SystemExit("policy?")
    pub = {}                      # (r, k) -> time block (r, k) is published (solved + stored)
    leaf_done = {}
    started = set()
    qi = [0] * len(queues)
    nslots = cus * slots_per_cu
    slot_task = [None] * nslots   # running task state per slot
    t = 0.0
    done = 0
    total = sum(len(q) for q in queues)
    busy_prod = 0.0               # slot-us in product phase
    idle = wait = solve_t = 0.0
    leaf_times = []

    def next_task():
        if len(queues) == 1:
            q = queues[0]
            if qi[0] < len(q):
                qi[0] += 1
                return q[qi[0] - 1]
            return None
        cq, bq = queues
        # bulk frontier column: column of the next bulk task (all bulk tasks of earlier columns have started)
        fcol = bq[qi[1]][2] if qi[1] < len(bq) else 10 ** 9
        if qi[0] < len(cq) and cq[qi[0]][2] <= fcol + 1:
            qi[0] += 1
            return cq[qi[0] - 1]
        if qi[1] < len(bq):
            qi[1] += 1
            return bq[qi[1] - 1]
        if qi[0] < len(cq):
            qi[0] += 1
            return cq[qi[0] - 1]
        return None

    while done < total:
        # fill free slots
        for s in range(nslots):
            if slot_task[s] is None:
                tk = next_task()
                if tk is None:
                    break
                kind, r, k = tk
                slot_task[s] = {"kind": kind, "r": r, "k": k, "j": 0, "rem": 0.0, "phase": "prod" if kind == "b" else "syrk", "t_end": None}
        # per CU: count product-phase tasks that are runnable now
        for cu in range(cus):
            sl = [slot_task[cu * slots_per_cu + i] for i in range(slots_per_cu)]
            runnable = []
            for st in sl:
                if st is None:
                    idle += dt
                    continue
                if st["kind"] == "b" and st["phase"] == "prod":
                    r, k = st["r"], st["k"]
                    if st["j"] >= k and st["rem"] <= 0:
                        st["phase"] = "solve"; st["t_end"] = None
                    else:
                        if st["rem"] <= 0:
                            j = st["j"]
                            if pub.get((r, j), 1e18) <= t and (j == k or pub.get((k, j), 1e18) <= t or k == j):
                                st["rem"] = 1.0; st["j"] = j + 1
                        if st["rem"] > 0:
                            runnable.append(st)
                        else:
                            wait += dt
            share = 1.0 / max(len(runnable), 1)
            for st in runnable:
                st["rem"] -= dt * share / PROD
                busy_prod += dt
            nprod = len(runnable)
            for st in sl:
                if st is None:
                    continue
                if st["kind"] == "b" and st["phase"] == "solve":
                    k = st["k"]
                    if st["t_end"] is None:
                        if leaf_done.get(k, 1e18) <= t + 8.0:          # the solve follows the leaf: ends ~8 us after it at the earliest
                            dur = SOLVE2 if nprod > 0 else SOLVE
                            st["t_end"] = max(t + dur, leaf_done[k] + 8.0)
                        else:
                            wait += dt
                    if st["t_end"] is not None:
                        solve_t += dt
                        if t >= st["t_end"]:
                            pub[(st["r"], k)] = t
                            st["done"] = True
                elif st["kind"] == "d":
                    k = st["k"]
                    if st["phase"] == "syrk":
                        if all(pub.get((k, j), 1e18) <= t for j in range(k)):
                            st["phase"] = "leaf"; st["t_end"] = t + (SYRK if k > 0 else 0.0) + (LEAF2 if nprod > 0 else LEAF)
                        else:
                            wait += dt
                    elif t >= st["t_end"]:
                        leaf_done[k] = t; leaf_times.append(t); st["done"] = True
        for s in range(nslots):
            st = slot_task[s]
            if st is not None and st.get("done"):
                slot_task[s] = None
                done += 1
        t += dt
        if t > 1e5:
            raise SystemExit("stuck")
    tot = t * nslots
    return {"span_us": t, "leaves": leaf_times, "prod_frac": busy_prod / tot, "idle_frac": idle / tot, "wait_frac": wait / tot, "solve_frac": solve_t / tot}


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 235
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    pols = sys.argv[3:] or ["col", "sq", "crit2", "crit4"]
    ideal = sum((R - k - 1) * k for k in range(n)) * 13.65 / 256
    print(f"R={R} n={n}: flop time on 256 CUs {ideal:.0f} us")
    for p in pols:
        r = simulate(R, n, p)
        lv = r["leaves"]
        cad = (lv[-1] - lv[0]) / max(len(lv) - 1, 1)
        print(f"  {p:6s}: span {r['span_us']:7.0f} us  products {r['prod_frac']:.2f} solves {r['solve_frac']:.2f} waiting {r['wait_frac']:.2f} empty {r['idle_frac']:.2f}   leaf cadence {cad:5.0f} us, last leaf {lv[-1]:6.0f}")

"""Model check of the persistent GSIP kernel's queue protocol (k_gsip) -- CPU only, no HIP.

The kernel's correctness argument has a sequential part (the per-point arithmetic, shared with the launch chain and
covered by the bit-identity tests on the GPU) and a concurrent part: tagged ring entries, ticket claims that may run
ahead of the reservations, per-point `pending` counters, finished points handed in when a wave runs dry, the stop
word.  This test runs that protocol, restated step by step (one shared-memory access per step), under many random
interleavings of the waves and checks what the kernel relies on: every task is consumed exactly once, no ring slot is
overwritten before it was taken (rings sized by the kernel's rule, so they wrap many times), every point finishes,
every wave terminates, and the finished count is complete when the stop word goes up.  It models the protocol, not the HIP code."""
import random

import pytest

SOLVE, ROUND = 0, 1


class Shard:
    def __init__(self, qs):
        self.head = 0
        self.reserve = 0
        self.stop = 0
        self.ring = [0] * qs            # 0 = free, else (pos + 1, payload)


def wave(w, st, rng):
    """One persistent wave as a generator: every `yield` is a point where other waves may run."""
    nq, qs = st["nq"], st["qs"]
    home = st["shards"][w % nq]
    rr = w * 40503
    my_done = 0
    c0 = w                               # static chunks of the initial list
    while True:
        from_list = c0 * 16 < len(st["list0"])
        if from_list:
            t0, tn = c0 * 16, 16
            c0 += st["n_waves"]
        else:
            r = home.reserve; yield
            h = home.head; yield
            backlog = r - h
            tn = 16 if backlog >= 16 else 8 if backlog >= 8 else 4 if backlog >= 4 else 2
            t0 = home.head; home.head += tn; yield          # atomicAdd
        todo = set(range(tn))
        if from_list:
            todo = {e for e in todo if t0 + e < len(st["list0"])}
        held = {}
        polls = waited = 0
        quit_ = False
        while todo:
            for e in sorted(todo - set(held)):
                pos = t0 + e
                if from_list:
                    held[e] = st["list0"][pos]
                else:
                    v = home.ring[pos % qs]; yield
                    if v and v[0] == pos + 1:
                        held[e] = v[1]
                        home.ring[pos % qs] = 0; yield      # taken
            if held and tn > 2 and waited < st["grace"] and len(held) < len(todo):
                waited += 1; yield
                continue
            if not held:
                polls += 1
                if polls == 1 and my_done:
                    before = st["done"]; st["done"] += my_done; yield   # atomicAdd
                    if before + my_done >= st["n_int"]:
                        assert st["done"] == st["n_int"]
                        for s in st["shards"]:
                            s.stop = 1
                        my_done = 0
                        quit_ = True
                        break
                    my_done = 0
                if home.stop:
                    quit_ = True
                    break
                assert polls < 200000, "poll cap: the model deadlocked"
                yield
                continue
            polls = 0
            got, held = held, {}
            todo -= set(got)
            last_points = []
            for e, task in got.items():                     # SOLVE entries
                kind, a = task
                st["consumed"][task_key(task, st)] += 1
                if kind == SOLVE:
                    pt = st["points"][a[0]]
                    old = pt["pending"]; pt["pending"] = old - 1; yield   # atomicSub returns the old value
                    if old == 1:
                        last_points.append(a[0])
            if last_points:
                ts = st["shards"][rr % nq]; rr += 1
                base = ts.reserve; ts.reserve += len(last_points); yield
                for i, p in enumerate(last_points):
                    yield from put(ts, qs, base + i, (ROUND, (p, st["points"][p]["round"])), st)
            rounds = [task for task in got.values() if task[0] == ROUND]
            for i in range(0, len(rounds), 2):              # ROUND entries, two at a time
                push = []
                for _, (p, _r) in rounds[i:i + 2]:
                    pt = st["points"][p]
                    pt["round"] += 1
                    if pt["round"] > pt["n_rounds"]:
                        my_done += 1
                        pt["finished"] += 1
                    else:
                        k = rng.randint(1, 5)
                        pt["pending"] = k; yield
                        push += [(SOLVE, (p, pt["round"], j)) for j in range(k)]
                if push:
                    ts = st["shards"][rr % nq]; rr += 1
                    base = ts.reserve; ts.reserve += len(push); yield
                    for j, t in enumerate(push):
                        yield from put(ts, qs, base + j, t, st)
        if quit_:
            return


def task_key(task, st):
    st["consumed"].setdefault(task, 0)
    return task


def put(shard, qs, pos, task, st):
    spins = 0
    while shard.ring[pos % qs] != 0:                        # previous lap not taken yet
        spins += 1
        assert spins < 200000, "ring full forever"
        yield
    shard.ring[pos % qs] = (pos + 1, task)
    st["produced"][task] = st["produced"].get(task, 0) + 1
    yield


@pytest.mark.parametrize("seed", range(40))
def test_queue_protocol_random_interleavings(seed):
    rng = random.Random(seed)
    n_points = rng.choice([1, 3, 17, 60, 150])
    n_waves = rng.choice([1, 2, 5, 16])
    nq = rng.choice([1, 2, 4])
    nq = min(nq, n_waves)
    # the kernel's sizing rule with the model's numbers: twice an even share of the tasks that can be outstanding
    # (here 5 per point, 24 in the kernel) plus a slack no single push exceeds (here 12 >= 2 x 5, 2048 >= 48 there)
    qs = (2 * 5 * n_points) // nq + 12
    points = [dict(round=1, n_rounds=rng.randint(1, 4), pending=0, finished=0) for _ in range(n_points)]
    list0 = []
    for p, pt in enumerate(points):                         # k_round(0): round 1 open, its solves in the initial list
        k = rng.randint(1, 3)
        pt["pending"] = k
        list0 += [(SOLVE, (p, 1, j)) for j in range(k)]
    st = dict(nq=nq, qs=qs, shards=[Shard(qs) for _ in range(nq)], list0=list0, n_waves=n_waves, grace=3, done=0,
              n_int=n_points, points=points, consumed={}, produced={t: 1 for t in list0})
    waves = [wave(w, st, rng) for w in range(n_waves)]
    alive = list(range(n_waves))
    steps = 0
    while alive:
        w = rng.choice(alive)
        try:
            next(waves[w])
        except StopIteration:
            alive.remove(w)
        steps += 1
        assert steps < 5_000_000
    assert st["done"] == n_points and all(pt["finished"] == 1 for pt in points)
    assert st["consumed"] == st["produced"]                 # every task exactly once
    assert all(v == 1 for v in st["consumed"].values())
    assert all(slot == 0 for s in st["shards"] for slot in s.ring)   # the rings are left zeroed

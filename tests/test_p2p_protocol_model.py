"""Model of the peer-mailbox exchange protocol (csrc/p2p.hip) under adversarial interleavings: each rank is a thread that
performs the kernel's four phases (write own vector into every mailbox -> publish flags -> wait for all flags in its own
mailbox -> sum) for a sequence of steps, with random delays everywhere.  The property: every rank always sums exactly the
vectors of the CURRENT generation - also with ONE exchange per step, where only the two-parity layout keeps a fast peer
that is already in the next step from overwriting an entry a slow rank has not read yet (running the harness with
parities=1 shows exactly that overwrite within a few seeds; not asserted here because it needs a lucky interleaving)."""
import random
import threading
import time


def _run(world, steps, slots, parities, seed):
    rnd = random.Random(seed)
    delays = [[rnd.random() * 2e-4 for _ in range(8)] for _ in range(world)]
    # mailbox[r]["data"][parity][slot][src], mailbox[r]["flag"][parity][slot][src]
    box = [{"data": [[[None] * world for _ in range(slots)] for _ in range(parities)],
            "flag": [[[0] * world for _ in range(slots)] for _ in range(parities)]} for _ in range(world)]
    bad = []

    def rank_main(r):
        lr = random.Random(seed * 131 + r)
        for step in range(steps):
            gen = step + 1
            par = gen % parities
            for slot in range(slots):
                mine = (r, step, slot)                                     # the "vector"
                for q in range(world):                                     # phase 1
                    box[q]["data"][par][slot][r] = mine
                    if lr.random() < 0.3:
                        time.sleep(delays[r][lr.randrange(8)])
                for q in range(world):                                     # phase 2 (after the fence)
                    box[q]["flag"][par][slot][r] = gen
                t0 = time.time()
                while any(box[r]["flag"][par][slot][q] != gen for q in range(world)):      # phase 3
                    if time.time() - t0 > 20:
                        bad.append(("timeout", r, step, slot))
                        return
                    time.sleep(0)
                if lr.random() < 0.5:
                    time.sleep(delays[r][lr.randrange(8)] * 3)             # a slow reader
                got = [box[r]["data"][par][slot][q] for q in range(world)]  # phase 4
                want = [(q, step, slot) for q in range(world)]
                if got != want:
                    bad.append((r, step, slot, got))

    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return bad


def test_two_parities_make_single_slot_reuse_safe():
    for seed in range(6):
        assert _run(world=3, steps=60, slots=1, parities=2, seed=seed) == []
    assert _run(world=4, steps=30, slots=3, parities=2, seed=11) == []

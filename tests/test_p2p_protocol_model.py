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


def _run_ll(world, steps, slots, blocks, parities, seed):
    """The LL-word form (csrc/p2p_ll.h, used inside the BatchNorm kernels): no flags and no block-wide phases - every BLOCK of a
    rank's kernel sends the words it owns ({value, generation} in one store) into every mailbox and polls the peers' words of the
    same indices, independently of the kernel's other blocks; a kernel (= one exchange `slot` of a step) ends when all its blocks
    have, and the next kernel of the rank starts only then (stream order).  Blocks of one kernel run concurrently (threads here)."""
    rnd = random.Random(seed)
    delays = [rnd.random() * 2e-4 for _ in range(16)]
    # word[r][parity][slot][src][block] = (value, gen)
    word = [[[[[(None, 0)] * blocks for _ in range(world)] for _ in range(slots)] for _ in range(parities)] for _ in range(world)]
    bad = []

    def block_main(r, step, slot, b, lr_seed):
        lr = random.Random(lr_seed)
        gen = step + 1
        par = gen % parities
        mine = (r, step, slot, b)
        for q in range(world):                                             # send: one store per mailbox
            word[q][par][slot][r][b] = (mine, gen)
            if lr.random() < 0.3:
                time.sleep(delays[lr.randrange(16)])
        got = []
        for q in range(world):                                             # poll rank q's word until it carries this generation
            t0 = time.time()
            while True:
                v, g = word[r][par][slot][q][b]
                if g == gen:
                    got.append(v)
                    break
                if time.time() - t0 > 20:
                    bad.append(("timeout", r, step, slot, b, q, g))
                    return
                time.sleep(0)
            if lr.random() < 0.2:
                time.sleep(delays[lr.randrange(16)] * 3)                   # a slow reader between two polls
        if got != [(q, step, slot, b) for q in range(world)]:
            bad.append((r, step, slot, b, got))

    def rank_main(r):
        for step in range(steps):
            for slot in range(slots):
                ths = [threading.Thread(target=block_main, args=(r, step, slot, b, seed * 977 + r * 131 + step * 17 + slot * 5 + b))
                       for b in range(blocks)]
                for t in ths:
                    t.start()
                for t in ths:                                              # the kernel ends when all of its blocks have
                    t.join()

    ranks = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ranks:
        t.start()
    for t in ranks:
        t.join()
    return bad


def test_ll_words_per_block_exchange_is_safe_with_two_parities():
    """every block always sums exactly the current generation's words, with ONE exchange per step (the hardest reuse pattern:
    generation g + 2 lands in the words generation g was read from) and with several; a rank that is a whole step ahead cannot
    overwrite a word a slow block of a peer has not read, because it cannot finish step g + 1 without that peer's step-(g + 1) words,
    which the peer's kernels send only after ALL blocks of its step-g kernel have ended"""
    for seed in range(4):
        assert _run_ll(world=3, steps=25, slots=1, blocks=3, parities=2, seed=seed) == []
    assert _run_ll(world=4, steps=12, slots=3, blocks=2, parities=2, seed=7) == []
    assert _run_ll(world=2, steps=40, slots=2, blocks=4, parities=2, seed=3) == []

"""XCD-aware block order of the weight-gradient launches (csrc/wgrad.hip: wg_logical_block), restated on the CPU: the map from
the physical block id (XCD = id & 7, round-robin dispatch) to the logical tile id must be a bijection for every grid size,
and each XCD must receive runs of WG_RUN consecutive tiles spread over the whole (sorted) problem list."""
WG_RUN = 8


def logical(pb, total):
    full = total // (8 * WG_RUN) * (8 * WG_RUN)
    if pb >= full:
        return pb
    xcd, j = pb & 7, pb >> 3
    return ((j // WG_RUN) * 8 + xcd) * WG_RUN + j % WG_RUN


def test_bijection_for_every_grid_size():
    for total in list(range(1, 300)) + [511, 512, 513, 2304, 4097]:
        got = sorted(logical(p, total) for p in range(total))
        assert got == list(range(total)), total


def test_runs_and_spread():
    total = 64 * 20
    per_xcd = {x: [logical(p, total) for p in range(total) if p & 7 == x] for x in range(8)}
    for x, ids in per_xcd.items():
        # consecutive blocks of an XCD walk runs of WG_RUN consecutive tiles
        for i in range(0, len(ids), WG_RUN):
            run = ids[i:i + WG_RUN]
            assert run == list(range(run[0], run[0] + WG_RUN))
        # and every XCD touches the first and the last eighth of the list (sorted problems: long and short reductions)
        assert min(ids) < total // 8 and max(ids) >= total - total // 8

"""Data parallelism for the native engine: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm,
over xGMI), mirroring the reference's DDP + SyncBatchNorm semantics (train.py:80-102):

  * SyncBN: per BatchNorm layer ONE all-reduce of [S1 | S2] (moments about the running mean) in forward and one of
    (sum g, sum g*xhat) in backward (engine.Engine uses `allreduce_sum` for both) - 142 small collectives per step
    instead of the reference's 71 all-gathers + 71 all-reduces; N-GPU training equals 1-GPU training on the
    concatenated batch;
  * gradients: the flat fp32 gradient arena is laid out stage by stage (stem+layer1, layer2, layer3, layer4+attnpool,
    text, neck, decoder, projector); as backward finishes a stage its range is all-reduced as ONE large message on a
    side HIP stream while the remaining backward still runs - projector, decoder, neck first, then the text encoder
    (from its own stream) and the visual layer groups 4..1 (xGMI is point-to-point and per-link bound: eight big
    messages of 24-254 MB, not DDP's 25 MB buckets); Adam applies the 1/world averaging through its grad_scale.
The path shards by sample only; there is no other data-path collective.
"""
import ctypes

import torch
import torch.distributed as dist


class PeerMailboxes:
    """The SyncBN exchange over peer-mapped mailboxes instead of an RCCL collective per BatchNorm layer (csrc/p2p.hip,
    csrc/p2p_ll.h, include/cris_hip.h cris_p2p_*): every rank allocates an uncached mailbox, the 64-byte IPC handles travel once
    through the communicator's host channel, every rank maps every peer's mailbox.  Construction happens in three steps so
    that a failure on ONE rank cannot leave the ranks in different collectives (the caller - TorchDistComm.enable_p2p - runs
    the same sequence of all-gathers on every rank whatever happens locally): `alloc_and_export()` (local, may raise) ->
    all-gather of (error, handle) -> `import_peers(handles)` (local, may raise) -> all-gather of errors -> barrier.
    `self_test()` runs a few exchanges with known data and a short poll limit, on both protocols (the flag-protocol kernel and
    the LL words the BatchNorm kernels use); the trainer switches to the mailboxes only when that passed on EVERY rank."""

    def __init__(self, rank, world, device, slots, max_floats):
        from . import hip
        self.hip, self.lib = hip, hip.load()
        self.rank, self.world, self.slots, self.max_floats, self.device = rank, world, slots, max_floats, device
        self.own, self.peers, self.boxes, self.err = None, [], None, None

    def alloc_and_export(self):
        """allocate this rank's mailbox; returns its 64-byte IPC handle"""
        nbytes = self.lib.cris_p2p_mailbox_bytes(self.world, self.slots, self.max_floats)
        own = ctypes.c_void_p()
        self.hip.check(self.lib.cris_p2p_alloc(nbytes, ctypes.byref(own)), "cris_p2p_alloc")
        self.own = own.value
        handle = (ctypes.c_ubyte * 64)()
        self.hip.check(self.lib.cris_p2p_export(self.own, handle), "cris_p2p_export")
        return bytes(handle)

    def import_peers(self, handles):
        ptrs = []
        for q, h in enumerate(handles):
            if q == self.rank:
                ptrs.append(self.own)
                continue
            peer = ctypes.c_void_p()
            buf = (ctypes.c_ubyte * 64).from_buffer_copy(h)
            self.hip.check(self.lib.cris_p2p_import(buf, ctypes.byref(peer)), "cris_p2p_import")
            self.peers.append(peer.value)
            ptrs.append(peer.value)
        self.boxes = torch.tensor(ptrs, dtype=torch.int64, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)

    def link(self, slot, gen_dev=None, gen_host=0, spin_limit=0):
        """the hip.P2PLink naming exchange `slot` of the step (what cris_bn_finalize_sync / cris_bn_bwd_reduce_sync take)"""
        assert 0 <= slot < self.slots
        l = self.hip.P2PLink()
        l.boxes, l.err = self.boxes.data_ptr(), self.err.data_ptr()
        l.gen_dev = None if gen_dev is None else gen_dev.data_ptr()
        l.gen_host, l.rank, l.world = gen_host, self.rank, self.world
        l.slot, l.slots, l.max_floats, l.spin_limit = slot, self.slots, self.max_floats, spin_limit
        return l

    def self_test(self, spin_limit=1 << 19):
        """three generations of one exchange with known data (rank r contributes r + 1 + i/7) on each protocol: True when every
        sum is right and no peer was reported missing.  The poll limit is short (a fraction of a second), so a mapping that
        does not propagate stores fails here instead of stalling the first training step."""
        n = min(self.max_floats, 1000)
        idx = torch.arange(n, device=self.boxes.device, dtype=torch.float32) / 7.0
        want = sum(float(q + 1) for q in range(self.world)) + self.world * idx
        ok = True
        for proto in ("flags", "ll"):
            for gen in range(3):
                t = (float(self.rank + 1) + idx + float(gen)).contiguous()
                if proto == "flags":
                    self.allreduce_sum(t, self.slots - 1, gen_host=1000 + gen, spin_limit=spin_limit)
                else:
                    self.ll_allreduce_sum(t, self.slots - 1, gen_host=1000 + gen, spin_limit=spin_limit)
                torch.cuda.synchronize()
                ok = ok and bool(torch.allclose(t, want + self.world * float(gen), rtol=0, atol=1e-3)) and int(self.err.item()) == 0
        self.err.zero_()
        return ok

    def allreduce_sum(self, t, slot, gen_dev=None, gen_host=0, spin_limit=0):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.max_floats and slot < self.slots
        prm = self.hip.P2PParams()
        prm.spin_limit = spin_limit
        prm.data, prm.boxes, prm.err = t.data_ptr(), self.boxes.data_ptr(), self.err.data_ptr()
        prm.gen_dev = None if gen_dev is None else gen_dev.data_ptr()
        prm.n, prm.rank, prm.world = t.numel(), self.rank, self.world
        prm.slot, prm.slots, prm.max_floats, prm.gen_host = slot, self.slots, self.max_floats, gen_host
        # not through hip.call: the caller (ops.torch_op) already puts this exchange on a command list being recorded
        self.hip.check(self.lib.cris_p2p_allreduce_sum(ctypes.byref(prm), torch.cuda.current_stream().cuda_stream),
                       "cris_p2p_allreduce_sum")

    def ll_allreduce_sum(self, t, slot, gen_dev=None, gen_host=0, spin_limit=0):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.max_floats and slot < self.slots
        l = self.link(slot, gen_dev=gen_dev, gen_host=gen_host, spin_limit=spin_limit)
        self.hip.check(self.lib.cris_p2p_ll_allreduce_sum(ctypes.byref(l), t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream),
                       "cris_p2p_ll_allreduce_sum")

    def close(self):
        for p in self.peers:
            self.lib.cris_p2p_close(p)
        self.peers = []
        if self.own:
            self.lib.cris_p2p_free(self.own)
            self.own = None


class TorchDistComm:
    def __init__(self, device=None):
        assert dist.is_initialized()
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.device = device
        self.side = torch.cuda.Stream(device=device) if (device is not None and torch.device(device).type == "cuda") else None
        self._pending = []
        # the gradient exchange gets its OWN communicator: collectives of one communicator execute in issue order, so on
        # the default group the tiny SyncBN all-reduces of the layers still in backward would queue behind 100+ MB
        # gradient messages and stall the compute stream
        self.grad_group = dist.new_group(ranks=list(range(self.world)))
        self.capturable = dist.get_backend() == "nccl"       # RCCL kernels can be captured into a HIP graph; gloo cannot
        self.p2p, self._gen_dev, self._slot, self._fused = None, None, 0, False
        self._arena, self._arena_ptrs, self._arena_slot0, self._arena_calls, self._arena_opened = None, None, 0, 0, []

    ARENA_STAGES = 10         # gradient-arena exchanges per step the mailbox reserves barrier slots for (three each; the engine has eight stages)

    def enable_p2p(self, slots, max_floats, gen_dev):
        """Route the SyncBN exchanges through peer mailboxes; `gen_dev` = the trainer's device counter of exchange generations (advances with every step, never rewound - unlike the optimizer step).  Collective: every
        rank runs the SAME sequence of host all-gathers whatever fails locally (a rank whose allocation or mapping fails
        reports it in the next all-gather instead of leaving the sequence).  Returns None when the mailboxes are in use, else
        the reason they are not (allocation / IPC mapping / self-test failed on some rank: every rank then keeps the RCCL
        collectives and frees what it had allocated or mapped)."""
        import os
        box, err, handle = None, None, None
        try:
            # (+ the barrier slots of the opt-in arena exchange, + 1: the self-test's slot, which is the last one)
            box = PeerMailboxes(self.rank, self.world, self.device, slots + 3 * self.ARENA_STAGES + 1, max_floats)
            self._arena_slot0 = slots
            handle = box.alloc_and_export()
        except Exception as ex:              # noqa: BLE001 - e.g. no fine-grained memory, no IPC export
            err = "rank %d: %r" % (self.rank, ex)
        got = self.all_gather_object((err, handle))                 # 1: everybody's handle or error
        errs = [e for e, _ in got if e]
        if not errs:
            try:
                box.import_peers([h for _, h in got])
            except Exception as ex:          # noqa: BLE001 - e.g. IPC handles not importable between these devices
                err = "rank %d: %r" % (self.rank, ex)
        errs = errs or [e for e in self.all_gather_object(err) if e]     # 2: mapping errors (every rank saw the same step-1 list, so all skip or none)
        if not errs:
            # (this all-gather is also the barrier "nobody writes into a mailbox that is not mapped yet")
            ok = box.self_test()
            errs = ["rank %d: self-test failed" % q for q, o in enumerate(self.all_gather_object(ok)) if not o]
        if errs:
            if box is not None:
                try:
                    box.close()
                except Exception:            # noqa: BLE001
                    pass
            return "; ".join(errs)[:300]
        self.p2p, self._gen_dev = box, gen_dev
        # CRIS_SYNCBN_FUSED=0: the exchange stays a kernel of its own between the BatchNorm launches (round 3's form)
        self._fused = os.environ.get("CRIS_SYNCBN_FUSED", "1") == "1"
        return None

    # ------------------------------------------------------------------------------------------------------------------
    # opt-in (CRIS_GRAD_EXCHANGE=p2p): the gradient all-reduce as a direct reduce-scatter + all-gather over the peer-mapped
    # gradient arenas (csrc/p2p.hip, include/cris_hip.h cris_p2p_arena_allreduce) instead of RCCL - every rank reads from all
    # peers at once (xGMI is a full mesh of point-to-point links), every element is summed once, on its owner, in rank order.
    # Never run on more than one GPU: correct by construction and bit-exact against a sequential sum with 2 / 4 / 8 ranks
    # sharing one device (tests/test_p2p_gpu.py), which says nothing about the links - hence not the default.
    # ------------------------------------------------------------------------------------------------------------------
    def enable_arena_exchange(self, arena):
        """Collective (the same sequence of host all-gathers on every rank whatever fails locally).  Maps every rank's `arena`
        (flat fp32 tensor, the same size everywhere) into this process and self-tests the exchange on a small range.  Returns
        None when allreduce_async() will take the direct path for ranges of `arena`, else the reason it will not."""
        if self.p2p is None:
            return "no peer mailboxes"
        lib, hip = self.p2p.lib, self.p2p.hip
        err, handle = None, None
        try:
            assert arena.dtype == torch.float32 and arena.is_contiguous() and arena.data_ptr() % 8 == 0
            h = (ctypes.c_ubyte * 64)()
            hip.check(lib.cris_p2p_export(ctypes.c_void_p(arena.data_ptr()), h), "cris_p2p_export(arena)")
            handle = bytes(h)
        except Exception as ex:              # noqa: BLE001
            err = "rank %d: %r" % (self.rank, ex)
        got = self.all_gather_object((err, handle, int(arena.numel())))
        errs = [e for e, _, _ in got if e]
        if not errs and any(n != got[0][2] for _, _, n in got):
            errs = ["arena sizes differ across ranks: %s" % [n for _, _, n in got]]
        ptrs, opened = [], []
        if not errs:
            try:
                for q, (_, hq, _) in enumerate(got):
                    if q == self.rank:
                        ptrs.append(arena.data_ptr())
                        continue
                    peer = ctypes.c_void_p()
                    hip.check(lib.cris_p2p_import((ctypes.c_ubyte * 64).from_buffer_copy(hq), ctypes.byref(peer)), "cris_p2p_import(arena)")
                    opened.append(peer.value)
                    ptrs.append(peer.value)
            except Exception as ex:          # noqa: BLE001
                err = "rank %d: %r" % (self.rank, ex)
        errs = errs or [e for e in self.all_gather_object(err) if e]
        ok = False
        if not errs:
            self._arena, self._arena_ptrs = arena, torch.tensor(ptrs, dtype=torch.int64, device=arena.device)
            ok = self._arena_self_test()
            errs = ["rank %d: arena exchange self-test failed" % q for q, o in enumerate(self.all_gather_object(ok)) if not o]
        if errs:
            self._arena, self._arena_ptrs = None, None
            for p_ in opened:
                try:
                    lib.cris_p2p_close(p_)
                except Exception:            # noqa: BLE001
                    pass
            return "; ".join(errs)[:300]
        self._arena_opened = opened
        return None

    def _arena_launch(self, lo, n, slot, gen_dev=None, gen_host=0, spin_limit=0):
        prm = self.p2p.hip.P2PArenaParams()
        prm.arenas, prm.lo, prm.n, prm.blocks = self._arena_ptrs.data_ptr(), lo, n, 0
        l = self.p2p.link(slot, gen_dev=gen_dev, gen_host=gen_host, spin_limit=spin_limit)
        prm.link = l
        self.p2p.hip.check(self.p2p.lib.cris_p2p_arena_allreduce(ctypes.byref(prm), torch.cuda.current_stream().cuda_stream),
                           "cris_p2p_arena_allreduce")

    def _arena_self_test(self, spin_limit=1 << 21):
        """two generations of the exchange on the arena's first words with known data (rank q contributes q + 1 + i / 7, then twice
        that): the result must be the sequential sum over the ranks, bit for bit, and no peer may be reported missing.  The words
        are restored afterwards (the exchange's last barrier says that every peer has finished reading them)."""
        a = self._arena
        n = min(int(a.numel()) & ~1, 4096)
        if n < 2:
            return False
        keep = a[:n].clone()
        idx = torch.arange(n, device=a.device, dtype=torch.float32) / 7.0
        ok = True
        for t in range(2):
            a[:n] = (float(self.rank + 1) + idx) * float(t + 1)
            torch.cuda.synchronize()
            self.all_gather_object(0)                          # every rank's pattern is in place before anybody reads
            self._arena_launch(0, n, self._arena_slot0, gen_host=3000 + t, spin_limit=spin_limit)
            torch.cuda.synchronize()
            want = torch.zeros(n, device=a.device)
            for q in range(self.world):
                want = want + (float(q + 1) + idx) * float(t + 1)
            ok = ok and bool(torch.equal(a[:n], want)) and int(self.p2p.err.item()) == 0
        a[:n] = keep
        torch.cuda.synchronize()
        self.p2p.err.zero_()
        return ok

    def check_peer_timeout(self):
        """COLLECTIVE (host all-gather; every rank must call it, e.g. at a logging interval or an epoch boundary - it costs one
        device synchronisation): raises RuntimeError on EVERY rank when a mailbox exchange of ANY rank gave up waiting for a peer
        (csrc/p2p_ll.h: the poll limit; the statistics of that rank are NaN from there on).  A rank that lost a peer therefore
        never trains on alone, and no rank hangs: the exchange kernels themselves are bounded by the poll limit."""
        if self.p2p is None:
            return
        torch.cuda.synchronize()
        flags = self.all_gather_object(int(self.p2p.err.item()))
        bad = [q for q, f in enumerate(flags) if f]
        if bad:
            raise RuntimeError("SyncBN peer-mailbox exchange: rank(s) %s gave up waiting for a peer (poll limit reached) - the statistics "
                               "of this run are invalid from that step on" % bad)

    def next_link(self):
        if self.p2p is None or not self._fused:
            return None
        slot = self._slot                            # exchange number inside the step, fixed at schedule time
        self._slot += 1
        return self.p2p.link(slot, gen_dev=self._gen_dev)

    def begin_step(self):
        self._slot = 0
        self._arena_calls = 0

    def allreduce_sum_op(self, t):
        if self.p2p is None:
            return lambda: self.allreduce_sum(t)
        slot = self._slot                            # exchange number inside the step, fixed at schedule time
        self._slot += 1
        return lambda: self.p2p.allreduce_sum(t, slot, gen_dev=self._gen_dev)

    def broadcast(self, t, src=0):
        dist.broadcast(t, src)

    def all_gather_object(self, obj):
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    # SyncBN exchanges run inline on the compute stream (they sit on the critical path by construction)
    def allreduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    supports_max_u8 = True       # allreduce_async(op="max") on uint8 tensors (the Adam row marks of the token embedding)

    # gradient exchange: overlapped
    def allreduce_async(self, t, op="sum"):
        rop = dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX
        a = getattr(self, "_arena", None)
        if a is not None and op == "sum" and t.dtype == torch.float32 and t.is_contiguous() and self.side is not None:
            # a range of the mapped gradient arena: the direct exchange (three launches on the side stream, no c10d work).  The
            # decision depends on the range and the call count only - identical on every rank
            lo = (t.data_ptr() - a.data_ptr()) // 4
            n = int(t.numel())
            if 0 <= lo and lo + n <= a.numel() and lo % 2 == 0 and n % 2 == 0 and n > 0 and self._arena_calls < self.ARENA_STAGES:
                slot = self._arena_slot0 + 3 * self._arena_calls      # exchange number inside the step, fixed at schedule time
                self._arena_calls += 1
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    self._arena_launch(lo, n, slot, gen_dev=self._gen_dev)
                return
        if self.side is None:
            self._pending.append(dist.all_reduce(t, op=rop, group=self.grad_group, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self._pending.append(dist.all_reduce(t, op=rop, group=self.grad_group, async_op=True))

    def wait_all(self):
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


class RcclComm:
    """The same exchanges on communicators the LIBRARY owns (csrc/comm.hip, include/cris_hip.h cris_comm_*): RCCL called from
    libcris_hip.so, no torch.distributed process group on the data path.  `store`: any c10d Store (TCPStore / FileStore /
    HashStore for one rank) - host-side bootstrap only: it carries rank 0's RCCL ids to the other ranks and the few host
    objects the trainer gathers (batch-size check, graph-capture agreement).  Interface = TorchDistComm's, so
    NativeTrainer(comm=RcclComm(...)) is the only change; all calls enqueue on HIP streams and are captured into the step's
    graph like every other launch.

        store = torch.distributed.TCPStore("127.0.0.1", port, world, is_master=(rank == 0))
        comm = RcclComm(rank, world, torch.device("cuda", local_rank), store)
    """
    capturable = True

    def __init__(self, rank, world, device, store, prefix="cris_comm"):
        from . import hip
        self.hip, self.lib = hip, hip.load()
        self.rank, self.world, self.device, self.store, self.prefix = rank, world, device, store, prefix
        torch.cuda.set_device(device)                      # cris_comm_init binds to the current device
        nbytes = 256                                       # CRIS_COMM_ID_BYTES
        key = prefix + "/id"
        if rank == 0:
            buf = (ctypes.c_ubyte * nbytes)()
            hip.check(self.lib.cris_comm_unique_id(buf), "cris_comm_unique_id")
            store.set(key, bytes(buf))
        ids = bytes(store.get(key))
        assert len(ids) == nbytes
        handle = ctypes.c_void_p()
        hip.check(self.lib.cris_comm_init(rank, world, (ctypes.c_ubyte * nbytes).from_buffer_copy(ids), ctypes.byref(handle)),
                  "cris_comm_init")
        self.handle = handle
        self._gathers = 0
        self.p2p = None

    def rccl_path(self):
        p = self.lib.cris_comm_rccl_path()
        return p.decode() if p else None

    def begin_step(self):
        return

    # --- SyncBN exchanges: inline on the compute stream ---
    def allreduce_sum(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self.hip.check(self.lib.cris_comm_syncbn_exchange(self.handle, t.data_ptr(), t.numel(),
                                                          torch.cuda.current_stream().cuda_stream), "cris_comm_syncbn_exchange")

    def allreduce_sum_op(self, t):
        return lambda: self.allreduce_sum(t)

    def next_link(self):
        return None

    # --- gradient exchange: own communicator + side stream inside the library ---
    def allreduce_async(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self.hip.check(self.lib.cris_comm_allreduce_bucket(self.handle, t.data_ptr(), t.numel(),
                                                           torch.cuda.current_stream().cuda_stream), "cris_comm_allreduce_bucket")

    def wait_all(self):
        self.hip.check(self.lib.cris_comm_wait(self.handle, torch.cuda.current_stream().cuda_stream), "cris_comm_wait")

    def broadcast(self, t, src=0):
        assert t.is_contiguous()
        self.hip.check(self.lib.cris_comm_broadcast(self.handle, t.data_ptr(), t.numel() * t.element_size(), src,
                                                    torch.cuda.current_stream().cuda_stream), "cris_comm_broadcast")

    # --- host objects (set-up time only) through the store ---
    def all_gather_object(self, obj):
        import pickle
        n = self._gathers
        self._gathers += 1
        self.store.set("%s/g%d/%d" % (self.prefix, n, self.rank), pickle.dumps(obj))
        return [pickle.loads(bytes(self.store.get("%s/g%d/%d" % (self.prefix, n, q)))) for q in range(self.world)]

    def close(self):
        if self.handle:
            torch.cuda.synchronize(self.device)
            self.lib.cris_comm_destroy(self.handle)
            self.handle = None


def merge_batchnorm_partials(sum_l, m2_l, n_l, comm, ref=None):
    """Reference arithmetic of the SyncBN forward exchange on plain tensors (used by the CPU/gloo tests): local
    (sum, M2 about the local mean, count) -> global (mean, biased var) with ONE all-reduce of the moments about `ref`
    (any vector identical on all ranks; the engine uses the running mean) - cris_bn_sync_pack / _unpack."""
    ref = torch.zeros_like(sum_l) if ref is None else ref
    mean_l = sum_l / n_l
    packed = torch.cat([sum_l - n_l * ref, m2_l + n_l * (mean_l - ref) ** 2])
    comm.allreduce_sum(packed)
    n_g = n_l * comm.world
    s1, s2 = packed[:sum_l.numel()], packed[sum_l.numel():]
    mean_g = ref + s1 / n_g
    m2_g = (s2 - s1 * s1 / n_g).clamp_min(0)
    return mean_g, m2_g / n_g

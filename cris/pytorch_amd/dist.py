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
import torch
import torch.distributed as dist


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

    # SyncBN exchanges run inline on the compute stream (they sit on the critical path by construction)
    def allreduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    # gradient exchange: overlapped
    def allreduce_async(self, t):
        if self.side is None:
            self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.grad_group, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self._pending.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.grad_group, async_op=True))

    def wait_all(self):
        for w in self._pending:
            w.wait()
        self._pending = []
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


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

"""Thin functional wrappers: torch tensors (device memory only) -> C-ABI launchers of libcris_hip.so.

No arithmetic happens in Python/torch here; torch supplies buffers (`torch.empty/zeros`) and the
current HIP stream.  Every function launches on `torch.cuda.current_stream()`.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import hip
from .hip import ptr

BF16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


@dataclass(frozen=True)
class Geom:
    """im2col geometry of an NHWC input [Bn, H, W, C] -> rows (b, oh, ow), k = tap*C + c."""
    Bn: int
    H: int
    W: int
    C: int
    KH: int = 1
    KW: int = 1
    stride: int = 1
    pad: int = 0

    @property
    def OH(self):
        return (self.H + 2 * self.pad - self.KH) // self.stride + 1

    @property
    def OW(self):
        return (self.W + 2 * self.pad - self.KW) // self.stride + 1

    @property
    def M(self):
        return self.Bn * self.OH * self.OW

    @property
    def K(self):
        return self.KH * self.KW * self.C

    @staticmethod
    def linear(M: int, C: int) -> "Geom":
        return Geom(M, 1, 1, C)


@dataclass(frozen=True)
class Drop:
    p: float
    seed: int
    stream: int
    dev: Optional[torch.Tensor] = None       # optional uint32/int32 device word added to `seed` (HIP-graph replay)

    @property
    def thresh(self):
        return hip.dropout_threshold(self.p)


NO_DROP = Drop(0.0, 0, 0)


class GemmQueue:
    """Independent forward / input-gradient GEMMs launched together (cris_conv_gemm_group_launch): conv_gemm(queue=q) fills the
    parameter block (and allocates the statistics partials for the tile variant that will run) but defers the launch to
    q.flush(), which issues one grouped launch per (tile variant, epilogue) present in the queue - longest reductions first.
    The caller guarantees independence: no queued problem reads or accumulates into what another one writes.  Problems whose
    own tile choice cannot be grouped (skinny kernels, the larger 8-wave tiles) are launched one by one at the flush.
    CRIS_GEMM_GROUPS=0 launches every problem alone (A/B)."""

    GROUPABLE = None          # variant ids a group launch can run (filled on first use)
    enabled = os.environ.get("CRIS_GEMM_GROUPS", "1") != "0"

    def __init__(self):
        self.items = []        # (params, variant, epilogue, keep-alive tensors, flops, bytes, tag)

    @classmethod
    def groupable(cls):
        if cls.GROUPABLE is None:
            names = gemm_variants()
            cls.GROUPABLE = {names.index(n) for n in ("128x64", "64x64", "64x128", "128x128", "8w128x128")}
        return cls.GROUPABLE

    def add(self, p, variant, keep, flops, nbytes, tag):
        epi = C.c_int(0)
        v = hip.load().cris_conv_gemm_plan(C.byref(p), variant, C.byref(epi))
        if v < 0:
            raise ValueError("GEMM tile variant %r cannot run this problem" % (variant,))
        self.items.append((p, v, epi.value, keep, flops, nbytes, tag))

    def __len__(self):
        return len(self.items)

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        groups = {}
        for it in items:
            key = (it[1], it[2]) if (self.enabled and it[1] in self.groupable()) else ("solo", id(it))
            groups.setdefault(key, []).append(it)
        for key, its in groups.items():
            if key[0] == "solo":
                _launch_gemm(its[0][0], its[0][1], its[0][4], its[0][5], its[0][6])
                continue
            its.sort(key=lambda it: -it[0].K)                    # stable: longest reductions first
            for i in range(0, len(its), hip.GEMM_GROUP_MAX):
                chunk = its[i:i + hip.GEMM_GROUP_MAX]
                if len(chunk) == 1:
                    _launch_gemm(chunk[0][0], chunk[0][1], chunk[0][4], chunk[0][5], chunk[0][6])
                    continue
                grp = hip.ConvGemmGroup()
                grp.n = len(chunk)
                for j, it in enumerate(chunk):
                    grp.prob[j] = it[0]
                if KERNEL_TIMER is not None:
                    KERNEL_TIMER.launch("conv_gemm", sum(it[4] for it in chunk), sum(it[5] for it in chunk), "cris_conv_gemm_group_launch",
                                        C.byref(grp), key[0], tag="group of %d: %s" % (len(chunk), " + ".join(it[6] for it in chunk)), tile=True)
                else:
                    hip.call("cris_conv_gemm_group_launch", C.byref(grp), key[0], _stream())


def _launch_gemm(p, variant, flops, nbytes, tag):
    if KERNEL_TIMER is not None:
        # (classified by the problem, not by the kernel that runs it: M <= 144 linears are the text encoder's / per-sample vectors)
        v = hip.load().cris_conv_gemm_plan(C.byref(p), variant, None)
        KERNEL_TIMER.launch("skinny_gemm" if (p.M <= 144 and p.KH == 1) else "conv_gemm", flops, nbytes, "cris_conv_gemm_variant",
                            C.byref(p), variant, tag=tag, tile=gemm_variants()[v] not in ("skinny1", "skinny9", "skinny9s"))
        return
    hip.call("cris_conv_gemm_variant", C.byref(p), variant, _stream())


def conv_gemm(A, Wt, g: Geom, N: int, *, lda=None, a_coff=0, ldb=None, bias=None, act=0, resid=None, ldr=None, r_coff=0,
              out=None, ldc=None, c_coff=0, outT=None, T_L=0, T_Lpad=0, T_E=0, T_sec_stride=0, stats=False,
              drop: Drop = NO_DROP, variant: int = -1, queue: Optional[GemmQueue] = None, bnr: Optional[dict] = None):
    """out[M,N] = epi(A_im2col @ Wt^T).  A bf16 NHWC buffer (row stride lda), Wt bf16 [N][ldb].
    stats=True: also returns the BatchNorm statistics partials (Stats) of the output columns.
    variant: tile variant (index or name, see gemm_variants()); -1 = the library's choice for the problem size.
    queue: defer the launch to the queue's flush (grouped with the other independent problems waiting there).
    bnr: dict(y, ldy, coff, mean, invstd, scale, shift) - this GEMM's output is the gradient of relu(bn(y)): its epilogue also
    writes the BatchNorm-backward partial sums; returns a BnrParts (table [parts][2N] for bn_bwd(pre_reduced=...)), or None when
    the problem runs on a kernel without that epilogue (the GEMM is then launched plainly and the caller reduces as usual)."""
    p = hip.ConvGemmParams()
    p.A, p.Wt, p.bias = ptr(A), ptr(Wt), ptr(bias)
    p.lda = lda if lda is not None else A.shape[-1]
    p.a_coff = a_coff
    p.Bn, p.H, p.W, p.C = g.Bn, g.H, g.W, g.C
    p.OH, p.OW, p.KH, p.KW, p.stride, p.pad = g.OH, g.OW, g.KH, g.KW, g.stride, g.pad
    p.ldb = ldb if ldb is not None else Wt.shape[-1]
    p.M, p.N, p.K = g.M, N, g.K
    p.act = act
    if resid is not None:
        p.resid = ptr(resid)
        p.ldr = ldr if ldr is not None else resid.shape[-1]
        p.r_coff = r_coff
        p.resid_f32 = 1 if resid.dtype == torch.float32 else 0
    if out is not None:
        p.out = ptr(out)
        p.ldc = ldc if ldc is not None else out.shape[-1]
        p.c_coff = c_coff
        p.out_f32 = 1 if out.dtype == torch.float32 else 0
    if outT is not None:
        p.outT = ptr(outT)
        p.T_L, p.T_Lpad, p.T_E, p.T_sec_stride = T_L, T_Lpad, T_E, T_sec_stride
    p.drop_p, p.drop_thresh, p.drop_seed, p.drop_stream = drop.p, drop.thresh, drop.seed & 0xFFFFFFFF, drop.stream
    p.drop_seed_dev = ptr(drop.dev)
    st = None
    if isinstance(variant, str):
        variant = gemm_variants().index(variant)
    rows = hip.load().cris_conv_gemm_variant_stat_rows(C.byref(p), variant)   # depends on the tile variant that will run
    if rows < 0:
        raise ValueError("GEMM tile variant %r cannot run this problem" % (variant,))
    parts = None
    if bnr is not None:
        assert not stats
        epi = C.c_int(0)
        v = hip.load().cris_conv_gemm_plan(C.byref(p), variant, C.byref(epi))
        nparts = (g.M + rows - 1) // rows
        if gemm_variants()[v] not in ("128x64", "64x64", "64x128", "128x128", "8w128x128") or epi.value != 1 or nparts > BNR_MAX_PARTS:
            bnr = None                           # no such epilogue on this kernel / too long a list: plain launch
        else:
            parts = BnrParts(torch.empty(nparts, 2 * N, dtype=torch.float32, device=A.device), nparts)
            p.colsum, p.colsq, p.stat_ld = ptr(parts.t), parts.t.data_ptr() + 4 * N, 2 * N
            p.bnr_y, p.bnr_ldy, p.bnr_coff = ptr(bnr["y"]), bnr["ldy"], bnr["coff"]
            p.bnr_mean, p.bnr_invstd, p.bnr_scale, p.bnr_shift = ptr(bnr["mean"]), ptr(bnr["invstd"]), ptr(bnr["scale"]), ptr(bnr["shift"])
    if stats:
        st = Stats((g.M + rows - 1) // rows, N, rows, A.device)
        p.colsum, p.colsq = ptr(st[0]), ptr(st[1])
    nws = hip.load().cris_conv_gemm_ws_floats(C.byref(p), variant)
    ws = torch.empty(nws, dtype=torch.float32, device=A.device) if nws else None       # (stream-ordered: freed after the launches)
    p.ws = ptr(ws)
    flops, nbytes, tag = 2.0 * g.M * N * g.K, 2.0 * (g.M * g.C + N * g.K + g.M * N), "M%d N%d K%d k%d" % (g.M, N, g.K, g.KH)
    if queue is not None:
        assert bnr is None and parts is None, "a queued launch cannot feed a BatchNorm backward that runs before the flush"
        queue.add(p, variant, (A, Wt, bias, resid, out, outT, drop.dev, ws, st.t if st is not None else None), flops, nbytes, tag)
        return st
    _launch_gemm(p, variant, flops, nbytes, tag)
    return parts if bnr is not None else st


BNR_MAX_PARTS = int(os.environ.get("CRIS_BNR_MAX_PARTS", "512"))     # longer partial lists (the 86528-pixel maps) keep the separate reduce launch
BNR_FUSE = os.environ.get("CRIS_BNR_FUSE", "1") == "1"                 # 0: BatchNorm backward always reduces in its own launch (A/B)


class BnrParts:
    """BatchNorm-backward partial sums written by an input-gradient GEMM's epilogue: t [nparts][2C] = (sum g | sum g xhat)"""
    __slots__ = ("t", "nparts")

    def __init__(self, t, nparts):
        self.t, self.nparts = t, nparts


def gemm_variants():
    """names of the conv_gemm tile variants, by index (cris_conv_gemm_variant)"""
    lib = hip.load()
    return [lib.cris_conv_gemm_variant_name(i).decode() for i in range(lib.cris_conv_gemm_num_variants())]


class KernelTimer:
    """HIP-event timing of individual launches on the launch stream (bench.py roofline leg)."""

    def __init__(self):
        self.records = []        # (name, flops, bytes, start_event, end_event)
        self.tile_flags = []     # per record: ran on a tile kernel

    def launch(self, name, flops, nbytes, fn, *args, tag="", tile=False):
        """tile: the launch runs one of the forward / input-gradient TILE kernels (conv_gemm_kernel / conv_gemm8_kernel and
        their grouped forms) - the family whose memory-side traffic the PMC passes report (tile_family())"""
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        hip.call(fn, *args, s.cuda_stream)
        e1.record(s)
        self.records.append((name, flops, nbytes, e0, e1, tag))
        self.tile_flags.append(bool(tile))

    def tile_family(self):
        """{launches, ms, flops, bytes} over every launch of a tile kernel, whatever the problem's class - the same set of
        launches the PMC summary's "conv_gemm (all tile kernels)" entry covers"""
        d = dict(launches=0, ms=0.0, flops=0.0, bytes=0.0)
        for (name, fl, nb, e0, e1, _), t in zip(self.records, self.tile_flags):
            if t:
                d["launches"] += 1
                d["ms"] += e0.elapsed_time(e1)
                d["flops"] += fl
                d["bytes"] += nb
        return d

    def by_shape(self):
        """{(kernel, shape tag): {launches, ms, flops, bytes}} - which problem shapes the time goes to"""
        out = {}
        for name, fl, nb, e0, e1, tag in self.records:
            d = out.setdefault((name, tag), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out

    def summary(self):
        out = {}
        for name, fl, nb, e0, e1, _ in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


KERNEL_TIMER = None


_WGRAD_BLOCKS = int(os.environ.get("CRIS_WGRAD_BLOCKS", "512"))      # launch-geometry knobs (never change results)
_WGRAD_GROUP_M = int(os.environ.get("CRIS_WGRAD_GROUP_M", "8192"))    # problems up to this many pixel rows are queued
_WGRAD_MIN_STEPS = int(os.environ.get("CRIS_WGRAD_MIN_STEPS", "4"))    # 128-row steps per split at least (4 vs 8: -0.1 ms per step, r03 call Q)
_WGRAD_FLUSH_BLOCKS = int(os.environ.get("CRIS_WGRAD_FLUSH_BLOCKS", "3072"))


_WGRAD8_BLOCKS = int(os.environ.get("CRIS_WGRAD8_BLOCKS", "256"))    # 8-wave 256x256 tile: one block (128 KB of LDS) per CU
# 1 (default since round 4, call r04a: 12.233 against 12.280 ms per step): the split reductions of the large layers (32 launches
# of ~6.6 us per step) are deferred to the queue's flush and run as grouped launches (cris_wgrad_reduce_group); 0: one by one
_WGRAD_REDUCE_GROUP = os.environ.get("CRIS_WGRAD_REDUCE_GROUP", "1") == "1"


def wgrad_splits(M: int, N: int, K: int, tile: int = 128) -> int:
    """split the pixel reduction only as far as needed to put ~2 blocks (128x128 tile) / 1 block (256x256 tile) on each of the
    256 CUs; every split costs a partial tile written to and read back from the workspace, so long reductions per block win"""
    tiles = ((N + tile - 1) // tile) * ((K + tile - 1) // tile)
    steps = (M + 127) // 128
    blocks = _WGRAD_BLOCKS if tile == 128 else _WGRAD8_BLOCKS
    want = max(1, (blocks + tiles // 2) // tiles)
    return max(1, min(want, (steps + _WGRAD_MIN_STEPS - 1) // _WGRAD_MIN_STEPS, 256))


def _wgrad_params(dY, X, g: Geom, N: int, dW, ldy, y_coff, N_ld, ldx, x_coff, ldw, splits, dbias, tile=0):
    p = hip.WgradParams()
    p.tile = tile
    p.dY, p.X, p.dW = ptr(dY), ptr(X), ptr(dW)
    p.ldy = ldy if ldy is not None else dY.shape[-1]
    p.y_coff = y_coff
    p.N_ld = N_ld if N_ld is not None else pad8(N)
    p.ldx = ldx if ldx is not None else X.shape[-1]
    p.x_coff = x_coff
    p.Bn, p.H, p.W, p.C = g.Bn, g.H, g.W, g.C
    p.OH, p.OW, p.KH, p.KW, p.stride, p.pad = g.OH, g.OW, g.KH, g.KW, g.stride, g.pad
    p.M, p.N, p.K = g.M, N, g.K
    p.ldw = ldw if ldw is not None else dW.shape[-1]
    p.dbias = ptr(dbias)
    p.splits = splits
    return p


class WgradQueue:
    """Weight-gradient problems of the mid-size layers, launched together (cris_conv_wgrad_group): such a layer has 16-600
    output tiles of 128x128 - alone it would have to split its pixel reduction 4-11 ways to occupy 256 CUs.  The engine
    queues them while backward walks an arena stage and flushes at the stage boundary (or when enough blocks are waiting),
    longest reductions first.  Operand tensors are kept alive until their launch has been issued."""

    def __init__(self, on_full=None):
        self.items = []
        self.blocks = 0
        self.on_full = on_full        # called instead of flush() when enough blocks are waiting (the engine picks the stream)
        self.reduces = []             # (params, operand tensors) of launches whose split reduction was deferred to the flush

    def add_reduce(self, p, keep):
        """a problem launched with defer_reduce: its slabs are added up by the next flush (same stream as the launch)"""
        self.reduces.append((p, keep))

    def add(self, p, keep, flops, nbytes):
        tile = hip.load().cris_conv_wgrad_tile(C.byref(p))       # 128 or 256: a group launch runs ONE tile kernel
        self.items.append((p, keep, flops, nbytes, tile))
        self.blocks += ((p.N + 127) // 128) * ((p.K + 127) // 128)
        if self.blocks >= _WGRAD_FLUSH_BLOCKS:
            (self.on_full or self.flush)()

    def flush(self):
        """launch everything queued on the current stream; returns the operand tensors of the launched problems (a caller
        that launches on a side stream keeps them alive until that stream has been joined)"""
        items, self.items, self.blocks = self.items, [], 0
        reduces, self.reduces = self.reduces, []
        for i in range(0, len(reduces), hip.WGRAD_GROUP_MAX):
            chunk = reduces[i:i + hip.WGRAD_GROUP_MAX]
            grp = hip.WgradGroup()
            grp.n = len(chunk)
            for j, it in enumerate(chunk):
                grp.prob[j] = it[0]
            hip.call("cris_wgrad_reduce_group", C.byref(grp), _stream())
        if not items:
            return [it[1] for it in reduces]
        items.sort(key=lambda it: (-it[4], -it[0].M))          # by tile kernel; stable: longest pixel reductions first
        bounds = [i for i in range(len(items)) if i == 0 or items[i][4] != items[i - 1][4]] + [len(items)]
        starts = [i for a, b in zip(bounds, bounds[1:]) for i in range(a, b, hip.WGRAD_GROUP_MAX)]
        ends = [min(i + hip.WGRAD_GROUP_MAX, next(b for b in bounds[1:] if b > i)) for i in starts]
        for i, j in zip(starts, ends):
            chunk = items[i:j]
            grp = hip.WgradGroup()
            grp.n = len(chunk)
            for j, it in enumerate(chunk):
                grp.prob[j] = it[0]
            if KERNEL_TIMER is not None:
                KERNEL_TIMER.launch("conv_wgrad", sum(it[2] for it in chunk), sum(it[3] for it in chunk), "cris_conv_wgrad_group",
                                    C.byref(grp), tag="group of %d (M %d..%d)" % (len(chunk), chunk[-1][0].M, chunk[0][0].M))
            else:
                hip.call("cris_conv_wgrad_group", C.byref(grp), _stream())
        return [it[1] for it in items] + [it[1] for it in reduces]


def conv_wgrad(dY, X, g: Geom, N: int, dW, *, ldy=None, y_coff=0, N_ld=None, ldx=None, x_coff=0, ldw=None, splits=None, dbias=None,
               queue: Optional[WgradQueue] = None, tile: int = 0):
    """dW[N][K] = dY^T X_im2col (GEMM layout), dbias = column sums of dY.  With a `queue`, problems of up to
    CRIS_WGRAD_GROUP_M pixel rows are deferred to the queue's next grouped launch (unsplit); larger ones (and every call
    without a queue) launch now, their pixel range split over blocks with a deterministic workspace reduction."""
    flops, nbytes = 2.0 * g.M * N * g.K, 2.0 * (g.M * N + g.M * g.C) + 4.0 * N * g.K
    if queue is not None and splits is None and g.M <= _WGRAD_GROUP_M:
        p = _wgrad_params(dY, X, g, N, dW, ldy, y_coff, N_ld, ldx, x_coff, ldw, 1, dbias, tile)
        queue.add(p, (dY, X, dW, dbias), flops, nbytes)
        return
    p = _wgrad_params(dY, X, g, N, dW, ldy, y_coff, N_ld, ldx, x_coff, ldw, 1, dbias, tile)
    p.splits = splits if splits is not None else wgrad_splits(g.M, N, g.K, hip.load().cris_conv_wgrad_tile(C.byref(p)))
    nws = hip.load().cris_wgrad_ws_floats(p.M, p.N, p.ldw, p.splits)
    ws = torch.empty(nws, dtype=torch.float32, device=dW.device) if nws else None
    p.ws = ptr(ws)
    if KERNEL_TIMER is not None:
        KERNEL_TIMER.launch("conv_wgrad", flops, nbytes, "cris_conv_wgrad", C.byref(p), tag="M%d N%d K%d k%d s%d" % (g.M, N, g.K, g.KH, p.splits))
        return
    if queue is not None and _WGRAD_REDUCE_GROUP and p.splits > 1:
        p.defer_reduce = 1
        hip.call("cris_conv_wgrad", C.byref(p), _stream())
        queue.add_reduce(p, (dY, X, dW, dbias, ws))
        return
    hip.call("cris_conv_wgrad", C.byref(p), _stream())


PACK_SKEW = os.environ.get("CRIS_PACK_SKEW", "0") == "1"


def pack_row_stride(k):
    """row stride (elements) of a bf16 weight pack whose rows hold `k` values.  Rows a multiple of 256 B apart land on few of an
    L2's sixteen 128-byte-interleaved channels (K = 4608: 9216 B = 72 lines, every row on one of two channels); one more line
    makes the count odd and the rows walk through all of them.  Measured STANDALONE (tools/stride_skew_probe.py, call r06p: operands
    hot in L2): the K = 4608 convolutions -5 ... -10 %, everything else within +-1 %.  Measured IN THE STEP (call r06q, three
    interleaved runs each): 11.726 / 11.745 / 11.745 ms with, 11.722 / 11.737 / 11.760 without - nothing: in the step the weights
    come from HBM, not from a warm L2, and the channel spread of a resident panel is not what bounds the launch.  Hence OFF by
    default (CRIS_PACK_SKEW=1 switches it on; results are bit-identical either way)."""
    return k + 64 if (PACK_SKEW and k % 128 == 0) else k


class PackTable:
    """Device-resident table of cris_pack_desc: one launch repacks every fp32 weight into its bf16
    forward (F) / dgrad (D) layouts."""

    def __init__(self):
        self.descs = []
        self.info = []       # per tensor: (dstF, dstD, N, Cin, taps, Cpad, Npad, transposed) - what AdamTable(packs=) takes
        self.keep = []       # tensors referenced by the table
        self.dev = None
        self.total_blocks = 0

    def add(self, src, N, Cin, taps, Cpad=None, Npad=None, want_F=True, want_D=True, src_transposed=False, row_scale=None):
        Cpad = Cpad if Cpad is not None else pad8(Cin)
        Npad = Npad if Npad is not None else pad8(N)
        # row strides of an ODD number of 128-byte lines (pack_row_stride): the GEMMs read the packs with ldb = shape[1]
        ldF, ldD = pack_row_stride(taps * Cpad), pack_row_stride(taps * Npad)
        dstF = torch.zeros(N, ldF, dtype=BF16, device=src.device) if want_F else None
        dstD = torch.zeros(Cin, ldD, dtype=BF16, device=src.device) if want_D else None
        d = hip.PackDesc()
        d.src, d.dstF, d.dstD, d.row_scale = ptr(src), ptr(dstF), ptr(dstD), ptr(row_scale)
        d.N, d.Cin, d.taps, d.Cpad, d.Npad, d.src_transposed = N, Cin, taps, Cpad, Npad, int(src_transposed)
        d.ldF, d.ldD = ldF, ldD
        self.descs.append(d)
        self.keep.append((src, dstF, dstD, row_scale))
        self.info.append((dstF, dstD, N, Cin, taps, Cpad, Npad, bool(src_transposed)))
        self.dev = None
        return dstF, dstD

    def finalize(self, device):
        lib = hip.load()
        n = len(self.descs)
        arr = (hip.PackDesc * n)()
        start = 0
        for i, d in enumerate(self.descs):
            d.block_start = start
            start += lib.cris_pack_blocks(C.byref(d))
            arr[i] = d
        self.total_blocks = start
        raw = bytes(arr)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self.dev = host.to(device)

    def run(self):
        if not self.descs:
            return
        if self.dev is None:
            self.finalize(self.keep[0][0].device)
        hip.call("cris_pack_weights", ptr(self.dev), len(self.descs), self.total_blocks, _stream())

    def refresh_sources(self, new_srcs):
        """Re-point descriptors at new parameter storage (e.g. after model.cuda())."""
        for d, s, k in zip(self.descs, new_srcs, range(len(self.keep))):
            d.src = ptr(s)
            self.keep[k] = (s,) + self.keep[k][1:]
        self.dev = None


# ---- BatchNorm ---------------------------------------------------------------------------------

def partials_rows(nparts: int) -> int:
    """rows a partials buffer needs (room for bn_finalize's first-level merge; cris_bn_partials_rows)"""
    return hip.load().cris_bn_partials_rows(nparts)


class Stats:
    """BatchNorm statistics partials: t [2][rows >= nparts][C] (sum, M2 about the part mean), `rows_per_part` rows each."""
    __slots__ = ("t", "nparts", "rows_per_part")

    def __init__(self, nparts, C_, rows_per_part, device):
        self.t = torch.empty(2, partials_rows(nparts), C_, dtype=torch.float32, device=device)
        self.nparts, self.rows_per_part = nparts, rows_per_part

    def __getitem__(self, i):
        return self.t[i]




def bn_finalize(st: Optional[Stats], count_local, count, gamma, beta, rmean, rvar, momentum, eps, C_, scale, shift, mean, invstd,
                merged=None, global_stats=None, link=None):
    """link (hip.P2PLink): SyncBatchNorm - the statistics exchange happens inside this launch (cris_bn_finalize_sync)"""
    if link is not None:
        assert merged is None and global_stats is None and st is not None
        hip.call("cris_bn_finalize_sync", ptr(st[0]), ptr(st[1]), st.nparts, st.rows_per_part, float(count_local), float(count),
                 ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), float(momentum), float(eps), C_, ptr(scale), ptr(shift), ptr(mean),
                 ptr(invstd), C.byref(link), _stream())
        return
    hip.call("cris_bn_finalize", ptr(st[0]) if st is not None else None, ptr(st[1]) if st is not None else None,
             st.nparts if st is not None else 0, st.rows_per_part if st is not None else 0, float(count_local), float(count),
             ptr(gamma), ptr(beta),
             ptr(rmean), ptr(rvar), float(momentum), float(eps), C_, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(merged),
             ptr(global_stats), _stream())


def bn_sync_pack(merged, mean_local, ref, n_local, C_):
    hip.call("cris_bn_sync_pack", ptr(merged), ptr(mean_local), ptr(ref), float(n_local), C_, _stream())


def bn_sync_unpack(merged, ref, count_global, C_):
    hip.call("cris_bn_sync_unpack", ptr(merged), ptr(ref), float(count_global), C_, _stream())


def colstats(x, M, C_, rows_per_part, device, ldx=None, coff=0) -> Stats:
    st = Stats((M + rows_per_part - 1) // rows_per_part, C_, rows_per_part, device)
    hip.call("cris_colstats_bf16", ptr(x), ldx if ldx is not None else x.shape[-1], coff, M, C_, rows_per_part, ptr(st[0]),
             ptr(st[1]), _stream())
    return st


def bn_eval_coeffs(gamma, beta, rmean, rvar, eps, C_, scale, shift):
    hip.call("cris_bn_eval_coeffs", ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), float(eps), C_, ptr(scale), ptr(shift),
             _stream())


def bn_apply(y, scale, shift, z, Bn, H, W, C_, *, ldy=None, y_coff=0, ldz=None, z_coff=0, relu=True, pool=False, y2=None,
             ldy2=None, y2_coff=0, scale2=None, shift2=None, ident=None, ldi=None, i_coff=0, mul=None):
    p = hip.BnApplyParams()
    p.y, p.ldy, p.y_coff = ptr(y), ldy if ldy is not None else y.shape[-1], y_coff
    p.scale, p.shift = ptr(scale), ptr(shift)
    if y2 is not None:
        p.y2, p.ldy2, p.y2_coff = ptr(y2), ldy2 if ldy2 is not None else y2.shape[-1], y2_coff
        p.scale2, p.shift2 = ptr(scale2), ptr(shift2)
    if ident is not None:
        p.ident, p.ldi, p.i_coff = ptr(ident), ldi if ldi is not None else ident.shape[-1], i_coff
    p.mul = ptr(mul)
    p.z, p.ldz, p.z_coff = ptr(z), ldz if ldz is not None else z.shape[-1], z_coff
    p.Bn, p.H, p.W, p.C = Bn, H, W, C_
    p.relu, p.pool = int(relu), int(pool)
    hip.call("cris_bn_apply", C.byref(p), _stream())


def bn_bwd(dz, y, scale, shift, mean, invstd, sums, dy, Bn, H, W, C_, count, *, lddz=None, dz_coff=0, ldy=None, y_coff=0,
           lddy=None, dy_coff=0, relu=True, pool=False, z=None, ldz=None, z_coff=0, y2=None, ldy2=None, y2_coff=0, mean2=None,
           invstd2=None, scale2=None, dy2=None, lddy2=None, dy2_coff=0, mul=None, dmul=None, dident=None, lddi=None,
           di_coff=0, dident_accum=False, between=None, link=None, local_sums=None, pre_reduced: Optional[BnrParts] = None):
    """reduce + apply.  `between(sums)` (optional) runs between the two launches (SyncBN all-reduce by a collective); with
    `link` (hip.P2PLink) the summation launch itself adds this rank's sums into `local_sums` and exchanges them
    (cris_bn_bwd_reduce_sync): `sums` then receives the sums over all ranks."""
    p = hip.BnBwdParams()
    p.dz, p.lddz, p.dz_coff = ptr(dz), lddz if lddz is not None else dz.shape[-1], dz_coff
    if z is not None:
        p.z, p.ldz, p.z_coff = ptr(z), ldz if ldz is not None else z.shape[-1], z_coff
    p.y, p.ldy, p.y_coff = ptr(y), ldy if ldy is not None else y.shape[-1], y_coff
    p.scale, p.shift, p.mean, p.invstd = ptr(scale), ptr(shift), ptr(mean), ptr(invstd)
    if y2 is not None:
        p.y2, p.ldy2, p.y2_coff = ptr(y2), ldy2 if ldy2 is not None else y2.shape[-1], y2_coff
        p.mean2, p.invstd2, p.scale2 = ptr(mean2), ptr(invstd2), ptr(scale2)
    p.mul, p.sums, p.dmul = ptr(mul), ptr(sums), ptr(dmul)
    p.dy, p.lddy, p.dy_coff = ptr(dy), lddy if lddy is not None else dy.shape[-1], dy_coff
    if dy2 is not None:
        p.dy2, p.lddy2, p.dy2_coff = ptr(dy2), lddy2 if lddy2 is not None else dy2.shape[-1], dy2_coff
    if dident is not None:
        p.dident, p.lddi, p.di_coff = ptr(dident), lddi if lddi is not None else dident.shape[-1], di_coff
        p.dident_accum = int(dident_accum)
    p.Bn, p.H, p.W, p.C = Bn, H, W, C_
    p.relu, p.pool = int(relu), int(pool)
    p.count = float(count)
    s = _stream()
    if pre_reduced is not None:
        # the partial rows came out of the epilogue of the GEMM that produced dz: only their summation (+ exchange) is left
        p.part = ptr(pre_reduced.t)
        if link is not None:
            hip.call("cris_bn_bwd_sum_sync", C.byref(p), pre_reduced.nparts, ptr(local_sums), C.byref(link), s)
        else:
            hip.call("cris_bn_bwd_sum", C.byref(p), pre_reduced.nparts, s)
            if between is not None:
                between(sums)
        hip.call("cris_bn_bwd_apply", C.byref(p), s)
        return
    part = torch.empty(hip.load().cris_bn_bwd_ws_floats(C.byref(p)), dtype=torch.float32, device=dz.device)
    p.part = ptr(part)
    if link is not None:
        hip.call("cris_bn_bwd_reduce_sync", C.byref(p), ptr(local_sums), C.byref(link), s)
    else:
        hip.call("cris_bn_bwd_reduce", C.byref(p), s)
        if between is not None:
            between(sums)
    hip.call("cris_bn_bwd_apply", C.byref(p), s)


# ---- LayerNorm ---------------------------------------------------------------------------------
def ln_fwd(x, gamma, beta, rows, C_, mean, rstd, *, ldx=None, y=None, ypos=None, pos=None, pos_rows=0, resid=None,
           out_f32=None, in_relu=False, in_drop: Drop = NO_DROP, out_drop: Drop = NO_DROP, eps=1e-5):
    p = hip.LnFwdParams()
    p.x, p.x_f32, p.ldx = ptr(x), int(x.dtype == torch.float32), ldx if ldx is not None else C_
    p.gamma, p.beta = ptr(gamma), ptr(beta)
    p.pos, p.pos_rows = ptr(pos), pos_rows
    p.resid, p.y, p.ypos, p.out_f32 = ptr(resid), ptr(y), ptr(ypos), ptr(out_f32)
    p.mean, p.rstd = ptr(mean), ptr(rstd)
    p.rows, p.C, p.in_relu = rows, C_, int(in_relu)
    p.in_drop_p, p.in_thresh, p.in_seed, p.in_stream = in_drop.p, in_drop.thresh, in_drop.seed & 0xFFFFFFFF, in_drop.stream
    p.out_drop_p, p.out_thresh, p.out_seed, p.out_stream = out_drop.p, out_drop.thresh, out_drop.seed & 0xFFFFFFFF, out_drop.stream
    p.seed_dev = ptr(in_drop.dev if in_drop.dev is not None else out_drop.dev)
    p.eps = eps
    hip.call("cris_ln_fwd", C.byref(p), _stream())


class SumQueue:
    """Ordered column sums of per-block partial tables (cris_sum_tables), deferred and launched together: the LayerNorm
    parameter gradients are only read by the exchange / optimizer, so the engine flushes one group per arena stage."""

    def __init__(self):
        self.items = []

    def add(self, part, out, nparts, ncol, ld, part_off=0):
        e = hip.SumEntry()
        e.part, e.out = part.data_ptr() + 4 * part_off, ptr(out)
        e.nparts, e.ncol, e.ld = nparts, ncol, ld
        self.items.append((e, part, out))

    def flush(self):
        items, self.items = self.items, []
        for i in range(0, len(items), hip.SUM_GROUP_MAX):
            chunk = items[i:i + hip.SUM_GROUP_MAX]
            grp = hip.SumGroup()
            grp.n = len(chunk)
            for j, it in enumerate(chunk):
                grp.e[j] = it[0]
            hip.call("cris_sum_tables", C.byref(grp), _stream())


def ln_bwd(x, gamma, mean, rstd, rows, C_, dx, *, ldx=None, dy=None, dypos=None, dout_f32=None, dgamma=None, dbeta=None,
           dx_accum=False, in_relu=False, in_drop: Drop = NO_DROP, out_drop: Drop = NO_DROP, queue: Optional[SumQueue] = None):
    """dx now; dgamma / dbeta (overwritten) = ordered sum of the kernel's per-block partial rows - through `queue` at its next
    flush, or right away without one."""
    p = hip.LnBwdParams()
    p.x, p.x_f32, p.ldx = ptr(x), int(x.dtype == torch.float32), ldx if ldx is not None else C_
    p.gamma, p.mean, p.rstd = ptr(gamma), ptr(mean), ptr(rstd)
    p.dy, p.dypos, p.dout_f32 = ptr(dy), ptr(dypos), ptr(dout_f32)
    nparts = hip.load().cris_ln_bwd_parts(rows)
    part = torch.empty(nparts, 2 * C_, dtype=torch.float32, device=x.device)
    p.part = ptr(part)
    p.dx, p.dx_f32, p.dx_accum = ptr(dx), int(dx.dtype == torch.float32), int(dx_accum)
    p.rows, p.C, p.in_relu = rows, C_, int(in_relu)
    p.in_drop_p, p.in_thresh, p.in_seed, p.in_stream = in_drop.p, in_drop.thresh, in_drop.seed & 0xFFFFFFFF, in_drop.stream
    p.out_drop_p, p.out_thresh, p.out_seed, p.out_stream = out_drop.p, out_drop.thresh, out_drop.seed & 0xFFFFFFFF, out_drop.stream
    p.seed_dev = ptr(in_drop.dev if in_drop.dev is not None else out_drop.dev)
    hip.call("cris_ln_bwd", C.byref(p), _stream())
    q = queue if queue is not None else SumQueue()
    if dgamma is not None:
        q.add(part, dgamma, nparts, C_, 2 * C_, 0)
    if dbeta is not None:
        q.add(part, dbeta, nparts, C_, 2 * C_, C_)
    if queue is None:
        q.flush()


# ---- attention ---------------------------------------------------------------------------------
def attn_params(Q, K, V, Vt, B, Hn, Lq, Lk, Lk_pad, scale, *, ldq=None, ldk=None, ldv=None, Kt=None, Qt=None, Lq_pad=0,
                key_tokens=None, causal=False, drop: Drop = NO_DROP):
    p = hip.AttnParams()
    p.Q, p.ldq = ptr(Q), ldq if ldq is not None else Q.shape[-1]
    p.K, p.ldk = ptr(K), ldk if ldk is not None else K.shape[-1]
    p.V, p.ldv = ptr(V), ldv if ldv is not None else V.shape[-1]
    p.Vt, p.Kt, p.Qt, p.Lk_pad, p.Lq_pad = ptr(Vt), ptr(Kt), ptr(Qt), Lk_pad, Lq_pad
    p.key_tokens = ptr(key_tokens)
    p.B, p.Hn, p.Lq, p.Lk, p.causal, p.scale = B, Hn, Lq, Lk, int(causal), float(scale)
    p.drop_p, p.drop_thresh, p.drop_seed, p.drop_stream = drop.p, drop.thresh, drop.seed & 0xFFFFFFFF, drop.stream
    p.drop_seed_dev = ptr(drop.dev)
    return p


def attn_fwd(p, O, lse, ldo=None):
    p.O, p.ldo, p.lse = ptr(O), ldo if ldo is not None else O.shape[-1], ptr(lse)
    hip.call("cris_attn_fwd", C.byref(p), _stream())


def attn_bwd(p, O, lse, dO, dOt, delta, dQ, dK, dV, *, ldo=None, lddo=None, lddq=None, lddk=None, lddv=None):
    p.O, p.ldo, p.lse = ptr(O), ldo if ldo is not None else O.shape[-1], ptr(lse)
    p.dO, p.lddo, p.dOt, p.delta = ptr(dO), lddo if lddo is not None else dO.shape[-1], ptr(dOt), ptr(delta)
    p.dQ, p.lddq = ptr(dQ), lddq if lddq is not None else dQ.shape[-1]
    p.dK, p.lddk = ptr(dK), lddk if lddk is not None else dK.shape[-1]
    p.dV, p.lddv = ptr(dV), lddv if lddv is not None else dV.shape[-1]
    s = _stream()
    hip.call("cris_attn_bwd_dq", C.byref(p), s)
    hip.call("cris_attn_bwd_dkv", C.byref(p), s)


# ---- elementwise -------------------------------------------------------------------------------
def stem_im2col(img, out):
    Bn, _, H, W = img.shape
    hip.call("cris_stem_im2col", ptr(img), Bn, H, W, ptr(out), _stream())


def avgpool2_fwd(x, Bn, H, W, C_, y, ldx=None, xcoff=0, ldy=None, ycoff=0):
    hip.call("cris_avgpool2_fwd", ptr(x), ldx if ldx is not None else x.shape[-1], xcoff, Bn, H, W, C_, ptr(y),
             ldy if ldy is not None else y.shape[-1], ycoff, _stream())


def avgpool2_bwd(dy, Bn, H, W, C_, dx, lddy=None, dycoff=0, lddx=None, dxcoff=0, accum=False):
    hip.call("cris_avgpool2_bwd", ptr(dy), lddy if lddy is not None else dy.shape[-1], dycoff, Bn, H, W, C_, ptr(dx),
             lddx if lddx is not None else dx.shape[-1], dxcoff, int(accum), _stream())


def upsample2_fwd(x, Bn, H, W, C_, y, ldx=None, xcoff=0, ldy=None, ycoff=0):
    hip.call("cris_upsample2_fwd", ptr(x), ldx if ldx is not None else x.shape[-1], xcoff, Bn, H, W, C_, ptr(y),
             ldy if ldy is not None else y.shape[-1], ycoff, _stream())


def upsample2_bwd(dy, Bn, H, W, C_, dx, lddy=None, dycoff=0, lddx=None, dxcoff=0, accum=False):
    hip.call("cris_upsample2_bwd", ptr(dy), lddy if lddy is not None else dy.shape[-1], dycoff, Bn, H, W, C_, ptr(dx),
             lddx if lddx is not None else dx.shape[-1], dxcoff, int(accum), _stream())


def fill_coords(x, ldx, coff, nfill, Bn, H, W):
    hip.call("cris_fill_coords", ptr(x), ldx, coff, nfill, Bn, H, W, _stream())


def add_bf16(a, y, M, C_, b=None, lda=None, acoff=0, ldb=None, bcoff=0, ldy=None, ycoff=0):
    hip.call("cris_add_bf16", ptr(a), lda if lda is not None else a.shape[-1], acoff, ptr(b),
             (ldb if ldb is not None else (b.shape[-1] if b is not None else 0)), bcoff, ptr(y),
             ldy if ldy is not None else y.shape[-1], ycoff, M, C_, _stream())


def add_rowtable(a, table, trows, y, M, C_, lda=None, ldy=None):
    hip.call("cris_add_rowtable", ptr(a), lda if lda is not None else a.shape[-1], ptr(table), trows, ptr(y),
             ldy if ldy is not None else y.shape[-1], M, C_, _stream())


def cast_f32_bf16(x, y):
    hip.call("cris_cast_f32_bf16", ptr(x), ptr(y), x.numel(), _stream())


def cast_bf16_f32(x, y, accum=False):
    hip.call("cris_cast_bf16_f32", ptr(x), ptr(y), x.numel(), int(accum), _stream())


def cast_f32_bf16_drop(x, y, drop: Drop):
    hip.call("cris_cast_f32_bf16_drop", ptr(x), ptr(y), x.numel(), drop.p, drop.thresh, drop.seed & 0xFFFFFFFF, drop.stream,
             ptr(drop.dev), _stream())


def step_advance(step, seed, exchange_gen=None):
    hip.call("cris_step_advance", ptr(step), ptr(seed), ptr(exchange_gen), _stream())


def counter_advance(counter, skip=None):
    """counter[0] += 1 on the device (unless skip[0] != 0)"""
    hip.call("cris_counter_advance_unless", ptr(counter), ptr(skip), _stream())


def axpy_f32(dst, src, alpha=1.0):
    hip.call("cris_axpy_f32", ptr(dst), ptr(src), float(alpha), dst.numel(), _stream())


def quickgelu_fwd(x, y):
    hip.call("cris_quickgelu_fwd", ptr(x), ptr(y), x.numel(), _stream())


def quickgelu_bwd(x, dy, dx):
    hip.call("cris_quickgelu_bwd", ptr(x), ptr(dy), ptr(dx), x.numel(), _stream())


def embed_fwd(tokens, table, pos, out):
    Bn, L = tokens.shape
    hip.call("cris_embed_fwd", ptr(tokens), ptr(table), ptr(pos), Bn, L, table.shape[1], ptr(out), _stream())


def embed_bwd(tokens, dx, dtable, dpos, row_live=None):
    """row_live: optional uint8 [vocabulary], set to 1 for the tokens of this batch (sticky; see AdamTable row_live)"""
    Bn, L = tokens.shape
    hip.call("cris_embed_bwd", ptr(tokens), ptr(dx), Bn, L, dtable.shape[1], ptr(dtable), ptr(dpos), ptr(row_live), _stream())


def eot_gather(tokens, x, D, out, eot_index):
    Bn, L = tokens.shape
    hip.call("cris_eot_gather", ptr(tokens), ptr(x), Bn, L, D, ptr(out), ptr(eot_index), _stream())


def eot_scatter_add(eot_index, drows, Bn, L, D, dx):
    hip.call("cris_eot_scatter_add", ptr(eot_index), ptr(drows), Bn, L, D, ptr(dx), _stream())


# ---- the sentence-vector path in fp32 (csrc/smallf32.hip): at most SMALL_MAX_ROWS rows ------------------------------------------
SMALL_MAX_ROWS = 16


def eot_gather_ln_f32(tokens, x, mean, rstd, gamma, beta, D, out, eot_index):
    Bn, L = tokens.shape
    hip.call("cris_eot_gather_ln_f32", ptr(tokens), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), Bn, L, D, ptr(out), ptr(eot_index),
             _stream())


def eot_scatter_add_f32(eot_index, drows, Bn, L, D, dx):
    hip.call("cris_eot_scatter_add_f32", ptr(eot_index), ptr(drows), Bn, L, D, ptr(dx), _stream())


def linear_f32_small(A, W, out, *, w_is_kn=False, bias=None, accumulate=False):
    """out[M][N] (+)= A[M][K] @ W^T (+ bias) with W [N][K], or A @ W with W [K][N] (w_is_kn); fp32 throughout, M <= 16"""
    M, K = A.shape
    N = W.shape[1] if w_is_kn else W.shape[0]
    assert (W.shape[0] if w_is_kn else W.shape[1]) == K and out.shape == (M, N)
    hip.call("cris_linear_f32_small", ptr(A), A.stride(0), ptr(W), W.stride(0), int(w_is_kn), ptr(bias), M, N, K, ptr(out), out.stride(0),
             int(accumulate), _stream())


def outer_sum_f32_small(X1, X2, G, rowsum=None):
    """G[R][C] = sum_m X1[m][r] * X2[m][c]; rowsum[r] = sum_m X1[m][r]"""
    M, R = X1.shape
    Cc = X2.shape[1]
    assert X2.shape[0] == M and G.shape == (R, Cc)
    hip.call("cris_outer_sum_f32_small", ptr(X1), X1.stride(0), ptr(X2), X2.stride(0), M, R, Cc, ptr(G), G.stride(0), ptr(rowsum), _stream())


def colstats_f32_small(y, device) -> "Stats":
    M, C_ = y.shape
    st = Stats(1, C_, M, device)
    hip.call("cris_colstats_f32_small", ptr(y), y.stride(0), M, C_, ptr(st[0]), ptr(st[1]), _stream())
    return st


def bn_relu_f32_small(y, scale, shift, z):
    M, C_ = y.shape
    hip.call("cris_bn_relu_f32_small", ptr(y), y.stride(0), ptr(scale), ptr(shift), M, C_, ptr(z), z.stride(0), _stream())


def bn_relu_bwd_f32_small(dz, y, scale, shift, mean, invstd, sums, count, dy, *, between=None, link=None, local_sums=None):
    """backward of z = relu(bn(y)) over M <= 16 fp32 rows: one partial row -> cris_bn_bwd_sum(_sync) into `sums` (+= / exchanged,
    exactly as ops.bn_bwd does with its partial table) -> dy"""
    M, C_ = y.shape
    part = torch.empty(1, 2 * C_, dtype=torch.float32, device=y.device)
    s = _stream()
    hip.call("cris_bn_relu_bwd_reduce_f32_small", ptr(dz), dz.stride(0), ptr(y), y.stride(0), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
             M, C_, ptr(part), s)
    p = hip.BnBwdParams()
    p.part, p.sums, p.C = ptr(part), ptr(sums), C_
    if link is not None:
        hip.call("cris_bn_bwd_sum_sync", C.byref(p), 1, ptr(local_sums), C.byref(link), s)
    else:
        hip.call("cris_bn_bwd_sum", C.byref(p), 1, s)
        if between is not None:
            between(sums)
    hip.call("cris_bn_relu_bwd_apply_f32_small", ptr(dz), dz.stride(0), ptr(y), y.stride(0), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
             ptr(sums), float(count), M, C_, ptr(dy), dy.stride(0), s)


def posresize_fwd(R, pos, T, G, C_, posr):
    hip.call("cris_posresize_fwd", ptr(R), ptr(pos), T, G, C_, ptr(posr), _stream())


def posresize_bwd(R, dposr, T, G, C_, dpos):
    hip.call("cris_posresize_bwd", ptr(R), ptr(dposr), T, G, C_, ptr(dpos), _stream())


def batch_rowsum(dx, Bn, T, C_, out, ldx=None):
    hip.call("cris_batch_rowsum", ptr(dx), ldx if ldx is not None else dx.shape[-1], Bn, T, C_, ptr(out), _stream())


def dynconv_fwd(x, Bn, H, W, C_, wb, pred):
    hip.call("cris_dynconv_fwd", ptr(x), Bn, H, W, C_, ptr(wb), wb.shape[-1], ptr(pred), _stream())


def dynconv_bwd(x, dpred, Bn, H, W, C_, wb, dx, dwb):
    ws = torch.empty(hip.load().cris_dynconv_bwd_ws_floats(Bn, H, W, wb.shape[-1]), dtype=torch.float32, device=dx.device)
    hip.call("cris_dynconv_bwd", ptr(x), ptr(dpred), Bn, H, W, C_, ptr(wb), wb.shape[-1], ptr(dx), ptr(dwb), ptr(ws), _stream())


def mask_resize_nearest(mask, OH, OW, out):
    Bn, _, IH, IW = mask.shape
    hip.call("cris_mask_resize_nearest", ptr(mask), Bn, IH, IW, OH, OW, ptr(out), _stream())


def bce_fwd(logits, target, loss):
    ws = torch.empty(hip.load().cris_bce_ws_floats(), dtype=torch.float32, device=logits.device)
    hip.call("cris_bce_fwd", ptr(logits), ptr(target), logits.numel(), ptr(loss), ptr(ws), _stream())


def bce_bwd(logits, target, gscale, dlogits):
    hip.call("cris_bce_bwd", ptr(logits), ptr(target), logits.numel(), ptr(gscale), ptr(dlogits), _stream())


def train_metric(logits, target, Bn, HW, out, thr=0.35, pr_iou=0.5):
    hip.call("cris_train_metric", ptr(logits), ptr(target), Bn, HW, float(thr), float(pr_iou), ptr(out), _stream())


def zero_(t):
    """zero fill through the library: recorded / captured like every other launch"""
    hip.call("cris_zero_bytes", ptr(t), t.numel() * t.element_size(), _stream())
    return t


def zero_ranges(tensors):
    """zero several tensors (contiguous views) with one launch per CRIS_ZERO_RANGES_MAX of them"""
    for i in range(0, len(tensors), hip.ZERO_RANGES_MAX):
        chunk = tensors[i:i + hip.ZERO_RANGES_MAX]
        zr = hip.ZeroRanges()
        zr.n = len(chunk)
        for j, t in enumerate(chunk):
            zr.r[j].p, zr.r[j].nbytes = ptr(t), t.numel() * t.element_size()
        hip.call("cris_zero_many", C.byref(zr), _stream())


def torch_op(fn):
    """Run a torch-level op of the step now (stream wait, collective) and, while a step is being recorded, put it on the
    command list bound to the stream that is current now."""
    fn()
    rec = hip.RECORDER
    if rec is not None:
        s = torch.cuda.current_stream()

        def run():
            if torch.cuda.current_stream() == s:
                fn()
            else:
                with torch.cuda.stream(s):
                    fn()
        rec.cmds.append((run, None, "torch_op"))


class _AdamDeviceTable:
    def __init__(self, arr, n, total_blocks, device):
        self.arr, self.n, self.total_blocks = arr, n, total_blocks
        self.dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device) if n else None

    def upload(self, device):
        if self.n:
            self.dev = torch.frombuffer(bytearray(bytes(self.arr)), dtype=torch.uint8).to(device)


class AdamTable:
    """Device tables of {p, g, m, v, n, lr [, bf16 packs]}: torch.optim.Adam.step() over every tensor in two launches - the
    3x3 convolution weights whose bf16 operand copies are refreshed by the update (9-tap tiles: 74 KB of LDS per block) and
    everything else (1-tap packed weights and plain tensors)."""

    def __init__(self, params, grads, lrs, layouts=None, packs=None, row_live=None):
        """layouts[i]: None (gradient in the parameter layout) or (N, Cin, taps, Cpad) (GEMM layout, see cris_conv_wgrad).
        packs[i]: None or (dstF, dstD, N, Cin, taps, Cpad, Npad, transposed) - the bf16 GEMM-operand copies of tensor i
        (PackTable layouts) that the update rewrites from the new values.
        row_live: {i: uint8 tensor [rows of tensor i]} - rows whose byte is 0 have never had a gradient and are skipped
        (bit-identical to the dense update while weight_decay == 0; cris_adam_desc.row_live)."""
        lib = hip.load()
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.params, self.grads, self.lrs = list(params), list(grads), list(lrs)
        self.packs = list(packs) if packs is not None else [None] * len(self.params)
        self.device = self.params[0].device
        groups = {9: [], 1: []}
        for i in range(len(self.params)):
            pk = self.packs[i]
            groups[9 if (pk is not None and pk[4] == 9) else 1].append(i)
        self.index = groups
        self.tables = {}
        for taps, idx in groups.items():
            arr = (hip.AdamDesc * max(len(idx), 1))()
            start = 0
            for j, i in enumerate(idx):
                p, g = self.params[i], self.grads[i]
                d = arr[j]
                d.p, d.g, d.m, d.v, d.n, d.lr = ptr(p), ptr(g), ptr(self.m[i]), ptr(self.v[i]), p.numel(), self.lrs[i]
                lay = layouts[i] if layouts is not None else None
                if lay is not None and not (lay[2] == 1 and lay[3] == lay[1]):
                    d.taps, d.cin, d.cpad = lay[2], lay[1], lay[3]
                pk = self.packs[i]
                if pk is not None:
                    dstF, dstD, N, Cin, ptaps, Cpad, Npad, transposed = pk
                    assert ptaps in (1, 9), "packed weights have 1 or 9 taps"
                    assert d.taps == ptaps or (d.taps == 0 and ptaps == 1), "a 9-tap packed weight takes its gradient in the GEMM layout"
                    d.dstF, d.dstD = ptr(dstF), ptr(dstD)
                    d.N, d.cin, d.cpad, d.npad, d.transposed = N, Cin, Cpad, Npad, int(transposed)
                    d.ldF = dstF.shape[1] if dstF is not None else 0       # (row strides of the packs: ops.pack_row_stride)
                    d.ldD = dstD.shape[1] if dstD is not None else 0
                if row_live is not None and i in row_live:
                    assert pk is None and d.taps == 0 and p.dim() == 2 and row_live[i].numel() == p.shape[0]
                    d.row_live, d.row_len = ptr(row_live[i]), p.shape[1]
                d.block_start = start
                start += lib.cris_adam_blocks(C.byref(d))
            self.tables[taps] = _AdamDeviceTable(arr, len(idx), start, self.device)
        self.row_live = dict(row_live) if row_live else {}
        self.keep = [pk[:2] for pk in self.packs if pk is not None]
        self.step_count = 0

    def set_lrs(self, lrs):
        self.lrs = list(lrs)
        for taps, idx in self.index.items():
            t = self.tables[taps]
            for j, i in enumerate(idx):
                t.arr[j].lr = self.lrs[i]
            t.upload(self.device)

    @property
    def refreshes_packs(self):
        return any(pk is not None for pk in self.packs)

    def step(self, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, step_dev=None, loss_scale_dev=None, skip_dev=None):
        """step_dev: optional int32 device tensor holding the 1-based step count (graph replay); else a host counter.
        loss_scale_dev / skip_dev: GradScaler's scale and found_inf (device fp32 scalars): gradients are divided by the scale
        inside the update, a non-zero found_inf skips it (cris_adam_step_amp)"""
        self.step_count += 1
        bc1 = 1.0 - beta1 ** self.step_count
        bc2 = 1.0 - beta2 ** self.step_count
        for taps in (9, 1):
            t = self.tables[taps]
            if t.n:
                hip.call("cris_adam_step_amp", ptr(t.dev), t.n, t.total_blocks, beta1, beta2, eps, weight_decay, bc1, bc2,
                         grad_scale, ptr(step_dev), ptr(loss_scale_dev), ptr(skip_dev), taps, _stream())


class UnpackTable:
    """Device table for cris_unpack_grads: GEMM-layout gradients (srcs) -> parameter-layout tensors (dsts)."""

    def __init__(self, srcs, dsts, layouts):
        lib = hip.load()
        be = lib.cris_adam_block_elems()
        self.keep = (list(srcs), list(dsts))
        n = len(dsts)
        arr = (hip.AdamDesc * n)()
        start = 0
        for i, (s_, d_, lay) in enumerate(zip(srcs, dsts, layouts)):
            d = arr[i]
            d.p, d.g, d.n = ptr(d_), ptr(s_), d_.numel()
            d.taps, d.cin, d.cpad = lay[2], lay[1], lay[3]
            d.block_start = start
            start += (d_.numel() + be - 1) // be
        self.n, self.total_blocks = n, start
        self.dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dsts[0].device)

    def run(self):
        hip.call("cris_unpack_grads", ptr(self.dev), self.n, self.total_blocks, _stream())

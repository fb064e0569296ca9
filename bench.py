#!/usr/bin/env python
"""CRIS-R50 train-step throughput on MI355X (BASELINE.json metric: train-step samples/sec, CRIS-R50 416x416).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) the process IS a rank;
started plainly (`python bench.py --gpus 8`, no WORLD_SIZE) it re-launches itself under torch.distributed.run with N ranks on
127.0.0.1 and a free port, and relays rank 0's JSON line.

One step = the whole hot path on one synthetic RefCOCO-shaped batch already resident in HBM: weight repack, forward,
BCE loss, backward, (N > 1: SyncBN exchanges + overlapped gradient all-reduce over RCCL), fused Adam, train metric.
Per-GPU batch is fixed at 8 (global 8*N): weak scaling, the reference's own recipe (global 64 on 8 GPUs).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_SAMPLE = 395.0e9          # fwd+bwd, contractions only, R50/416/L17 (SURVEY.md 8d / BASELINE.md 2)
# the other BASELINE configurations (fwd+bwd 395.0 / 469.2 / 529.6 GFLOP per sample); anything else: None
FLOP_PER_SAMPLE_BY_CONFIG = {("r50", 416): 395.0e9, ("r101", 416): 469.2e9, ("r50", 480): 529.6e9}
ALG_BYTES_PER_SAMPLE = 0.98e9      # bf16 contraction operands+outputs under perfect fusion
ADAM_BYTES_PER_STEP = 4.11e9
MFMA_PEAK = 2500.0                 # TFLOP/s dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK = 8000.0                  # GB/s spec
DUAL_CEILING = {("r50", 416): 4305.0, ("r101", 416): 3678.0, ("r50", 480): 3164.0}   # samples/s per GPU: per-layer max(MFMA, HBM) roofline


_RESULT_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner when its
    first communicator comes up, from C stdio: buffered until the process exits, i.e. AFTER anything Python printed).  So a rank
    process keeps the real stdout for the result only: fd 1 is pointed at stderr for everything else, and emit_line() writes the line to
    the saved descriptor.  Not done in the process that merely re-launches itself under torch.distributed.run (spawn_ranks): its
    children inherit fd 1."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(record):
    line = (json.dumps(record) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def physical_cores():
    """physical cores of the host (lscpu: sockets x cores per socket; SURVEY.md 8d asks for physical, not logical, CPUs)"""
    import subprocess
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = dict((a.strip(), b.strip()) for a, b in (ln.split(":", 1) for ln in out.splitlines() if ":" in ln))
        n = int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])
        if n > 0:
            return n
    except Exception:               # noqa: BLE001
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def gpu_state(dev_index=0):
    """shader / memory clock, power and temperature of the GPU from rocm-smi (one call, ~0.1 s): logged at the start and the end
    of a run so that a throttled box can be told from a regression (profiles/r03_kernel_stats_slow_box.csv)"""
    import re
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True,
                             timeout=20).stdout
    except Exception as ex:         # noqa: BLE001
        return {"error": repr(ex)[:80]}
    st = {}
    for key, pat in (("sclk_mhz", r"sclk clock level: \S+ \((\d+)Mhz\)"), ("mclk_mhz", r"mclk clock level: \S+ \((\d+)Mhz\)"),
                     ("power_w", r"Power \(W\): ([0-9.]+)"), ("temp_junction_c", r"Sensor junction\) \(C\): ([0-9.]+)"),
                     ("temp_memory_c", r"Sensor memory\) \(C\): ([0-9.]+)")):
        m = re.search(pat, out)
        if m:
            st[key] = float(m.group(1))
    return st


def cpu_baseline(spec, batch, size, word_len, physical):
    """The CPU oracle (oracle/cris_oracle.py: fp32 restatement of the reference forward/loss, autograd backward) timed on the
    host cores.  kind "port": the unmodified reference (/root/reference) does not exist on the GPU box, only its restatement
    travels.  The port runs the way the reference's modules would: BatchNorm / LayerNorm through F.batch_norm / F.layer_norm
    (cris_oracle.NATIVE_NORMS; the hand-spelled forms exist for the bf16-emulation study) and WITHOUT the counter-hash dropout
    masks (numpy hashes of 3 x [64, 676, 676] probabilities per step are test infrastructure, not what a CPU user would run).
    (i) BASELINE.json configs[0]: eval forward of one image + expression at 8, 32 and all physical cores - a
    batch-1 forward does not scale over 128 threads (round 4 reported 1128 ms at 128 threads where 8 threads give ~230 ms) - the
    best is reported with its thread count; (ii) one train forward+backward at the bench batch, at 32 threads and at all
    physical cores, best reported.  A reported baseline, not the optimisation target."""
    from cris.pytorch_amd import arch, synth
    from oracle import cris_oracle as O
    clip, head = arch.specs_by_name(spec)
    sd = arch.synthetic_state_dict(clip, head, 0)
    img, word, mask = synth.make_batch(batch, size, word_len, 0, 0)
    img1, word1 = img[:1].contiguous(), word[:1].contiguous()
    sweep = sorted({t for t in (8, 32, physical) if 1 <= t <= physical} or {physical})
    O.NATIVE_NORMS = True
    try:
        evals = {}
        with torch.no_grad():
            for th in sweep:
                torch.set_num_threads(th)
                for _ in range(5):
                    O.cris_forward(sd, clip, head, img1, word1, None, training=False)
                t1 = time.time()
                n_eval = 20
                for _ in range(n_eval):
                    O.cris_forward(sd, clip, head, img1, word1, None, training=False)
                evals[th] = 1000.0 * (time.time() - t1) / n_eval
        leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        trains, warm = {}, {}
        for th in sorted({t for t in (32, physical) if t <= physical} or {physical}):
            torch.set_num_threads(th)
            dts = []
            for it in range(2):                  # one warm-up step (allocator, thread pool, autograd graph caches), one timed
                for v in leaf.values():
                    if v.is_floating_point():
                        v.grad = None
                t0 = time.time()
                _, _, loss = O.cris_forward(leaf, clip, head, img, word, mask, training=True, drop_seed=None)
                loss.backward()
                dts.append(time.time() - t0)
            trains[th], warm[th] = dts[-1], dts[0]
        # the reported figure: the MEDIAN of three more timed steps at the best thread count of the sweep (round-5 review: one
        # timed step is a thin sample) - ~10 s of CPU work at R50 / 416 / batch 8
        bt = min(trains, key=trains.get)
        torch.set_num_threads(bt)
        best_runs = [trains[bt]]
        for it in range(3):
            for v in leaf.values():
                if v.is_floating_point():
                    v.grad = None
            t0 = time.time()
            _, _, loss = O.cris_forward(leaf, clip, head, img, word, mask, training=True, drop_seed=None)
            loss.backward()
            best_runs.append(time.time() - t0)
    finally:
        O.NATIVE_NORMS = False
    med = sorted(best_runs)[len(best_runs) // 2]
    be = min(evals, key=evals.get)
    return {"value": batch / med, "unit": "samples/s", "cores": bt, "cores_are": "threads used (torch.set_num_threads) of %d physical cores (lscpu)" % physical,
            "kind": "port",
            "sample": "oracle port (the reference itself is absent on the GPU box), F.batch_norm / F.layer_norm, no dropout masks: 1 warm-up + 1 "
                      "timed train step (fwd+loss+bwd, fp32, no optimizer) at batch %d, %dx%d, L=%d per thread count %s -> seconds %s; then "
                      "the MEDIAN of 4 timed steps at the best count (%d threads: %s s) is `value`; eval forward bs=1: 5 warm-up + 20 "
                      "timed iterations per thread count"
                      % (batch, size, size, word_len, sorted(trains), {k: round(v, 2) for k, v in trains.items()}, bt, [round(x, 2) for x in best_runs]),
            "train_step_s_by_threads": {str(k): v for k, v in trains.items()}, "train_step_s_median": med, "train_step_s_runs": best_runs,
            "eval_forward_bs1_ms": evals[be], "eval_forward_bs1_threads": be,
            "eval_forward_bs1_ms_by_threads": {str(k): v for k, v in evals.items()}}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run this very command line under torch.distributed.run, one rank per GPU
    (the driver's form for N > 1 is exactly this command); rank 0 prints the JSON line, its stdout is ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def launch_check(rank, world, args):
    """The launch protocol alone (CPU, gloo): rendezvous, barrier on both sides of a 'timed region', MAX over ranks, one line."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0, float(rank)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit_line({"launch_check": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "max_rank_seen": int(t[1]), "seconds": float(t[0])})
    if world > 1:
        dist.destroy_process_group()


def module_path(args, rank, world, dev, optimizer_name=None, steps=None, warmup=None, ddp_one_rank=False, emit=True):
    """The drop-in module under the reference's own recipe (train.py:96-111) and loop body (engine/engine.py:37-73), timed with the
    same protocol as the native path.  What the unchanged engine imposes is inside the timed region: torch's Adam + GradScaler
    (unscale / inf check over every gradient), DDP's bucket copies, the bf16 operand re-pack every step (a torch optimizer
    changed the parameters), gradient export to `.grad`, trainMetricGPU with its three `.item()` host syncs.  Only the
    logging (meters, wandb) and the DataLoader are left out: batches are resident in HBM.
    `ddp_one_rank`: one process, but with the process group + SyncBatchNorm + DistributedDataParallel wrap train.py:80-102
    always builds (also on one GPU).  `emit=False`: return the record instead of printing it (the native bench line embeds
    these runs as `module_path`)."""
    from types import SimpleNamespace as NS
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    optimizer_name = args.optimizer if optimizer_name is None else optimizer_name
    own_pg = False
    if ddp_one_rank and world == 1 and not dist.is_initialized():
        import tempfile
        pg_dir = tempfile.mkdtemp(prefix="cris_bench_pg_")
        dist.init_process_group("nccl", init_method="file://" + os.path.join(pg_dir, "pg"), rank=0, world_size=1, device_id=dev)
        own_pg = True
    from torch import nn
    from cris.pytorch_amd import arch, synth
    from cris.pytorch_amd.model import build_segmenter
    assert args.spec == "r50", "--path module builds config/refcoco/cris_r50.yaml"
    word_len = args.word_len if args.word_len is not None else (22 if args.size == 480 else 17)
    cfg = NS(clip_pretrain="synthetic", word_len=word_len, fpn_in=[512, 1024, 1024], fpn_out=[256, 512, 1024], num_layers=3, vis_dim=512,
             num_head=8, dim_ffn=2048, dropout=0.1, intermediate=False, word_dim=1024, base_lr=1e-4, lr_multi=0.1, sync_bn=True)
    torch.manual_seed(1234)                                                               # (train.py:56-57 seeds from the yaml: the head is randomly initialised)
    model, param_list = build_segmenter(cfg)                                              # train.py:96
    if world > 1 or own_pg:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)                             # train.py:97-98
        model = nn.parallel.DistributedDataParallel(model.cuda(), device_ids=[dev.index], find_unused_parameters=True)   # :100-102
    else:
        model = model.cuda()
    if optimizer_name == "cris":
        from cris.pytorch_amd import optim as cris_optim                                   # the optional one-line change
        optimizer = cris_optim.Adam(param_list, lr=cfg.base_lr, weight_decay=0.0)
    else:
        optimizer = torch.optim.Adam(param_list, lr=cfg.base_lr, weight_decay=0.0)         # train.py:105-107
    scaler = torch.amp.GradScaler("cuda")                                                 # train.py:111
    nb = 4
    batches = [tuple(t.to(dev) for t in synth.make_batch(args.batch, args.size, word_len, rank, s)) for s in range(nb)]

    phases = ("forward", "zero_grad", "backward", "scaler.step", "scaler.update", "metric", "item")
    marks = []                                   # --phase-times: (host time, device event) at every phase boundary of every timed step

    def mark():
        if args.phase_times:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((time.perf_counter(), ev))

    def step(i):
        image, text, target = batches[i % nb]
        target = target if target.dim() == 4 else target.unsqueeze(1)
        mark()
        with torch.autocast("cuda"):                                                      # engine/engine.py:48
            pred, target, loss = model(image, text, target)
        mark()
        optimizer.zero_grad()
        mark()
        scaler.scale(loss).backward()
        mark()
        scaler.step(optimizer)
        mark()
        scaler.update()
        mark()
        o = (torch.sigmoid(pred.flatten(1)) >= 0.35)                                       # utils/misc.py:114-129
        t = target.flatten(1).bool()
        ious = (o & t).sum(1) / ((o | t).sum(1) + 1e-6)
        iou, pr5 = 100.0 * ious.mean(), 100.0 * (ious > 0.5).float().mean()
        l = loss.detach().clone()
        if world > 1:
            dist.all_reduce(l)
            dist.all_reduce(iou)
            dist.all_reduce(pr5)
        mark()
        r = l.item() / world, iou.item() / world, pr5.item() / world                      # the meters' .item() syncs (:67-69)
        mark()
        return r

    model.train()
    first = None
    for i in range(max(warmup, 3)):               # (first step eager, second captures the graphs / records the lists)
        r = step(i)
        first = first if first is not None else r[0]
    del marks[:]
    if args.pyprof:
        # where the HOST time of the loop body goes (the device idles while the host prepares a launch after a step's syncs)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(10):
            step(i)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
        del marks[:]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        r = step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    fused = bool(optimizer._usable()) if optimizer_name == "cris" else None
    inner = model.module if hasattr(model, "module") else model
    if own_pg:
        dist.destroy_process_group()
    if not emit:
        return {"ms_per_step": 1000.0 * dt / steps, "samples_per_s": world * args.batch * steps / dt, "steps": steps,
                "optimizer": "cris.pytorch_amd.optim.Adam" if optimizer_name == "cris" else "torch.optim.Adam", "fused_update": fused,
                "ddp_one_rank": bool(own_pg), "replay": os.environ.get("CRIS_MODULE_REPLAY", "graph"), "final_loss": r[0],
                "graph_error": getattr(inner, "graph_error", None)}
    phase_ms = None
    if args.phase_times and marks:
        # per phase: host time spent in it, and device time between the events recorded at its two ends (the device works
        # through what the host queued; a phase whose device time exceeds its host time ran behind the host, and vice versa)
        n = len(phases) + 1
        host, devt = [0.0] * len(phases), [0.0] * len(phases)
        for s0 in range(0, len(marks) - n + 1, n):
            for k in range(len(phases)):
                host[k] += 1e3 * (marks[s0 + k + 1][0] - marks[s0 + k][0])
                devt[k] += marks[s0 + k][1].elapsed_time(marks[s0 + k + 1][1])
        cnt = len(marks) // n
        phase_ms = {ph: {"host_ms": host[k] / cnt, "device_ms": devt[k] / cnt} for k, ph in enumerate(phases)}
    if rank == 0:
        sps = world * args.batch * steps / dt
        emit_line({
            "phase_times": phase_ms,
            "metric": "train-step samples/sec, CRIS-R50 416x416 bs=64; loss parity vs ref", "value": sps, "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1000.0 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "drop-in module cris.pytorch_amd.model.CRIS under the reference's recipe and loop body (torch Adam, "
                                   "GradScaler, fp16 autocast outside / bf16 HIP engine inside, trainMetricGPU + .item() syncs%s), "
                                   "CRIS-R50 %dx%d, per-GPU bs=%d, %d-token text, batches resident in HBM"
                                   % ("; SyncBatchNorm + DistributedDataParallel" if world > 1 else "", args.size, args.size, args.batch, word_len),
                       "path": "module", "replay": os.environ.get("CRIS_MODULE_REPLAY", "graph"), "optimizer": "cris.pytorch_amd.optim.Adam (fused update: %s)" % fused if optimizer_name == "cris" else "torch.optim.Adam(param_list, lr, weight_decay) as train.py:105 builds it (torch's fused implementation: %s - asked for by the parameter groups build_segmenter returns)" % bool(optimizer.param_groups[0].get("fused")), "ddp_one_rank": bool(own_pg),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world, "first_loss": first,
                       "final_loss": r[0], "grad_scale": float(scaler.get_scale()),
                       "graph_error": getattr(inner, "graph_error", None), "syncbn_exchange": getattr(inner, "syncbn_exchange", None),
                       "own_gradient_exchange": bool(getattr(inner, "_self_exchange", False)),
                       "gradient_exchange_in_this_run": inner._exchange_comm_for_this_step() is not None,
                       "ddp_managed_parameters": len(model._module_parameters) if hasattr(model, "_module_parameters") else None},
            "step_roofline": {"mfma_frac": sps / world * FLOP_PER_SAMPLE_BY_CONFIG.get((args.spec, args.size), FLOP_PER_SAMPLE) / (MFMA_PEAK * 1e12)}})
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()


def module_path_subprocess(args, optimizer_name):
    """`bench.py --path module --ddp-one-rank` in a process of its own; returns the record the in-process runs return"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--path", "module", "--ddp-one-rank", "--optimizer", optimizer_name, "--steps",
           str(args.module_steps), "--warmup", "5", "--batch", str(args.batch), "--size", str(args.size), "--spec", args.spec]
    if args.word_len is not None:
        cmd += ["--word-len", str(args.word_len)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)     # (a healthy run takes ~25 s)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
    d = json.loads(lines[0])                           # (a line that was printed is a finished measurement, whatever the teardown did)
    return {"ms_per_step": d["ms_per_step"], "samples_per_s": d["value"], "steps": d["steps"], "optimizer": d["config"]["optimizer"],
            "ddp_one_rank": d["config"].get("ddp_one_rank"), "replay": d["config"].get("replay"), "final_loss": d["config"].get("final_loss"),
            "own_process": True, "exit_code": r.returncode, "graph_error": d["config"].get("graph_error"),
            "own_gradient_exchange": d["config"].get("own_gradient_exchange"), "ddp_managed_parameters": d["config"].get("ddp_managed_parameters")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spec", default="r50")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--word-len", type=int, default=None, help="tokens per expression (default 17; 22 with --size 480 = BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the Python schedule (no graph / command list)")
    ap.add_argument("--launch", default=None, choices=["graph", "cmdlist", "eager"], help="default: graph (N > 1: falls back to cmdlist if the capture fails on any rank)")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"],
                    help="N > 1: torch = torch.distributed process groups (RCCL through c10d; the default: the only one that has run on "
                         "more than one rank - two ranks over gloo); native = dist.RcclComm, communicators owned by libcris_hip.so "
                         "(cris_comm_*: exercised with one rank only so far)")
    ap.add_argument("--grad-exchange", default=None, choices=["rccl", "p2p"],
                    help="N > 1: rccl = eight staged all-reduces through torch.distributed (the default); p2p = the direct reduce-scatter + "
                         "all-gather over the peer-mapped gradient arenas (csrc/p2p.hip; sets CRIS_GRAD_EXCHANGE: opt-in, never run on "
                         "more than one GPU)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets two ranks share one GPU in tests)")
    ap.add_argument("--shape-table", default=None, help="write the per-shape GEMM timing table (tsv) here")
    ap.add_argument("--path", default="native", choices=["native", "module"],
                    help="native: NativeTrainer (one captured HIP graph per step; the default bench line).  module: the path the north "
                         "star names - build_segmenter(args) -> torch.optim.Adam over its two groups -> GradScaler, driven by the "
                         "reference's loop body (engine/engine.py:37-73: fp16 autocast, scaled backward, scaler.step / update, "
                         "trainMetricGPU + .item() syncs); N > 1: SyncBatchNorm + DistributedDataParallel as train.py:97-102")
    ap.add_argument("--optimizer", default="torch", choices=["torch", "cris"],
                    help="--path module: torch = torch.optim.Adam as train.py:105 builds it (the unchanged loop); cris = "
                         "cris.pytorch_amd.optim.Adam, the optional one-line replacement whose step() is the library's fused update")
    ap.add_argument("--ddp-one-rank", action="store_true",
                    help="--path module with --gpus 1: build the process group + SyncBatchNorm + DistributedDataParallel wrap that train.py:80-102 "
                         "always builds, with one rank")
    ap.add_argument("--no-module-path", action="store_true",
                    help="native path, N = 1: skip the four short runs of the drop-in module (unchanged loop / optional optimizer, bare / "
                         "under a one-rank DistributedDataParallel) whose step times the bench line carries as `module_path`")
    ap.add_argument("--module-steps", type=int, default=20, help="timed steps of each `module_path` run")
    ap.add_argument("--pyprof", action="store_true", help="--path module: cProfile of ten loop bodies (stderr) before the timed region")
    ap.add_argument("--phase-times", action="store_true", help="--path module: host and device time of every phase of the loop body")
    ap.add_argument("--launch-check", action="store_true",
                    help="exercise only the multi-rank launch protocol (spawn, rendezvous, barrier, max-over-ranks, one JSON "
                         "line from rank 0) without touching a GPU - what tests/test_bench_launch.py runs on the CPU")
    args = ap.parse_args()

    if args.grad_exchange is not None:
        os.environ["CRIS_GRAD_EXCHANGE"] = args.grad_exchange
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d (torch.distributed.run --nproc-per-node must equal --gpus)" % (world, args.gpus))
    if args.path != "module":
        # (the `--path module` form prints its line the plain way: as a child of the default run its stdout is a pipe the parent
        # scans for the JSON line anyway)
        claim_stdout()
    if args.launch_check:
        return launch_check(rank, world, args)
    ngpu = torch.cuda.device_count()
    if ngpu == 0:
        sys.exit("bench.py: no GPU visible - the HIP path has no CPU fallback")
    if world > ngpu and args.backend == "nccl":
        sys.exit("bench.py: %d ranks but %d GPUs: RCCL needs one GPU per rank (ranks may share a GPU only with --backend gloo)" % (world, ngpu))
    local = local % ngpu
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        from cris.pytorch_amd.dist import RcclComm, TorchDistComm
        if args.comm == "native" and args.backend == "nccl":
            store = dist.distributed_c10d._get_default_store()
            comm = RcclComm(rank, world, dev, store)
        else:
            comm = TorchDistComm(dev)

    import __graft_entry__ as g
    g.build()
    from cris.pytorch_amd import arch, synth, ops
    from cris.pytorch_amd.trainer import NativeTrainer
    if args.path == "module":
        return module_path(args, rank, world, dev, ddp_one_rank=args.ddp_one_rank)

    import dataclasses
    clip, head = arch.specs_by_name(args.spec)
    word_len = args.word_len if args.word_len is not None else (22 if args.size == 480 else head.word_len)
    head = dataclasses.replace(head, word_len=word_len)
    sd = arch.synthetic_state_dict(clip, head, 0)
    smi0 = gpu_state(local) if rank == 0 else None
    tr = NativeTrainer(clip, head, sd, dev, comm=comm, sync_bn=world > 1, use_graph=not args.no_graph, launch=args.launch)
    del sd
    nb = 4
    batches = [tuple(t.to(dev) for t in synth.make_batch(args.batch, args.size, head.word_len, rank, s)) for s in range(nb)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not a benchmark step: the launch schedule is captured (HIP graph) / recorded (command list) by running it -
    # one eager pass that fills the host-side caches, one captured pass - so that even --warmup 0 times replays only
    for i in range(2):
        tr.train_step(*batches[i % nb])
    first_loss = None
    for i in range(args.warmup):
        l0, _ = tr.train_step(*batches[i % nb])
        if first_loss is None:
            first_loss = float(l0)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = tr.train_step(*batches[i % nb])
    sync()
    dt = time.perf_counter() - t0
    loss_v = float(loss)
    smi1 = gpu_state(local) if rank == 0 else None
    # per-kernel HIP-event timing needs individual launches: an extra instrumented EAGER pass right after the timed
    # region (the timed region itself replays the captured HIP graph - one host call per step)
    timer, timer_steps = None, 0
    if not args.no_kernel_timer:                 # every rank runs the pass (its collectives must match across ranks)
        timer = ops.KernelTimer()
        ops.KERNEL_TIMER = timer
        timer_steps = min(3, args.steps)
        for i in range(timer_steps):
            tr.train_step(*batches[i % nb])
        torch.cuda.synchronize()
        ops.KERNEL_TIMER = None
    if world > 1:
        dist.barrier()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    if rank == 0:
        sps = world * args.batch * args.steps / dt
        steps_s = args.steps / dt
        out = {
            "metric": "train-step samples/sec, CRIS-R50 416x416 bs=64; loss parity vs ref",
            "value": sps, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "CRIS-%s bf16 training step (fwd+BCE+bwd+Adam), %dx%d, per-GPU bs=%d, %d-token text, "
                                   "synthetic RefCOCO-shape batch resident in HBM (BASELINE.json configs[1]%s)"
                                   % (args.spec.upper(), args.size, args.size, args.batch, head.word_len,
                                      "" if world == 1 else "; x%d GPUs = configs[2] recipe: SyncBN + gradient all-reduce over RCCL" % world),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world, "first_loss": first_loss, "final_loss": loss_v,
                       "launch": tr.launch, "graph_captured": tr._graph is not None, "graph_error": tr.graph_error,
                       "syncbn_exchange": tr.syncbn_exchange, "grad_exchange": tr.grad_exchange, "comm": type(comm).__name__ if comm is not None else None,
                       "syncbn_peer_timeout": (bool(int(comm.p2p.err.item())) if getattr(comm, "p2p", None) is not None else None),
                       "source_commit": (open(os.path.join(ROOT, ".source_commit")).read().strip()
                                         if os.path.exists(os.path.join(ROOT, ".source_commit")) else None),
                       "gpu_state_start": smi0, "gpu_state_end": smi1},
            "step_roofline": {"mfma_frac": (sps / world * FLOP_PER_SAMPLE_BY_CONFIG[(args.spec, args.size)] / (MFMA_PEAK * 1e12)
                                            if (args.spec, args.size) in FLOP_PER_SAMPLE_BY_CONFIG else None),
                              "flop_per_sample": FLOP_PER_SAMPLE_BY_CONFIG.get((args.spec, args.size)),
                              # (the algorithmic byte count exists for the benchmarked configuration only)
                              "hbm_frac_alg": ((sps / world * ALG_BYTES_PER_SAMPLE + steps_s * ADAM_BYTES_PER_STEP) / (HBM_PEAK * 1e9)
                                               if (args.spec, args.size) == ("r50", 416) else None)},
        }
        if (args.spec, args.size) in DUAL_CEILING and args.batch == 8:
            # the per-layer dual (MFMA / HBM) roofline of SURVEY.md 8d for this configuration at 8 samples per GPU
            out["step_roofline"]["dual_ceiling_samples_s_per_gpu"] = DUAL_CEILING[(args.spec, args.size)]
            out["step_roofline"]["dual_ceiling_frac"] = sps / world / DUAL_CEILING[(args.spec, args.size)]
        if timer is not None:
            summ = timer.summary()
            dom = max(summ, key=lambda k: summ[k]["ms"])
            d = summ[dom]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": MFMA_PEAK, "unit": "TFLOP/s",
                               "frac": ach / MFMA_PEAK, "traffic": None,
                               "launches_per_step": d["launches"] / timer_steps, "avg_launch_us": 1000.0 * d["ms"] / d["launches"],
                               "share_of_step": (d["ms"] / timer_steps) / (1000.0 * dt / args.steps)}
            # HBM bytes per launch of the same kernel from the PMC passes of tools/gpu_pmc.sh (rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE in separate runs, FETCH doubled per the gfx950 correction): a committed measurement, not taken live
            # ONE set of launches for both figures: every launch of a forward / input-gradient TILE kernel (the convolution
            # GEMMs and the linears of M > 16 rows that run on them) - what the PMC summary's family entry covers.  Per launch
            # both are averages over that set; traffic_ratio compares the two per STEP (a grouped launch is one launch of
            # several problems, so launch counts of different builds are not comparable, bytes per step are)
            fam = timer.tile_family()
            for pmf in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json"):
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", pmf)))
                    pm = pj["kernels"]["conv_gemm (all tile kernels)"]
                    pmc_steps = pj.get("steps") or 5          # (r03's file: 2 set-up + 1 warm-up + 2 timed steps)
                    out["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                    # (a constant read from a committed file, NOT a measurement of this run: `traffic_commit` = the commit the PMC
                    # passes were taken at, as the file records it)
                    out["roofline"]["traffic_commit"] = pj.get("commit")
                    out["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc, eager launches; average over the %d launches per step of the tile kernels)" % (pmf, round(pm["launches"] / pmc_steps))
                    out["roofline"]["algorithmic_bytes_per_launch"] = fam["bytes"] / max(fam["launches"], 1)
                    out["roofline"]["algorithmic_launch_set"] = "the %d launches per step of the tile kernels in this run" % round(fam["launches"] / timer_steps)
                    t_step, a_step = pm["traffic_bytes_per_launch"] * pm["launches"] / pmc_steps, fam["bytes"] / timer_steps
                    out["roofline"]["traffic_per_step"], out["roofline"]["algorithmic_bytes_per_step"] = t_step, a_step
                    out["roofline"]["traffic_ratio"] = t_step / a_step
                    break
                except Exception:               # noqa: BLE001
                    pass
            # where the parity measurements of this path are (a pointer, not a measurement of this run)
            out["config"]["parity"] = "profiles/parity_r05.md + profiles/r06/teacher_forced_r50_fp64_call_e_final.json (float64-teacher TEACHER-FORCED runs of configs[1] / [3] / [4] inside pytest -m gpu: mean abs(dloss) over the 100 states of configs[1] 9.15e-4, bound 1.0e-3; the free-running curve is gpu_long and is NOT within 1e-3), profiles/parity_r04.md (per-stage bf16 error budget), tests/test_parity_long_gpu.py, tests/golden/grad_cos_r50_config1.json (ten worst gradient tensors, +-0.02 bands)"
            out["roofline"]["timing"] = "HIP events around each launch, %d-step eager pass after the timed region" % timer_steps
            out["kernels"] = {k: {"ms_per_step": v["ms"] / timer_steps, "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                  "launches_per_step": v["launches"] / timer_steps} for k, v in summ.items()}
        if timer is not None and args.shape_table:
            rows = sorted(timer.by_shape().items(), key=lambda kv: -kv[1]["ms"])
            with open(args.shape_table, "w") as f:
                f.write("kernel\tshape\tlaunches/step\tms/step\tus/launch\tTFLOP/s\talgGB/s\n")
                for (kn, tag), v in rows:
                    f.write("%s\t%s\t%.1f\t%.3f\t%.1f\t%.1f\t%.0f\n" % (
                        kn, tag, v["launches"] / timer_steps, v["ms"] / timer_steps, 1e3 * v["ms"] / v["launches"],
                        v["flops"] / (v["ms"] * 1e-3) / 1e12, v["bytes"] / (v["ms"] * 1e-3) / 1e9))
        if world == 1 and not args.no_module_path and args.spec == "r50":
            # the path the north star names - build_segmenter -> optimizer -> GradScaler under the reference's loop body - in the
            # same record (round-4 review: "neither number is in a driver record"): the unchanged loop, the one-line optimizer
            # swap, and both again under the one-rank DistributedDataParallel wrap the reference's train.py always builds
            del tr
            torch.cuda.empty_cache()
            mp = {}
            for key, opt_name, ddp1 in (("unchanged_loop", "torch", False), ("cris_optimizer", "cris", False),
                                        ("unchanged_loop_ddp_one_rank", "torch", True), ("cris_optimizer_ddp_one_rank", "cris", True)):
                try:
                    if ddp1:
                        # a process of its own: RCCL prints its version banner on stdout when a communicator comes up, and this
                        # process's stdout carries ONE JSON line
                        mp[key] = module_path_subprocess(args, opt_name)
                    else:
                        mp[key] = module_path(args, rank, world, dev, optimizer_name=opt_name, steps=args.module_steps, warmup=5,
                                              ddp_one_rank=False, emit=False)
                except Exception as ex:          # noqa: BLE001 - the native line must not depend on these runs
                    mp[key] = {"error": repr(ex)[:300]}
                torch.cuda.empty_cache()
            # a broken drop-in must not hide behind a green native line: every failed run is an `error` entry AND counts here
            mp["module_path_rc"] = sum(1 for v in mp.values() if isinstance(v, dict) and ("error" in v or v.get("exit_code") not in (None, 0)))
            mp["ms_per_step"] = mp["unchanged_loop"].get("ms_per_step")
            mp["ms_per_step_cris_optimizer"] = mp["cris_optimizer"].get("ms_per_step")
            mp["what"] = ("cris.pytorch_amd.model.CRIS under the reference's loop body (engine/engine.py:37-73: fp16 autocast, GradScaler, "
                          "trainMetricGPU + three .item() syncs), %d timed steps each, same batch shape as the native line" % args.module_steps)
            out["module_path"] = mp
            out["module_path_rc"] = mp["module_path_rc"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.spec, args.batch, args.size, head.word_len, physical_cores())
        emit_line(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

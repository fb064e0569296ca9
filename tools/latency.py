"""Inference speed of the HIP path - the measurement of the reference's tools/latency.py:38-72 (one 416x416 image + one
expression, 500 iterations of which the first 100 are warm-up, FPS and memory) on cris.pytorch_amd.infer.InferenceRunner,
plus the throughput at larger batches.  Prints one JSON line.
    python tools/latency.py [--spec r50] [--size 416] [--word-len 17] [--batches 1,8,32] [--iters 500]"""
import argparse
import dataclasses
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import arch                       # noqa: E402
from cris.pytorch_amd.infer import InferenceRunner      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spec", default="r50")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--word-len", type=int, default=17)
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--modes", default="fold+graph,nofold+graph,fold+eager")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    clip, head = arch.specs_by_name(args.spec)
    head = dataclasses.replace(head, word_len=args.word_len)
    sd = arch.synthetic_state_dict(clip, head, 0)
    out = {"spec": args.spec, "size": args.size, "word_len": args.word_len, "iters": args.iters, "warmup": args.iters // 5,
           "params_M": round(sum(v.numel() for k, v in sd.items() if v.is_floating_point() and "running_" not in k) * 1e-6, 2),
           "runs": []}
    for mode in args.modes.split(","):
        fold, graph = mode.split("+")[0] == "fold", mode.split("+")[1] == "graph"
        mem0 = torch.cuda.max_memory_allocated()
        r = InferenceRunner(clip, head, sd, dev, fold_bn=fold, use_graph=graph)
        for b in (int(x) for x in args.batches.split(",")):
            img = torch.randn(b, 3, args.size, args.size, device=dev)                 # tools/latency.py:51-52
            word = torch.randint(1, 4096, (b, args.word_len), device=dev).long()
            warm = args.iters // 5
            # (a) the reference's protocol: host clock around every call, device sync per iteration
            t_sum = 0.0
            for i in range(args.iters):
                t0 = time.time()
                r(img, word)
                torch.cuda.synchronize()
                if i >= warm:
                    t_sum += time.time() - t0
            lat = t_sum / (args.iters - warm)
            # (b) back-to-back replays, one sync at the end
            torch.cuda.synchronize()
            t0 = time.time()
            for i in range(args.iters - warm):
                r(img, word)
            torch.cuda.synchronize()
            thr = (time.time() - t0) / (args.iters - warm)
            out["runs"].append({"mode": mode, "batch": b, "latency_ms": round(lat * 1e3, 3), "fps_synced": round(b / lat, 1),
                                "ms_back_to_back": round(thr * 1e3, 3), "samples_per_s": round(b / thr, 1),
                                "graph_error": r.graph_error})
        out["peak_mem_GB_" + mode] = round((torch.cuda.max_memory_allocated() - mem0) / 1.073742e9, 2)
        del r
    print("LATENCY " + json.dumps(out))


if __name__ == "__main__":
    main()

"""Writes the per-tensor gradient-cosine table of BASELINE.json configs[1] (R50, 416x416, batch 8, state 0, dropout 0, seed 3 -
the inputs of tests/test_engine_gpu.py::test_config1_r50_416_batch8_step_matches_oracle): HIP path against the fp32 CPU oracle,
the ten worst tensors.  Run on the GPU box; the output is committed as tests/golden/grad_cos_r50_config1.json and read by the
test with a +-0.02 band per tensor.   python tools/grad_cos_table.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cris.pytorch_amd import selfcheck      # noqa: E402

rep = selfcheck.run("r50", batch=8, size=416, dropout=0.0, seed=3, return_all_cos=True)
allcos = rep.pop("grad_cos_all")
stamp = os.path.join(ROOT, ".source_commit")
out = {"what": "gradient cosine per parameter tensor, HIP path vs fp32 CPU oracle, CRIS-R50 416x416 batch 8, synthetic state 0, dropout 0, seed 3",
       "commit": open(stamp).read().strip() if os.path.exists(stamp) else None, "band": 0.02,
       "worst10": [[n, round(c, 6)] for n, c in sorted(allcos.items(), key=lambda kv: kv[1])[:10]],
       "median": rep["grad_cos_median"], "n_tensors": rep["n_grads"], "loss_hip": rep["loss_hip"], "loss_oracle": rep["loss_oracle"],
       "below_0.95": sum(1 for c in allcos.values() if c < 0.95), "below_0.99": sum(1 for c in allcos.values() if c < 0.99)}
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "grad_cos_r50_config1.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Per-layer forward / backward call times of the bf16-policy two-EPS models (diagnostic):  python tools/layers_bf16.py [cfg3a_bf16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3a_bf16"
dev = torch.device("cuda:0")
bench.cpu_baseline_eps_model = lambda *a, **k: None
e = bench.extra_eps_model(name, dev, 10)
print(json.dumps({k: e[k] for k in ("workload", "ms_per_step", "fwd_ms")}))
for l in e["layers"]:
    print(json.dumps(l))

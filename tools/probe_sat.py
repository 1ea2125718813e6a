"""Time the processor attention forward at batch 256 (bench.roofline_probe) under tile-shape overrides:
   PIT_FORCE_RT=2 PIT_FORCE_TPWG=8 python tools/probe_sat.py"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch, bench
from position_induced_transformer_amd import tasks
model, _, _ = tasks.make_task("darcy", seed=0)
r = bench.roofline_probe(model, 256)
print(os.environ.get("PIT_FORCE_RT"), os.environ.get("PIT_FORCE_TPWG"), r["us_per_launch"], r["frac"])

"""BASELINE.json config 4 (Mixtral-8x7B Q4_K_M expert path) as a stand-alone run: python tools/bench_mixtral.py [--layers 32]"""
import argparse, json, sys
sys.path.insert(0, ".")
from llamafile_amd import sgemm, mixtral_bench

p = argparse.ArgumentParser()
p.add_argument("--layers", type=int, default=32)
p.add_argument("--iters", type=int, default=50)
p.add_argument("--prefill", type=int, default=512)
a = p.parse_args()
sgemm.init(0)
print(json.dumps(mixtral_bench.run(a.layers, a.iters, a.prefill)))

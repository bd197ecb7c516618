#!/usr/bin/env python3
"""BASELINE configs[4]: inference-only predict path at batch 1024 -- the same leg bench.py reports as `predict`
(bench.predict_leg), stand-alone.  Prints one JSON line.  usage: predict_bench.py [--batch 1024] [--iters 20] [--precision bf16s]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "crnn-ocr-lite_amd")]

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--precision", default="bf16s")
ap.add_argument("--cpu-sample", type=int, default=32)
args = ap.parse_args()

from bench import predict_leg
print(json.dumps(predict_leg(args.batch, args.iters, args.precision, args.cpu_sample)))

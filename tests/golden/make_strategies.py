#!/usr/bin/env python
"""make_strategies.py genstrat.json out.json max_block pre_gap pre_min expectation_scale

Derives the strategies files of the bkzs_* fixtures from the output of `ref_driver genstrat`
(pruning coefficients from the reference's own pruner): block sizes up to max_block, preprocessing
block size [b - pre_gap] for b >= pre_min, every expectation multiplied by expectation_scale (a
scale < 1 pushes the success probability of a single enumeration below
BKZ_DEF_MIN_SUCCESS_PROBABILITY, so that svp_reduction loops and rerandomises: bkz.cpp:299-345)."""
import json
import sys

src, dst, max_b, gap, pre_min, scale = sys.argv[1:7]
max_b, gap, pre_min, scale = int(max_b), int(gap), int(pre_min), float(scale)
out = []
for s in json.load(open(src)):
    b = s["block_size"]
    if b > max_b:
        continue
    out.append({
        "block_size": b,
        "preprocessing_block_sizes": [b - gap] if b >= pre_min else [],
        "pruning_parameters": [[p[0], p[1], max(1e-9, min(1.0, p[2] * scale))]
                               for p in s["pruning_parameters"]],
    })
json.dump(out, open(dst, "w"))

#!/usr/bin/env python3
"""Stamps a PMC summary with the hash of the kernel sources it was taken from (bench.py quotes `roofline.traffic` only from a
summary whose stamp names the sources of the tree it runs in).  usage: pmc_stamp.py <summary.csv>  -> <summary>.json"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
out = sys.argv[1][:-4] + ".json"
json.dump({"kernel_source_sha16": bench.kernel_source_hash(), "taken": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
           "command": "scripts/pmc_passes.sh (bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dropin --serial --no-graph)"}, open(out, "w"))
print(out)

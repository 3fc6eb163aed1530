#!/usr/bin/env python3
"""Parses the output of tools/_bin/inst_probe into {waves per SIMD: {form: ns per wave-instruction per SIMD}}.
    python tools/inst_costs.py gpurun_out/<tag>_inst_probe.log > gpurun_out/<tag>_inst_costs.json"""
import json
import re
import sys

table, current = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"== (\d+) wave", line)
    if m:
        current = table.setdefault(m.group(1), {})
        continue
    m = re.match(r"(\S+)\s+[\d.]+ ms = [\d.]+\s+\(([\d.]+) ns", line)
    if m and current is not None:
        current[m.group(1)] = float(m.group(2))
json.dump(table, sys.stdout, indent=1)

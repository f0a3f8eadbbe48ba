#!/usr/bin/env python3
"""Run the REFERENCE's own CSV consumer (/root/reference/plot_results.py:3-46, parse_results) in this
container on CSV files our driver wrote, and store what it returned. Data only is committed:
  benchmark_results_mi355x.csv   written by driver/fa_driver on an MI355X (gpurun)
  csv_synthetic.csv              hand-made edge cases (naive = 0 row, short row, junk row, extra columns)
  csv_expected.json              parse_results() outputs for both
"""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_plot", "/root/reference/plot_results.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

SYN = """Using device: x
N,Naive(ms),Flash(ms),FlashV2(ms),FlashV3(ms),FlashV4(ms),SpeedupV1,SpeedupV2,SpeedupV3,SpeedupV4
128,1.0,2.0,0.5,0.9,1.4,0.5,2,1.11,0.71
256,4.0,3.0,1.0,2.0,2.2,1.33,4,2,1.8,EXTRA,COLUMNS,12.5
512,1,2,3
junk,1,2,3,4,5,6,7,8,9

1024,8.0,0,2.0,4.0,1.0,0,4,2,8
16384,0,30.0,20.0,10.0,5.0,0,0,0,0
"""
open(os.path.join(HERE, "csv_synthetic.csv"), "w").write(SYN)
out = {}
for name in ("benchmark_results_mi355x.csv", "csv_synthetic.csv"):
    out[name] = [list(x) for x in ref.parse_results(os.path.join(HERE, name))]
json.dump(out, open(os.path.join(HERE, "csv_expected.json"), "w"), indent=1)
print({k: [len(x) for x in v] for k, v in out.items()})
ref.generate_svg(*ref.parse_results(os.path.join(HERE, "benchmark_results_mi355x.csv")), filename="/tmp/speedup_plot_test.svg")
print("reference generate_svg ok:", os.path.getsize("/tmp/speedup_plot_test.svg"), "bytes")

"""Pretty-print a bench.py JSON line: python tools/show_bench.py file.json [...]"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{f}: {d['value']} img/s  {d['ms_per_step']} ms/step  conv family {r['achieved']} {r['unit']} ({r['frac'] * 100:.2f}% of the {r['bound']} roof)"
          f"  whole step {d['whole_step']['frac_of_mfma_peak'] * 100:.2f}%  eval fwd {d['forward_eval']['ms_per_batch']} ms ({d['forward_eval']['frac_of_mfma_peak'] * 100:.2f}%)")
    for k, v in d["kernel_classes"].items():
        print(f"   {k:12s} {v['ms_per_step']:8.3f} ms  {v['launches_per_step']:4d} launches  tflops {v['tflops']}  alg GB/s {v['algorithmic_gbs']}")
    if "cpu_baseline" in d:
        print("   cpu_baseline:", d["cpu_baseline"])

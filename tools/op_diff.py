"""Which main-chain kernels does the weight-gradient stream slow down?  Joins two tools/op_profile.py tables (with / without the
weight-gradient launches: CVX_TUNE_SKIP_WGRAD=1 on the tuning library) per (class, op).   python tools/op_diff.py with.txt without.txt"""
import collections
import re
import sys


def load(f):
    d = {}
    for line in open(f):
        p = line.split()
        if len(p) < 6 or p[0] not in ("conv_dgrad", "bn_bwd", "conv_wgrad", "misc", "conv_fwd", "bn_fwd"):
            continue
        try:
            op = int(p[1])
        except ValueError:
            continue
        d[(p[0], op)] = (float(p[-5]), " ".join(p[2:-5]))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(((a[k][0] - b[k][0], k, a[k][0], b[k][0], a[k][1]) for k in a if k in b and k[0] in ("conv_dgrad", "bn_bwd", "misc")), reverse=True)
print("delta us  class       op   with  without  shape")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%7.1f  %-10s %3d %6.1f %7.1f  %s" % (r[0], r[1][0], r[1][1], r[2], r[3], r[4]))
print("sum of deltas %.1f us over %d kernels" % (sum(r[0] for r in rows), len(rows)))
by = collections.defaultdict(float)
for r in rows:
    m = re.search(r"(\d+)x(\d+)", r[4])
    by[m.group(1) if m else "?"] += r[0]
print("by input resolution:", {k: round(v, 1) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})
print("\nweight-gradient launches (us inside the step):")
for k in sorted((k for k in a if k[0] == "conv_wgrad"), key=lambda k: -a[k][0])[:24]:
    print("  %3d %7.1f  %s" % (k[1], a[k][0], a[k][1]))

"""Per-shape device time of the GEMM-shaped conv kernel from a rocprofv3 kernel trace of tools/gemm_probe.py (13 calls per shape: 3 warm-up,
10 timed): the probe's own event timing includes the unit entry's tap upload and stream synchronisation, the trace does not.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gemm_trace -- python3 tools/gemm_probe.py
    python tools/gemm_trace.py gpurun_out/gemm_trace
"""
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tools.gemm_probe import SHAPES  # noqa: E402


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    conv = [r for r in rows if "conv_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
    pack = [r for r in rows if "gemm_pack" in r["Kernel_Name"]]
    per = len(conv) // len(SHAPES)
    pper = len(pack) // len(SHAPES) if pack else 0
    for i, (B, H, W, Ci, Co, k, s, what) in enumerate(SHAPES):
        mine = conv[i * per + 3:(i + 1) * per]
        us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine) / len(mine) / 1e3
        pus = 0.0
        if pper:
            pm = pack[i * pper + 3:(i + 1) * pper]
            pus = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in pm) / len(pm) / 1e3
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        gf = 2.0 * B * Ho * Wo * Co * Ci * k * k / 1e9
        name = mine[0]["Kernel_Name"].split("(")[0][-40:]
        print(f"{what:24s} B{B} {H}x{W} {Ci}->{Co} k{k}: conv {us:7.1f} us  {gf / us * 1e3:7.1f} TF/s ({gf / us * 1e3 / 2516.6 * 100:4.1f} %)  pack {pus:5.1f} us  [{name}]")


if __name__ == "__main__":
    main(sys.argv[1])

"""Folds the stamp files of tools/sweeps/conv_floor.sh into one table (profiles/r04_conv_floor.txt):
    python tools/sweeps/conv_floor_table.py gpurun_out/r04_floor > profiles/r04_conv_floor.txt"""
import re
import sys

d = sys.argv[1]
MODES = {1: "eval epilogue (folded BN + SiLU, fp16 out)", 3: "train epilogue (raw fp32 + statistics)", 0: "plain (data gradient)"}


def read(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\d+x\d+x\d+ \d+->\d+ k\ds\d)\s+(\d+)\s+[\d.]+\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+)\s+([\d.]+) \|\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            out[m.group(1)] = [float(v) for v in m.groups()[1:]]
    return out


print("In-kernel phase stamps of the <= 40x40 convolution launches of YOLOv8-n (VERDICT r03 item 1), batch 32, MI355X, tuning library")
print("(s_memrealtime at: block start | operand DMA issued | first operands landed (first barrier) | K loop done | block end).")
print("span = first block's start -> last block's end of ONE launch (us).  r03 = the round-3 kernels (mode bit 0x4000: conv_halo / conv_gemm /")
print("conv_pw / conv_igemm_dma), r04 = the round-4 dispatcher (conv_tile where it applies: 3x3 stride 1, 32 <= Cin <= 144, maps <= 40x40).")
print("Block means of the r04 launch: issue / land / kloop / epi (us).  Rows whose kernel did not change repeat within noise.")
print("Written by tools/sweeps/conv_floor.sh + conv_floor_table.py.\n")
for mode, title in MODES.items():
    new, old = read(f"{d}/new_mode{mode}.txt"), read(f"{d}/old_mode{mode}.txt")
    print(f"--- {title} ---")
    print(f"{'shape':26s} {'r03 blocks':>10s} {'r03 span':>9s} | {'r04 blocks':>10s} {'r04 span':>9s} {'issue':>6s} {'land':>6s} {'kloop':>6s} {'epi':>6s} | {'r04/r03':>7s}  3x3<=40x40: span<=10us?")
    for k, n in new.items():
        o = old.get(k)
        if not o:
            continue
        is3 = " k3s1" in k and ("x40x40 " in k or "x20x20 " in k)
        flag = ("yes" if n[1] <= 10.0 else "NO") if is3 else ""
        print(f"{k:26s} {int(o[0]):10d} {o[1]:9.1f} | {int(n[0]):10d} {n[1]:9.1f} {n[5]:6.2f} {n[6]:6.2f} {n[7]:6.2f} {n[8]:6.2f} | {n[1] / o[1]:7.2f}  {flag}")
    print()

"""Time-ordered kernels of ONE train step from a rocprofv3 --kernel-trace CSV, with the queue they ran on, their grid and LDS, and -- for
every main-stream kernel -- the gap to its predecessor and the side-stream kernels that overlapped it.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --profile-steps 1
    python tools/step_timeline.py gpurun_out/trace/*/*_kernel_trace.csv [t0_us t1_us]
"""
import re
import sys

import pandas as pd


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I|E)", n)
    if m:
        return m.group(1)
    return re.split(r"[(]", n)[0][:44]


df = pd.read_csv(sys.argv[1])
df["name"] = df["Kernel_Name"].map(short)
adam = df[df["name"].str.contains("adam_kernel")].sort_values("Start_Timestamp")
a0, a1 = adam.iloc[2]["End_Timestamp"], adam.iloc[3]["End_Timestamp"]
st = df[(df["Start_Timestamp"] >= a0) & (df["End_Timestamp"] <= a1)].sort_values("Start_Timestamp").copy()
st["dur"] = (st["End_Timestamp"] - st["Start_Timestamp"]) / 1e3
st["t"] = (st["Start_Timestamp"] - a0) / 1e3
st["te"] = (st["End_Timestamp"] - a0) / 1e3
qcol = "Queue_Id" if "Queue_Id" in st.columns else "Stream_Id"
main = st.groupby(qcol)["dur"].sum().idxmax()
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
print(f"step wall {(a1 - a0) / 1e3:.1f} us; queues: {dict(st.groupby(qcol)['dur'].sum().round(0))}; main = {main}")
prev_end = None
for _, r in st.iterrows():
    if r["te"] < t0 or r["t"] > t1:
        if r[qcol] == main:
            prev_end = r["te"]
        continue
    wsz = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))
    wgs = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, wsz)
    lds = int(r.get("LDS_Block_Size", 0))
    tag = "MAIN" if r[qcol] == main else f"  q{int(r[qcol])}"
    gap = ""
    if r[qcol] == main:
        gap = f"gap {r['t'] - prev_end:5.1f}" if prev_end is not None else ""
        prev_end = r["te"]
    print(f"{r['t']:8.1f} -> {r['te']:8.1f} {r['dur']:7.1f} us {tag} {r['name']:34s} wgs {wgs:5d} x {wsz:4d} lds {lds // 1024:3d}K vgpr {int(r.get('VGPR_Count', 0)):3d} {gap}")

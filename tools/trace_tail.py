"""Timeline summary of one train step from a rocprofv3 --kernel-trace CSV: per-stream busy time, main-stream gaps and the tail
after the last main-stream backward kernel.      python tools/trace_tail.py <kernel_trace.csv>"""
import re
import sys

import numpy as np
import pandas as pd


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.split(r"[<(]", n)[0]
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)E", n)
    return m.group(1) if m else n


df = pd.read_csv(sys.argv[1])
df["name"] = df["Kernel_Name"].map(short)
adam = df[df["name"].str.contains("adam_kernel")].sort_values("Start_Timestamp")
a0, a1 = adam.iloc[3]["End_Timestamp"], adam.iloc[4]["End_Timestamp"]
st = df[(df["Start_Timestamp"] >= a0) & (df["End_Timestamp"] <= a1)].sort_values("Start_Timestamp").copy()
st["dur"] = (st["End_Timestamp"] - st["Start_Timestamp"]) / 1e3
st["t"] = (st["Start_Timestamp"] - a0) / 1e3
print(f"step wall {(a1 - a0) / 1e6:.3f} ms; {len(st)} kernels; sum of durations {st['dur'].sum() / 1e3:.3f} ms")
print(st.groupby("Stream_Id")["dur"].agg(["count", "sum"]))
main = st.groupby("Stream_Id")["dur"].sum().idxmax()
m = st[st["Stream_Id"] == main]
gaps = (m["Start_Timestamp"].values[1:] - m["End_Timestamp"].values[:-1]) / 1e3
print(f"main stream busy {m['dur'].sum() / 1e3:.3f} ms, gaps {gaps[gaps > 0].sum() / 1e3:.3f} ms, largest {np.round(np.sort(gaps)[-4:], 1)}")
print(m.tail(5)[["name", "dur", "t"]].to_string())
print(st[st["Stream_Id"] != main].tail(7)[["name", "Stream_Id", "dur", "t"]].to_string())
print(st.groupby("name")["dur"].agg(["count", "sum"]).sort_values("sum", ascending=False).head(16))

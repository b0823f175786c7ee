"""HBM traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `python bench.py --steps 3 --warmup 2`.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

The JSON records the SHA-256 of the library the passes ran on (the in-tree libcvx_engine.so at the time this tool runs:
run it on the GPU box right after the passes); bench.py refuses the figure when it does not match the running library.

One timed step (between two Adam launches) is summed.  Correction as MI355X_MICROARCH.md (section HBM) prescribes for gfx950:
bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE tallies 128-byte requests at 64 bytes).
"""
import collections
import hashlib
import os
import csv
import json
import sys

FAMILIES = ("conv_wgrad_k3_kernel", "conv_wgrad_stream_kernel", "conv_wgrad_gemm_kernel", "conv_gemm_kernel", "gemm_pack", "conv_tile_kernel", "tile_pack", "conv_chain_kernel", "chain_pack_kernel", "bn_act_apply", "conv_halo_kernel", "conv_pw_kernel", "conv_igemm_dma_kernel", "stem_stats_kernel", "stem_apply_kernel", "stem_wgrad_kernel", "conv_wgrad_halo_kernel", "conv_wgrad_kernel", "bn_bwd_reduce",
            "bn_bwd_apply", "bn_silu_apply", "reduce_slabs")
CONV = ("conv_gemm_kernel", "conv_tile_kernel", "conv_chain_kernel", "conv_halo_kernel", "conv_pw_kernel", "conv_igemm_dma_kernel", "stem_stats_kernel", "stem_apply_kernel")


def family(name):
    for k in FAMILIES:
        if k in name:
            return k
    return "other"


def one_step(path, counter):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    ids = list(per)
    adam = [i for i in ids if "adam_kernel" in per[i][0]]
    a, b = adam[2], adam[3]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for i in ids:
        if a < i <= b:
            k = family(per[i][0])
            agg[k][0] += per[i][1]
            agg[k][1] += 1
            if k == "other":  # the kernels behind "other", by name
                OTHER[counter][per[i][0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]] += per[i][1]
    return agg


OTHER = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}


def _src_sha256():
    """the key bench.py looks profiles/ files up by (bench.py: src_sha256)"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    csrc = os.path.join(root, "computervision.pytorch_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(root, "include", "cvx_engine.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def main():
    fetch, write = one_step(sys.argv[1], "FETCH_SIZE"), one_step(sys.argv[2], "WRITE_SIZE")
    fam = {k: {"fetch_kib": fetch[k][0], "write_kib": write[k][0], "launches": fetch[k][1],
               "hbm_bytes": (2 * fetch[k][0] + write[k][0]) * 1024} for k in fetch}
    n = sum(fam[k]["launches"] for k in CONV if k in fam)
    tot = sum(fam[k]["hbm_bytes"] for k in CONV if k in fam)
    import hashlib
    import os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computervision.pytorch_amd", "lib", "libcvx_engine.so")
    out = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "src_sha256": _src_sha256(), "model": "n", "batch": 32,
           "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) -- python bench.py "
                     "--no-cpu-baseline --steps 3 --warmup 2 --profile-steps 1; one timed step",
           "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters), MI355X_MICROARCH.md section HBM",
           "conv_family": [k for k in CONV if k in fam], "launches_per_step": n, "hbm_bytes_per_step": tot, "hbm_bytes_per_launch": tot / n,
           "whole_step_hbm_bytes": sum(v["hbm_bytes"] for v in fam.values()), "families": fam,
           "other_by_kernel_mb": {k: round((2 * OTHER["FETCH_SIZE"].get(k, 0.0) + OTHER["WRITE_SIZE"].get(k, 0.0)) * 1024 / 1e6, 1)
                                  for k in sorted(set(OTHER["FETCH_SIZE"]) | set(OTHER["WRITE_SIZE"]),
                                                  key=lambda k: -(2 * OTHER["FETCH_SIZE"].get(k, 0.0) + OTHER["WRITE_SIZE"].get(k, 0.0)))[:20]}}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"conv family: {n} launches, {tot / 1e9:.2f} GB per step, {tot / n / 1e6:.1f} MB per launch; whole step {out['whole_step_hbm_bytes'] / 1e9:.1f} GB")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
        print(f"  {k:24s} {v['launches']:4d} launches  {v['hbm_bytes'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()

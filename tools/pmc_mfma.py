"""MFMA utilisation per kernel family from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, with --kernel-trace).

    python tools/pmc_mfma.py <counter_collection.csv> <out.json> [first_marker_kernel [last_marker_kernel]]

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the matrix-busy cycles summed over the chip's 1024 SIMDs against
the cycles the dispatch was active (rocprofv3 reports GRBM_GUI_ACTIVE as the sum over the 8 XCDs -- MI355X_MICROARCH.md, DVFS give-back).
A v_mfma_f32_32x32x16_f16 holds its SIMD's matrix pipe for 32 cycles, a 16x16x32 for 16: busy / 32 (or 16) is the instruction count.
The JSON carries the SHA-256 of the library the pass ran on; bench.py quotes the figure only when it matches the running one.
"""
import collections
import csv
import hashlib
import json
import os
import sys

FAMILIES = ("conv_wgrad_gemm_kernel", "conv_gemm_kernel", "gemm_pack", "conv_tile_kernel", "tile_pack", "conv_chain_kernel", "conv_halo_kernel", "conv_pw_kernel", "conv_igemm_dma_kernel", "stem_", "conv_wgrad_halo_kernel", "conv_wgrad_kernel",
            "bn_act_apply", "bn_bwd_reduce", "bn_bwd_apply", "chain_pack_kernel", "pack_weights")
CONV = ("conv_gemm_kernel", "conv_tile_kernel", "conv_chain_kernel", "conv_halo_kernel", "conv_pw_kernel", "conv_igemm_dma_kernel")


def family(name):
    for k in FAMILIES:
        if k in name:
            return k
    return "other"


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
    per = collections.OrderedDict()
    for r in csv.DictReader(open(sys.argv[1])):
        d = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    fam = collections.defaultdict(lambda: {"mfma_busy": 0.0, "gui_active": 0.0, "launches": 0})
    for d in per.values():
        f = fam[family(d["name"])]
        f["mfma_busy"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        f["gui_active"] += d.get("GRBM_GUI_ACTIVE", 0.0)
        f["launches"] += 1
    out_f = {}
    for k, f in fam.items():
        act = f["gui_active"] / 8.0 * 1024.0
        out_f[k] = {"launches": f["launches"], "mfma_busy_cycles": f["mfma_busy"], "active_simd_cycles": act,
                    "mfma_busy_frac": (f["mfma_busy"] / act) if act > 0 else None}
    conv_busy = sum(fam[k]["mfma_busy"] for k in CONV if k in fam)
    conv_act = sum(fam[k]["gui_active"] for k in CONV if k in fam) / 8.0 * 1024.0
    all_act = sum(f["gui_active"] for f in fam.values()) / 8.0 * 1024.0
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "computervision.pytorch_amd", "lib", "libcvx_engine.so")
    out = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "src_sha256": _src_sha256(),
           "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace (a pass of its own); every dispatch of the process",
           "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)",
           "conv_family": [k for k in CONV if k in fam], "conv_mfma_busy_frac": conv_busy / conv_act if conv_act > 0 else None,
           "all_kernels_mfma_busy_frac": conv_busy / all_act if all_act > 0 else None, "families": out_f}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("conv_mfma_busy_frac", "all_kernels_mfma_busy_frac")}))


if __name__ == "__main__":
    main()

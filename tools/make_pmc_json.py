#!/usr/bin/env python3
"""Merge the three PMC pass summaries of tools/profile_round.sh into
profiles/<tag>_pmc.json (what bench.py quotes as roofline.traffic) and copy the
round's markdown summaries from gpurun_out/profiles_<tag>/ into profiles/.

    python tools/make_pmc_json.py r01
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    dst = os.path.join(ROOT, "profiles")
    out = {}
    for name, rename in (("pmc_fetch", {"FETCH_SIZE": "FETCH_SIZE_KB_raw"}),
                         ("pmc_write", {"WRITE_SIZE": "WRITE_SIZE_KB"}), ("pmc_mfma", {}), ("pmc_mfma_f16", {}),
                         ("pmc_mfma_gemm", {})):
        if not os.path.exists(os.path.join(src, name + ".json")):
            continue
        d = json.load(open(os.path.join(src, name + ".json")))
        for k, c in d["counters"].items():
            e = out.setdefault(k, {})
            for cn, v in c.items():
                e[rename.get(cn, cn)] = v
        if name in ("pmc_mfma", "pmc_mfma_f16", "pmc_mfma_gemm"):
            for k, t in d["kernels"].items():
                if k in out:
                    out[k]["avg_us_mfma_pass"] = t["avg_us"]
    for k, e in out.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            # GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            e["MfmaUtil_pct"] = round(100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] /
                                      (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 2)
            # one MOPS unit = 512 flops (MI355X_MICROARCH.md, rocprofv3 section)
            e["mfma_flops_per_dispatch"] = e.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0
    # FETCH_SIZE calibration (tools/fetch_calib.hip: every kernel streams 1 GiB = 1048576 KB once):
    # reported KB / true KB per access pattern; bench.py multiplies FETCH by 1 / that factor
    calib, factors = {}, {}
    cpath = os.path.join(src, "pmc_fetch_calib.json")
    if os.path.exists(cpath):
        d = json.load(open(cpath))
        for k, c in d["counters"].items():
            if "FETCH_SIZE" in c:
                calib[k] = round(c["FETCH_SIZE"] / 1048576.0, 4)
        pat = {"lds_dma_b32": "calib_lds_dma<4>", "lds_dma_b128": "calib_lds_dma<16>",
               "load_b128": "calib_load<HIP_vector_type<float, 4u> >", "load_b32": "calib_load<float>"}
        got = {k: calib.get(v) for k, v in pat.items()}
        # which pattern dominates each kernel's reads
        for kern, key in (("wino_conv_z_kernel", "lds_dma_b32"), ("wino_wgrad_kernel", "load_b32"),
                          ("gemm_conv_nn_kernel", "lds_dma_b128"), ("gemm_conv_nt_kernel", "lds_dma_b128"),
                          ("conv3x3_f16_kernel", "load_b128"), ("conv3x3_wgrad_f16_kernel", "lds_dma_b128"),
                          ("cls_losses_fused_kernel", "load_b128"), ("pow_sum_kernel", "load_b128"),
                          ("distill_fwd_kernel", "load_b128"), ("distill_bwd_kernel", "load_b128")):
            if got.get(key):
                factors[kern] = round(1.0 / got[key], 4)
    note = ("per-dispatch averages over python tools/kbench.py / tools/gemm_pmc.py; FETCH_SIZE raw: gfx950 "
            "reports 1/2 of the bytes actually fetched for every pattern calibrated here (4-B and 16-B "
            "LDS-DMA, 4-B and 16-B global loads: fetch_calibration_reported_over_true), so true read bytes "
            "= FETCH_SIZE_KB_raw x fetch_calibration[kernel]; GRBM_GUI_ACTIVE is summed over the 8 XCDs")
    json.dump({"kernels": out, "note": note, "fetch_calibration_reported_over_true": calib,
               "fetch_calibration": factors},
              open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    for f in sorted(os.listdir(src)):
        if f.endswith(".md"):
            shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))
    print("wrote", os.path.join(dst, tag + "_pmc.json"))


if __name__ == "__main__":
    main()

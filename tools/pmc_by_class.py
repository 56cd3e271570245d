#!/usr/bin/env python3
"""Hardware counters of the launches bench.py actually times, keyed by timing class.

    python tools/pmc_by_class.py --out profiles/r03_pmc_classes.json --md profiles/r03_pmc_classes.md \
        fetch=/tmp/prof_pmc_fetch/x.db write=/tmp/prof_pmc_write/x.db mfma=/tmp/prof_pmc_mfma/x.db

Each .db is a rocprofv3 `--kernel-trace --pmc ...` capture of
`python bench.py --workload heads --steps K ...` (separate passes per counter group, as
MI355X_MICROARCH.md prescribes).  One kernel serves several timing classes (wino_conv_z_kernel:
tower forward = class 2, cls_pred = 3, bbox_pred = 4, data gradients = 16), so dispatches are
attributed by ORDER: the step's program (built here on the CPU with batch 1 -- the launch sequence
does not depend on the batch) gives, per kernel name, the sequence of classes of one step; a step
ends with its sgd_flat_kernel dispatch; launches of one kernel name are issued on one stream, so
their dispatch ids are in program order.
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"[<(].*", "", name)
    # the fp16 3x3 filter gradient is one of two kernels (all nine taps per workgroup since the end of round 3,
    # SSAD_F16_WGRAD9=0: one filter row per workgroup); same op, same class, same stream
    return {"wgrad9_f16_kernel": "conv3x3_wgrad_f16_kernel"}.get(name, name)


def class_sequences(f16=False):
    """{kernel name: [class of its 1st, 2nd, ... launch within one step]} for the subnets step."""
    import ssad_amd  # noqa: F401
    from ssad_amd import program as PR, synth
    from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    h = (DistillHeadsF16 if f16 else DistillHeads)(HeadConfig(num_gpus=1), N=1, shapes=synth.LEVEL_SHAPES_600,
                                                   device="cpu")
    names = {PR.CONV3X3: "wino_conv_z_kernel", PR.CONV3X3_WGRAD: "wino_wgrad_kernel", PR.POW_SUM: "pow_sum_kernel",
             PR.CLS_LOSSES_FUSED: "cls_losses_fused_kernel", PR.SGD_FLAT: "sgd_flat_kernel",
             PR.F16_CONV3X3: "conv3x3_f16_kernel", PR.F16_WGRAD: "conv3x3_wgrad_f16_kernel"}
    from ssad_amd import kernels as K
    seq = {}
    for op in h.prog.ops:
        if op.code == PR.SPLIT_ABSMAX_LEVELS:      # the |max| passes of the pipeline's table: a class of their own
            seq.setdefault("ssad_split::split_absmax_kernel", []).append((op.klass, True))
            continue
        n = names.get(op.code)
        if n is None or (op.code == PR.CONV3X3 and not op.i[4]):
            continue
        launches = 1
        if op.code == PR.CONV3X3:
            # one call = one launch per staging geometry present among its levels (8 x 16 patches / sub-patch
            # pairs, conv3x3_winograd.hip); the levels of the batch-1 program have the bench's map sizes
            import ctypes as C
            # (+ the split-tail launch of a geometry whose partial round is split, round 5: that depends on the
            # batch, so the batch-1 program's levels are asked about at the bench's batch of 16)
            src = C.cast(op.p[0], C.POINTER(K.ConvLevel))
            lv16 = (K.ConvLevel * op.i[0])()
            for q in range(op.i[0]):
                lv16[q] = src[q]
                lv16[q].N = 16
            launches = K.lib().ssad_conv3x3_forward_wino_launches_for(lv16, op.i[0], op.i[1], op.i[2], op.i[3])
            if op.i[4] == 3:      # the split-operand engine: a call = (|max| pass +) split pass + convolution, all its class
                if not op.p[4]:   # (handed the words of the pipeline's |max| table: no pass of its own)
                    seq.setdefault("ssad_split::split_absmax_kernel", []).append((op.klass, False))
                seq.setdefault("ssad_split::split_pack_act_kernel", []).append((op.klass, False))
                seq.setdefault("conv3x3_split_kernel", []).append((op.klass, True))
                continue
            if op.i[4] == 2:      # the F(2x4) engine: another kernel name, one launch per staging geometry
                launches = K.lib().ssad_conv3x3_forward_wino_launches(lv16, op.i[0])
                for k in range(launches):
                    seq.setdefault("wino24_conv_kernel", []).append((op.klass, k == 0))
                continue
        if op.code == PR.CONV3X3_WGRAD and op.i[4] == 1:
            # the split-operand filter gradient: |max| pass + main kernel + slab reduction, all its class (the bias
            # gradient's two launches follow every filter gradient and stay out of the counters, as before)
            if not op.p[4]:
                seq.setdefault("wsplit_absmax_kernel", []).append((op.klass, False))
            seq.setdefault("wsplit_kernel", []).append((op.klass, True))
            seq.setdefault("wsplit_reduce_kernel", []).append((op.klass, False))
            continue
        for k in range(launches):
            seq.setdefault(n, []).append((op.klass, k == 0))
    return seq


def read_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection").fetchall()
    disp = {}
    for did, kn, cn, v in rows:
        d = disp.setdefault(did, {"name": short(kn), "c": {}})
        d["c"][cn] = d["c"].get(cn, 0.0) + v          # summed over XCDs / SEs
    return [dict(id=k, **v) for k, v in sorted(disp.items())]


def attribute(dispatches, seq):
    """-> {klass: {"kernel", "dispatches", counter: per-dispatch average}}; steps = segments that end
    with sgd_flat_kernel and contain exactly the expected number of launches of every kernel."""
    steps, cur = [], []
    for d in dispatches:
        cur.append(d)
        if d["name"] == "sgd_flat_kernel":
            steps.append(cur)
            cur = []
    acc, used = {}, 0
    for st in steps:
        per = {}
        for d in st:
            if d["name"] in seq:
                per.setdefault(d["name"], []).append(d)
        if any(len(per.get(n, [])) != len(s) for n, s in seq.items()):
            continue                                   # warm-up step with extra (teacher pack) or partial capture
        used += 1
        for n, s in seq.items():
            for d, (k, first) in zip(per[n], s):
                a = acc.setdefault(k, {"kernel": n, "dispatches": 0, "sum": {}})
                if first:
                    a["kernel"] = n                    # the call's main kernel names the row (not a helper pass)
                a["dispatches"] += int(first)          # per CALL: a call's launches are summed
                for cn, v in d["c"].items():
                    a["sum"][cn] = a["sum"].get(cn, 0.0) + v
    out = {}
    for k, a in acc.items():
        e = {"kernel": a["kernel"], "dispatches": a["dispatches"]}
        for cn, v in a["sum"].items():
            e[cn] = v / a["dispatches"]
        out[k] = e
    return out, used, len(steps)


def main():
    args = sys.argv[1:]
    out_path = md_path = None
    f16 = False
    passes = {}
    i = 0
    while i < len(args):
        if args[i] == "--out":
            out_path = args[i + 1]; i += 2
        elif args[i] == "--md":
            md_path = args[i + 1]; i += 2
        elif args[i] == "--f16":
            f16 = True; i += 1
        else:
            k, v = args[i].split("=", 1)
            passes[k] = v; i += 1
    seq = class_sequences(f16)
    merged, notes = {}, []
    for pname, path in passes.items():
        got, used, total = attribute(read_db(path), seq)
        notes.append("%s: %d of %d captured steps matched the program's launch sequence" % (pname, used, total))
        for k, e in got.items():
            m = merged.setdefault(k, {"kernel": e["kernel"]})
            m["dispatches_" + pname] = e["dispatches"]
            for cn, v in e.items():
                if cn not in ("kernel", "dispatches"):
                    m[cn] = v
    for k, e in merged.items():
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            # KB per dispatch; gfx950 reports half of the bytes fetched for every access pattern this
            # repo uses (profiles/r02_pmc_fetch_calib.md, MI355X_MICROARCH.md HBM section): x 2
            e["hbm_read_bytes"] = 2.0 * 1024.0 * e.get("FETCH_SIZE", 0.0)
            e["hbm_write_bytes"] = 1024.0 * e.get("WRITE_SIZE", 0.0)
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if e.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            # GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            e["MfmaUtil_pct"] = round(100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 2)
        if e.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in e:
            e["wait_inst_any_pct"] = round(100.0 * e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"], 1)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_pct"] = round(100.0 * e["TCC_HIT_sum"] / max(e["TCC_HIT_sum"] + e["TCC_MISS_sum"], 1.0), 1)
        if "TCC_EA0_RDREQ_sum" in e:
            # every fabric read request of these kernels is a 128-byte line fill (TCC_EA0_RDREQ_32B = 0;
            # PowSum: 4.30 M requests for 550 MB)
            e["l2_fill_bytes"] = 128.0 * e["TCC_EA0_RDREQ_sum"]
    doc = {"workload": "python bench.py --workload heads%s (bs 16, 600 px), per-dispatch averages by timing class "
                       "(ssad_amd/program.py: KLASS)" % (" --precision f16" if f16 else ""),
           "attribution": notes, "classes": {str(k): merged[k] for k in sorted(merged)},
           # the build the counters were taken on: bench.py refuses to quote them for other kernel sources
           "kernel_sources": kernel_source_hashes()}
    if out_path:
        json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    lines = ["# Counters of the timed launches, by timing class", "", doc["workload"], ""] + ["* " + n for n in notes]
    cols = ["hbm_read_bytes", "hbm_write_bytes", "MfmaUtil_pct", "wait_inst_any_pct", "SQ_LDS_BANK_CONFLICT",
            "l2_hit_pct", "l2_fill_bytes"]
    cols = [c for c in cols if any(c in e for e in merged.values())]
    lines += ["", "| class | kernel | " + " | ".join(cols) + " |", "|---|---|" + "---|" * len(cols)]
    for k in sorted(merged):
        e = merged[k]
        lines.append("| %d | %s | %s |" % (k, e["kernel"], " | ".join(
            ("%.4g" % e[c]) if c in e else "" for c in cols)))
    text = "\n".join(lines) + "\n"
    if md_path:
        open(md_path, "w").write(text)
    print(text)


def kernel_source_hashes():
    """sha256[:16] of every kernel source of the product library (csrc/kernels/*): what `traffic` in bench.py's
    line is valid for."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kdir = os.path.join(root, "semi-supervised-adaptive-distillation_amd", "csrc", "kernels")
    return {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
            for f in sorted(glob.glob(os.path.join(kdir, "*")))}


if __name__ == "__main__":
    main()

"""Numerical check of every tuned GEMM pick in harness/tunableop_gfx950.csv: the same
strided-batched product with TunableOp's pick and in float64 must agree to fp32 round-off
(a pick that used a reduced-precision path -- xf32, split fp16 -- would show as ~1e-3).
    python tools/check_tuned_gemms.py"""
import csv
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd  # noqa
from tools.harness import full_model as fm


def main():
    path = fm._TUNABLEOP_CSV
    rows = [r for r in csv.reader(open(path)) if r[0] != "Validator"]
    import torch.cuda.tunable as T
    T.enable(True)
    T.tuning_enable(False)
    T.set_filename(path)
    worst = 0.0
    g = torch.Generator(device="cuda").manual_seed(0)
    for op, key, sol, _ in rows:
        p = key.split("_")
        ta, tb = p[0][0], p[0][1]
        m, n, k = int(p[1]), int(p[2]), int(p[3])
        batch = int(p[p.index("B") + 1]) if "B" in p else 1
        # column-major C[m x n] = op(A)[m x k] op(B)[k x n]  ==  row-major C^T = op(B)^T op(A)^T:
        # build the torch call that produces exactly this key: out[n x m] = X[n x k] @ Y[k x m]
        X = torch.randn(batch, n, k, device="cuda", generator=g)
        Y = torch.randn(batch, k, m, device="cuda", generator=g)
        if tb == "t":
            X = torch.randn(batch, k, n, device="cuda", generator=g).transpose(1, 2)
        if ta == "t":
            Y = torch.randn(batch, m, k, device="cuda", generator=g).transpose(1, 2)
        if batch == 1:
            out = torch.mm(X[0], Y[0])
            ref = torch.mm(X[0].double(), Y[0].double())
        else:
            out = torch.bmm(X, Y)
            ref = torch.bmm(X.double(), Y.double())
        rel = float((out.double() - ref).abs().max() / ref.abs().max())
        worst = max(worst, rel)
        print("%-44s %-26s max rel err %.2e" % (key[:44], sol, rel), flush=True)
    print("worst %.2e over %d picks; tuned results recorded by torch: %d" % (worst, len(rows), len(T.get_results())))
    assert worst < 2e-5, "a tuned pick is not an fp32 GEMM"


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for greedy NMS from the reference's own code.

detectron/lib/utils/cython_nms.pyx does not build in this image (Cython 3 / numpy 2 have no `np.int_t` / `np.int`), so its
`nms` (lines 37-92) is EXECUTED AS PYTHON instead: this script reads the .pyx where it lies under /root/reference at
generation time (nothing of it is stored in the repository), removes what only the C compiler needs -- the list of
substitutions below is the whole transformation -- and runs the result on seeded detections.  With numpy >= 2 scalar
arithmetic keeps float32 when a float32 meets a Python number (NEP 50), which is what the typed C variables do.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_nms_golden.py     -> tests/golden/nms_ref.npz
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/detectron/lib/utils/cython_nms.pyx"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nms_ref.npz")


def reference_nms():
    src = open(REF).read().split("\n")
    # the two inline helpers and `nms` itself: from the first `cdef inline` to the line before the Soft-NMS banner
    start = next(i for i, l in enumerate(src) if l.startswith("cdef inline"))
    end = next(i for i, l in enumerate(src) if "Soft-NMS" in l) - 1
    out = []
    for line in src[start:end]:
        s = line
        if s.strip().startswith(("@cython.", "cimport ")):
            continue
        if s.strip().startswith("#"):
            continue
        # cdef inline T name(T a, T b) nogil:  ->  def name(a, b):
        m = re.match(r"cdef inline \S+ (\w+)\(\S+ (\w+), \S+ (\w+)\) nogil:", s)
        if m:
            out.append("def %s(%s, %s):" % m.groups())
            continue
        # def nms(<typed args>):  ->  def nms(dets, thresh):
        s = re.sub(r"np\.ndarray\[[^\]]*\] ", "", s)
        s = re.sub(r"np\.float32_t ", "", s)
        # bare declarations (`cdef int _i, _j`, `cdef np.float32_t w, h` after the line above) carry no value
        if re.match(r"\s*cdef (int )?[\w, ]+$", s):
            continue
        s = re.sub(r"^(\s*)cdef (int )?", r"\1", s)            # typed assignments keep their right-hand side
        s = s.replace("dtype=np.int)", "dtype=np.int64)")     # numpy 2 spelling of the same dtype
        s = re.sub(r"^(\s*)with nogil:", r"\1if True:", s)
        out.append(s)
    text = "\n".join(out)
    ns = {"np": np}
    exec(compile(text, "<cython_nms.pyx:nms as python>", "exec"), ns)
    return ns["nms"], text


def cases(rng):
    out = []
    for n, spread, size in ((1, 50, 30), (2, 5, 30), (60, 120, 40), (200, 200, 60), (400, 80, 50)):
        ctr = rng.uniform(0, spread, size=(n, 2)).astype(np.float32)
        wh = rng.uniform(4, size, size=(n, 2)).astype(np.float32)
        scores = rng.permutation(n).astype(np.float32) / np.float32(n) + np.float32(0.001)    # distinct
        dets = np.concatenate([ctr - wh / 2, ctr + wh / 2, scores[:, None]], axis=1).astype(np.float32)
        out.append(dets)
    # duplicates and an exact-threshold pair: two 10 x 10 boxes shifted by 5 -> inter 50+..., and identical boxes
    out.append(np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [5, 0, 14, 9, 0.7], [20, 20, 29, 29, 0.6],
                         [0, 0, 19, 9, 0.5]], np.float32))
    return out


def main():
    nms, text = reference_nms()
    assert "suppressed[j] = 1" in text and "cdef" not in text and "np.float32_t" not in text
    rng = np.random.default_rng(20260930)
    blobs = {}
    k = 0
    for dets in cases(rng):
        for thresh in (0.3, 0.5, 0.7):
            keep = nms(dets.copy(), np.float32(thresh))
            blobs["dets_%d" % k] = dets
            blobs["thresh_%d" % k] = np.float32(thresh)
            blobs["keep_%d" % k] = np.asarray(keep, np.int64)
            k += 1
    blobs["n_cases"] = np.int64(k)
    np.savez_compressed(OUT, **blobs)
    print("wrote %s: %d cases, survivors %s" % (OUT, k, [int(blobs["keep_%d" % i].size) for i in range(k)]))


if __name__ == "__main__":
    sys.dont_write_bytecode = True
    main()

"""Generates tests/golden/contract_golden.json.

The reference tree has no contraction path and no vectors for it, so these
fixtures are NOT reference outputs: inputs come from the shared splitmix64
generator, expected values from numpy.einsum in float64 (an independent
implementation of the same mathematical definition).  They pin the oracle's
restatement (tests/test_oracle_golden.py) - parity with the reference itself
stays unpinned.

Run from the repo root:  python oracle/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref  # noqa: E402

SEED_A, SEED_B = 0xC7C0A, 0xC7C0B

CASES = [
    # (name, spec, extents, dtype)
    ("plumbing_small", "adc,db->abc", dict(a=8, b=6, c=5, d=7), "f64"),
    ("plumbing_medium", "adc,db->abc", dict(a=32, b=48, c=16, d=24), "f64"),
    ("ccsd_n8", "aeim,ebmj->abij", dict(a=8, b=8, i=8, j=8, e=8, m=8), "f64"),
    ("ccsd_ragged", "aeim,ebmj->abij", dict(a=5, b=7, i=3, j=6, e=9, m=4), "f64"),
    ("ccsd_n16_f32", "aeim,ebmj->abij", dict(a=16, b=16, i=16, j=16, e=16, m=16), "f32"),
    ("adversarial_f32", "dkca,bk->abcd", dict(a=16, b=8, c=12, d=16, k=8), "f32"),
    ("six_index", "adij,bcdk->abcijk", dict(a=4, b=5, c=3, i=6, j=4, k=5, d=8), "f64"),
    ("batched", "zak,zkb->zab", dict(z=7, a=24, b=16, k=32), "f64"),
    ("batch_last", "akz,kbz->abz", dict(z=4, a=10, b=12, k=6), "f64"),
    ("outer_product", "a,b->ab", dict(a=17, b=9), "f64"),
    ("matvec", "ak,k->a", dict(a=33, k=21), "f64"),
    ("dot_like", "ak,bk->ab", dict(a=1, b=1, k=64), "f64"),
    ("c_fast_in_a", "ka,kb->ba", dict(a=20, b=12, k=18), "f64"),
    ("ccsd_bf16", "aeim,ebmj->abij", dict(a=4, b=4, i=8, j=8, e=4, m=8), "bf16"),
]

NP = {"f64": np.float64, "f32": np.float32}
CODE = {"f64": ref.F64, "f32": ref.F32, "bf16": ref.BF16}


def operand(letters, extents, seed, dtype):
    n = int(np.prod([extents[ch] for ch in letters])) if letters else 1
    raw = ref.fill(n, seed, CODE[dtype])
    vals = ref.bf16_to_f32(raw).astype(np.float64) if dtype == "bf16" else raw.astype(np.float64)
    return raw, vals.reshape([extents[ch] for ch in letters])


def classify_py(spec):
    """Independent pure-Python statement of the classification and canonical transposes."""
    ins, c = spec.split("->")
    a, b = ins.split(",")
    names = []
    for ch in a + b + c:
        if ch not in names:
            names.append(ch)
    cls = {}
    for ch in names:
        key = (ch in a, ch in b, ch in c)
        cls[ch] = {(True, True, True): "batch", (True, True, False): "contracted",
                   (True, False, True): "free_m", (False, True, True): "free_n"}[key]
    batch = "".join(ch for ch in c if cls[ch] == "batch")
    m = "".join(ch for ch in c if cls[ch] == "free_m")
    n = "".join(ch for ch in c if cls[ch] == "free_n")
    k = "".join(ch for ch in a if cls[ch] == "contracted")
    return {
        "names": "".join(names), "classes": cls, "a": a, "b": b, "c": c,
        "batch": batch, "m": m, "n": n, "k": k,
        "perm_a": [a.index(ch) for ch in batch + m + k],
        "perm_b": [b.index(ch) for ch in batch + k + n],
        "perm_c": [(batch + m + n).index(ch) for ch in c],
    }


def main():
    out = {"generator": {}, "cases": []}
    for seed in (0, 1, SEED_A, SEED_B, 0xFFFFFFFFFFFFFFFF):
        out["generator"][hex(seed)] = {
            "splitmix64": [hex(ref.splitmix64(seed + k)) for k in range(4)],
            "f64": [float(v).hex() for v in ref.fill(4, seed, ref.F64)],
            "f32": [float(v).hex() for v in ref.fill(4, seed, ref.F32)],
            "bf16": [int(v) for v in ref.fill(4, seed, ref.BF16)],
        }
    rng = np.random.RandomState(12345)
    for name, spec, ext, dtype in CASES:
        ins, outm = spec.split("->")
        la, lb = ins.split(",")
        _, A = operand(la, ext, SEED_A, dtype)
        _, B = operand(lb, ext, SEED_B, dtype)
        E = np.einsum(spec, A, B).reshape(-1)  # float64 math on the (rounded) inputs
        n = E.size
        idx = sorted(set([0, n - 1] + [int(i) for i in rng.randint(0, n, size=min(16, n))]))
        out["cases"].append({
            "name": name, "spec": spec, "extents": ext, "dtype": dtype,
            "seed_a": SEED_A, "seed_b": SEED_B,
            "n_out": n,
            "samples": [[i, float(E[i]).hex()] for i in idx],
            "sum": float(E.sum()).hex(),
            "sum_abs": float(np.abs(E).sum()).hex(),
            "max_abs": float(np.abs(E).max()).hex(),
            "sha256_inputs": hashlib.sha256(A.tobytes() + B.tobytes()).hexdigest(),
            "classification": classify_py(spec),
        })
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "tests", "golden", "contract_golden.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, len(out["cases"]), "cases")


if __name__ == "__main__":
    main()

"""Generates tests/golden/fastr_objects_golden.json: realignment regions built out of the REFERENCE's own
objects (BamAlignment, AlignedRead, ReadBin, Consensus, ConsensusHashTable), flattened by the reference's
getters in clean_gpu_prep's order and scored by the reference's findBestOffset
(oracle/_ref/libfastr_ref.so: oracle/fastr_ref_harness.cpp, fastr_ref_regions).  The fixture holds the
region descriptions, the flattened arrays and the scores - data only.

Run from the repo root:  python oracle/make_fastr_objects_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fastr  # noqa: E402

STRIDE = 48


def make_regions(seed=0x0B1EC7):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGT", np.uint8)
    regions = []
    for t in range(5):
        span = int(rng.integers(90, 170))
        genome = alphabet[rng.integers(0, 4, span)]
        if t == 2:
            genome[rng.integers(0, span, 6)] = ord("N")
        start = int(rng.integers(1000, 50000))
        text = lambda a: bytes(a).decode()  # noqa: E731

        def read_at(off, length, clip, errors):
            seq = genome[off:off + length].copy()
            for e in rng.integers(0, length, errors):
                seq[e] = alphabet[rng.integers(0, 4)]
            lead = alphabet[rng.integers(0, 4, clip)]
            phred = [int(x) for x in rng.integers(2, 42, length + clip)]
            return (start + off, clip, text(np.concatenate([lead, seq])), phred)

        # the first read of the bin fixes its start; it is a reference-matching read left of the others
        bin_reads = [read_at(0, 36, 0, 0)] + [read_at(int(rng.integers(0, 20)), 36, 0, 0) for _ in range(int(rng.integers(0, 3)))]
        alt = []
        for _ in range(int(rng.integers(3, 9))):
            length = int(rng.integers(20, 40))
            clip = int(rng.integers(0, 6)) if rng.random() < 0.4 else 0
            alt.append(read_at(int(rng.integers(0, span - length - 8)), length, clip, int(rng.integers(0, 4))))
        if t == 3:   # a read hanging over the end of every consensus
            alt.append(read_at(span - 20, 20, 0, 1))
        cands = []
        for _ in range(int(rng.integers(1, 4))):
            cut = int(rng.integers(20, span - 30))
            gap = int(rng.integers(1, 9))
            if rng.random() < 0.5:
                cands.append(text(np.concatenate([genome[:cut], genome[cut + gap:]])))              # deletion
            else:
                cands.append(text(np.concatenate([genome[:cut], alphabet[rng.integers(0, 4, gap)], genome[cut:]])))
        cands.append(cands[0])                      # a duplicate: the table keeps the first only
        if t == 1:
            cands.insert(1, cands[-1])
        regions.append({"bin_reads": bin_reads, "alt_reads": alt, "candidates": cands})
    # a region whose only scored read is also the bin's first read, and one with no consensus at all
    g = text(alphabet[rng.integers(0, 4, 60)])
    regions.append({"bin_reads": [], "alt_reads": [(777, 0, g[5:35], [30] * 30)], "candidates": [g, g[:30] + g[34:]]})
    regions.append({"bin_reads": [(900, 0, g[:36], [20] * 36)], "alt_reads": [(905, 2, g[3:33], [11] * 30)], "candidates": []})
    return regions


def main():
    if not fastr.reference_available():
        sys.exit("build the reference first: bash oracle/build_ref.sh")
    regions = make_regions()
    out = fastr.reference_regions(regions, STRIDE)
    fixture = {"regions": [{"bin_reads": [list(r) for r in reg["bin_reads"]], "alt_reads": [list(r) for r in reg["alt_reads"]],
                            "candidates": reg["candidates"]} for reg in regions],
               "read_stride": STRIDE,
               "flattened": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in out.items()}}
    path = os.path.join(ROOT, "tests", "golden", "fastr_objects_golden.json")
    with open(path, "w") as f:
        json.dump(fixture, f, separators=(",", ":"))
    print("wrote", path, len(regions), "regions,", len(out["best_index"]), "pairs,", os.path.getsize(path), "bytes")
    print("n_consensus", out["n_consensus"].tolist(), "alt_reads", out["alt_reads"].tolist())


if __name__ == "__main__":
    main()

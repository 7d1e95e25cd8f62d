"""Generates tests/golden/fastr_golden.json: inputs and the (best_index, best_score) pairs returned
by the REFERENCE's own findBestOffset (oracle/_ref/libfastr_ref.so, compiled from
/root/reference/apps/fastr by oracle/build_ref.sh).  Data only; no reference source is stored.

Run from the repo root:  python oracle/make_fastr_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fastr  # noqa: E402


def hand_cases():
    """(name, ref, read, quals, original_alignment) edge cases."""
    ref = "ACGTACGTTTGACCAGTNACGGATCCATGCAAGT"
    out = [
        ("exact_at_original", ref, ref[5:15], [30] * 10, 5),
        ("exact_elsewhere_left", ref, ref[2:12], [30] * 10, 9),
        ("exact_elsewhere_right", ref, ref[20:30], [30] * 10, 3),
        ("tie_prefers_original", "AAAAAAAAAAAAAAAA", "AAAC", [10, 10, 10, 7], 6),
        ("tie_prefers_lowest_left", "CAAACAAACAAAGGGG", "AAAC", [9, 9, 9, 9], 12),
        ("tie_right_side_lowest", "GGGGCAAACAAAC", "AAAC", [9, 9, 9, 9], 0),
        ("n_in_reference_is_free", ref, "AGTCACGG", [20] * 8, 14),
        ("n_in_read_is_free", ref, "ACGNACGT", [20] * 8, 0),
        ("lowercase_is_regular_but_distinct", "ACGTacgtACGT", "acgt", [11, 12, 13, 14], 0),
        ("original_past_last_start_hangs", ref, ref[-6:] + "AC", [5] * 8, len(ref) - 6),
        ("read_longer_than_reference", "ACGT", "ACGTAC", [7] * 6, 0),
        ("single_base", ref, "G", [40], 7),
        ("zero_qualities", ref, "TTTTTTTT", [0] * 8, 4),
        ("original_zero", ref, ref[10:20], [25] * 10, 0),
        ("all_mismatch", "AAAAAAAAAAAA", "CCCC", [3, 4, 5, 6], 4),
        # UNDEFINED BEHAVIOUR OF THE REFERENCE, pinned as observed: its `byte` is a signed char
        # (apps/fastr/Types.h:51), so a quality byte above 127 enters the sums as a negative number (defined,
        # if surprising) and a BASE byte above 127 would index `static char ch_lookup[128]`
        # (apps/fastr/IndelRealigner.cpp:96,282) below its start.  The two cases here use high QUALITY bytes
        # only; no committed vector holds a base byte above 127, because what the reference returns for one
        # depends on whatever the linker placed in front of that array.  The oracle and the kernel treat such
        # a base as "not a regular base" (the value the reference reads while the bytes before the array are
        # zero); that choice is theirs, not a fact about the reference.
        ("quality_above_127", "ACGTACGTACGTACGTACGT", "ACTTAC", [30, 30, 200, 30, 30, 30], 8),
        ("quality_above_127_negative_total", "AAAAAAAAAAAAAAAA", "CACA", [250, 10, 250, 10], 5),
        # the original alignment hangs past the end, offsets left of it still hang, others do not
        ("hanging_left_offsets", ref, ref[-12:-4] + "GGGGGGGG", [7] * 16, len(ref) - 8),
        ("long_read_many_steps", ref * 3, (ref * 3)[17:17 + 61], [(i * 7) % 41 for i in range(61)], 40),
    ]
    return out


def main():
    if not fastr.reference_available():
        sys.exit("build the reference first: bash oracle/build_ref.sh")
    cases = []
    for name, ref, read, quals, orig in hand_cases():
        i, s = fastr.reference_find_best_offset(np.frombuffer(ref.encode(), np.uint8),
                                                np.frombuffer(read.encode(), np.uint8), np.array(quals, np.uint8), orig)
        cases.append({"name": name, "ref": ref, "read": read, "quals": quals, "original_alignment": orig,
                      "best_index": i, "best_score": s})
    batch = fastr.make_batch(seed=0xFA57, n_regions=3, max_consensus=4, reads=(6, 14), read_len=36, ref_len=(80, 160))
    idx, sc = [], []
    for r, rd, q, o in fastr.pair_iter(batch):
        i, s = fastr.reference_find_best_offset(r, rd, q, o)
        idx.append(i)
        sc.append(s)
    out = {
        "cases": cases,
        "batch": {
            "generator": {"seed": 0xFA57, "n_regions": 3, "max_consensus": 4, "reads": [6, 14], "read_len": 36,
                          "ref_len": [80, 160]},
            "ref": batch["ref"].tobytes().decode(), "ref_lens": batch["ref_lens"].tolist(),
            "n_consensus": batch["n_consensus"].tolist(), "alt_reads": batch["alt_reads"].tolist(),
            "read_seq": batch["read_seq"].tobytes().decode(), "quals": batch["quals"].tolist(),
            "read_len": batch["read_len"].tolist(), "original_alignment": batch["original_alignment"].tolist(),
            "read_stride": batch["read_stride"],
            "best_index": idx, "best_score": sc,
        },
    }
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fastr_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, len(cases), "cases +", len(idx), "batch pairs")
    for c in cases:
        print(c["name"], c["best_index"], c["best_score"])


if __name__ == "__main__":
    main()

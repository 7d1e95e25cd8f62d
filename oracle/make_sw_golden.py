"""Writes tests/golden/sw_golden.json: inputs and the outputs of the REFERENCE's own
SWPairwiseAlignment (apps/fastr/SWPairwiseAlignment.cpp, through oracle/_ref/libfastr_ref.so,
built by oracle/build_ref.sh where /root/reference exists).  TEST INFRASTRUCTURE ONLY.

  python -m oracle.make_sw_golden
"""
import json
import os

from . import sw

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "sw_golden.json")

TOY = [
    ("AAAAACCCCCGGGGGTTTTT", "CCCCCGGGGG"), ("ACGTACGTACGTTTTACGT", "ACGTACGTACGT"), ("ACGT", "ACGT"), ("A", "A"),
    ("A", "C"), ("ACGTTTTT", "GGGGACGT"), ("AAAACCCCGGGGTTTT", "AAAACCCCTTTT"), ("AAAACCCCTTTT", "AAAACCCCGGGGTTTT"),
    ("GATTACA", "TTTTGATTACATTTT"), ("ACGTACGTAC", "TTTTTTTTTT"), ("AAAAAAAAAAAAAAAA", "AAAA"), ("ACACACACACACAC", "ACACAC"),
    ("TTTTACGTACGTTTTT", "ACGTNCGT"), ("CCCCCCCCCCACGTGGGGGGGGGG", "ACGTACGTACGTACGTACGT"),
]
FRACTIONAL = (1.0, -1.0 / 3.0, -1.0 - 1.0 / 3.0, -1.0 / 3.0)  # the class's own default weights (:73-81)


def main():
    if not sw.reference_available():
        raise SystemExit("oracle/_ref/libfastr_ref.so missing: run oracle/build_ref.sh")
    cases = []
    for ref, read in TOY:
        for w in (sw.DEFAULT_WEIGHTS, FRACTIONAL):
            off, cigar = sw.reference_align(ref, read, w)
            cases.append({"ref": ref, "read": read, "weights": list(w), "offset": off, "cigar": cigar})
    b = sw.make_pairs(seed=0x5117, n_windows=6, reads_per_window=(6, 10), ref_len=(150, 700))
    for p in range(b["read_len"].size):
        ref = bytes(b["ref"][b["ref_off"][p]:b["ref_off"][p] + b["ref_len"][p]]).decode()
        read = bytes(b["reads"][p, :b["read_len"][p]]).decode()
        off, cigar = sw.reference_align(ref, read)
        cases.append({"ref": ref, "read": read, "weights": list(sw.DEFAULT_WEIGHTS), "offset": off, "cigar": cigar})
    with open(OUT, "w") as f:
        json.dump({"source": "apps/fastr/SWPairwiseAlignment.cpp via oracle/fastr_ref_harness.cpp (fastr_ref_sw_align)",
                   "generator": "oracle/make_sw_golden.py", "cases": cases}, f, indent=0)
    print("wrote %d cases to %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()

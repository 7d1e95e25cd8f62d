"""Short-K contraction abcijk = adij,bcdk (a = 5, d = 64) over free extents that are and are not multiples of 8:
the one-tile-per-workgroup kernel, the persistent kernel and the automatic choice.  One JSON line each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import stream_cmp  # noqa: E402

if __name__ == "__main__":
    extents = [int(x) for x in sys.argv[1:]] or [40, 42, 44, 46, 48, 50]
    for n in extents:
        stream_cmp.run("slab_a5_n%d_d64" % n, "adij,bcdk->abcijk", dict(a=5, b=n, c=n, i=n, j=n, k=n, d=64), "f64", iters=6)

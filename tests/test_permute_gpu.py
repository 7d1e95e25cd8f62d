"""Mode permutation kernels (through the C ABI) against the CPU oracle: bit-exact."""
import numpy as np
import pytest
import torch

import ctc_amd

pytestmark = pytest.mark.gpu

CASES = [
    ((3, 4, 5), (2, 0, 1)),
    ((64, 64), (1, 0)),
    ((70, 130), (1, 0)),
    ((7, 1, 6, 2), (3, 2, 1, 0)),
    ((8,), (0,)),
    ((5, 6, 7), (0, 1, 2)),
    ((2, 3, 4, 5, 6), (0, 4, 1, 3, 2)),
    ((16, 8, 12, 16), (3, 1, 2, 0)),
    ((32, 32, 32, 32), (3, 2, 1, 0)),
    ((9, 300, 3), (1, 0, 2)),
    ((4, 5, 2000), (1, 0, 2)),
    ((33, 3, 65), (2, 1, 0)),
    ((6, 5, 4, 3, 2, 7), (5, 0, 3, 1, 4, 2)),
    # fastest extents that are no multiple of a 16-byte vector: whole vectors at element-aligned addresses,
    # element tails (4- and 8-byte elements), the element kernel for 2-byte ones
    ((97, 33, 35), (2, 1, 0)),
    ((40, 111), (1, 0)),
    ((5, 67, 3, 66), (1, 3, 2, 0)),
    ((129, 65), (1, 0)),
    ((33, 7, 130), (2, 1, 0)),
    # patches that would hang over the edge: further modes are fused into the patch sides
    ((12, 16, 80, 112), (1, 0, 3, 2)),
    ((12, 16, 80, 112), (3, 2, 1, 0)),
    ((10, 3, 100, 36), (3, 1, 2, 0)),
    ((6, 40, 24, 72), (0, 3, 2, 1)),
]


@pytest.mark.parametrize("shape,perm", CASES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.int16])
def test_permute_matches_oracle(cuda, oracle, shape, perm, dtype):
    n = int(np.prod(shape))
    host = (np.arange(n) * 2654435761 % 65521).astype({torch.float64: np.float64, torch.float32: np.float32,
                                                       torch.int16: np.int16}[dtype])
    x = torch.from_numpy(host).to(cuda).reshape(shape)
    got = ctc_amd.permute(x, perm)
    assert tuple(got.shape) == tuple(shape[p] for p in perm)
    want = oracle.permute(host, shape, perm)
    assert np.array_equal(got.cpu().numpy().reshape(-1), want)


def test_full_size_reversal_round_trip(cuda):
    """128^4 fp32 (BASELINE config 3): dcba <- abcd applied twice is the identity,
    and it agrees with torch's own permute on a sampled slab."""
    x = torch.empty(128 ** 4, dtype=torch.float32, device=cuda)
    ctc_amd.fill(x, 0xC7C0A)
    x = x.reshape(128, 128, 128, 128)
    y = ctc_amd.permute(x, (3, 2, 1, 0))
    assert torch.equal(y[3, 5], x[:, :, 5, 3].t())
    z = ctc_amd.permute(y, (3, 2, 1, 0))
    assert torch.equal(z, x)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("shape,perm", [((24, 36, 64), (2, 1, 0)), ((48, 128), (1, 0)), ((6, 64, 68), (1, 2, 0))])
def test_permute_of_views_that_are_not_16_byte_aligned(cuda, oracle, shape, perm, dtype):
    """Input and output begin one element into their allocations."""
    n = int(np.prod(shape))
    host = (np.arange(n) * 40503 % 65521).astype({torch.float64: np.float64, torch.float32: np.float32}[dtype])
    xbuf = torch.zeros(n + 1, dtype=dtype, device=cuda)
    xbuf[1:] = torch.from_numpy(host).to(cuda)
    obuf = torch.full((n + 2,), -7.0, dtype=dtype, device=cuda)
    out = obuf[1:n + 1].reshape([shape[p] for p in perm])
    ctc_amd.permute(xbuf[1:].reshape(shape), perm, out=out)
    got = obuf.cpu().numpy()
    assert got[0] == -7.0 and got[-1] == -7.0
    assert np.array_equal(got[1:n + 1], oracle.permute(host, shape, perm))

"""faiss_amd/datasets.py -- synthetic inputs shaped like the reference's benchmarks.

``synthetic_dataset`` restates the reference's own SyntheticDataset recipe
(contrib/datasets.py:89-105): a 10-dimensional Gaussian pushed through a random linear map,
per-dimension scaling and sin().  It is the generator SURVEY.md section 8(d) prescribes for the
"SIFT1M-shaped" workload (d=128, nt=100k, nb=1M, nq=10k, seed 1338): clustered enough that IVF
recall is meaningful, unlike uniform noise in 128 dimensions.
"""
import numpy as np


def synthetic_dataset(d, nt, nb, nq, seed=1338, return_map=False):
    """Returns (xt, xb, xq) float32, identical to SyntheticDataset(d, nt, nb, nq, seed=seed).
    return_map=True also returns (proj, scale), the random map of this dataset, for synthetic_more()."""
    d1 = 10  # intrinsic dimension (more or less)
    n = nb + nt + nq
    rs = np.random.RandomState(seed)
    x = rs.normal(size=(n, d1))
    proj = rs.rand(d1, d)
    x = np.dot(x, proj)
    scale = rs.rand(d) * 4 + 0.1
    x = x * scale
    x = np.sin(x).astype("float32")
    if return_map:
        return x[:nt], x[nt:nt + nb], x[nt + nb:], (proj, scale)
    return x[:nt], x[nt:nt + nb], x[nt + nb:]


def synthetic_more(dmap, n, seed):
    """n further vectors of the SAME distribution as a synthetic_dataset(..., return_map=True) call (same
    low-dimensional map, fresh latent draws): databases of 10M-1B rows are produced chunk by chunk
    with seed = base + chunk, never materialised on the host at once."""
    proj, scale = dmap
    rs = np.random.RandomState(seed)
    x = rs.normal(size=(n, proj.shape[0]))
    return np.sin(np.dot(x, proj) * scale).astype("float32")


def synthetic_more_device(dmap, n, seed, device):
    """synthetic_more() drawn ON THE DEVICE: the same low-dimensional map (proj, scale) applied in fp64 to fresh latent
    draws of a torch generator seeded `seed` on `device`; returns a contiguous float32 torch tensor [n][d] that add()
    takes as a device pointer.  The host recipe costs ~2 s per million rows (numpy normal + a 10 x d matmul in fp64),
    which is all of the build time of a 100M-1B row database; the rows differ from synthetic_more()'s (another
    generator), their distribution does not."""
    import torch
    proj = torch.from_numpy(np.ascontiguousarray(dmap[0], dtype=np.float64)).to(device)
    scale = torch.from_numpy(np.ascontiguousarray(dmap[1], dtype=np.float64)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    lat = torch.randn((n, proj.shape[0]), generator=g, device=device, dtype=torch.float64)
    x = torch.sin(torch.matmul(lat, proj) * scale).to(torch.float32).contiguous()
    torch.cuda.synchronize(device)
    return x


def integer_dataset(d, nb, nq, seed=7, hi=16):
    """Small-integer coordinates: every partial sum is exact in fp32, so every summation order
    gives identical bits and exact distance ties are frequent (tie-rule stress)."""
    rs = np.random.RandomState(seed)
    xb = rs.randint(0, hi, size=(nb, d)).astype("float32")
    xq = rs.randint(0, hi, size=(nq, d)).astype("float32")
    return xb, xq

"""Deterministic weights shared by every side of a parity check (TEST INFRASTRUCTURE).

The reference model, the CPU oracle and the HIP product all have to hold identical weights
for a parity comparison, and 1.4 GB of weights cannot be committed.  Every tensor is
therefore regenerated from its *name*: seed = crc32(name) + base, PCG64 standard normals,
scaled by the kind of tensor.  LayerNorm gains/biases are perturbed away from (1, 0) and
the zero-initialised tokens of the reference (cls/act/rel tokens, biases) get small
non-zero values so that a wiring mistake cannot hide behind an identity.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math
import zlib

import numpy as np


def tensor_for(name, shape, base_seed=2024):
    shape = tuple(int(s) for s in shape)
    rng = np.random.default_rng(zlib.crc32(name.encode()) + base_seed)
    z = rng.standard_normal(shape, dtype=np.float32)
    if len(shape) == 1:
        if name.endswith("weight"):          # every 1-D "weight" on the path is a LayerNorm gain
            return (1.0 + 0.05 * z).astype(np.float32)
        return (0.02 * z).astype(np.float32)
    if len(shape) == 5:                      # Conv3d weight (out, in, kt, kh, kw)
        fan_in = shape[1] * shape[2] * shape[3] * shape[4]
        return (z / math.sqrt(fan_in)).astype(np.float32)
    if name.endswith("in_proj_weight"):
        return (0.03 * z).astype(np.float32)
    return (0.02 * z).astype(np.float32)


def fill(named_shapes, base_seed=2024):
    """named_shapes: iterable of (name, shape) -> dict name -> float32 ndarray."""
    return {n: tensor_for(n, s, base_seed) for n, s in named_shapes}

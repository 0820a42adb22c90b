"""CPU restatement of the reference's pyramid pooling rules (TEST INFRASTRUCTURE ONLY).

mean: scripts/create_mouse_multiscale.py:23-54 — ``reshape(n0,2,n1,2,n2,2).mean(axis=(1,3,5))``; the
product stores the density level in the source dtype: uint8 -> floor of the mean, float32 -> the mean
evaluated as ((a+b)+(c+d)) + ((e+f)+(g+h)) times 0.125 (numpy's own f32 summation order is an
implementation detail; this fixed order is the product's contract).
max:  scripts/create_platynereis_multiscale.py:86-134 — block-wise maximum of the labels."""
import numpy as np


def _blocks(a):
    n0, n1, n2 = (s // 2 for s in a.shape)
    return a.reshape(n0, 2, n1, 2, n2, 2)


def mean_u8(a):
    return (_blocks(a).astype(np.uint32).sum(axis=(1, 3, 5)) // 8).astype(np.uint8)


def mean_f32(a):
    b = _blocks(a.astype(np.float32))
    s0 = (b[:, 0, :, 0, :, 0] + b[:, 0, :, 0, :, 1]) + (b[:, 0, :, 1, :, 0] + b[:, 0, :, 1, :, 1])
    s1 = (b[:, 1, :, 0, :, 0] + b[:, 1, :, 0, :, 1]) + (b[:, 1, :, 1, :, 0] + b[:, 1, :, 1, :, 1])
    return ((s0 + s1) * np.float32(0.125)).astype(np.float32)


def max_u32(a):
    return _blocks(a).max(axis=(1, 3, 5))

"""Shared helpers for the test-suite: fixtures of the reference's ring-buffer tests."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

with open(os.path.join(GOLDEN, "ring_known_answers.json")) as f:
    KNOWN = json.load(f)


def fixture_arrays(name):
    fx = KNOWN["fixtures"][name]
    shape = tuple(fx["data_shape"])
    data = np.arange(int(np.prod(shape)), dtype=fx["dtype"]).reshape(shape)
    seg = np.zeros(shape, dtype=fx["dtype"])
    return data, seg, tuple(fx["ring_chunks"]), tuple(fx["chunk"])


def slices(r):
    return tuple(slice(o, o + s) for o, s in zip(r[0], r[1]))


def as_pair(r):
    return (tuple(r[0]), tuple(r[1]))

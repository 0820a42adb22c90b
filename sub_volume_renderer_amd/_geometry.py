"""Integer ROI algebra used by the ring buffers.

The reference delegates this to ``funlib.geometry`` 0.3.0 (``Roi``,
``Coordinate``; pixi.lock:449), which is not available on the target.  Only the
calls the reference actually makes are provided (call sites:
``_wrapping_buffer.py:132-141,173,283-309,344-375`` and
``_wobject.py:173-204``): ``offset, shape, begin, end, empty, size, dims,
intersect, intersects, contains, snap_to_grid(mode="grow")``, ``Roi +- Coordinate``,
``Roi * / Coordinate`` and ``==``.  Unbounded (``None``) extents are not
supported: the ring buffers never use them.
"""

from __future__ import annotations

from math import prod
from typing import Iterable


class Coordinate(tuple):
    """An integer point / extent; arithmetic is element-wise."""

    def __new__(cls, *values):
        if len(values) == 1 and not isinstance(values[0], (int, float)) and values[0] is not None:
            values = values[0]
        return super().__new__(cls, (int(v) for v in values))

    @property
    def dims(self) -> int:
        return len(self)

    def _zip(self, other):
        if isinstance(other, (tuple, list)):
            if len(other) != len(self):
                raise AssertionError(
                    f"can not combine coordinates of different dimensions: {self} and {tuple(other)}"
                )
            return zip(self, other)
        return ((a, other) for a in self)

    def __add__(self, other):
        return Coordinate(a + b for a, b in self._zip(other))

    __radd__ = __add__

    def __sub__(self, other):
        return Coordinate(a - b for a, b in self._zip(other))

    def __rsub__(self, other):
        return Coordinate(b - a for a, b in self._zip(other))

    def __mul__(self, other):
        return Coordinate(a * b for a, b in self._zip(other))

    __rmul__ = __mul__

    def __neg__(self):
        return Coordinate(-a for a in self)

    def __abs__(self):
        return Coordinate(abs(a) for a in self)

    def __truediv__(self, other):
        # funlib truncates the true quotient back to int; for the chunk-aligned
        # operands used here the quotient is exact, so do it without floats
        # (stays exact for arbitrarily large ints).
        out = []
        for a, b in self._zip(other):
            q = abs(a) // abs(b)
            out.append(q if (a >= 0) == (b >= 0) else -q)
        return Coordinate(out)

    def __floordiv__(self, other):
        return Coordinate(a // b for a, b in self._zip(other))

    def __mod__(self, other):
        return Coordinate(a % b for a, b in self._zip(other))

    def __repr__(self):
        return "(" + ", ".join(str(a) for a in self) + ")"


def _ceil_div(a: int, b: int) -> int:
    return -((-a) // b)


class Roi:
    """Axis-aligned box ``[offset, offset + shape)`` of integer coordinates."""

    __slots__ = ("_offset", "_shape")

    def __init__(self, offset: Iterable[int] | None = None, shape: Iterable[int] | None = None):
        if shape is None:
            raise ValueError("a Roi needs a shape")
        self._shape = Coordinate(shape)
        self._offset = Coordinate(offset) if offset is not None else Coordinate((0,) * len(self._shape))
        if len(self._offset) != len(self._shape):
            raise AssertionError("offset dimension and shape dimension do not match")
        if any(s < 0 for s in self._shape):
            raise AssertionError(f"negative shape {self._shape}")

    # -- plain accessors -------------------------------------------------
    @property
    def offset(self) -> Coordinate:
        return self._offset

    @property
    def shape(self) -> Coordinate:
        return self._shape

    @property
    def begin(self) -> Coordinate:
        return self._offset

    @property
    def end(self) -> Coordinate:
        return self._offset + self._shape

    @property
    def dims(self) -> int:
        return len(self._shape)

    @property
    def size(self) -> int:
        return prod(self._shape)

    @property
    def empty(self) -> bool:
        return self.size == 0

    # -- set algebra -----------------------------------------------------
    def intersects(self, other: "Roi") -> bool:
        if self.dims != other.dims:
            raise AssertionError("ROIs must have the same number of dimensions")
        if self.empty or other.empty:
            return False
        return all(
            b1 < e2 and b2 < e1
            for b1, e1, b2, e2 in zip(self.begin, self.end, other.begin, other.end)
        )

    def intersect(self, other: "Roi") -> "Roi":
        if not self.intersects(other):
            return Roi((0,) * self.dims, (0,) * self.dims)
        begin = Coordinate(max(b1, b2) for b1, b2 in zip(self.begin, other.begin))
        end = Coordinate(min(e1, e2) for e1, e2 in zip(self.end, other.end))
        return Roi(begin, end - begin)

    def contains(self, other) -> bool:
        if isinstance(other, Roi):
            if other.empty:
                return self.contains(other.begin)
            return all(
                b1 <= b2 and e2 <= e1
                for b1, e1, b2, e2 in zip(self.begin, self.end, other.begin, other.end)
            )
        return all(b <= c < e for c, b, e in zip(other, self.begin, self.end))

    def snap_to_grid(self, voxel_size, mode: str = "grow") -> "Roi":
        voxel_size = Coordinate(voxel_size)
        if mode == "grow":
            begin = Coordinate(b // v for b, v in zip(self.begin, voxel_size))
            end = Coordinate(_ceil_div(e, v) for e, v in zip(self.end, voxel_size))
        elif mode == "shrink":
            begin = Coordinate(_ceil_div(b, v) for b, v in zip(self.begin, voxel_size))
            end = Coordinate(e // v for e, v in zip(self.end, voxel_size))
            end = Coordinate(max(b, e) for b, e in zip(begin, end))
        elif mode == "closest":
            begin = Coordinate((2 * b + v) // (2 * v) for b, v in zip(self.begin, voxel_size))
            end = Coordinate((2 * e + v) // (2 * v) for e, v in zip(self.end, voxel_size))
        else:
            raise RuntimeError(f"unknown mode {mode} for snap_to_grid")
        return Roi(begin * voxel_size, (end - begin) * voxel_size)

    # -- arithmetic ------------------------------------------------------
    def __add__(self, other):
        return Roi(self._offset + other, self._shape)

    def __sub__(self, other):
        return Roi(self._offset - other, self._shape)

    def __mul__(self, other):
        return Roi(self._offset * other, self._shape * other)

    def __truediv__(self, other):
        return Roi(self._offset / other, self._shape / other)

    def __floordiv__(self, other):
        return Roi(self._offset // other, self._shape // other)

    def __eq__(self, other):
        return isinstance(other, Roi) and self._offset == other._offset and self._shape == other._shape

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash((tuple(self._offset), tuple(self._shape)))

    def to_slices(self) -> tuple[slice, ...]:
        return tuple(slice(int(o), int(o) + int(s)) for o, s in zip(self._offset, self._shape))

    def __repr__(self):
        return f"[{', '.join(f'{b}:{e}' for b, e in zip(self.begin, self.end))}] ({self.shape})"

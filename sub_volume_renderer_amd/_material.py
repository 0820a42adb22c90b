"""``SubVolumeMaterial`` — the parameter block of the LMIP march.

Drop-in for the reference's ``SubVolumeMaterial(gfx.VolumeMipMaterial)``
(``src/sub_volume/_material.py``): same constructor arguments and defaults, same property
names, same exception types on bad values.  The reference stores the values in a pygfx
uniform buffer; here they live in a plain dict of numpy values with the field types of that
uniform block (``_material.py:6-24``: f4 / i4 / 3xf4 / u4 / n*4xf4) and reach the device
through ``svr_set_material`` (include/svr.h) whenever ``_version`` moved.

The scalar uniforms are declared once, as data descriptors (``_Scalar``); the three
structured ones (``clim``, ``fog_color``, ``colors``) are properties with their own checks.
"""

from __future__ import annotations

import numbers

import numpy as np

# the hues pygfx-side code falls back to when no colours are given (_material.py:51-57)
_DEFAULT_HUES = (0.0, 0.25, 0.5, 0.75)


class _Scalar:
    """One scalar uniform: converts on assignment to the uniform's storage type and bumps the
    material's version so that the next draw re-sends the block."""

    def __init__(self, storage, doc, lo=None, hi=None):
        self.storage, self.lo, self.hi = storage, lo, hi
        self.__doc__ = doc

    def __set_name__(self, owner, name):
        self.field = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        value = obj._u[self.field]
        return float(value) if self.lo is not None else value      # clamped fields read back as python floats

    def __set__(self, obj, value):
        number = int(value) if np.issubdtype(self.storage, np.integer) else float(value)
        if self.lo is not None:
            number = min(max(number, self.lo), self.hi)
        obj._store(self.field, self.storage(number))


class SubVolumeMaterial:
    # inherited from pygfx's VolumeMipMaterial in the reference
    gamma = _Scalar(np.float32, "Exponent applied to the contrast-limited value before colouring.", -np.inf, np.inf)
    opacity = _Scalar(np.float32, "Alpha written for hit pixels (fs_main.wgsl:86), kept inside [0, 1].", 0.0, 1.0)
    # the LMIP uniforms (_material.py:60-99)
    lmip_threshold = _Scalar(np.float32, "The minimum intensity considered significant for the LMIP algorithm.")
    lmip_fall_off = _Scalar(np.float32, "The fraction of the maximum intensity that is still considered significant.")
    lmip_max_samples = _Scalar(np.int32, "How many samples are examined after the first significant one (i32 in the shader).")
    fog_density = _Scalar(np.float32, "The density of the fog effect applied to the volume.")

    def __init__(
        self,
        lmip_threshold: float,
        lmip_fall_off: float = 0.5,
        lmip_max_samples: int = 10,
        fog_density: float = 0.5,
        fog_color: tuple[float, float, float] = (0.5, 0.5, 0.5),
        colors: list[tuple[float, float, float]] | None = None,
        clim: tuple[float, float] = (0, 1),
        gamma: float = 1.0,
        opacity: float = 1.0,
    ):
        self._u = {}
        self._version = 0
        self.depth_test = True          # the reference insists on depth testing (_material.py:39-45)
        # inherited from pygfx's Material: no clipping planes unless the caller sets some
        self.clipping_planes = ()
        self.clipping_mode = "ANY"
        self.render_mode = "lmip"
        self.weight_falloff = 0.5
        arguments = dict(clim=clim, gamma=gamma, opacity=opacity, lmip_threshold=lmip_threshold,
                         lmip_fall_off=lmip_fall_off, lmip_max_samples=lmip_max_samples, fog_density=fog_density,
                         fog_color=fog_color,
                         colors=[(h, 1.0, 1.0) for h in _DEFAULT_HUES] if colors is None else colors)
        for name, value in arguments.items():
            setattr(self, name, value)

    def _store(self, field, value):
        self._u[field] = value
        self._version += 1

    # -- clim (VolumeMipMaterial) ----------------------------------------------------------------
    @property
    def clim(self) -> tuple[float, float]:
        """The contrast limits applied before colouring (sampled_value_to_color)."""
        lo, hi = self._u["clim"]
        return float(lo), float(hi)

    @clim.setter
    def clim(self, limits) -> None:
        if not isinstance(limits, (tuple, list)) or len(limits) != 2:
            raise TypeError("Material.clim must be a 2-tuple")
        self._store("clim", np.asarray([float(v) for v in limits], np.float32))

    # -- fog colour: three numbers in [0, 1], anything else is a ValueError (_material.py:100-117) --------
    @property
    def fog_color(self) -> tuple[float, float, float]:
        """The color of the fog effect applied to the volume."""
        return tuple(self._u["fog_color"])

    @fog_color.setter
    def fog_color(self, rgb) -> None:
        components = list(rgb)
        if len(components) != 3:
            raise ValueError(f"fog_color needs exactly three components (r, g, b), got {len(components)}")
        for c in components:
            if not isinstance(c, numbers.Real):
                raise ValueError(f"fog_color components must be numbers, not {type(c).__name__}")
        packed = np.asarray(components, np.float32)
        if packed.min() < 0.0 or packed.max() > 1.0:
            raise ValueError("fog_color components must lie in [0, 1]")
        self._store("fog_color", packed)

    # -- clipping planes (pygfx Material; the shader includes pygfx.clipping_planes.wgsl, fs_main.wgsl:8) ----
    MAX_CLIPPING_PLANES = 8

    @property
    def clipping_planes(self) -> list[tuple[float, float, float, float]]:
        """World-space planes (a, b, c, d).  The draw skips a pixel whose ray leaves the volume's box at a
        point p with ``dot(p, abc) < d`` for ANY plane (``clipping_mode == "ANY"``) or for ALL of them."""
        return [tuple(map(float, row)) for row in self._u["clipping_planes"]]

    @clipping_planes.setter
    def clipping_planes(self, planes) -> None:
        rows = []
        for plane in planes:
            if isinstance(plane, (str, bytes)) or len(plane) != 4:
                raise TypeError(f"Each clipping plane must be an abcd tuple, not {plane}")
            rows.append([float(v) for v in plane])
        if len(rows) > self.MAX_CLIPPING_PLANES:
            raise ValueError(f"at most {self.MAX_CLIPPING_PLANES} clipping planes are supported")
        self._store("clipping_planes", np.asarray(rows, np.float32).reshape(-1, 4))

    @property
    def clipping_mode(self) -> str:
        """"ANY": a point is clipped if it is behind any plane; "ALL": only if it is behind all of them."""
        return self._u["clipping_mode"]

    @clipping_mode.setter
    def clipping_mode(self, mode) -> None:
        mode = str(mode).upper()
        if mode not in ("ANY", "ALL"):
            raise ValueError(f"Unexpected clipping_mode: {mode}")
        self._store("clipping_mode", mode)

    # -- render mode: the swappable raycast the reference wishes for (FUTURE.md:97-120) -------------------------
    RENDER_MODES = ("lmip", "mip", "weighted_average")

    @property
    def render_mode(self) -> str:
        """"lmip" (the reference's raycast.wgsl) or "mip": the maximum over the WHOLE ray, first occurrence —
        what pygfx's own ``VolumeMipMaterial`` raycast selects (without its sub-step refinement).  MIP is the
        LMIP state machine with every sample significant, no fall-off and no sample limit, so it runs on the
        same kernel: the draw sends threshold = -inf, fall_off = 0, max_samples = 2**31 - 1 and leaves the
        ``lmip_*`` properties untouched.

        "weighted_average": the mode FUTURE.md:97-109 wishes for ("weight each sample by distance ... sampling a
        finite number of points based on distance") and gives no formula for; defined in ``include/svr.h``
        (``SVR_MODE_WEIGHTED_AVERAGE``): sample i weighs ``max(1 - weight_falloff * d_i, 0) ** 2``, the pixel shows
        the weighted mean of the ray's samples at the sample that contributes most."""
        return self._u["render_mode"]

    @render_mode.setter
    def render_mode(self, mode) -> None:
        mode = str(mode).lower()
        if mode not in self.RENDER_MODES:
            raise ValueError(f"render_mode must be one of {self.RENDER_MODES}, not {mode!r}")
        self._store("render_mode", mode)

    @property
    def weight_falloff(self) -> float:
        """"weighted_average" mode: how fast a sample's weight falls with its distance d from the ray's entry
        into the volume (d in the fog's unit: 1 = one volume edge); samples at d >= 1 / weight_falloff weigh
        nothing and are not taken.  0 = plain mean of the whole ray."""
        return float(self._u["weight_falloff"])

    @weight_falloff.setter
    def weight_falloff(self, value) -> None:
        value = float(value)
        if not (0.0 <= value < float("inf")):
            raise ValueError(f"weight_falloff must be finite and >= 0, not {value!r}")
        self._store("weight_falloff", np.float32(value))

    def lmip_uniforms(self) -> tuple[float, float, int]:
        """(threshold, fall_off, max_samples) as the draw sends them for the current render mode."""
        if self._u["render_mode"] == "mip":
            return float("-inf"), 0.0, 2**31 - 1
        return float(self._u["lmip_threshold"]), float(self._u["lmip_fall_off"]), int(self._u["lmip_max_samples"])

    # -- label hues: list of (h, s, v); stored vec4-padded like the reference's n*4xf4 array (:119-159) ---
    @property
    def _color_count(self):
        return np.uint32(len(self._u["colors"]))

    @property
    def colors(self) -> list[tuple[float, float, float]]:
        """The list of HSV colors used for rendering labels (each padded to four components)."""
        return [tuple(map(float, row)) for row in self._u["colors"]]

    @colors.setter
    def colors(self, hsv_list) -> None:
        if not isinstance(hsv_list, (list, tuple)):
            raise TypeError("Colors must be a list.")
        table = np.ones((len(hsv_list), 4), np.float32)               # fourth component: padding, always 1
        for row, hsv in zip(table, hsv_list):
            if not isinstance(hsv, (list, tuple)) or len(hsv) != 3:
                raise TypeError(f"Each color must be an hsv tuple, not {hsv}")
            row[:3] = hsv
        self._store("colors", table)

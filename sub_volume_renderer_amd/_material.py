"""``SubVolumeMaterial`` — the parameter block of the LMIP march.

Mirror of the reference's ``SubVolumeMaterial(gfx.VolumeMipMaterial)``
(``src/sub_volume/_material.py:5-159``): same constructor, defaults, property
names, validation and exception types.  Where the reference keeps the values in
a pygfx uniform buffer, this class keeps them in a small numpy record with the
same field types (``_material.py:6-24``) and hands them to the device through
``svr_set_material`` (include/svr.h) whenever they changed.
"""

from __future__ import annotations

import numpy as np


class SubVolumeMaterial:
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
        # the fields of the reference's uniform block (_material.py:6-24) plus the
        # inherited VolumeMipMaterial ones (clim, gamma, opacity)
        self._u = {
            "clim": np.zeros(2, np.float32),
            "gamma": np.float32(1.0),
            "opacity": np.float32(1.0),
            "lmip_threshold": np.float32(0.0),
            "lmip_fall_off": np.float32(0.0),
            "lmip_max_samples": np.int32(0),
            "fog_density": np.float32(0.0),
            "fog_color": np.zeros(3, np.float32),
            "color_count": np.uint32(0),
            "colors": np.zeros((0, 4), np.float32),
        }
        self._version = 0
        # the reference forces depth testing on (_material.py:39-45); kept as an attribute
        self.depth_test = True
        self.clim = clim
        self.gamma = gamma
        self.opacity = opacity
        self.lmip_threshold = lmip_threshold
        self.lmip_fall_off = lmip_fall_off
        self.lmip_max_samples = lmip_max_samples
        self.fog_density = fog_density
        self.fog_color = fog_color
        if colors is None:
            # _material.py:51-57
            colors = [
                (0.0, 1.0, 1.0),
                (0.25, 1.0, 1.0),
                (0.5, 1.0, 1.0),
                (0.75, 1.0, 1.0),
            ]
        self.colors = colors

    def _touch(self):
        self._version += 1

    # -- inherited from pygfx VolumeMipMaterial ----------------------------
    @property
    def clim(self) -> tuple[float, float]:
        """The contrast limits applied before colouring (sampled_value_to_color)."""
        c = self._u["clim"]
        return float(c[0]), float(c[1])

    @clim.setter
    def clim(self, clim) -> None:
        if not (isinstance(clim, (tuple, list)) and len(clim) == 2):
            raise TypeError("Material.clim must be a 2-tuple")
        self._u["clim"] = np.array((float(clim[0]), float(clim[1])), np.float32)
        self._touch()

    @property
    def gamma(self) -> float:
        return float(self._u["gamma"])

    @gamma.setter
    def gamma(self, value: float) -> None:
        self._u["gamma"] = np.float32(float(value))
        self._touch()

    @property
    def opacity(self) -> float:
        return float(self._u["opacity"])

    @opacity.setter
    def opacity(self, value: float) -> None:
        self._u["opacity"] = np.float32(min(max(float(value), 0.0), 1.0))
        self._touch()

    # -- _material.py:60-89 --------------------------------------------------
    @property
    def lmip_threshold(self) -> float:
        """The minimum intensity considered significant for the LMIP algorithm."""
        return self._u["lmip_threshold"]

    @lmip_threshold.setter
    def lmip_threshold(self, value: float) -> None:
        self._u["lmip_threshold"] = np.float32(float(value))
        self._touch()

    @property
    def lmip_fall_off(self) -> float:
        """The fraction of the maximum intensity that is still considered significant."""
        return self._u["lmip_fall_off"]

    @lmip_fall_off.setter
    def lmip_fall_off(self, value: float) -> None:
        self._u["lmip_fall_off"] = np.float32(float(value))
        self._touch()

    @property
    def lmip_max_samples(self) -> int:
        """The maximum number of samples to consider after detecting a significant intensity."""
        return self._u["lmip_max_samples"]

    @lmip_max_samples.setter
    def lmip_max_samples(self, value: int) -> None:
        self._u["lmip_max_samples"] = np.int32(int(value))
        self._touch()

    # -- _material.py:90-117 -------------------------------------------------
    @property
    def fog_density(self) -> float:
        """The density of the fog effect applied to the volume."""
        return self._u["fog_density"]

    @fog_density.setter
    def fog_density(self, value: float) -> None:
        self._u["fog_density"] = np.float32(float(value))
        self._touch()

    @property
    def fog_color(self) -> tuple[float, float, float]:
        """The color of the fog effect applied to the volume."""
        return tuple(self._u["fog_color"])

    @fog_color.setter
    def fog_color(self, fog_color: tuple[float, float, float]) -> None:
        if len(fog_color) != 3:
            raise ValueError("fog_color must be a tuple of three floats (r, g, b)")
        if not all(isinstance(c, (int, float)) for c in fog_color):
            raise ValueError("fog_color must contain only numeric values")
        fog_color = np.array(fog_color, dtype=np.float32)
        if np.any(fog_color < 0) or np.any(fog_color > 1):
            raise ValueError("fog_color values must be in the range [0, 1]")
        self._u["fog_color"] = fog_color
        self._touch()

    # -- _material.py:119-159 ------------------------------------------------
    @property
    def _color_count(self) -> int:
        return self._u["color_count"]

    @property
    def colors(self) -> list[tuple[float, float, float]]:
        """The list of HSV colors used for rendering labels (vec4-padded like the reference)."""
        return [tuple(float(f) for f in row) for row in self._u["colors"]]

    @colors.setter
    def colors(self, colors: list[tuple[float, float, float]]):
        if not isinstance(colors, (tuple, list)):
            raise TypeError("Colors must be a list.")
        colors2 = []
        for color in colors:
            if isinstance(color, (tuple, list)) and len(color) == 3:
                # the reference pads every colour to a vec4 (_material.py:146-149)
                colors2.append((*color, 1))
            else:
                raise TypeError(f"Each color must be an hsv tuple, not {color}")
        self._u["colors"] = np.array(colors2, np.float32).reshape(len(colors2), 4)
        self._u["color_count"] = np.uint32(len(colors2))
        self._touch()

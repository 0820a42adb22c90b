"""Second, independent restatement of the reference's shader in vectorised numpy (float32).

TEST INFRASTRUCTURE ONLY.  Two restatements written separately (this one array-at-a-time in
numpy, ``lmip_oracle.c`` pixel-at-a-time in C) agreeing bit for bit on flags / labels / step
counts is the only available substitute for the un-importable reference;
``tests/test_oracle_lmip.py`` checks that agreement.  Render parity itself stays *unpinned* by
the reference (no rendered fixture exists upstream).

Follows: vs_main.wgsl:6-50, fs_main.wgsl:4-101, raycast.wgsl:11-88, sample_vol.wgsl:1-94,
hsv_selection.wgsl:1-41; pygfx ``sampled_value_to_color`` / ``srgb2physical`` restated from their
published text.  Same operation-order contract as the C oracle (see its header).
"""

from __future__ import annotations

import numpy as np

f32 = np.float32


def _mv(m, x, y, z, w):
    """column-major-equivalent M*v with the contract's association: ((M0*x + M1*y) + M2*z) + M3*w.
    ``m`` is a row-major 4x4 float32 array."""
    return [((m[r, 0] * x + m[r, 1] * y) + m[r, 2] * z) + m[r, 3] * w for r in range(4)]


def _mm(a, b):
    out = np.zeros((4, 4), f32)
    for c in range(4):
        col = _mv(a, b[0, c], b[1, c], b[2, c], b[3, c])
        for r in range(4):
            out[r, c] = col[r]
    return out


def render(rings, matrices, volume_dimensions_shader, material, width, height, colorspace_srgb=True, pick_id=None):
    """Full-frame render.  ``rings`` as in ``oracle.lmip.render``.  Returns dict of arrays."""
    with np.errstate(all="ignore"):
        return _render(rings, matrices, volume_dimensions_shader, material, width, height, colorspace_srgb, pick_id)


def _render(rings, M, vdim, mat, W, H, srgb, pick_id=None):
    world = np.asarray(M["world"], f32)
    ndc_to_data = _mm(_mm(np.asarray(M["world_inv"], f32), np.asarray(M["cam_inv"], f32)), np.asarray(M["proj_inv"], f32))
    pc = _mm(np.asarray(M["proj"], f32), np.asarray(M["cam"], f32))
    size = [f32(v) for v in vdim]
    rel = f32(min(max(np.sqrt(f32(max(size))) / f32(20.0), f32(0.1)), f32(0.8)))       # fs_main.wgsl:20

    jj, ii = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    px = (f32(2.0) * (ii + f32(0.5))) / f32(W) - f32(1.0)
    py = f32(1.0) - (f32(2.0) * (jj + f32(0.5))) / f32(H)
    one = np.ones_like(px)
    n4 = _mv(ndc_to_data, px, py, -one, one)
    f4 = _mv(ndc_to_data, px, py, one, one)
    far = [f4[k] / f4[3] for k in range(3)]
    near = [n4[k] / n4[3] for k in range(3)]
    d = [far[k] - near[k] for k in range(3)]
    ln = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    ray = [d[k] / ln for k in range(3)]
    lo = f32(-0.5)
    t1 = [(lo - near[k]) / ray[k] for k in range(3)]
    t2 = [((size[k] - f32(0.5)) - near[k]) / ray[k] for k in range(3)]
    tmax = [np.fmax(t1[k], t2[k]) for k in range(3)]
    tmin = [np.fmin(t1[k], t2[k]) for k in range(3)]
    t_exit = np.fmin(np.fmin(tmax[0], tmax[1]), tmax[2])
    t_enter = np.fmax(np.fmax(tmin[0], tmin[1]), tmin[2])
    frag = t_enter <= t_exit
    back = [near[k] + ray[k] * t_exit for k in range(3)]
    bw = _mv(world, back[0], back[1], back[2], one)
    bc = _mv(pc, bw[0], bw[1], bw[2], bw[3])
    frag &= (bc[3] > 0) & (bc[2] >= 0) & (bc[2] <= bc[3])
    # fs_main.wgsl:8, pygfx.clipping_planes.wgsl (restated, assumption A6): discard where the back-face world
    # position lies behind ANY / ALL planes: dot(world_pos, abc) < d
    planes = np.array(mat.get("clipping_planes", ()), f32).reshape(-1, 4)
    if len(planes):
        behind = [((bw[0] * p[0] + bw[1] * p[1]) + bw[2] * p[2]) < p[3] for p in planes]
        if str(mat.get("clipping_mode", "ANY")).upper() == "ALL":
            frag &= ~np.logical_and.reduce(behind)
        else:
            frag &= ~np.logical_or.reduce(behind)
    nb = [near[k] - back[k] for k in range(3)]
    dist = (nb[0] * ray[0] + nb[1] * ray[1]) + nb[2] * ray[2]
    for k in range(3):
        dist = np.fmax(dist, np.fmin((f32(-0.5) - back[k]) / ray[k], (size[k] - f32(0.5) - back[k]) / ray[k]))
    front = [back[k] + ray[k] * dist for k in range(3)]
    nf = -dist / rel + f32(0.5)
    frag &= nf >= 1.0
    nf = np.where(frag, np.fmin(nf, f32(16777216.0)), f32(1.0))
    nsteps = nf.astype(np.int32)                                   # trunc (values are >= 1)
    nstepsf = nsteps.astype(f32)
    start = [(front[k] + f32(0.5)) / size[k] for k in range(3)]
    step = [((back[k] - front[k]) / size[k]) / nstepsf for k in range(3)]

    thr, fall, maxs = f32(mat["lmip_threshold"]), f32(mat["lmip_fall_off"]), int(mat["lmip_max_samples"])
    found = np.zeros((H, W), bool)
    finished = ~frag
    lmax = np.zeros((H, W), f32)
    samp = np.zeros((H, W), f32)
    since = np.zeros((H, W), np.int32)
    hit_off = [np.zeros((H, W), f32) for _ in range(3)]
    hit_coord = [np.zeros((H, W), f32) for _ in range(3)]
    steps = np.zeros((H, W), np.uint32)

    def texel_index(coord, labels):
        """sample_vol / sample_segmentations_vol (sample_vol.wgsl:51-77) for arrays of coords."""
        val = np.zeros(coord[0].shape, np.uint32 if labels else f32)
        done = np.zeros(coord[0].shape, bool)
        dd = [coord[k] * size[k] for k in range(3)]
        for r in rings:
            sd = [dd[k] * f32(r["scale"][k]) for k in range(3)]
            ic = [s.astype(np.int32) for s in sd]
            inb = np.ones(coord[0].shape, bool)
            for k in range(3):
                inb &= (r["offset"][k] <= ic[k]) & (ic[k] < r["offset"][k] + r["shape"][k])
            sel = inb & ~done
            if sel.any():
                tex = r["labels"] if labels else r["density"]
                rz, ry, rx = tex.shape
                val[sel] = tex[ic[2][sel] % rz, ic[1][sel] % ry, ic[0][sel] % rx]
            done |= inb
        return val

    mode = str(mat.get("render_mode", "lmip"))
    mip = mode == "mip"
    wavg = mode == "weighted_average"
    it = 0
    if wavg:
        # SVR_MODE_WEIGHTED_AVERAGE (include/svr.h; this project's definition, the reference has none):
        # w = max(1 - k d, 0)^2, value = sum(w s) / sum(w), shown at the first sample with the largest w |s|
        k = f32(mat.get("weight_falloff", 0.5))
        steplen = np.sqrt((step[0] * step[0] + step[1] * step[1]) + step[2] * step[2])
        num = np.zeros((H, W), f32)
        den = np.zeros((H, W), f32)
        best = np.zeros((H, W), f32)
        going = frag.copy()
        while True:
            tw = f32(1.0) - k * (f32(it) * steplen)
            going &= (it < nsteps) & (tw > 0)
            if not going.any():
                break
            idx = np.nonzero(going)
            off = [f32(it) * step[k_][idx] for k_ in range(3)]
            coord = [start[k_][idx] + off[k_] for k_ in range(3)]
            s = texel_index(coord, labels=False)
            steps[idx] += 1
            w = tw[idx] * tw[idx]
            num[idx] = num[idx] + w * s
            den[idx] = den[idx] + w
            contribution = w * np.abs(s)
            better = contribution > best[idx]
            best[idx] = np.where(better, contribution, best[idx])
            for k_ in range(3):
                hit_off[k_][idx] = np.where(better, off[k_], hit_off[k_][idx])
                hit_coord[k_][idx] = np.where(better, coord[k_], hit_coord[k_][idx])
            it += 1
        found = best > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            samp = np.where(found, num / den, f32(0.0)).astype(f32)
    while mip:
        # MIP stated directly (not through the LMIP state machine): the largest |sample| of the whole ray,
        # first occurrence; every fragment "finds" one (its first sample starts the running maximum)
        act = frag & (it < nsteps)
        if not act.any():
            break
        idx = np.nonzero(act)
        off = [f32(it) * step[k][idx] for k in range(3)]
        coord = [start[k][idx] + off[k] for k in range(3)]
        s = texel_index(coord, labels=False)
        inten = np.abs(s)
        steps[idx] += 1
        better = ~found[idx] | (inten > lmax[idx])
        lmax[idx] = np.where(better, inten, lmax[idx])
        samp[idx] = np.where(better, s, samp[idx])
        for k in range(3):
            hit_off[k][idx] = np.where(better, off[k], hit_off[k][idx])
            hit_coord[k][idx] = np.where(better, coord[k], hit_coord[k][idx])
        found[idx] = True
        it += 1
    while not (mip or wavg):
        act = ~finished & (it < nsteps)
        if not act.any():
            break
        idx = np.nonzero(act)
        iterf = f32(it)
        off = [iterf * step[k][idx] for k in range(3)]
        coord = [start[k][idx] + off[k] for k in range(3)]
        s = texel_index(coord, labels=False)
        inten = np.abs(s)
        steps[idx] += 1
        was = found[idx]
        first_hit = ~was & (inten >= thr)
        since_a = since[idx] + was.astype(np.int32)
        take = first_hit | (was & (inten > lmax[idx]))
        lm = np.where(take, inten, lmax[idx])
        brk = was & ((since_a >= maxs) | (inten < lm * fall))
        lmax[idx] = lm
        samp[idx] = np.where(take, s, samp[idx])
        since[idx] = since_a
        for k in range(3):
            hit_off[k][idx] = np.where(take, off[k], hit_off[k][idx])
            hit_coord[k][idx] = np.where(take, coord[k], hit_coord[k][idx])
        found[idx] = was | first_hit
        finished[idx] = brk
        it += 1

    flags = np.where(frag, np.where(found, 2, 1), 0).astype(np.uint8)
    label = np.zeros((H, W), np.uint32)
    rgba = np.zeros((H, W, 4), f32)
    depth = np.zeros((H, W), f32)
    rgba[frag & ~found] = (0, 0, 0, 1)
    hit = np.nonzero(found)
    if len(hit[0]):
        c = [hit_coord[k][hit] for k in range(3)]
        lab = texel_index(c, labels=True)
        label[hit] = lab
        v = (samp[hit] - f32(mat["clim"][0])) / (f32(mat["clim"][1]) - f32(mat["clim"][0]))
        v = np.power(v, f32(mat["gamma"]), dtype=f32)
        if srgb:
            v = np.where(v <= f32(0.04045), v / f32(12.92), np.power((v + f32(0.055)) / f32(1.055), f32(2.4), dtype=f32))
        wp = _mv(world, c[0] - f32(0.5), c[1] - f32(0.5), c[2] - f32(0.5), np.ones_like(c[0]))
        ndc = _mv(pc, wp[0], wp[1], wp[2], wp[3])
        depth[hit] = ndc[2] / np.fmax(ndc[3], f32(0.001))
        colors = np.asarray(mat["colors"], f32)
        hs = colors[lab % np.uint32(len(colors))]
        h_, s_ = hs[:, 0], hs[:, 1]
        h6 = h_ * f32(6.0)
        fl = np.floor(h6)
        sector = fl.astype(np.int32)
        fr = h6 - fl
        p = v * (f32(1.0) - s_)
        q = v * (f32(1.0) - s_ * fr)
        t = v * (f32(1.0) - s_ * (f32(1.0) - fr))
        choices_r = [v, q, p, p, t]
        choices_g = [t, v, v, q, p]
        choices_b = [p, p, t, v, v]
        r = np.select([sector == k for k in range(5)], choices_r, v)
        g = np.select([sector == k for k in range(5)], choices_g, p)
        b = np.select([sector == k for k in range(5)], choices_b, q)
        grey = s_ == 0
        r, g, b = np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)
        o = [hit_off[k][hit] for k in range(3)]
        distance = np.sqrt((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2])
        fog = np.exp(-f32(mat["fog_density"]) * distance, dtype=f32)
        omf = f32(1.0) - fog
        fc = [f32(x) for x in mat["fog_color"]]
        rgba[hit] = np.stack([fc[0] * omf + r * fog, fc[1] * omf + g * fog, fc[2] * omf + b * fog,
                              np.full_like(r, f32(mat["opacity"]))], axis=-1)
    out = dict(rgba=rgba, depth=depth, label=label, flags=flags, steps=steps)
    if pick_id is not None:                                           # fs_main.wgsl:89-92, pygfx pick_pack restated
        pick = np.zeros((H, W), np.uint64)
        if len(hit[0]):
            word = np.full(len(hit[0]), min(int(pick_id), 0xFFFFF), np.uint64)
            for k, shift in enumerate((20, 34, 48)):
                f = (hit_coord[k][hit] * f32(16383.0)).astype(np.float64)
                u = np.where(f > 0, np.minimum(np.floor(np.nan_to_num(f, nan=0.0)), 16383.0), 0.0).astype(np.uint64)
                word |= u << np.uint64(shift)
            pick[hit] = word
        out["pick"] = pick
    return out


def render_spec(spec, pick_id=None):
    from . import lmip

    vol = lmip.oracle_volume(spec)
    m = dict(lmip.DEFAULT_MATERIAL)
    m.update(spec.material)
    if m["colors"] is None:
        m["colors"] = lmip.DEFAULT_COLORS
    m["colors"] = [(*c, 1.0) for c in m["colors"]]
    return render(lmip.rings_of(vol), spec.matrices(), vol.volume_dimensions_shader, m, spec.width, spec.height,
                  colorspace_srgb=(spec.colorspace == "srgb"), pick_id=pick_id)

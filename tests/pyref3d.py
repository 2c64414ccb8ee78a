"""Second, independent restatement of the 3D step (oracle/sph_oracle3d.cpp's header is the statement: the 2D shader
shapes of compute.wgsl:8-299 / funcs.wgsl:72-218 with a third coordinate) in pure Python with np.float32 scalars —
every operation rounds to f32.  Only for tiny N: it cross-checks the C++ 3D oracle, which has no reference
counterpart to be pinned by.  The three kernel constants (poly6, spiky, viscosity: libm powf on the host) are inputs."""
import numpy as np

from pyref import EPS, bitonic, sign, u32sat, wmax

f = np.float32


def cell_xyz(u, pt):
    return tuple((u32sat(np.floor((pt[a] + u["size"][a] * f(0.5)) / u["h"])) + 1) & 0xFFFFFFFF for a in range(3))


def step3(parts, u):
    """parts: list of dicts(pos, pred, vel, density, grid) with 3-tuples of np.float32; in place."""
    n = len(parts)
    gw, gh, gd = u["grid"]
    h, dt = u["h"], u["dt"]
    bs = tuple(u["size"][a] * f(0.5) for a in range(3))
    for p in parts:                                            # predict + key
        pr = [p["pos"][a] + p["vel"][a] * dt for a in range(3)]
        for a in range(3):
            if abs(pr[a]) > bs[a]:
                pr[a] = bs[a] * sign(pr[a])
        p["pred"] = tuple(f(x) for x in pr)
        cx, cy, cz = cell_xyz(u, p["pred"])
        p["grid"] = ((cz * gh + cy) * gw + cx) & 0xFFFFFFFF
    bitonic(parts, lambda r: r["grid"], n)
    starts = {}
    for i in range(n):                                         # clean rebuild: first index of every cell
        if i == 0 or parts[i]["grid"] != parts[i - 1]["grid"]:
            starts[parts[i]["grid"]] = i

    def walk(cid):
        k = starts.get(cid)
        while k is not None and k < n and parts[k]["grid"] == cid:
            yield k
            k += 1

    def cells(c):
        for oz in (-1, 0, 1):
            for oy in (-1, 0, 1):
                for ox in (-1, 0, 1):
                    x, y, z = c[0] + ox, c[1] + oy, c[2] + oz
                    if 0 <= x < gw and 0 <= y < gh and 0 <= z < gd:
                        yield (z * gh + y) * gw + x

    h2 = h * h
    for p in parts:                                            # density
        me = p["pred"]
        rho = f(0.0)
        for cid in cells(cell_xyz(u, me)):
            for k in walk(cid):
                q = parts[k]["pred"]
                dx, dy, dz = q[0] - me[0], q[1] - me[1], q[2] - me[2]
                r2 = dx * dx + dy * dy + dz * dz
                kern = f(0.0)
                if not (r2 > h2):
                    d = h2 - r2
                    kern = u["poly6"] * d * d * d
                rho = rho + u["mass"] * kern * f(1.0)
        p["density"] = wmax(wmax(rho, EPS), f(0.1))
    snap = [dict(p) for p in parts]
    for i, q in enumerate(snap):                               # force + integrate (Jacobi snapshot)
        me = q["pred"]
        pressure = u["k"] * (q["density"] - u["rho0"])
        seed = (i * 12 + u["tick"] * 69) & 0xFFFFFFFF
        fp = [f(0.0)] * 3
        fv = [f(0.0)] * 3
        for cid in cells(cell_xyz(u, me)):
            for k in walk(cid):
                if k == i:
                    continue
                nb = snap[k]
                o = tuple(nb["pred"][a] - me[a] for a in range(3))
                r2 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2]
                if r2 > h2:
                    continue
                dst = np.sqrt(r2)
                if dst == 0:
                    r = []
                    for _ in range(3):
                        seed ^= (seed << 13) & 0xFFFFFFFF
                        seed ^= seed >> 17
                        seed ^= (seed << 5) & 0xFFFFFFFF
                        r.append(f(seed) / f(4294967296.0))
                    ln = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
                    d = tuple(r[a] / ln for a in range(3))
                else:
                    d = tuple(o[a] / dst for a in range(3))
                nrho = nb["density"]
                npress = u["k"] * (nrho - u["rho0"])
                kern = (-(h - dst)) * u["spiky"] if dst <= h else f(0.0)
                shared = (pressure + npress) * f(0.5)
                kv = f(0.0)
                if dst <= h:
                    kv = u["visc"] if dst == 0 else u["visc"] * ((-(dst * dst * dst) / (f(2.0) * h * h * h)) + ((dst * dst) / (h * h))
                                                                 + (h / (f(2.0) * dst)) - f(1.0))
                for a in range(3):
                    fp[a] = fp[a] + d[a] * kern * shared / nrho
                    fv[a] = fv[a] + (nb["vel"][a] - q["vel"][a]) / nrho * kv
        v = list(q["vel"])
        x = list(q["pos"])
        for a in range(3):
            acc = fp[a] + fv[a] * u["visc_coeff"]
            v[a] = v[a] + (acc / q["density"]) * dt
            v[a] = v[a] + u["gravity"][a] * dt
        if not (v[0] == v[0] and v[1] == v[1] and v[2] == v[2]):
            v = [f(0.0)] * 3
        sp = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        if sp > f(500.0):
            v = [(v[a] / sp) * f(500.0) for a in range(3)]
        for a in range(3):
            x[a] = x[a] + v[a] * dt
        for a in range(3):
            if abs(x[a]) > bs[a]:
                x[a] = bs[a] * sign(x[a])
                v[a] = v[a] * (f(-1.0) * u["damping"])
        parts[i] = dict(q, pos=tuple(f(t) for t in x), vel=tuple(f(t) for t in v))

"""Second, independent restatement of the reference step in pure Python with np.float32
scalars (every operation rounds to f32), written directly from the WGSL:
compute.wgsl:8-299, funcs.wgsl:72-218, sort.wgsl:27-51.  Only for tiny N — it exists to
cross-check the C++ oracle, not to be fast."""
import numpy as np

f = np.float32
PI = f(3.14159265359)
EPS = f(1.19209290e-07)


def set_float(dtype):
    """Switch the scalar type of the whole restatement (np.float32 = the reference arithmetic; np.float64 =
    the run the stated float tolerance is confirmed against).  Returns the previous type."""
    global f, PI, EPS
    prev = f
    f = dtype
    PI = f(3.14159265359)
    EPS = f(1.19209290e-07)
    return prev


def u32sat(x):
    x = float(x)
    if not (x > 0.0):
        return 0
    if x >= 4294967296.0:
        return 0xFFFFFFFF
    return int(x)


def wmax(a, b):
    """WGSL max() on floats: "If one operand is a NaN, the other is returned" (Python's max() would keep a NaN
    first operand)."""
    if a != a:
        return b
    if b != b:
        return a
    return b if a < b else a


def sign(x):
    return f(1.0) if x > 0 else (f(-1.0) if x < 0 else f(0.0))


def xy_of_point(u, pt):
    cx = u32sat(np.floor((pt[0] + u["bounds"][0] * f(0.5)) / u["h"])) + 1
    cy = u32sat(np.floor((pt[1] + u["bounds"][1] * f(0.5)) / u["h"])) + 1
    return cx & 0xFFFFFFFF, cy & 0xFFFFFFFF


def bitonic(rec, key, n):
    p2 = 1
    while p2 < n:
        p2 <<= 1
    num_pairs = p2 // 2
    threads = -(-num_pairs // 128) * 128
    stages = p2.bit_length() - 1
    for stage in range(stages):
        for step in range(stage + 1):
            gw = 1 << (stage - step)
            gh = 2 * gw - 1
            for i in range(threads):
                h = i & (gw - 1)
                lo = h + (gh + 1) * (i // gw)
                hi = lo + (gh - 2 * h if step == 0 else (gh + 1) // 2)
                if hi >= n:
                    continue
                if key(rec[lo]) > key(rec[hi]):
                    rec[lo], rec[hi] = rec[hi], rec[lo]


def step(parts, start_indices, u):
    """parts: list of dicts(pos, pred, vel, density, grid) with np.float32 pairs; in place."""
    n = len(parts)
    bs = (u["bounds"][0] * f(0.5), u["bounds"][1] * f(0.5))
    W = u["grid_w"]
    for p in parts:  # predict
        pr = [p["pos"][0] + p["vel"][0] * u["dt"], p["pos"][1] + p["vel"][1] * u["dt"]]
        for a in (0, 1):
            if abs(pr[a]) > bs[a]:
                pr[a] = bs[a] * sign(pr[a])
        p["pred"] = (f(pr[0]), f(pr[1]))
    for p in parts:  # key
        cx, cy = xy_of_point(u, p["pred"])
        p["grid"] = (cy * W + cx) & 0xFFFFFFFF
    bitonic(parts, lambda r: r["grid"], n)
    for i in range(1, n):  # cell starts (never cleared, i = 0 skipped)
        if parts[i]["grid"] != parts[i - 1]["grid"] and parts[i]["grid"] < len(start_indices):
            start_indices[parts[i]["grid"]] = i

    def walk(cid):
        if cid >= len(start_indices):
            return
        k = int(start_indices[cid])
        while k < n and parts[k]["grid"] == cid:
            yield k
            k += 1

    h = u["h"]
    h2 = h * h
    norm = f(4.0) / (PI * u["pow_h8"])
    for p in parts:  # density, 7x7 as written
        cx, cy = xy_of_point(u, p["pred"])
        rho = f(0.0)
        for oy in range(-3, 4):
            for ox in range(-3, 4):
                cid = (((cy + oy) & 0xFFFFFFFF) * W + ((cx + ox) & 0xFFFFFFFF)) & 0xFFFFFFFF
                for k in walk(cid):
                    q = parts[k]["pred"]
                    dx, dy = q[0] - p["pred"][0], q[1] - p["pred"][1]
                    r2 = dx * dx + dy * dy
                    kern = f(0.0) if r2 > h2 else norm * (h2 - r2) * (h2 - r2) * (h2 - r2)
                    rho = rho + u["mass"] * kern * f(1.0)
        p["density"] = wmax(wmax(rho, EPS), f(0.1))           # funcs.wgsl:202, compute.wgsl:70
    snap = [dict(p) for p in parts]
    for pid, p in enumerate(snap):  # move
        pos = p["pred"]
        pressure = u["k"] * (p["density"] - u["rho0"])
        seed = (pid * 12 + u["frame"] * 69) & 0xFFFFFFFF
        fp = [f(0.0), f(0.0)]
        fv = [f(0.0), f(0.0)]
        cx, cy = xy_of_point(u, pos)
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                cid = (((cy + oy) & 0xFFFFFFFF) * W + ((cx + ox) & 0xFFFFFFFF)) & 0xFFFFFFFF
                for k in walk(cid):
                    if k == pid:
                        continue
                    nb = snap[k]
                    o = (nb["pred"][0] - pos[0], nb["pred"][1] - pos[1])
                    r2 = o[0] * o[0] + o[1] * o[1]
                    if r2 > u["sqr_radius"]:
                        continue
                    dst = np.sqrt(r2)
                    if dst == 0:
                        rr = []
                        for _ in range(2):
                            seed ^= (seed << 13) & 0xFFFFFFFF
                            seed ^= seed >> 17
                            seed ^= (seed << 5) & 0xFFFFFFFF
                            rr.append(f(seed) / f(4294967296.0))
                        ln = np.sqrt(rr[0] * rr[0] + rr[1] * rr[1])
                        d = (rr[0] / ln, rr[1] / ln)
                    else:
                        d = (o[0] / dst, o[1] / dst)
                    npress = u["k"] * (nb["density"] - u["rho0"])
                    kern = (-(h - dst)) * u["spiky"] if dst <= h else f(0.0)
                    shared = (pressure + npress) * f(0.5)
                    fp[0] = fp[0] + d[0] * kern * shared / nb["density"]
                    fp[1] = fp[1] + d[1] * kern * shared / nb["density"]
                    if dst <= h:
                        if dst == 0:
                            kv = u["visc"]
                        else:
                            kv = u["visc"] * ((-(dst * dst * dst) / (f(2.0) * h * h * h)) + ((dst * dst) / (h * h))
                                              + (h / (f(2.0) * dst)) - f(1.0))
                    else:
                        kv = f(0.0)
                    fv[0] = fv[0] + (nb["vel"][0] - p["vel"][0]) / nb["density"] * kv
                    fv[1] = fv[1] + (nb["vel"][1] - p["vel"][1]) / nb["density"] * kv
        fv = [fv[0] * u["visc_coeff"], fv[1] * u["visc_coeff"]]
        v = [p["vel"][0], p["vel"][1]]
        x = [p["pos"][0], p["pos"][1]]
        for a in (0, 1):
            v[a] = v[a] + ((fp[a] + fv[a]) / p["density"]) * u["dt"]
            v[a] = v[a] + u["gravity"][a] * u["dt"]
        if u.get("mouse_state", 0) != 0:                      # compute.wgsl:99-108
            pp = p["pred"]
            diff = (u["mouse_pos"][0] - pp[0], u["mouse_pos"][1] - pp[1])
            dist = np.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
            if dist <= u["mouse_radius"]:
                d = ((diff[0] / dist) / dist, (diff[1] / dist) / dist)
                ratio = dist / u["mouse_radius"]
                ms = f(u["mouse_state"])
                v[0] = v[0] + d[0] * u["mouse_power"] * ms * ratio
                v[1] = v[1] + d[1] * u["mouse_power"] * ms * ratio
        if not (v[0] == v[0] and v[1] == v[1]):
            v = [f(0.0), f(0.0)]
        sp = np.sqrt(v[0] * v[0] + v[1] * v[1])
        if sp > f(500.0):
            v = [(v[0] / sp) * f(500.0), (v[1] / sp) * f(500.0)]
        for a in (0, 1):
            x[a] = x[a] + v[a] * u["dt"]
        tex = u.get("texture")                                # compute.wgsl:127-140 (force-field push-out)
        if tex is not None:
            pp = p["pred"]
            ts = u["texture_size"]                            # (w, h) as f32, like the uniform
            uv = ((pp[0] / u["bounds"][0]) * f(1.0) + f(0.5), (pp[1] / u["bounds"][1]) * f(1.0) + f(0.5))
            px, py = u32sat(uv[0] * ts[0]), u32sat(uv[1] * ts[1])
            ti = (py * u32sat(ts[0]) + px) & 0xFFFFFFFF
            force = (f(tex[ti][0]), f(tex[ti][1])) if ti < len(tex) else (f(0.0), f(0.0))   # OOB read -> zero
            if force[0] != 0 or force[1] != 0:
                p2w = ((u["bounds"][0] * f(2.0)) / ts[0], (u["bounds"][1] * f(2.0)) / ts[1])
                fw = (force[0] * p2w[0], force[1] * p2w[1])
                ln = np.sqrt(force[0] * force[0] + force[1] * force[1])
                nn = (force[0] / ln, force[1] / ln)
                x[0] = x[0] + fw[0]
                x[1] = x[1] + fw[1]
                vn = v[0] * nn[0] + v[1] * nn[1]
                v[0] = v[0] - (f(1.0) - u["damping"]) * vn * nn[0]
                v[1] = v[1] - (f(1.0) - u["damping"]) * vn * nn[1]
        for a in (0, 1):                                      # walls, compute.wgsl:143-153
            if abs(x[a]) > bs[a]:
                x[a] = bs[a] * sign(x[a])
                v[a] = v[a] * (f(-1.0) * u["damping"])
        parts[pid] = dict(p, pos=(f(x[0]), f(x[1])), vel=(f(v[0]), f(v[1])))

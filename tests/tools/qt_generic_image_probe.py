"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7): pins what QPainter::drawImage(QRectF, QImage) does for a source
image in Format_ARGB32 (NOT premultiplied) -- the format of the reference's generated assets (basic-abstract-game.cpp:102-107)
-- drawn without antialiasing onto an RGB32 frame.  qScaleFunctions / qTransformFunctions have no entry for that source
format, so QRasterPaintEngine::drawImage takes its generic route (qpaintengine_raster.cpp):
  coverage : QRasterizer::rasterizeLine(a, b, h / w) with a / b the mapped midpoints of the rect's left / right edge; for an
             axis-aligned line that is the pixel box [int(left + .5), int(right - .5)] x [int(top + .5), int(bottom - .5)] of
             values recomputed from midpoint and half extent in double arithmetic (clamped to the clip rect);
  sampling : QSpanData::setupMatrix -- inverse of (translate(1/65536) * translate(r.x, r.y) * scale(r.w / 64, r.h / 64)) --
             and fetchTransformed: fx = int((m11 * (x + .5) + dx) * 65536) at the span start, += int(m11 * 65536) per
             pixel, source coordinate (fx >> 16) clamped to the image;
  blend    : comp_func_SourceOver with const_alpha = (255 * intOpacity) >> 8.
Also covers the painter turned by a multiple of 90 degrees (BAG:902-906), where the line is still axis aligned.
"""
import os, sys, math
os.environ["QT_QPA_PLATFORM"] = "offscreen"
import numpy as np

W = H = 64
SW = SH = 64


def c_int(v):
    return int(v)  # C truncation


def model_axis_aligned(rx, ry, rw, rh, cw=W, ch=H, sw=SW, sh=SH):
    """returns (mask[H][W] bool, srcx[H][W], srcy[H][W]) for the untransformed painter"""
    mask = np.zeros((ch, cw), bool); sx_ = np.zeros((ch, cw), int); sy_ = np.zeros((ch, cw), int)
    if rw <= 0 or rh <= 0:  # drawImage: r.isEmpty()
        return mask, sx_, sy_
    # ---- QRasterPaintEngine::drawImage: a, b, width
    l, t = rx, ry
    r_, b_ = rx + rw, ry + rh
    ax, ay = (l + l) * 0.5, (t + b_) * 0.5
    bx, by = (r_ + r_) * 0.5, (t + b_) * 0.5
    width = rh / rw
    sp = rasterize_line(ax, ay, bx, by, width, cw, ch)
    if sp is None:
        return mask, sx_, sy_
    iLeft, iRight, iTop, iBottom = sp
    # ---- sampling matrix
    scx, scy = rw / sw, rh / sh
    d = 1.0 / 65536
    m11 = 1.0 * scx; m22 = 1.0 * scy
    m31 = d * scx + rx; m32 = d * scy + ry
    i11 = 1.0 / m11; i22 = 1.0 / m22
    idx = -m31 * i11; idy = -m32 * i22
    fdx = c_int(i11 * 65536.)
    for y in range(iTop, iBottom + 1):
        cy = y + 0.5
        fy = c_int((i22 * cy + 0.0 * (iLeft + 0.5) + idy) * 65536.)
        fx = c_int((0.0 * cy + i11 * (iLeft + 0.5) + idx) * 65536.)
        py = min(max(fy >> 16, 0), sh - 1)
        for x in range(iLeft, iRight + 1):
            px = min(max(fx >> 16, 0), sw - 1)
            mask[y, x] = True; sx_[y, x] = px; sy_[y, x] = py
            fx += fdx
    return mask, sx_, sy_


def rasterize_line(ax, ay, bx, by, width, cw, ch):
    """QRasterizer::rasterizeLine, non-antialiased, axis-aligned lines only -> (iLeft, iRight, iTop, iBottom) or None"""
    if (ax == bx and ay == by) or width == 0:
        return None
    pax, pay, pbx, pby = ax, ay, bx, by
    offx = abs(by - ay) * width * 0.5; offy = abs(bx - ax) * width * 0.5
    cl, ct, cr, cb = 0 - offx, 0 - offy, (cw - 1) + 1 + offx, (ch - 1) + 1 + offy   # QRectF(topLeft - offs, bottomRight + (1, 1) + offs)
    def contains(px, py):
        return cl <= px <= cr and ct <= py <= cb
    if not contains(pax, pay) or not contains(pbx, pby):
        t1 = 0.0; t2 = 1.0
        o = (pax, pay); dd = (pbx - pax, pby - pay); low = (cl, ct); high = (cr, cb)
        for i_ in range(2):
            if dd[i_] == 0:
                if o[i_] <= low[i_] or o[i_] >= high[i_]:
                    return None
                continue
            d_inv = 1 / dd[i_]
            t_low = (low[i_] - o[i_]) * d_inv
            t_high = (high[i_] - o[i_]) * d_inv
            if t_low > t_high:
                t_low, t_high = t_high, t_low
            if t1 < t_low: t1 = t_low
            if t2 > t_high: t2 = t_high
            if t1 >= t2:
                return None
        pax, pay, pbx, pby = pax + (pbx - pax) * t1, pay + (pby - pay) * t1, pax + (pbx - pax) * t2, pay + (pby - pay) * t2
    d0x, d0y = ax - bx, ay - by
    w0 = d0x * d0x + d0y * d0y
    dx_, dy_ = pax - pbx, pay - pby
    w = dx_ * dx_ + dy_ * dy_
    if w == 0:
        return None
    width *= math.sqrt(w0 / w)
    def q26eq(p, q):
        return c_int((p - q) * 64) == 0
    if q26eq(pay, pby):
        x = (pax + pbx) * 0.5
        dx = abs(pbx - pax) * 0.5
        y = pay
        dy = width * dx
        pax, pay = x, y - dy
        pbx, pby = x, y + dy
        width = 1 / width
    if not q26eq(pax, pbx):
        raise ValueError("not axis aligned")
    if pay > pby:
        pax, pay, pbx, pby = pbx, pby, pax, pay
    dy = pby - pay
    half = 0.5 * width * dy
    left = pax - half; right = pax + half
    left = min(max(left, 0.0), float(cw)); right = min(max(right, 0.0), float(cw))
    pay = min(max(pay, 0.0), float(ch)); pby = min(max(pby, 0.0), float(ch))
    if q26eq(left, right) or q26eq(pay, pby):
        return None
    iTop = c_int(pay + 0.5)
    iBottom = -1 if pby < 0.5 else c_int(pby - 0.5)
    iLeft = c_int(left + 0.5)
    iRight = -1 if right < 0.5 else c_int(right - 0.5)
    return iLeft, iRight, iTop, iBottom


SNAPX = SNAPY = lambda v: int(math.floor(v))  # the corners go to 26.6 by flooring (probe: truncation and qRound both miss)


def rasterize_line_general(ax, ay, bx, by, width, cw, ch):
    """QRasterizer::rasterizeLine, non-antialiased: list of (y, x1, x2) spans (x2 inclusive); handles every direction"""
    if (ax == bx and ay == by) or width == 0:
        return []
    pax, pay, pbx, pby = ax, ay, bx, by
    offx = abs(by - ay) * width * 0.5; offy = abs(bx - ax) * width * 0.5
    cl, ct, cr, cb = 0 - offx, 0 - offy, (cw - 1) + 1 + offx, (ch - 1) + 1 + offy
    def contains(px, py):
        return cl <= px <= cr and ct <= py <= cb
    if not contains(pax, pay) or not contains(pbx, pby):
        t1 = 0.0; t2 = 1.0
        o = (pax, pay); dd = (pbx - pax, pby - pay); low = (cl, ct); high = (cr, cb)
        for i_ in range(2):
            if dd[i_] == 0:
                if o[i_] <= low[i_] or o[i_] >= high[i_]:
                    return []
                continue
            d_inv = 1 / dd[i_]
            t_low = (low[i_] - o[i_]) * d_inv
            t_high = (high[i_] - o[i_]) * d_inv
            if t_low > t_high:
                t_low, t_high = t_high, t_low
            if t1 < t_low: t1 = t_low
            if t2 > t_high: t2 = t_high
            if t1 >= t2:
                return []
        pax, pay, pbx, pby = pax + (pbx - pax) * t1, pay + (pby - pay) * t1, pax + (pbx - pax) * t2, pay + (pby - pay) * t2
    d0x, d0y = ax - bx, ay - by
    w0 = d0x * d0x + d0y * d0y
    dx_, dy_ = pax - pbx, pay - pby
    w = dx_ * dx_ + dy_ * dy_
    if w == 0:
        return []
    width *= math.sqrt(w0 / w)
    def q26eq(p, q):
        return c_int((p - q) * 64) == 0
    if q26eq(pay, pby):
        if q26eq(pax, pbx):
            return []
        x = (pax + pbx) * 0.5
        dx = abs(pbx - pax) * 0.5
        y = pay
        dy = width * dx
        pax, pay = x, y - dy
        pbx, pby = x, y + dy
        width = 1 / width
    spans = []
    if q26eq(pax, pbx):
        if pay > pby:
            pax, pay, pbx, pby = pbx, pby, pax, pay
        dy = pby - pay
        half = 0.5 * width * dy
        left = pax - half; right = pax + half
        left = min(max(left, 0.0), float(cw)); right = min(max(right, 0.0), float(cw))
        pay = min(max(pay, 0.0), float(ch)); pby = min(max(pby, 0.0), float(ch))
        if q26eq(left, right) or q26eq(pay, pby):
            return []
        iTop = c_int(pay + 0.5)
        iBottom = -1 if pby < 0.5 else c_int(pby - 0.5)
        iLeft = c_int(left + 0.5)
        iRight = -1 if right < 0.5 else c_int(right - 0.5)
        for y in range(iTop, iBottom + 1):
            if iRight >= iLeft:
                spans.append((y, iLeft, iRight))
        return spans
    if pay > pby:
        pax, pay, pbx, pby = pbx, pby, pax, pay
    dlx = (pbx - pax) * (0.5 * width); dly = (pby - pay) * (0.5 * width)
    perpx, perpy = dly, -dlx
    if pax < pbx:
        top = (pax + perpx, pay + perpy); left = (pax - perpx, pay - perpy); right = (pbx + perpx, pby + perpy); bottom = (pbx - perpx, pby - perpy)
    else:
        top = (pax - perpx, pay - perpy); left = (pbx - perpx, pby - perpy); right = (pax + perpx, pay + perpy); bottom = (pbx + perpx, pby + perpy)
    # the four corners go to 26.6 by truncation and through the scan converter (pixel centres, 16.16 edge walkers)
    P = [(SNAPX(px * 64.), SNAPY(py * 64.)) for (px, py) in (top, right, bottom, left)]
    rows = {}
    for i in range(4):
        a_, b_ = P[i], P[(i + 1) % 4]
        if a_[1] > b_[1]:
            a_, b_ = b_, a_
        itop = max(0, (a_[1] + 32) >> 6); ibot = min(ch - 1, (b_[1] - 32) >> 6)
        if itop > ibot:
            continue
        xfp = 32768 + (a_[0] << 10); slope = 0
        if b_[0] != a_[0]:
            slope = c_int((b_[0] - a_[0]) / float(b_[1] - a_[1]) * 65536.)
            xfp += (slope * ((itop << 16) + 32768 - (a_[1] << 10))) >> 16
        for y in range(itop, ibot + 1):
            rows.setdefault(y, []).append(xfp >> 16); xfp += slope
    for y in sorted(rows):
        xs = sorted(rows[y])
        if len(xs) >= 2:
            x1 = max(xs[0], 0); x2 = min(xs[-1], cw) - 1
            if x2 >= x1:
                spans.append((y, x1, x2))
    return spans


def q_fuzzy_is_null(v):
    return abs(v) <= 0.000000000001


def model_rotated(cx, cy, w, h, deg, cw=W, ch=H, sw=SW, sh=SH):
    """p.translate(cx, cy); p.rotate(deg); p.drawImage(QRectF(-w/2, -h/2, w, h), img) (BAG:902-906)"""
    mask = np.zeros((ch, cw), bool); sx_ = np.zeros((ch, cw), int); sy_ = np.zeros((ch, cw), int)
    rx, ry, rw, rh = -w / 2, -h / 2, w, h
    if rw <= 0 or rh <= 0:
        return mask, sx_, sy_
    a = deg
    sina = cosa = 0.0
    if a == 0: cosa = 1.0
    elif a == 90. or a == -270.: sina = 1.
    elif a == 270. or a == -90.: sina = -1.
    elif a == 180.: cosa = -1.
    else:
        b = 0.017453292519943295769 * a
        sina = math.sin(b); cosa = math.cos(b)
    m11, m12, m21, m22, mdx, mdy = cosa, sina, -sina, cosa, cx, cy
    if not q_fuzzy_is_null(m12) or not q_fuzzy_is_null(m21): typ = 'rotate'
    elif not q_fuzzy_is_null(m11 - 1) or not q_fuzzy_is_null(m22 - 1): typ = 'scale'
    elif not q_fuzzy_is_null(mdx) or not q_fuzzy_is_null(mdy): typ = 'translate'
    else: typ = 'none'
    def mp(x, y):
        if typ == 'none': return x, y
        if typ == 'translate': return x + mdx, y + mdy
        if typ == 'scale': return m11 * x + mdx, m22 * y + mdy
        return m11 * x + m21 * y + mdx, m12 * x + m22 * y + mdy
    # sampling matrix: copy = matrix; copy.translate(r.x, r.y); copy.scale(r.w / sw, r.h / sh)
    c11, c12, c21, c22, cdx, cdy = m11, m12, m21, m22, mdx, mdy
    if typ == 'none': cdx, cdy = rx, ry; ctyp = 'translate'
    elif typ == 'translate': cdx += rx; cdy += ry; ctyp = 'translate'
    elif typ == 'scale': cdx += rx * c11; cdy += ry * c22; ctyp = 'scale'
    else: cdx += rx * c11 + ry * c21; cdy += ry * c22 + rx * c12; ctyp = 'rotate'
    scx, scy = rw / sw, rh / sh
    if ctyp == 'rotate':
        c12 *= scx; c21 *= scy
    c11 *= scx; c22 *= scy
    if ctyp in ('none', 'translate'): ctyp = 'scale'
    d = 1.0 / 65536
    if ctyp == 'scale':
        p11 = 1.0 * c11; p22 = 1.0 * c22; p31 = d * c11 + cdx; p32 = d * c22 + cdy
        i11 = 1. / p11; i22 = 1. / p22; i12 = i21 = 0.0
        idx = -p31 * i11; idy = -p32 * i22
    else:
        p11 = 1.0 * c11 + 0.0 * c21; p12 = 1.0 * c12 + 0.0 * c22
        p21 = 0.0 * c11 + 1.0 * c21; p22 = 0.0 * c12 + 1.0 * c22
        p31 = d * c11 + d * c21 + cdx; p32 = d * c12 + d * c22 + cdy
        dtr = p11 * p22 - p12 * p21
        dinv = 1.0 / dtr
        i11 = p22 * dinv; i12 = -p12 * dinv; i21 = -p21 * dinv; i22 = p11 * dinv
        idx = (p21 * p32 - p22 * p31) * dinv; idy = (p12 * p31 - p11 * p32) * dinv
    fdx = c_int(i11 * 65536.); fdy = c_int(i12 * 65536.)
    # coverage
    if typ == 'scale':
        x = m11 * rx + mdx; y = m22 * ry + mdy; ww = m11 * rw; hh = m22 * rh
        if ww < 0: ww = -ww; x -= ww
        if hh < 0: hh = -hh; y -= hh
        x1 = qround(x); y1 = qround(y); x2 = qround(x + ww); y2 = qround(y + hh)
        spans = [(yy, max(x1, 0), min(x2, cw) - 1) for yy in range(max(y1, 0), min(y2, ch))]
    else:
        l, t, r_, b_ = rx, ry, rx + rw, ry + rh
        ax, ay = mp((l + l) * 0.5, (t + b_) * 0.5)
        bx, by = mp((r_ + r_) * 0.5, (t + b_) * 0.5)
        spans = rasterize_line_general(ax, ay, bx, by, rh / rw, cw, ch)
    for (y, x1, x2) in spans:
        if x2 < x1: continue
        ccx = x1 + 0.5; ccy = y + 0.5
        fx = c_int((i21 * ccy + i11 * ccx + idx) * 65536.)
        fy = c_int((i22 * ccy + i12 * ccx + idy) * 65536.)
        for x in range(x1, x2 + 1):
            px = min(max(fx >> 16, 0), sw - 1); py = min(max(fy >> 16, 0), sh - 1)
            mask[y, x] = True; sx_[y, x] = px; sy_[y, x] = py
            fx += fdx; fy += fdy
    return mask, sx_, sy_


def qround(d):
    return int(d + 0.5) if d >= 0.0 else int(d - float(int(d - 1)) + 0.5) + int(d - 1)


def main():
    from PyQt5.QtGui import QImage, QPainter, QGuiApplication, QColor
    from PyQt5.QtCore import QRectF
    app = QGuiApplication(sys.argv[:1])
    # source: pixel (x, y) has colour (x * 4, y * 4, 7), opaque
    src = np.zeros((SH, SW), np.uint32)
    for y in range(SH):
        for x in range(SW):
            src[y, x] = 0xff000000 | ((x * 4) << 16) | ((y * 4) << 8) | 7
    simg = QImage(src.tobytes(), SW, SH, SW * 4, QImage.Format_ARGB32)
    simg._keep = src

    def qt_draw(rx, ry, rw, rh):
        img = QImage(W, H, QImage.Format_RGB32); img.fill(QColor(0, 0, 0))
        p = QPainter(img); p.drawImage(QRectF(rx, ry, rw, rh), simg); p.end()
        ptr = img.constBits(); ptr.setsize(W * H * 4)
        a = np.frombuffer(bytes(ptr), np.uint32).reshape(H, W) & 0xffffff
        return (a & 0xff) == 7, (a >> 16) // 4, ((a >> 8) & 0xff) // 4

    rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    bad = 0; shown = 0
    for i in range(n):
        k = i % 5
        if k == 0:
            rw = rng.uniform(.3, 30); rh = rng.uniform(.3, 30); rx = rng.uniform(-10, 70); ry = rng.uniform(-10, 70)
        elif k == 1:
            rw = float(np.float32(rng.uniform(2, 12))); rh = float(np.float32(rng.uniform(2, 12))); rx = float(np.float32(rng.uniform(0, 55))); ry = float(np.float32(rng.uniform(0, 55)))
        elif k == 2:
            rw = rng.randint(1, 1500) / 64.; rh = rng.randint(1, 1500) / 64.; rx = rng.randint(-200, 4000) / 64.; ry = rng.randint(-200, 4000) / 64.
        elif k == 3:
            rw = rng.uniform(40, 200); rh = rng.uniform(40, 200); rx = rng.uniform(-150, 30); ry = rng.uniform(-150, 30)
        else:
            rw = rng.randint(1, 40) + rng.choice([0, .5]); rh = rng.randint(1, 40) + rng.choice([0, .5]); rx = rng.randint(-5, 60) + rng.choice([0, .5]); ry = rng.randint(-5, 60) + rng.choice([0, .5])
        qm, qx, qy = qt_draw(rx, ry, rw, rh)
        mm, mx, my = model_axis_aligned(rx, ry, rw, rh)
        ok = np.array_equal(qm, mm) and np.array_equal(qx[qm], mx[qm]) and np.array_equal(qy[qm], my[qm])
        if not ok:
            bad += 1
            if shown < 6:
                shown += 1
                print("MISMATCH", repr((rx, ry, rw, rh)), "coverage equal:", np.array_equal(qm, mm))
                if not np.array_equal(qm, mm):
                    ys, xs = np.nonzero(qm); print("  qt box", (xs.min(), xs.max(), ys.min(), ys.max()) if len(xs) else None)
                    ys, xs = np.nonzero(mm); print("  model box", (xs.min(), xs.max(), ys.min(), ys.max()) if len(xs) else None)
                else:
                    dx = np.argwhere((qx != mx) & qm); dy = np.argwhere((qy != my) & qm)
                    print("  srcx diffs", len(dx), [(tuple(p), qx[tuple(p)], mx[tuple(p)]) for p in dx[:3]], "srcy diffs", len(dy), [(tuple(p), qy[tuple(p)], my[tuple(p)]) for p in dy[:3]])
    print("unrotated ARGB32 drawImage mismatches: %d of %d" % (bad, n))

    def qt_draw_rot(cx, cy, w, h, deg):
        img = QImage(W, H, QImage.Format_RGB32); img.fill(QColor(0, 0, 0))
        p = QPainter(img); p.translate(cx, cy); p.rotate(deg); p.drawImage(QRectF(-w / 2, -h / 2, w, h), simg); p.end()
        ptr = img.constBits(); ptr.setsize(W * H * 4)
        a = np.frombuffer(bytes(ptr), np.uint32).reshape(H, W) & 0xffffff
        return (a & 0xff) == 7, (a >> 16) // 4, ((a >> 8) & 0xff) // 4
    bad = 0; shown = 0
    for i in range(n):
        k = i % 4
        w = rng.uniform(1, 30); h = rng.uniform(1, 30); cx = rng.uniform(-5, 69); cy = rng.uniform(-5, 69)
        if k == 0: deg = rng.uniform(-720, 720)
        elif k == 1: deg = float(np.float32(rng.uniform(-6.3, 6.3)) * np.float32(180) / np.float32(3.14159265))  # rotation * 180 / PI in float
        elif k == 2: deg = float(rng.choice([90, -90, 180, 270, -270, 360, -180, 450]))
        else: deg = rng.choice([90, 180, 0.5, 45, 135, 89.999, 179.99999]) + rng.choice([0, 1e-9, -1e-7, 1e-13])
        qm, qx, qy = qt_draw_rot(cx, cy, w, h, deg)
        try:
            mm, mx, my = model_rotated(cx, cy, w, h, deg)
        except Exception as e:
            mm = np.zeros_like(qm); mx = my = np.zeros((H, W), int); print("model error", e)
        ok = np.array_equal(qm, mm) and np.array_equal(qx[qm], mx[qm]) and np.array_equal(qy[qm], my[qm])
        if not ok:
            bad += 1
            if shown < 6:
                shown += 1
                print("MISMATCH rot", repr((cx, cy, w, h, deg)), "coverage equal:", np.array_equal(qm, mm), "qt px", int(qm.sum()), "model px", int(mm.sum()))
                if not np.array_equal(qm, mm):
                    dd = np.argwhere(qm != mm); print("   coverage diffs", len(dd), [tuple(p) for p in dd[:6]])
                else:
                    dx = np.argwhere((qx != mx) & qm); dy = np.argwhere((qy != my) & qm)
                    print("   srcx diffs", len(dx), [(tuple(p), qx[tuple(p)], mx[tuple(p)]) for p in dx[:3]], "srcy diffs", len(dy), [(tuple(p), qy[tuple(p)], my[tuple(p)]) for p in dy[:3]])
    print("rotated ARGB32 drawImage mismatches: %d of %d" % (bad, n))


if __name__ == "__main__":
    main()

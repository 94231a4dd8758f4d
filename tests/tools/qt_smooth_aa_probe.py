"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7): pins what QPainter does with
    setRenderHint(Antialiasing); setRenderHint(SmoothPixmapTransform)
(reference src/game.cpp:84-87, the render_human 512 x 512 frame) for the calls game_draw makes on an RGB32 frame:
    drawImage(QRectF, QImage)  with ARGB32_Premultiplied / RGB32 sources, opacity, untransformed or rotated painter
    fillRect(QRectF, QColor)
Qt 5.9 route (qpaintengine_raster.cpp, qrasterizer.cpp, qdrawhelper.cpp), restated in model_*():
  coverage : QRasterizer::rasterizeLine(a, b, h / w), antialiased: 16.16 fixed point.  Axis aligned: per row
             rowHeight = min(y + 1, bottom) - max(y, top), per column the left / right partial widths * 255, coverage =
             (rowHeight * colCoverage) >> 16 >> 16.  (A rect inside one pixel column gets (leftWidth + rightWidth) * 255,
             Qt's own quirk.)  Rotated: the trapezoid walker with intersectPixelFP.
  sampling : QSpanData::setupMatrix(bilinear) and fetchTransformedBilinearARGB32PM: fx = int((m11 * (x + .5) + dx) * 65536)
             - 32768 at each span's first pixel, += int(m11 * 65536) per pixel; scale-up on x (0 < fdx <= 65536): rows
             blended with 8-bit disty first, then columns with 8-bit distx; otherwise 4-bit distances
             (interpolate_4_pixels_16) unless the zoom exceeds 8x (8-bit interpolate_4_pixels); source coordinates clamped.
  blend    : comp_func_SourceOver with const_alpha = (coverage * intOpacity) >> 8.
usage: qt_smooth_aa_probe.py [n_cases] [seed]
"""
import os, sys, math
os.environ["QT_QPA_PLATFORM"] = "offscreen"
import numpy as np

CW = CH = 96


def c_int(v):
    return int(v)


def F16(v):  # FloatToQ16Dot16
    return c_int(v * 65536.0)


def mul16(a, b):  # Q16Dot16Multiply
    return (a * b) >> 16


def byte_mul(x, a):
    t = (x & 0xff00ff) * a
    t = (t + ((t >> 8) & 0xff00ff) + 0x800080) >> 8
    t &= 0xff00ff
    x = ((x >> 8) & 0xff00ff) * a
    x = (x + ((x >> 8) & 0xff00ff) + 0x800080)
    x &= 0xff00ff00
    return (x | t) & 0xffffffff


def source_over(d, s, ca):
    if ca != 255:
        s = byte_mul(s, ca)
    return (s + byte_mul(d, 255 - (s >> 24))) & 0xffffffff


def clip_line(ax, ay, bx, by, width, cw, ch):
    """common head of QRasterizer::rasterizeLine: returns (pax, pay, pbx, pby, width) or None"""
    if (ax == bx and ay == by) or width == 0:
        return None
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
    return pax, pay, pbx, pby, width


def q26eq(p, q):
    return c_int((p - q) * 64) == 0


def aa_line_spans(ax, ay, bx, by, width, cw=CW, ch=CH):
    """QRasterizer::rasterizeLine, antialiased -> list of (y, x, len, coverage 0..255)"""
    c = clip_line(ax, ay, bx, by, width, cw, ch)
    if c is None:
        return []
    pax, pay, pbx, pby, width = c
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
    def add(x, ln, y, cov):
        if cov and ln and 0 <= y < ch:
            spans.append((y, x, ln, cov))
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
        iLeft = c_int(left); iRight = c_int(right)
        leftWidth = ((iLeft + 1) << 16) - F16(left)
        rightWidth = F16(right) - (iRight << 16)
        cov = []; xs = []; ln = []
        if iLeft == iRight:
            cov.append((leftWidth + rightWidth) * 255); xs.append(iLeft); ln.append(1)
        else:
            cov.append(leftWidth * 255); xs.append(iLeft); ln.append(1)
            if leftWidth == 65536:
                ln[0] = iRight - iLeft
            elif iRight - iLeft > 1:
                cov.append(255 << 16); xs.append(iLeft + 1); ln.append(iRight - iLeft - 1)
            if rightWidth:
                cov.append(rightWidth * 255); xs.append(iRight); ln.append(1)
        iTopFP = c_int(pay) << 16; iBottomFP = c_int(pby) << 16
        yPa = F16(pay); yPb = F16(pby)
        yFP = iTopFP
        while yFP <= iBottomFP:
            rowHeight = min(yFP + 65536, yPb) - max(yFP, yPa)
            y = yFP >> 16
            if y > ch - 1:
                break
            for i in range(len(cov)):
                add(xs[i], ln[i], y, mul16(rowHeight, cov[i]) >> 16)
            yFP += 65536
        return spans
    raise ValueError("general lines: see aa_line_spans_general")


def setup_matrix(rx, ry, rw, rh, sw, sh):
    """QSpanData::setupMatrix for the untransformed painter: inverse of translate(1/65536) * translate(r) * scale(r / s)"""
    scx, scy = rw / sw, rh / sh
    d = 1.0 / 65536
    m11 = 1.0 * scx; m22 = 1.0 * scy
    m31 = d * scx + rx; m32 = d * scy + ry
    i11 = 1.0 / m11; i22 = 1.0 / m22
    return i11, i22, -m31 * i11, -m32 * i22


def lerp256(a, b, d):  # per channel (a * (256 - d) + b * d) >> 8
    out = 0
    for sh in (0, 8, 16, 24):
        ca = (a >> sh) & 255; cb = (b >> sh) & 255
        out |= (((ca * (256 - d) + cb * d) >> 8) & 255) << sh
    return out


def interp16(tl, tr, bl, br, dx, dy):  # interpolate_4_pixels_16
    dxy = dx * dy
    w = (256 - 16 * dx - 16 * dy + dxy, 16 * dx - dxy, 16 * dy - dxy, dxy)
    out = 0
    for sh in (0, 8, 16, 24):
        c = sum(((p >> sh) & 255) * wi for p, wi in zip((tl, tr, bl, br), w))
        out |= ((c >> 8) & 255) << sh
    return out


def interp8(tl, tr, bl, br, dx, dy):  # interpolate_4_pixels (8-bit distances): rows first, then columns
    return lerp256(lerp256(tl, bl, dy), lerp256(tr, br, dy), dx)


def fetch_bilinear_scale(src, y, x0, length, i11, i22, idx, idy):
    """fetchTransformedBilinearARGB32PM<BlendTransformedBilinear>, fdy == 0, one call per span (x0, length)"""
    sh, sw = src.shape
    fdx = c_int(i11 * 65536.)
    cx = x0 + 0.5; cy = y + 0.5
    fx = c_int((0.0 * cy + i11 * cx + idx) * 65536.) - 32768
    fy = c_int((i22 * cy + 0.0 * cx + idy) * 65536.) - 32768
    y1 = fy >> 16
    if y1 < 0: y1 = y2 = 0
    elif y1 >= sh - 1: y1 = y2 = sh - 1
    else: y2 = y1 + 1
    out = []
    def bx(x1):
        if x1 < 0: return 0, 0
        if x1 >= sw - 1: return sw - 1, sw - 1
        return x1, x1 + 1
    def px4(fx):
        x1, x2 = bx(fx >> 16)
        return int(src[y1, x1]), int(src[y1, x2]), int(src[y2, x1]), int(src[y2, x2]), x1, x2
    if 0 < fdx <= 65536:  # scale up on x: rows blended first (8-bit disty), then columns (8-bit distx)
        for i in range(length):
            tl, tr, bl, br, _, _ = px4(fx)
            out.append(interp8(tl, tr, bl, br, (fx & 0xffff) >> 8, (fy & 0xffff) >> 8))
            fx += fdx
    elif (fdx < 0 and fdx > -(65536 // 8)) or abs(i22) < 1. / 8.:  # scale up more than 8x
        for i in range(length):
            tl, tr, bl, br, _, _ = px4(fx)
            out.append(interp8(tl, tr, bl, br, (fx & 0xffff) >> 8, (fy & 0xffff) >> 8))
            fx += fdx
    else:  # scale down: scalar prolog while clamped, SSE2 groups of four with rounded 4-bit distances, scalar tail
        b = 0
        while b < length:
            tl, tr, bl, br, x1, x2 = px4(fx)
            if x1 != x2:
                break
            out.append(interp8(tl, tr, bl, br, (fx & 0xffff) >> 8, (fy & 0xffff) >> 8)); fx += fdx; b += 1
        bounded = length
        if fdx > 0:
            bounded = min(bounded, b + c_int(((sw - 1) * 65536 - fx) / fdx))
        elif fdx < 0:
            bounded = min(bounded, b + c_int((0 - fx) / fdx))
        bounded -= 3
        dy4 = ((fy & 0xffff) + 0x800) >> 12
        while b < bounded:
            for k in range(4):
                x1 = fx >> 16
                out.append(interp16(int(src[y1, x1]), int(src[y1, x1 + 1]), int(src[y2, x1]), int(src[y2, x1 + 1]), ((fx & 0xffff) + 0x800) >> 12, dy4))
                fx += fdx
            b += 4
        while b < length:
            tl, tr, bl, br, _, _ = px4(fx)
            out.append(interp8(tl, tr, bl, br, (fx & 0xffff) >> 8, (fy & 0xffff) >> 8)); fx += fdx; b += 1
    return out


def interpolate_pixel_255(x, a, y, b):
    t = (x & 0xff00ff) * a + (y & 0xff00ff) * b
    t = (t + ((t >> 8) & 0xff00ff) + 0x800080) >> 8
    t &= 0xff00ff
    x = ((x >> 8) & 0xff00ff) * a + ((y >> 8) & 0xff00ff) * b
    x = (x + ((x >> 8) & 0xff00ff) + 0x800080)
    x &= 0xff00ff00
    return (x | t) & 0xffffffff


def blend_runs(dst, spans, fetch, io, opaque_source):
    """blend_src_generic / handleSpans: the spans of one row that touch are fetched as ONE run (which matters: the fetch
    treats the head, the groups of four and the tail of a run differently), then blended span by span with
    const_alpha = (coverage * intOpacity) >> 8.  A source without alpha channel turns SourceOver into Source
    (getOperator): d = INTERPOLATE_PIXEL_255(s, ca, d, 255 - ca)."""
    i = 0
    while i < len(spans):
        y, x, ln, cov = spans[i]
        j = i + 1; right = x + ln
        while j < len(spans) and spans[j][0] == y and spans[j][1] == right:
            right += spans[j][2]; j += 1
        px = fetch(y, x, right - x)
        for k in range(i, j):
            _, sx, sl, sc = spans[k]
            ca = (sc * io) >> 8
            for t in range(sl):
                s_ = px[sx - x + t]; d_ = int(dst[y, sx + t])
                if opaque_source:
                    dst[y, sx + t] = s_ if ca == 255 else interpolate_pixel_255(s_, ca, d_, 255 - ca)
                else:
                    dst[y, sx + t] = source_over(d_, s_, ca)
        i = j


def model_draw_image(dst, src, rx, ry, rw, rh, opacity=1.0, opaque_source=False):
    ch, cw = dst.shape
    sh, sw = src.shape
    if rw <= 0 or rh <= 0:
        return
    io = c_int(min(max(opacity, 0.0), 1.0) * 256)
    if rw == sw and rh == sh:
        # not stretched (QRasterPaintEngine::drawImage, translate only): fillRect_normalized over the ROUNDED rect with the
        # untransformed image filler -- no antialiasing, no filtering: source pixel (x - qRound(r.x), y - qRound(r.y))
        qr = lambda v: int(v + 0.5) if v >= 0 else int(v - float(int(v - 1)) + 0.5) + int(v - 1)
        x1, y1, x2, y2 = qr(rx), qr(ry), qr(rx + rw), qr(ry + rh)
        ca = (255 * io) >> 8
        for y in range(max(y1, 0), min(y2, ch)):
            for x in range(max(x1, 0), min(x2, cw)):
                sx, sy = x - x1, y - y1
                if 0 <= sx < sw and 0 <= sy < sh:
                    s_ = int(src[sy, sx]); d_ = int(dst[y, x])
                    if opaque_source and io == 256:
                        dst[y, x] = s_
                    else:
                        dst[y, x] = source_over(d_, s_, ca)
        return
    l, t, r_, b_ = rx, ry, rx + rw, ry + rh
    ax, ay = (l + l) * 0.5, (t + b_) * 0.5
    bx, by = (r_ + r_) * 0.5, (t + b_) * 0.5
    spans = aa_line_spans(ax, ay, bx, by, rh / rw, cw, ch)
    i11, i22, idx, idy = setup_matrix(rx, ry, rw, rh, sw, sh)
    # QSpanData::initTexture: hasAlpha = image.hasAlphaChannel() || intOpacity != 256
    blend_runs(dst, spans, lambda y, x, n: fetch_bilinear_scale(src, y, x, n, i11, i22, idx, idy), io, opaque_source and io == 256)


def model_fill_rect(dst, rx, ry, rw, rh, color):
    ch, cw = dst.shape
    l, t, r_, b_ = rx, ry, rx + rw, ry + rh
    ax, ay = (l + l) * 0.5, (t + b_) * 0.5
    bx, by = (r_ + r_) * 0.5, (t + b_) * 0.5
    for (y, x, ln, cov) in aa_line_spans(ax, ay, bx, by, rh / rw, cw, ch):
        for i in range(ln):
            dst[y, x + i] = source_over(int(dst[y, x + i]), color, cov)


# ---------------------------------------------------------------------------------------------------------------------
def qt_setup():
    from PyQt5.QtGui import QImage, QPainter, QColor, QGuiApplication
    from PyQt5.QtCore import QRectF
    global _app
    _app = QGuiApplication.instance() or QGuiApplication([])
    return QImage, QPainter, QColor, QRectF


def np_to_qimage(a, fmt):
    from PyQt5.QtGui import QImage
    a = np.ascontiguousarray(a.astype(np.uint32))
    raw = a.tobytes()
    img = QImage(raw, a.shape[1], a.shape[0], a.shape[1] * 4, fmt)
    img._keep = raw
    return img


def qt_draw(dst0, ops):
    QImage, QPainter, QColor, QRectF = qt_setup()
    h, w = dst0.shape
    img = np_to_qimage(dst0, QImage.Format_RGB32).copy()
    p = QPainter(img)
    p.setRenderHint(QPainter.Antialiasing, True)
    p.setRenderHint(QPainter.SmoothPixmapTransform, True)
    for op in ops:
        if op[0] == "image":
            _, src, fmt, r, opacity = op
            q = np_to_qimage(src, fmt)
            if opacity != 1:
                p.save(); p.setOpacity(opacity)
            p.drawImage(QRectF(*r), q)
            if opacity != 1:
                p.restore()
        elif op[0] == "fill":
            _, r, c = op
            p.fillRect(QRectF(*r), QColor((c >> 16) & 255, (c >> 8) & 255, c & 255, c >> 24))
    p.end()
    ptr = img.constBits(); ptr.setsize(w * h * 4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(h, w).copy()


def rand_src(rng, sw, sh, premul_alpha):
    a = rng.randint(0, 256, size=(sh, sw, 4)).astype(np.uint32)
    if premul_alpha:
        al = a[..., 3]
        # blocks of transparent / opaque / partial
        m = rng.randint(0, 3, size=(sh, sw))
        al = np.where(m == 0, 0, np.where(m == 1, 255, al))
        a[..., 3] = al
        for k in range(3):
            a[..., k] = a[..., k] * al // 255
    else:
        a[..., 3] = 255
    return (a[..., 3] << 24) | (a[..., 2] << 16) | (a[..., 1] << 8) | a[..., 0]


def main():
    from PyQt5.QtGui import QImage
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    miss = 0; worst = 0
    for case in range(n):
        dst0 = rand_src(rng, CW, CH, False)
        kind = case % 4
        sw, sh = [(8, 8), (40, 24), (128, 128), (300, 200)][rng.randint(0, 4)]
        premul = kind != 3
        src = rand_src(rng, sw, sh, premul)
        scale = [0.3, 0.9, 1.0, 2.5, 9.0][rng.randint(0, 5)]
        rw = sw * scale * rng.uniform(0.8, 1.2); rh = sh * scale * rng.uniform(0.8, 1.2)
        if rw > 3 * CW: rh *= 3 * CW / rw; rw = 3 * CW
        if rh > 3 * CH: rw *= 3 * CH / rh; rh = 3 * CH
        rx = rng.uniform(-rw * 0.5, CW - rw * 0.5); ry = rng.uniform(-rh * 0.5, CH - rh * 0.5)
        if rng.randint(0, 4) == 0:
            rx = float(int(rx)); ry = float(int(ry))
        if case % 7 == 3:  # drawn 1 : 1 -> Qt's untransformed route
            rw, rh = float(sw), float(sh)
        if kind == 2:
            rx, ry, rw, rh = float(np.float32(rx)), float(np.float32(ry)), float(np.float32(rw)), float(np.float32(rh))
        opacity = 1.0 if rng.randint(0, 3) else float(np.float32(rng.uniform(0, 1)))
        if kind == 1:
            color = int(rng.randint(0, 1 << 24)) | 0xff000000
            got = qt_draw(dst0, [("fill", (rx, ry, rw, rh), color)])
            want = dst0.copy().astype(np.uint32); model_fill_rect(want, rx, ry, rw, rh, color)
        else:
            fmt = QImage.Format_ARGB32_Premultiplied if premul else QImage.Format_RGB32
            got = qt_draw(dst0, [("image", src, fmt, (rx, ry, rw, rh), opacity)])
            want = dst0.copy().astype(np.uint32); model_draw_image(want, src, rx, ry, rw, rh, opacity, not premul)
        g = got.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int); w = want.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int)
        d = np.abs(g - w)
        if d.max() > 0:
            miss += 1; worst = max(worst, int(d.max()))
            ys, xs = np.nonzero(d.max(axis=2))
            print(f"case {case} kind {kind} src {sw}x{sh} rect ({rx:.4f},{ry:.4f},{rw:.4f},{rh:.4f}) op {opacity:.3f}: {len(ys)} px differ, max {d.max()}, "
                  f"first at (x={xs[0]}, y={ys[0]}) got {g[ys[0], xs[0]]} want {w[ys[0], xs[0]]}; rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}")
    print(f"{n} cases, {miss} with differences, worst channel difference {worst}")


if __name__ == "__main__":
    main()

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
usage: qt_smooth_aa_probe.py [n_cases] [seed] [untransformed|fills|images|all|lines|generic]
Last runs (seeds 1, 2, 3; 100-400 cases per mode): 0 misses.
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


def clip_line(ax, ay, bx, by, width, cw, ch, square_cap=False):
    """common head of QRasterizer::rasterizeLine: returns (pax, pay, pbx, pby, width) or None"""
    if (ax == bx and ay == by) or width == 0:
        return None
    pax, pay, pbx, pby = ax, ay, bx, by
    if square_cap:  # the line grows by half its width at either end
        dx_, dy_ = pbx - pax, pby - pay
        pax, pay = pax - (0.5 * width) * dx_, pay - (0.5 * width) * dy_
        pbx, pby = pbx + (0.5 * width) * dx_, pby + (0.5 * width) * dy_
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


def aa_line_spans(ax, ay, bx, by, width, cw=CW, ch=CH, square_cap=False):
    """QRasterizer::rasterizeLine, antialiased -> list of (y, x, len, coverage 0..255)"""
    c = clip_line(ax, ay, bx, by, width, cw, ch, square_cap)
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
            spans.append((y, x, ln, cov & 0xff))  # QT_FT_Span::coverage is an unsigned char
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


def fast_matrix(m11, m12, m21, m22, dx, dy):
    """QSpanData::setupMatrix: the 16.16 fixed-point fetch is only used while the inverse matrix is small"""
    return m11 * m11 + m21 * m21 < 1e4 and m12 * m12 + m22 * m22 < 1e4 and abs(dx) < 1e4 and abs(dy) < 1e4


def fetch_bilinear_slow(src, y, x0, length, T):
    """fetchTransformedBilinearARGB32PM without fast_matrix (a strongly scaled-down sprite far from the origin has an inverse
    translation beyond 1e4): floating-point source coordinates per pixel, 8-bit distances, interpolate_4_pixels"""
    m11, m12, m21, m22, dx, dy = T
    sh, sw = src.shape
    cx = x0 + 0.5; cy = y + 0.5
    fx = m21 * cy + m11 * cx + dx
    fy = m22 * cy + m12 * cx + dy
    out = []
    def bnd(v, n):
        if v < 0: return 0, 0
        if v >= n - 1: return n - 1, n - 1
        return v, v + 1
    for i in range(length):
        px = fx - 0.5; py = fy - 0.5
        x1 = c_int(px) - (1 if px < 0 else 0); y1 = c_int(py) - (1 if py < 0 else 0)
        distx = c_int((px - x1) * 256); disty = c_int((py - y1) * 256)
        x1, x2 = bnd(x1, sw); y1, y2 = bnd(y1, sh)
        tl, tr, bl, br = int(src[y1, x1]), int(src[y1, x2]), int(src[y2, x1]), int(src[y2, x2])
        out.append(interp8(tl, tr, bl, br, distx, disty))
        fx += m11; fy += m12
    return out


def fetch_bilinear_scale(src, y, x0, length, i11, i22, idx, idy):
    """fetchTransformedBilinearARGB32PM<BlendTransformedBilinear>, fdy == 0, one call per span (x0, length)"""
    sh, sw = src.shape
    if not fast_matrix(i11, 0.0, 0.0, i22, idx, idy):
        return fetch_bilinear_slow(src, y, x0, length, (i11, 0.0, 0.0, i22, idx, idy))
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
    treats the head, the groups of four and the tail of a run differently; a run never crosses a multiple of 256 in the draw's span
    count, QSpanBuffer's flush), then blended span by span with
    const_alpha = (coverage * intOpacity) >> 8.  A source without alpha channel turns SourceOver into Source
    (getOperator): d = INTERPOLATE_PIXEL_255(s, ca, d, 255 - ca)."""
    i = 0
    while i < len(spans):
        y, x, ln, cov = spans[i]
        j = i + 1; right = x + ln
        # (QSpanBuffer hands the spans on in batches of 256 -- SPAN_BUFFER_SIZE -- and runs only form within a batch)
        while j < len(spans) and spans[j][0] == y and spans[j][1] == right and j // 256 == i // 256:
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


def probe_untransformed(n, seed):
    from PyQt5.QtGui import QImage
    rng = np.random.RandomState(seed)
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




# ---- rotated painter (BAG:902-906: translate, rotate, drawImage on the centred rect) ---------------------------------------
def i32(v):
    v &= 0xffffffff
    return v - (1 << 32) if v & 0x80000000 else v


def fmul(a, b):  # Q16Dot16FastMultiply: 32-bit product
    return i32(a * b) >> 16


def intersect_pixel_fp(x, top, bottom, leftIntersectX, rightIntersectX, slope, invSlope):
    leftX = x << 16; rightX = (x << 16) + 65536
    if slope > 0:
        leftIntersectY = top + mul16(leftX - leftIntersectX, invSlope)
        rightIntersectY = leftIntersectY + invSlope
    else:
        leftIntersectY = top + mul16(leftX - rightIntersectX, invSlope)
        rightIntersectY = leftIntersectY + invSlope
    if leftIntersectX >= leftX and rightIntersectX <= rightX:
        return mul16(bottom - top, leftIntersectX - leftX + ((rightIntersectX - leftIntersectX) >> 1))
    elif leftIntersectX >= rightX:
        return bottom - top
    elif leftIntersectX >= leftX:
        if slope > 0:
            return (bottom - top) - fmul((rightX - leftIntersectX) >> 1, rightIntersectY - top)
        else:
            return (bottom - top) - fmul((rightX - leftIntersectX) >> 1, bottom - rightIntersectY)
    elif rightIntersectX <= leftX:
        return 0
    elif rightIntersectX <= rightX:
        if slope > 0:
            return fmul((rightIntersectX - leftX) >> 1, bottom - leftIntersectY)
        else:
            return fmul((rightIntersectX - leftX) >> 1, leftIntersectY - top)
    else:
        if slope > 0:
            return (bottom - rightIntersectY) + ((rightIntersectY - leftIntersectY) >> 1)
        else:
            return (rightIntersectY - top) + ((leftIntersectY - rightIntersectY) >> 1)


def safe_div(x, y):  # qSafeDivide
    if y == 0:
        return 1e9 if x > 0 else -1e9
    return x / y


def sF16(x):  # qSafeFloatToQ16Dot16
    return F16(min(max(x, -32768.0), 32767.0))


def aa_line_spans_any(ax, ay, bx, by, width, cw=CW, ch=CH, square_cap=False):
    """QRasterizer::rasterizeLine, antialiased, any direction.  General lines: the four corners are snapped DOWN to the 26.6
    grid (snapTo26Dot6Grid), every edge keeps its own slope, a 16.16 trapezoid walker with intersectPixelFP gives the coverage.
    (Qt 5.9 has no "is this part of the row empty" guards around intersectPixelFP: a side corner just above a clipped first row
    contributes a negative exclusion there -- later Qt versions guard it; pinned by the turned-fill probe, 400 / 400.)"""
    c = clip_line(ax, ay, bx, by, width, cw, ch, square_cap)
    if c is None:
        return []
    pax, pay, pbx, pby, w2 = c
    if q26eq(pay, pby) or q26eq(pax, pbx):
        return aa_line_spans(ax, ay, bx, by, width, cw, ch, square_cap)
    width = w2
    if pay > pby:
        pax, pay, pbx, pby = pbx, pby, pax, pay
    dlx = (pbx - pax) * (0.5 * width); dly = (pby - pay) * (0.5 * width)
    perpx, perpy = dly, -dlx
    if pax < pbx:
        top = (pax + perpx, pay + perpy); left = (pax - perpx, pay - perpy); right = (pbx + perpx, pby + perpy); bottom = (pbx - perpx, pby - perpy)
    else:
        top = (pax - perpx, pay - perpy); left = (pbx - perpx, pby - perpy); right = (pax + perpx, pay + perpy); bottom = (pbx + perpx, pby + perpy)
    snap = lambda p: (math.floor(p[0] * 64) * (1 / 64.), math.floor(p[1] * 64) * (1 / 64.))  # snapTo26Dot6Grid
    top, left, right, bottom = snap(top), snap(left), snap(right), snap(bottom)
    clipT, clipB, clipL, clipR = 0, ch - 1, 0, cw - 1
    topBound = min(max(top[1], float(clipT)), float(clipB)); bottomBound = min(max(bottom[1], float(clipT)), float(clipB))
    tlS = safe_div(left[0] - top[0], left[1] - top[1]); blS = safe_div(bottom[0] - left[0], bottom[1] - left[1])
    trS = safe_div(right[0] - top[0], right[1] - top[1]); brS = safe_div(bottom[0] - right[0], bottom[1] - right[1])
    tlFP, trFP, blFP, brFP = sF16(tlS), sF16(trS), sF16(blS), sF16(brS)
    itlFP, itrFP, iblFP, ibrFP = sF16(safe_div(1, tlS)), sF16(safe_div(1, trS)), sF16(safe_div(1, blS)), sF16(safe_div(1, brS))
    spans = []
    def add(x, ln, y, cov):
        if cov and ln and 0 <= y < ch:
            spans.append((y, x, ln, cov & 0xff))  # QT_FT_Span::coverage is an unsigned char: a degenerate first row can hand it a negative int
    iTopFP = c_int(topBound) << 16; iLeftFP = c_int(left[1]) << 16; iRightFP = c_int(right[1]) << 16; iBottomFP = c_int(bottomBound) << 16
    leftIntersectAf = sF16(top[0] + (c_int(topBound) - top[1]) * tlS)
    rightIntersectAf = sF16(top[0] + (c_int(topBound) - top[1]) * trS)
    leftIntersectBf = 0; rightIntersectBf = 0
    if iLeftFP < iTopFP:
        leftIntersectBf = sF16(left[0] + (c_int(topBound) - left[1]) * blS)
    if iRightFP < iTopFP:
        rightIntersectBf = sF16(right[0] + (c_int(topBound) - right[1]) * brS)
    yTopFP = sF16(top[1]); yLeftFP = sF16(left[1]); yRightFP = sF16(right[1]); yBottomFP = sF16(bottom[1])
    rowTop = max(iTopFP, yTopFP)
    topLeftIntersectAf = leftIntersectAf + mul16(tlFP, rowTop - iTopFP)
    topRightIntersectAf = rightIntersectAf + mul16(trFP, rowTop - iTopFP)
    yFP = iTopFP
    bound = lambda v: min(max(v, clipL), clipR)
    while yFP <= iBottomFP:
        rowBottomLeft = min(yFP + 65536, yLeftFP); rowBottomRight = min(yFP + 65536, yRightFP)
        rowTopLeft = max(yFP, yLeftFP); rowTopRight = max(yFP, yRightFP)
        rowBottom = min(yFP + 65536, yBottomFP)
        if yFP == iLeftFP:
            y = yFP >> 16
            leftIntersectBf = sF16(left[0] + (y - left[1]) * blS)
            topLeftIntersectBf = leftIntersectBf + mul16(blFP, rowTopLeft - yFP)
            bottomLeftIntersectAf = leftIntersectAf + mul16(tlFP, rowBottomLeft - yFP)
        else:
            topLeftIntersectBf = leftIntersectBf
            bottomLeftIntersectAf = leftIntersectAf + tlFP
        if yFP == iRightFP:
            y = yFP >> 16
            rightIntersectBf = sF16(right[0] + (y - right[1]) * brS)
            topRightIntersectBf = rightIntersectBf + mul16(brFP, rowTopRight - yFP)
            bottomRightIntersectAf = rightIntersectAf + mul16(trFP, rowBottomRight - yFP)
        else:
            topRightIntersectBf = rightIntersectBf
            bottomRightIntersectAf = rightIntersectAf + trFP
        if yFP == iBottomFP:
            bottomLeftIntersectBf = leftIntersectBf + mul16(blFP, rowBottom - yFP)
            bottomRightIntersectBf = rightIntersectBf + mul16(brFP, rowBottom - yFP)
        else:
            bottomLeftIntersectBf = leftIntersectBf + blFP
            bottomRightIntersectBf = rightIntersectBf + brFP
        if yFP < iLeftFP:
            leftMin = bottomLeftIntersectAf >> 16; leftMax = topLeftIntersectAf >> 16
        elif yFP == iLeftFP:
            leftMin = max(bottomLeftIntersectAf, topLeftIntersectBf) >> 16; leftMax = max(topLeftIntersectAf, bottomLeftIntersectBf) >> 16
        else:
            leftMin = topLeftIntersectBf >> 16; leftMax = bottomLeftIntersectBf >> 16
        leftMin = bound(leftMin); leftMax = bound(leftMax)
        if yFP < iRightFP:
            rightMin = topRightIntersectAf >> 16; rightMax = bottomRightIntersectAf >> 16
        elif yFP == iRightFP:
            rightMin = min(topRightIntersectAf, bottomRightIntersectBf) >> 16; rightMax = max(bottomRightIntersectAf, topRightIntersectBf) >> 16
        else:
            rightMin = bottomRightIntersectBf >> 16; rightMax = topRightIntersectBf >> 16
        rightMin = bound(rightMin); rightMax = bound(rightMax)
        if leftMax > rightMax: leftMax = rightMax
        if rightMin < leftMin: rightMin = leftMin
        rowHeight = rowBottom - rowTop
        yy = yFP >> 16
        def right_excl(x):
            e = 0
            if yFP <= iRightFP:
                e += (rowBottomRight - rowTop) - intersect_pixel_fp(x, rowTop, rowBottomRight, topRightIntersectAf, bottomRightIntersectAf, trFP, itrFP)
            if yFP >= iRightFP:
                e += (rowBottom - rowTopRight) - intersect_pixel_fp(x, rowTopRight, rowBottom, bottomRightIntersectBf, topRightIntersectBf, brFP, ibrFP)
            return e
        x = leftMin
        while x <= leftMax:
            excluded = 0
            if yFP <= iLeftFP:
                excluded += intersect_pixel_fp(x, rowTop, rowBottomLeft, bottomLeftIntersectAf, topLeftIntersectAf, tlFP, itlFP)
            if yFP >= iLeftFP:
                excluded += intersect_pixel_fp(x, rowTopLeft, rowBottom, topLeftIntersectBf, bottomLeftIntersectBf, blFP, iblFP)
            if x >= rightMin:
                excluded += right_excl(x)
            add(x, 1, yy, (255 * (rowHeight - excluded)) >> 16)
            x += 1
        if x < rightMin:
            add(x, rightMin - x, yy, (255 * rowHeight) >> 16)
            x = rightMin
        while x <= rightMax:
            add(x, 1, yy, (255 * (rowHeight - right_excl(x))) >> 16)
            x += 1
        leftIntersectAf += tlFP; leftIntersectBf += blFP
        rightIntersectAf += trFP; rightIntersectBf += brFP
        topLeftIntersectAf = leftIntersectAf; topRightIntersectAf = rightIntersectAf
        yFP += 65536
        rowTop = yFP
    return spans


def q_fuzzy_is_null(v):
    return abs(v) <= 0.000000000001


def painter_matrix(cx, cy, deg):
    """QTransform after translate(cx, cy); rotate(deg): (m11, m12, m21, m22, dx, dy, type)"""
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
    return m11, m12, m21, m22, mdx, mdy, typ


def map_point(M, x, y):
    m11, m12, m21, m22, mdx, mdy, typ = M
    if typ == 'none': return x, y
    if typ == 'translate': return x + mdx, y + mdy
    if typ == 'scale': return m11 * x + mdx, m22 * y + mdy
    return m11 * x + m21 * y + mdx, m12 * x + m22 * y + mdy


def model_fill_rotated(dst, cx, cy, w, h, deg, color):
    ch, cw = dst.shape
    M = painter_matrix(cx, cy, deg)
    rx, ry, rw, rh = -w / 2, -h / 2, w, h
    l, t, r_, b_ = rx, ry, rx + rw, ry + rh
    ax, ay = map_point(M, (l + l) * 0.5, (t + b_) * 0.5)
    bx, by = map_point(M, (r_ + r_) * 0.5, (t + b_) * 0.5)
    for (y, x, ln, cov) in aa_line_spans_any(ax, ay, bx, by, rh / rw, cw, ch):
        for i in range(ln):
            dst[y, x + i] = source_over(int(dst[y, x + i]), color, cov)


def qt_fill_rotated(dst0, cx, cy, w, h, deg, c):
    QImage, QPainter, QColor, QRectF = qt_setup()
    hh, ww = dst0.shape
    img = np_to_qimage(dst0, QImage.Format_RGB32).copy()
    p = QPainter(img)
    p.setRenderHint(QPainter.Antialiasing, True); p.setRenderHint(QPainter.SmoothPixmapTransform, True)
    p.translate(cx, cy); p.rotate(deg)
    p.fillRect(QRectF(-w / 2, -h / 2, w, h), QColor((c >> 16) & 255, (c >> 8) & 255, c & 255, c >> 24))
    p.end()
    ptr = img.constBits(); ptr.setsize(ww * hh * 4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(hh, ww).copy()


def rotated_matrices(cx, cy, w, h, deg, sw, sh):
    """painter matrix M and the texture matrix (inverse of translate(1/65536) * M * translate(r) * scale(r / s))"""
    M = painter_matrix(cx, cy, deg)
    m11, m12, m21, m22, mdx, mdy, typ = M
    rx, ry, rw, rh = -w / 2, -h / 2, w, h
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
    return M, (i11, i12, i21, i22, idx, idy)


def fetch_bilinear_any(src, y, x0, length, T):
    """fetchTransformedBilinearARGB32PM<BlendTransformedBilinear>, any affine matrix (fast_matrix)"""
    i11, i12, i21, i22, idx, idy = T
    sh, sw = src.shape
    if not fast_matrix(i11, i12, i21, i22, idx, idy):
        return fetch_bilinear_slow(src, y, x0, length, (i11, i12, i21, i22, idx, idy))
    fdx = c_int(i11 * 65536.); fdy = c_int(i12 * 65536.)
    if fdy == 0:
        assert i21 == 0 or True
    cx = x0 + 0.5; cy = y + 0.5
    fx = c_int((i21 * cy + i11 * cx + idx) * 65536.) - 32768
    fy = c_int((i22 * cy + i12 * cx + idy) * 65536.) - 32768
    if fdy == 0:
        return fetch_bilinear_scale_fxfy(src, length, fx, fy, fdx, i22)
    def bnd(v, n):
        if v < 0: return 0, 0
        if v >= n - 1: return n - 1, n - 1
        return v, v + 1
    def scalar(fx, fy):
        x1, x2 = bnd(fx >> 16, sw); y1, y2 = bnd(fy >> 16, sh)
        return interp8(int(src[y1, x1]), int(src[y1, x2]), int(src[y2, x1]), int(src[y2, x2]), (fx & 0xffff) >> 8, (fy & 0xffff) >> 8)
    out = []
    if abs(i11) < 1. / 8. or abs(i22) < 1. / 8.:  # zooming more than 8 times: 8-bit distances throughout
        for i in range(length):
            out.append(scalar(fx, fy)); fx += fdx; fy += fdy
        return out
    b = 0
    while b < length:  # head: while a coordinate pair is clamped
        x1, x2 = bnd(fx >> 16, sw); y1, y2 = bnd(fy >> 16, sh)
        if x1 != x2 and y1 != y2:
            break
        out.append(scalar(fx, fy)); fx += fdx; fy += fdy; b += 1
    bounded = length
    if fdx > 0: bounded = min(bounded, b + c_int(((sw - 1) * 65536 - fx) / fdx))
    elif fdx < 0: bounded = min(bounded, b + c_int((0 - fx) / fdx))
    if fdy > 0: bounded = min(bounded, b + c_int(((sh - 1) * 65536 - fy) / fdy))
    elif fdy < 0: bounded = min(bounded, b + c_int((0 - fy) / fdy))
    bounded -= 3
    while b < bounded:  # groups of four, rounded 4-bit distances
        for k in range(4):
            x1 = fx >> 16; y1 = fy >> 16
            out.append(interp16(int(src[y1, x1]), int(src[y1, x1 + 1]), int(src[y1 + 1, x1]), int(src[y1 + 1, x1 + 1]),
                                ((fx & 0xffff) + 0x800) >> 12, ((fy & 0xffff) + 0x800) >> 12))
            fx += fdx; fy += fdy
        b += 4
    while b < length:
        out.append(scalar(fx, fy)); fx += fdx; fy += fdy; b += 1
    return out


def fetch_bilinear_scale_fxfy(src, length, fx, fy, fdx, i22):
    """the fdy == 0 branch for either sign of fdx (a painter turned by 180 degrees mirrors the source: fdx < 0)"""
    sh, sw = src.shape
    y1 = fy >> 16
    if y1 < 0: y1 = y2 = 0
    elif y1 >= sh - 1: y1 = y2 = sh - 1
    else: y2 = y1 + 1
    def bx(x1):
        if x1 < 0: return 0, 0
        if x1 >= sw - 1: return sw - 1, sw - 1
        return x1, x1 + 1
    def scalar(fx):
        x1, x2 = bx(fx >> 16)
        return interp8(int(src[y1, x1]), int(src[y1, x2]), int(src[y2, x1]), int(src[y2, x2]), (fx & 0xffff) >> 8, (fy & 0xffff) >> 8)
    out = []
    if (0 < fdx <= 65536) or (fdx < 0 and fdx > -(65536 // 8)) or abs(i22) < 1. / 8.:
        for i in range(length):
            out.append(scalar(fx)); fx += fdx
        return out
    b = 0
    while b < length:
        x1, x2 = bx(fx >> 16)
        if x1 != x2:
            break
        out.append(scalar(fx)); fx += fdx; b += 1
    bounded = length
    if fdx > 0: bounded = min(bounded, b + c_int(((sw - 1) * 65536 - fx) / fdx))
    elif fdx < 0: bounded = min(bounded, b + c_int((0 - fx) / fdx))
    bounded -= 3
    dy4 = ((fy & 0xffff) + 0x800) >> 12
    while b < bounded:
        for k in range(4):
            x1, x2 = bx(fx >> 16)
            out.append(interp16(int(src[y1, x1]), int(src[y1, x2]), int(src[y2, x1]), int(src[y2, x2]), ((fx & 0xffff) + 0x800) >> 12, dy4))
            fx += fdx
        b += 4
    while b < length:
        out.append(scalar(fx)); fx += fdx; b += 1
    return out


def model_draw_rotated(dst, src, cx, cy, w, h, deg, opacity=1.0):
    ch, cw = dst.shape
    sh, sw = src.shape
    if w <= 0 or h <= 0:
        return
    M, T = rotated_matrices(cx, cy, w, h, deg, sw, sh)
    rx, ry, rw, rh = -w / 2, -h / 2, w, h
    l, t, r_, b_ = rx, ry, rx + rw, ry + rh
    ax, ay = map_point(M, (l + l) * 0.5, (t + b_) * 0.5)
    bx, by = map_point(M, (r_ + r_) * 0.5, (t + b_) * 0.5)
    spans = aa_line_spans_any(ax, ay, bx, by, rh / rw, cw, ch)
    io = c_int(min(max(opacity, 0.0), 1.0) * 256)
    blend_runs(dst, spans, lambda y, x, n: fetch_bilinear_any(src, y, x, n, T), io, False)


def qt_draw_rotated(dst0, src, cx, cy, w, h, deg, opacity):
    QImage, QPainter, QColor, QRectF = qt_setup()
    hh, ww = dst0.shape
    img = np_to_qimage(dst0, QImage.Format_RGB32).copy()
    q = np_to_qimage(src, QImage.Format_ARGB32_Premultiplied)
    p = QPainter(img)
    p.setRenderHint(QPainter.Antialiasing, True); p.setRenderHint(QPainter.SmoothPixmapTransform, True)
    if opacity != 1:
        p.setOpacity(opacity)
    p.translate(cx, cy); p.rotate(deg)
    p.drawImage(QRectF(-w / 2, -h / 2, w, h), q)
    p.end()
    ptr = img.constBits(); ptr.setsize(ww * hh * 4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(hh, ww).copy()


def probe_turned_fills(n, seed):
    """fillRect on a turned painter: the coverage of the antialiased trapezoid walker, pixel by pixel"""
    rng = np.random.RandomState(seed)
    miss = worst = 0
    for case in range(n):
        dst0 = np.full((CH, CW), 0xff000000, np.uint32)
        w = rng.uniform(2, 70); h = rng.uniform(2, 70); cx = rng.uniform(-10, CW + 10); cy = rng.uniform(-10, CH + 10)
        deg = float(np.float32(rng.uniform(-180, 180))) if case % 5 else [90., -90., 180., 45., 30.][case // 5 % 5]
        got = qt_fill_rotated(dst0, cx, cy, w, h, deg, 0xffffffff)
        want = dst0.copy(); model_fill_rotated(want, cx, cy, w, h, deg, 0xffffffff)
        d = np.abs((got & 255).astype(int) - (want & 255).astype(int))
        if d.max() > 0:
            miss += 1; worst = max(worst, int(d.max()))
            ys, xs = np.nonzero(d)
            print(f"case {case} c=({cx:.3f},{cy:.3f}) wh=({w:.3f},{h:.3f}) deg {deg:.4f}: {len(ys)} px differ, max {d.max()}, first (x={xs[0]},y={ys[0]})")
    print(f"turned fills: {n} cases, {miss} with differences, worst {worst}")


def probe_turned_images(n, seed):
    """drawImage on a turned painter (BAG:902-906): coverage + the rotation branch of the bilinear fetch + blend"""
    rng = np.random.RandomState(seed)
    miss = worst = 0
    for case in range(n):
        dst0 = rand_src(rng, CW, CH, False)
        sw, sh = [(16, 16), (40, 24), (100, 100)][rng.randint(0, 3)]
        src = rand_src(rng, sw, sh, True)
        sc = [0.4, 0.9, 1.5, 3.0][rng.randint(0, 4)]
        w = min(sw * sc * rng.uniform(.8, 1.2), 90); h = min(sh * sc * rng.uniform(.8, 1.2), 90)
        cx = rng.uniform(5, CW - 5); cy = rng.uniform(10, CH - 5)
        deg = float(np.float32(rng.uniform(-180, 180))) if case % 4 else [90., -90., 45., 180.][case // 4 % 4]
        op = 1.0 if case % 3 else float(np.float32(rng.uniform(0, 1)))
        got = qt_draw_rotated(dst0, src, cx, cy, w, h, deg, op)
        want = dst0.copy().astype(np.uint32); model_draw_rotated(want, src, cx, cy, w, h, deg, op)
        g = got.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int); wv = want.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int)
        d = np.abs(g - wv)
        if d.max() > 0:
            miss += 1; worst = max(worst, int(d.max()))
            ys, xs = np.nonzero(d.max(axis=2))
            print(f"case {case} src {sw}x{sh} c=({cx:.3f},{cy:.3f}) wh=({w:.3f},{h:.3f}) deg {deg:.4f} op {op:.2f}: {len(ys)} px differ, max {d.max()}, first (x={xs[0]},y={ys[0]})")
    print(f"turned images: {n} cases, {miss} with differences, worst {worst}")


def model_wide_line(dst, x1, y1, x2, y2, pen_width, color):
    """QPainter::drawLine(int, int, int, int) with a solid pen wider than 1 under Antialiasing (QRasterPaintEngine::stroke, LinesHint):
    rasterizeLine(p1, p2, width / length, squareCap) -- QPen's default cap is SquareCap; a zero-length line draws a pen-wide dash"""
    ch, cw = dst.shape
    if x1 == x2 and y1 == y2:
        spans = aa_line_spans_any(x1 - pen_width * 0.5, float(y1), x1 + pen_width * 0.5, float(y1), 1.0, cw, ch)
    else:
        length = math.sqrt(float(x2 - x1) ** 2 + float(y2 - y1) ** 2)
        spans = aa_line_spans_any(float(x1), float(y1), float(x2), float(y2), pen_width / length, cw, ch, square_cap=True)
    for (y, x, ln, cov) in spans:
        for i in range(ln):
            dst[y, x + i] = source_over(int(dst[y, x + i]), color, cov)


def qt_wide_line(dst0, x1, y1, x2, y2, pen_width, c):
    QImage, QPainter, QColor, QRectF = qt_setup()
    from PyQt5.QtGui import QPen
    hh, ww = dst0.shape
    img = np_to_qimage(dst0, QImage.Format_RGB32).copy()
    p = QPainter(img)
    p.setRenderHint(QPainter.Antialiasing, True); p.setRenderHint(QPainter.SmoothPixmapTransform, True)
    p.setPen(QPen(QColor((c >> 16) & 255, (c >> 8) & 255, c & 255), pen_width))
    p.drawLine(int(x1), int(y1), int(x2), int(y2))
    p.end()
    ptr = img.constBits(); ptr.setsize(ww * hh * 4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(hh, ww).copy()


def probe_wide_lines(n, seed):
    """jumper's compass needle at 512 pixels: an antialiased line with a pen of 2 * compass_dim pixels"""
    rng = np.random.RandomState(seed)
    miss = worst = 0
    for case in range(n):
        dst0 = np.full((CH, CW), 0xff000000, np.uint32)
        x1, y1, x2, y2 = [int(v) for v in rng.randint(-10, CW + 10, size=4)]
        if case % 7 == 0: x2 = x1
        if case % 11 == 0: y2 = y1
        pw = int(rng.randint(2, 9))
        got = qt_wide_line(dst0, x1, y1, x2, y2, pw, 0xfffcba03)
        want = dst0.copy(); model_wide_line(want, x1, y1, x2, y2, pw, 0xfffcba03)
        d = np.abs(got.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int) - want.view(np.uint8).reshape(CH, CW, 4)[..., :3].astype(int))
        if d.max() > 0:
            miss += 1; worst = max(worst, int(d.max()))
            ys, xs = np.nonzero(d.max(axis=2))
            if miss <= 10:
                print(f"case {case} line ({x1},{y1})-({x2},{y2}) pen {pw}: {len(ys)} px differ, max {d.max()}, first (x={xs[0]},y={ys[0]})")
    print(f"wide lines: {n} cases, {miss} with differences, worst {worst}")


def q_premultiply(p):
    a = p >> 24
    if a == 255: return p
    if a == 0: return 0
    t = (p & 0xff00ff) * a; t = ((t + ((t >> 8) & 0xff00ff) + 0x800080) >> 8) & 0xff00ff
    g = ((p >> 8) & 0xff) * a; g = (g + ((g >> 8) & 0xff) + 0x80) & 0xff00
    return g | t | (a << 24)


def fetch_generic_scale(src, y, x0, length, i11, i22, idx, idy):
    """the GENERIC fetchTransformedBilinear (a source that is not ARGB32_Premultiplied / RGB32, e.g. the reference's generated
    assets in Format_ARGB32): texels converted one by one (qPremultiply), then ONE formula for every pixel of the run -- 8-bit
    distances when scaling up on x or zooming more than 8 times, rounded 4-bit distances otherwise"""
    sh, sw = src.shape
    fdx = c_int(i11 * 65536.)
    fx = c_int((i11 * (x0 + .5) + idx) * 65536.) - 32768
    fy = c_int((i22 * (y + .5) + idy) * 65536.) - 32768
    y1 = fy >> 16
    if y1 < 0: y1 = y2 = 0
    elif y1 >= sh - 1: y1 = y2 = sh - 1
    else: y2 = y1 + 1
    out = []
    for i in range(length):
        x1 = fx >> 16
        if x1 < 0: x1 = x2 = 0
        elif x1 >= sw - 1: x1 = x2 = sw - 1
        else: x2 = x1 + 1
        tl, tr, bl, br = [q_premultiply(int(v)) for v in (src[y1, x1], src[y1, x2], src[y2, x1], src[y2, x2])]
        if 0 < fdx <= 65536 or (fdx < 0 and fdx > -8192) or abs(i22) < 1 / 8.:
            out.append(interp8(tl, tr, bl, br, (fx & 0xffff) >> 8, (fy & 0xffff) >> 8))
        else:
            out.append(interp16(tl, tr, bl, br, ((fx & 0xffff) + 0x800) >> 12, ((fy & 0xffff) + 0x800) >> 12))
        fx += fdx
    return out


def probe_generic_source(n, seed):
    """drawImage of a Format_ARGB32 (not premultiplied) source: use_generated_assets under render_human"""
    from PyQt5.QtGui import QImage
    rng = np.random.RandomState(seed)
    miss = 0
    for case in range(n):
        sw = sh = 64
        a = rng.randint(0, 256, size=(sh, sw, 4)).astype(np.uint32)
        m = rng.randint(0, 3, size=(sh, sw)); a[..., 3] = np.where(m == 0, 0, np.where(m == 1, 255, a[..., 3]))
        src = (a[..., 3] << 24) | (a[..., 2] << 16) | (a[..., 1] << 8) | a[..., 0]
        sc = [0.3, 0.6, 0.9, 1.5, 3.0][case % 5]
        rw = sw * sc * rng.uniform(.9, 1.1); rh = sh * sc * rng.uniform(.9, 1.1)
        rx = rng.uniform(0, CW - min(rw, 60)); ry = rng.uniform(0, CH - min(rh, 60))
        dst0 = rand_src(rng, CW, CH, False)
        got = qt_draw(dst0, [("image", src, QImage.Format_ARGB32, (rx, ry, rw, rh), 1.0)])
        l, t, r_, b_ = rx, ry, rx + rw, ry + rh
        spans = aa_line_spans(l, (t + b_) * .5, r_, (t + b_) * .5, rh / rw, CW, CH)
        i11, i22, idx, idy = setup_matrix(rx, ry, rw, rh, sw, sh)
        want = dst0.copy().astype(np.uint32)
        blend_runs(want, spans, lambda y, x, nn: fetch_generic_scale(src, y, x, nn, i11, i22, idx, idy), 256, False)
        if (want != got).any():
            miss += 1
    print(f"generic (ARGB32) sources: {n} cases, {miss} with differences")


if __name__ == "__main__":
    # usage: qt_smooth_aa_probe.py [n_cases] [seed] [untransformed|fills|images|all|lines|generic]
    n_ = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed_ = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    what = sys.argv[3] if len(sys.argv) > 3 else "all"
    if what in ("untransformed", "all"):
        probe_untransformed(n_, seed_)
    if what in ("fills", "all"):
        probe_turned_fills(n_, seed_)
    if what in ("images", "all"):
        probe_turned_images(n_, seed_)
    if what in ("generic",):  # use_generated_assets under render_human
        probe_generic_source(n_, seed_)
    if what in ("lines",):  # groundwork for jumper's compass under render_human (not in the product yet)
        probe_wide_lines(n_, seed_)

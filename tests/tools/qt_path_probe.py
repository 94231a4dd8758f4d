"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7): pins the non-antialiased path route that Qt 5.9 takes for
QPainter::drawEllipse(QRectF) on a rect that is not integer aligned (reference call sites: src/games/jumper.cpp:137-142
compass in easy mode / without center_agent, src/assetgen.cpp:99-105 generated assets):

  QRasterPaintEngine::drawEllipse -> QPaintEngineEx::drawEllipse: qt_curves_for_arc (4 cubics, kappa .5522847498)
    brush: QRasterPaintEngine::fill -> QOutlineMapper (each cubic flattened by QBezier::addToPolygon, threshold .25,
           points to 26.6 by qRound(v*64)) -> QRasterizer's QScanConverter (16.16 edge walkers sampled at pixel centres)
    pen  : width <= 1 -> QCosmeticStroker::drawPath: calculateLastPoint for the closed path, renderCubic (<= 6 levels of
           subdivision), drawLine<drawPixel, NoDasher> with its drop-out / duplicate-pixel control

The model below is the one restated in oracle/procgen_oracle.c (qt_path_ellipse) and procgen_amd/csrc/pg_qtpath.h.
"""
import os, sys, math
os.environ["QT_QPA_PLATFORM"] = "offscreen"
import numpy as np

W = H = 64
KAPPA = 0.5522847498
INT_MIN = -(1 << 31)


def qround(d):
    return int(d + 0.5) if d >= 0.0 else int(d - float(int(d - 1)) + 0.5) + int(d - 1)


def c_int(v):
    """C (int) conversion of a double: truncation."""
    return int(v)


def i32(v):
    v &= 0xffffffff
    return v - (1 << 32) if v & 0x80000000 else v


# ---------------------------------------------------------------- path construction
def curves_for_arc(x, y, w, h):
    """qt_curves_for_arc(rect, 0, -360): start point + 12 control points (qpainterpath.cpp)."""
    w2 = w / 2; w2k = w2 * KAPPA
    h2 = h / 2; h2k = h2 * KAPPA
    pts = [
        (x + w, y + h2),
        (x + w, y + h2 + h2k), (x + w2 + w2k, y + h), (x + w2, y + h),
        (x + w2 - w2k, y + h), (x, y + h2 + h2k), (x, y + h2),
        (x, y + h2 - h2k), (x + w2 - w2k, y), (x + w2, y),
        (x + w2 + w2k, y), (x + w, y + h2 - h2k), (x + w, y + h2),
    ]
    return pts


# ---------------------------------------------------------------- fill: outline mapper + scan converter
def bezier_split(b):
    x1, y1, x2, y2, x3, y3, x4, y4 = b
    c = (x2 + x3) * .5
    fx2 = (x1 + x2) * .5
    sx3 = (x3 + x4) * .5
    fx3 = (fx2 + c) * .5
    sx2 = (sx3 + c) * .5
    mx = (fx3 + sx2) * .5
    c = (y2 + y3) / 2
    fy2 = (y1 + y2) * .5
    sy3 = (y3 + y4) * .5
    fy3 = (fy2 + c) * .5
    sy2 = (sy3 + c) * .5
    my = (fy3 + sy2) * .5
    return (x1, y1, fx2, fy2, fx3, fy3, mx, my), (mx, my, sx2, sy2, sx3, sy3, x4, y4)


def add_to_polygon(b, out, thr):
    """QBezier::addToPolygon(QDataBuffer<QPointF>&, qreal)."""
    stack = [(b, 9)]
    while stack:
        b, lvl = stack[-1]
        x1, y1, x2, y2, x3, y3, x4, y4 = b
        y4y1 = y4 - y1; x4x1 = x4 - x1
        l = abs(x4x1) + abs(y4y1)
        if l > 1.:
            d = abs(x4x1 * (y1 - y2) - y4y1 * (x1 - x2)) + abs(x4x1 * (y1 - y3) - y4y1 * (x1 - x3))
        else:
            d = abs(x1 - x2) + abs(y1 - y2) + abs(x1 - x3) + abs(y1 - y3)
            l = 1.
        if d < thr * l or lvl == 0:
            out.append((x4, y4))
            stack.pop()
        else:
            first, second = bezier_split(b)
            stack[-1] = (second, lvl - 1)
            stack.append((first, lvl - 1))


def flatten_ellipse(x, y, w, h):
    p = curves_for_arc(x, y, w, h)
    elems = [p[0]]
    for k in range(4):
        a = elems[-1]
        add_to_polygon((a[0], a[1], p[1 + 3 * k][0], p[1 + 3 * k][1], p[2 + 3 * k][0], p[2 + 3 * k][1], p[3 + 3 * k][0], p[3 + 3 * k][1]), elems, 0.25)
    if elems[-1] != elems[0]:
        elems.append(elems[0])
    return elems


def scan_fill(elems, clip=(0, 0, W - 1, H - 1)):
    """QRasterizer::rasterize(outline) + QScanConverter, odd-even; returns list of (y, x0, x1) spans (x1 exclusive)."""
    cl, ct, cr, cb = clip
    pts = [(qround(px * 64), qround(py * 64)) for px, py in elems]
    min_y = min(p[1] for p in pts); max_y = max(p[1] for p in pts)
    top = max(ct, (min_y + 32) >> 6); bot = min(cb, (max_y - 32) >> 6)
    if top > bot:
        return []
    rows = {}
    for i in range(len(pts) - 1):
        a, b = pts[i], pts[i + 1]
        wind = 1
        if a[1] > b[1]:
            a, b = b, a; wind = -1
        itop = max(top, (a[1] + 32) >> 6); ibot = min(bot, (b[1] - 32) >> 6)
        if itop > ibot:
            continue
        afp = 32768 + (a[0] << 10)
        if b[0] == a[0]:
            xfp = afp; slope = 0
        else:
            s = (b[0] - a[0]) / float(b[1] - a[1])
            slope = c_int(s * 65536.)
            xfp = afp + ((slope * ((itop << 16) + 32768 - (a[1] << 10))) >> 16)
        for yy in range(itop, ibot + 1):
            rows.setdefault(yy, []).append((xfp >> 16, wind))
            xfp += slope
    spans = []
    for yy in sorted(rows):
        xs = sorted(rows[yy])
        wnd = 0; x = 0
        for cur, wd in xs:
            if wnd & 1:
                x0 = max(x, cl); x1 = min(cur, cr + 1)
                if x1 > x0:
                    spans.append((yy, x0, x1))
            x = cur; wnd += wd
    return spans


# ---------------------------------------------------------------- pen: cosmetic stroker
TB, BT, LR, RL = 1, 2, 4, 8
TURNCAP = int(os.environ.get("TURNCAP", "1"))


def tdiv(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def fixdiv(x, y):
    """F16Dot16FixedDiv: (qint64(x) << 16) / y with C truncation."""
    return i32(tdiv(x << 16, y))


def to26(v):
    return c_int(v * 64.)


class Stroker:
    def __init__(self, put):
        self.put = put
        self.lastDir = LR
        self.lastPixel = (INT_MIN, INT_MIN)
        self.lastAxisAligned = False

    def clip_line(self, x1, y1, x2, y2):
        """QCosmeticStroker::clipLine; bounds = device rect widened by one pixel (setup())."""
        xmin = -1.; xmax = W + 1.; ymin = -1.; ymax = H + 1.
        def kill():
            self.lastPixel = (INT_MIN, self.lastPixel[1])
        if x1 < xmin:
            if x2 <= xmin:
                kill(); return None
            y1 += (y2 - y1) / (x2 - x1) * (xmin - x1); x1 = xmin
        elif x1 > xmax:
            if x2 >= xmax:
                kill(); return None
            y1 += (y2 - y1) / (x2 - x1) * (xmax - x1); x1 = xmax
        if x2 < xmin:
            kill(); y2 += (y2 - y1) / (x2 - x1) * (xmin - x2); x2 = xmin
        elif x2 > xmax:
            kill(); y2 += (y2 - y1) / (x2 - x1) * (xmax - x2); x2 = xmax
        if y1 < ymin:
            if y2 <= ymin:
                kill(); return None
            x1 += (x2 - x1) / (y2 - y1) * (ymin - y1); y1 = ymin
        elif y1 > ymax:
            if y2 >= ymax:
                kill(); return None
            x1 += (x2 - x1) / (y2 - y1) * (ymax - y1); y1 = ymax
        if y2 < ymin:
            kill(); x2 += (x2 - x1) / (y2 - y1) * (ymin - y2); y2 = ymin
        elif y2 > ymax:
            kill(); x2 += (x2 - x1) / (y2 - y1) * (ymax - y2); y2 = ymax
        return x1, y1, x2, y2

    def calc_last_point(self, rx1, ry1, rx2, ry2):
        self.lastPixel = (INT_MIN, INT_MIN)
        c = self.clip_line(rx1, ry1, rx2, ry2)
        if c is None:
            return
        rx1, ry1, rx2, ry2 = c
        x1 = to26(rx1); y1 = to26(ry1); x2 = to26(rx2); y2 = to26(ry2)
        dx = abs(x2 - x1); dy = abs(y2 - y1)
        if dx < dy:
            swapped = False
            if y1 > y2:
                swapped = True; y1, y2 = y2, y1; x1, x2 = x2, x1
            xinc = fixdiv(x2 - x1, y2 - y1)
            x = x1 << 10
            y = (y1 + 32) >> 6; ys = (y2 + 32) >> 6
            rnd = 32 if xinc > 0 else 0
            if y != ys:
                x += (((y << 6) + rnd - y1) * xinc) >> 6
                if swapped:
                    self.lastPixel = (x >> 16, y); self.lastDir = BT
                else:
                    self.lastPixel = ((x + (ys - y - 1) * xinc) >> 16, ys - 1); self.lastDir = TB
                self.lastAxisAligned = abs(xinc) < (1 << 14)
        else:
            if not dx:
                return
            swapped = False
            if x1 > x2:
                swapped = True; x1, x2 = x2, x1; y1, y2 = y2, y1
            yinc = fixdiv(y2 - y1, x2 - x1)
            y = y1 << 10
            x = (x1 + 32) >> 6; xs = (x2 + 32) >> 6
            rnd = 32 if yinc > 0 else 0
            if x != xs:
                y += (((x << 6) + rnd - x1) * yinc) >> 6
                if swapped:
                    self.lastPixel = (x, y >> 16); self.lastDir = RL
                else:
                    self.lastPixel = (xs - 1, (y + (xs - x - 1) * yinc) >> 16); self.lastDir = LR
                self.lastAxisAligned = abs(yinc) < (1 << 14)

    def line(self, rx1, ry1, rx2, ry2, caps=0):
        c = self.clip_line(rx1, ry1, rx2, ry2)
        if c is None:
            return
        rx1, ry1, rx2, ry2 = c
        x1 = to26(rx1); y1 = to26(ry1); x2 = to26(rx2); y2 = to26(ry2)
        dx = abs(x2 - x1); dy = abs(y2 - y1)
        last = self.lastPixel
        lp = self.lastPixel
        if dx < dy:
            d = TB; swapped = False
            if y1 > y2:
                swapped = True; y1, y2 = y2, y1; x1, x2 = x2, x1; d = BT
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1)
            xinc = fixdiv(x2 - x1, y2 - y1)
            x = x1 << 10
            if TURNCAP and (self.lastDir ^ 3) == d:
                caps |= 2 if swapped else 1
            # capAdjust
            if caps & 1:
                y1 -= 32; x -= xinc >> 1
            if caps & 2:
                y2 += 32
            y = (y1 + 32) >> 6; ys = (y2 + 32) >> 6
            rnd = 32 if xinc > 0 else 0
            if (caps & 1) and lp[1] == y + 1:  # "capAdjust made us round away from what calculateLastPoint gave us"
                y += 1
            if y != ys:
                x += (((y << 6) + rnd - y1) * xinc) >> 6
                first = (x >> 16, y)
                last = ((x + (ys - y - 1) * xinc) >> 16, ys - 1)
                if swapped:
                    first, last = last, first
                axis = abs(xinc) < (1 << 14)
                if lp[0] > INT_MIN:
                    if first == lp:
                        if swapped:
                            ys -= 1
                        else:
                            y += 1; x += xinc
                    elif self.lastDir != d and ((axis and self.lastAxisAligned and lp[0] != first[0] and lp[1] != first[1]) or
                                                (abs(lp[0] - first[0]) > 1 or abs(lp[1] - first[1]) > 1)):
                        if swapped:
                            ys += 1
                        else:
                            y -= 1; x -= xinc
                    elif self.lastDir == d and (abs(lp[0] - first[0]) <= 1 and abs(lp[1] - first[1]) > 1):
                        x += xinc >> 1
                        if swapped:
                            last = (x >> 16, last[1])
                        else:
                            last = ((x + (ys - y - 1) * xinc) >> 16, last[1])
                self.lastDir = d; self.lastAxisAligned = axis
                while True:
                    if y >= ys and False:
                        break
                    self.put(x >> 16, y)
                    x += xinc; y += 1
                    if not y < ys:
                        break
        else:
            if not dx:
                return
            d = LR; swapped = False
            if x1 > x2:
                swapped = True; x1, x2 = x2, x1; y1, y2 = y2, y1; d = RL
                caps = ((caps & 1) << 1) | ((caps & 2) >> 1)
            yinc = fixdiv(y2 - y1, x2 - x1)
            y = y1 << 10
            if TURNCAP and (self.lastDir ^ 0xc) == d:
                caps |= 2 if swapped else 1
            if caps & 1:
                x1 -= 32; y -= yinc >> 1
            if caps & 2:
                x2 += 32
            x = (x1 + 32) >> 6; xs = (x2 + 32) >> 6
            rnd = 32 if yinc > 0 else 0
            if (caps & 1) and lp[0] == x + 1:
                x += 1
            if x != xs:
                y += (((x << 6) + rnd - x1) * yinc) >> 6
                first = (x, y >> 16)
                last = (xs - 1, (y + (xs - x - 1) * yinc) >> 16)
                if swapped:
                    first, last = last, first
                axis = abs(yinc) < (1 << 14)
                if lp[0] > INT_MIN:
                    if first == lp:
                        if swapped:
                            xs -= 1
                        else:
                            x += 1; y += yinc
                    elif self.lastDir != d and ((axis and self.lastAxisAligned and lp[0] != first[0] and lp[1] != first[1]) or
                                                (abs(lp[0] - first[0]) > 1 or abs(lp[1] - first[1]) > 1)):
                        if swapped:
                            xs += 1
                        else:
                            x -= 1; y -= yinc
                    elif self.lastDir == d and (abs(lp[0] - first[0]) <= 1 and abs(lp[1] - first[1]) > 1):  # sic: same test as the vertical branch
                        y += yinc >> 1
                        if swapped:
                            last = (last[0], y >> 16)
                        else:
                            last = (last[0], (y + (xs - x - 1) * yinc) >> 16)
                self.lastDir = d; self.lastAxisAligned = axis
                while True:
                    self.put(x, y >> 16)
                    y += yinc; x += 1
                    if not x < xs:
                        break
        self.lastPixel = last

    def cubic(self, p1, p2, p3, p4, caps=0):
        pts = [None] * (3 * 6 + 4)
        pts[3] = p1; pts[2] = p2; pts[1] = p3; pts[0] = p4
        self._sub(pts, 0, 6, caps)

    def _sub(self, pts, o, level, caps):
        if level:
            dx = pts[o + 3][0] - pts[o][0]; dy = pts[o + 3][1] - pts[o][1]
            ln = .25 * (abs(dx) + abs(dy))
            if abs(dx * (pts[o][1] - pts[o + 2][1]) - dy * (pts[o][0] - pts[o + 2][0])) >= ln or \
               abs(dx * (pts[o][1] - pts[o + 1][1]) - dy * (pts[o][0] - pts[o + 1][0])) >= ln:
                self._split(pts, o)
                self._sub(pts, o + 3, level - 1, caps & 1)
                self._sub(pts, o, level - 1, caps & 2)
                return
        self.line(pts[o + 3][0], pts[o + 3][1], pts[o][0], pts[o][1], caps)

    @staticmethod
    def _split(pts, o):
        res = []
        for k in (0, 1):
            p0, c, d, p3 = pts[o][k], pts[o + 1][k], pts[o + 2][k], pts[o + 3][k]
            p6 = p3
            a = (p0 + c) * .5; q1 = a
            b = (p3 + d) * .5; q5 = b
            c = (c + d) * .5
            a = (a + c) * .5; q2 = a
            b = (b + c) * .5; q4 = b
            q3 = (a + b) * .5
            res.append((p0, q1, q2, q3, q4, q5, p6))
        for j in range(7):
            pts[o + j] = (res[0][j], res[1][j])


def stroke_ellipse(x, y, w, h, put):
    p = curves_for_arc(x, y, w, h)
    s = Stroker(put)
    # closed path: last two points of the path (cp2 and end point of the last cubic)
    s.calc_last_point(p[11][0], p[11][1], p[12][0], p[12][1])
    cur = p[0]
    for k in range(4):
        s.cubic(cur, p[1 + 3 * k], p[2 + 3 * k], p[3 + 3 * k], 0)
        cur = p[3 + 3 * k]


def fill_culled(x, y, w, h):
    """QRasterPaintEngine::fill (5.9): controlPointRect().toRect() = QRect(qRound(x), qRound(y), qRound(w), qRound(h)) must
    intersect the device rect -- a rect rounded to 0 x 0 never does, one rounded to zero width at column 0 neither."""
    l = x; r = x + w; t = y; b = y + h          # QVectorPath::controlPointRect: min / max over the 13 points
    x1 = qround(l); y1 = qround(t); x2 = x1 + qround(r - l) - 1; y2 = y1 + qround(b - t) - 1
    if x2 == x1 - 1 and y2 == y1 - 1:
        return True
    if x1 > W - 1 or 0 > x2 or y1 > H - 1 or 0 > y2:
        return True
    return False


def model_ellipse(x, y, w, h, pen=True, brush=True):
    """0 = untouched, 1 = brush, 2 = pen."""
    m = np.zeros((H, W), np.uint8)
    if brush and not fill_culled(x, y, w, h):
        for yy, x0, x1 in scan_fill(flatten_ellipse(x, y, w, h)):
            m[yy, x0:x1] = 1
    if pen:
        def put(px, py):
            if 0 <= px < W and 0 <= py < H:
                m[py, px] = 2
        stroke_ellipse(x, y, w, h, put)
    return m


# ---------------------------------------------------------------- Qt side
def main():
    from PyQt5.QtGui import QImage, QPainter, QGuiApplication, QColor, QPen, QBrush
    from PyQt5.QtCore import QRectF, Qt
    app = QGuiApplication(sys.argv[:1])

    def qt_ellipse(x, y, w, h, pen=True, brush=True):
        img = QImage(W, H, QImage.Format_RGB32); img.fill(QColor(0, 0, 0))
        p = QPainter(img)
        p.setBrush(QBrush(QColor(0, 0, 255)) if brush else QBrush(Qt.NoBrush))
        p.setPen(QPen(QColor(255, 0, 0), 1) if pen else QPen(Qt.NoPen))
        p.drawEllipse(QRectF(x, y, w, h)); p.end()
        ptr = img.constBits(); ptr.setsize(W * H * 4)
        a = np.frombuffer(bytes(ptr), np.uint32).reshape(H, W) & 0xffffff
        return np.where(a == 0xff, 1, np.where(a == 0xff0000, 2, 0)).astype(np.uint8)

    golden = len(sys.argv) > 1 and sys.argv[1] == "golden"  # write tests/golden/qt_path_ellipses.npz (Qt's own pixels)
    if golden:
        sys.argv[1:] = ["20260924", "1200"]
    rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    cases = []
    if golden:  # the jumper compass rects: (mode visibility or world dim, compass dim), float arithmetic of BAG:819-838,803-805
        for vis, cd in ((12, 3), (16, 2), (20, 3), (40, 2), (45, 2)):
            raw = np.float32(64) / np.float32(vis); unit = np.float32(np.float64(raw) * 1.0); vd = np.float32(64.0 / np.float64(raw))
            cx = np.float32(np.float64(vd - np.float32(cd)) - .25)
            cases.append((float(cx * unit), float(np.float32(.25) * unit), float(np.float32(cd) * unit), float(np.float32(cd) * unit)))
    for i in range(n):
        k = i % 6
        if k == 0:    # anywhere, any size
            w = rng.uniform(.3, 40); h = rng.uniform(.3, 40); x = rng.uniform(-5, 64 - w + 5); y = rng.uniform(-5, 64 - h + 5)
        elif k == 1:  # small shapes (assetgen blotches)
            w = rng.uniform(.05, 6); h = rng.uniform(.05, 6); x = rng.uniform(0, 60); y = rng.uniform(0, 60)
        elif k == 2:  # knife edges: multiples of 1/64, 1/2
            w = rng.randint(1, 2000) / 64.; h = rng.randint(1, 2000) / 64.; x = rng.randint(0, 2000) / 64.; y = rng.randint(0, 2000) / 64.
        elif k == 3:  # float32-valued (the reference computes rects in float)
            w = float(np.float32(rng.uniform(1, 30))); h = float(np.float32(rng.uniform(1, 30)))
            x = float(np.float32(rng.uniform(0, 34))); y = float(np.float32(rng.uniform(0, 34)))
        elif k == 4:  # integer position, fractional size and the other way round
            w = rng.uniform(1, 30); h = w; x = float(rng.randint(0, 30)); y = float(rng.randint(0, 30))
            if rng.randint(2):
                w = float(int(w)); h = float(int(h)); x += rng.uniform(0, 1); y += rng.uniform(0, 1)
        elif k == 5 and i % 12 == 5:  # tiny / at the borders: the toRect() cull
            w = rng.uniform(.05, 3); h = rng.uniform(.05, 3); x = rng.choice([rng.uniform(-2, 1), rng.uniform(61, 65), rng.uniform(0, 64)]); y = rng.choice([rng.uniform(-2, 1), rng.uniform(61, 65), rng.uniform(0, 64)])
        else:         # compass-like: units 64/view_dim
            vd = rng.choice([9, 11, 13, 20, 30, 40, 64, 25]); unit = 64. / vd; cd = rng.choice([2, 3, 1.5])
            x = (vd - cd - .25) * unit; y = .25 * unit; w = cd * unit; h = cd * unit
        cases.append((x, y, w, h))
    if golden:
        rects = np.array(cases, np.float64)
        pix = np.stack([np.stack([qt_ellipse(x, y, w, h, True, True), qt_ellipse(x, y, w, h, False, True), qt_ellipse(x, y, w, h, True, False)]) for (x, y, w, h) in cases])
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "qt_path_ellipses.npz")
        np.savez_compressed(out, rects=rects, both=np.packbits(pix[:, 0] == 2, axis=-1), both_brush=np.packbits(pix[:, 0] == 1, axis=-1),
                            brush_only=np.packbits(pix[:, 1] == 1, axis=-1), pen_only=np.packbits(pix[:, 2] == 2, axis=-1))
        print("wrote", out, rects.shape)
        return
    for label, pen, brush in (("fill only", False, True), ("pen only", True, False), ("pen + brush", True, True)):
        bad = 0; shown = 0
        for (x, y, w, h) in cases:
            q = qt_ellipse(x, y, w, h, pen, brush); m = model_ellipse(x, y, w, h, pen, brush)
            if not np.array_equal(q, m):
                bad += 1
                if shown < 3:
                    shown += 1
                    print("MISMATCH", label, repr((x, y, w, h)))
                    ys, xs = np.nonzero(q != m)
                    y0 = max(0, ys.min() - 2); y1 = min(H, ys.max() + 3); x0 = max(0, xs.min() - 6); x1 = min(W, xs.max() + 7)
                    for yy in range(y0, y1):
                        print(''.join('.+#'[q[yy, xx]] for xx in range(x0, x1)), '  ', ''.join('.+#'[m[yy, xx]] for xx in range(x0, x1)))
        print("%-12s mismatches: %d of %d" % (label, bad, len(cases)))


if __name__ == "__main__":
    main()

"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7, the Qt the reference oracle links): pins the
unrotated, non-antialiased QPainter::drawImage(QRectF, QImage) sampling rule against two candidate models on
random and knife-edge rects.  Result on this image: ceil-1 matches 3000/3000, floor+1 (newer Qt) fails 3.
"""
import os, sys, math
os.environ["QT_QPA_PLATFORM"]="offscreen"
from PyQt5.QtGui import QImage, QPainter, QGuiApplication, QColor
from PyQt5.QtCore import QRectF
import numpy as np
app = QGuiApplication(sys.argv[:1])
def mk(sw,sh):
    xs = np.arange(sw, dtype=np.uint32)[None,:]; ys = np.arange(sh, dtype=np.uint32)[:,None]
    src = (0xff000000 | ((xs&0xff) << 16) | ((ys & 0xff) << 8) | ((xs>>8)<<4) | (ys >> 8)).astype(np.uint32)
    buf = src.tobytes()
    return src, buf, QImage(buf, sw, sh, sw*4, QImage.Format_ARGB32_Premultiplied)
def draw(img, L,T,W,H):
    dst = QImage(64, 64, QImage.Format_RGB32)
    dst.fill(QColor(0,0,255))
    p = QPainter(dst); p.drawImage(QRectF(L,T,W,H), img); p.end()
    ptr = dst.constBits(); ptr.setsize(64*64*4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(64,64).copy()
def qround(d): return int(d+0.5) if d>=0 else int(d-float(int(d-1))+0.5)+int(d-1)
def model(src, L,T,W,H, mode):
    sh, sw = src.shape
    out = np.full((64,64), 0xff0000ff, np.uint32)
    ix = int(65536/(W/sw)); iy=int(65536/(H/sh))
    tx1=max(qround(L),0); tx2=min(qround(L+W),64); ty1=max(qround(T),0); ty2=min(qround(T+H),64)
    w=tx2-tx1; h=ty2-ty1
    if w<=0 or h<=0: return out
    px=(tx1+0.5-L)*ix; py=(ty1+0.5-T)*iy
    if mode=='floor+1': bx=math.floor(px)+1; by=math.floor(py)+1
    else: bx=math.ceil(px)-1; by=math.ceil(py)-1
    bx&=0xffffffff; by&=0xffffffff
    yend=((by+iy*(h-1))&0xffffffff)>>16
    if yend>=sh: h-=1
    xend=((bx+ix*(w-1))&0xffffffff)>>16
    if xend>=sw: w-=1
    for y in range(h):
        sy=((by+y*iy)&0xffffffff)>>16
        for x in range(w):
            sx=((bx+x*ix)&0xffffffff)>>16
            out[ty1+y,tx1+x]=src[sy,sx]
    return out
rng=np.random.RandomState(1)
bad={'floor+1':0,'ceil-1':0}; n=0
for trial in range(3000):
    sw=int(rng.choice([17,60,64,100,128,256])); sh=int(rng.choice([17,64,128,256,200]))
    src,buf,img=mk(sw,sh)
    if trial%2==0:
        # knife-edge: unit such that sw/W integer-ish
        unit=np.float32(64)/np.float32(rng.choice([13,16,10,20,8,12]))
        cx=np.float32(rng.uniform(0,20)); 
        L=float(np.float32(np.float32(rng.randint(0,20))*unit - unit*(cx-np.float32(6.5))))
        T=float(np.float32(np.float32(rng.randint(0,20))*unit - unit*(np.float32(rng.uniform(0,20))-np.float32(6.5))))
        W=float(unit*np.float32(rng.choice([1,2,0.5,1.04]))); H=float(unit*np.float32(rng.choice([1,2,1.1574,1.04])))
    else:
        L=float(np.float32(rng.uniform(-30,60))); T=float(np.float32(rng.uniform(-30,60))); W=float(np.float32(rng.uniform(1,80))); H=float(np.float32(rng.uniform(1,80)))
    q=draw(img,L,T,W,H)
    n+=1
    for m in bad:
        if not np.array_equal(q, model(src,L,T,W,H,m)): bad[m]+=1
print(n,bad)

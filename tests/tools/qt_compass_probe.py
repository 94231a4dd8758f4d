"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7): pins the two Qt raster primitives of the jumper
compass (reference src/games/jumper.cpp:134-169) for the non-antialiased engine:
  * QPainter::drawLine(int,int,int,int) with a width-0 pen -> QCosmeticStroker::drawLine (26.6 end points, 16.16
    minor-axis walker, half-pixel square caps): every end point offset in [-9,9]^2 matches, a point draws one pixel;
  * drawEllipse on an integer-aligned rect -> drawEllipse_midpoint_i / drawEllipsePoints, with pen+brush and with
    NoPen + a translucent brush: 1500 random integer rects match (fractional rects take the path-based
    QPaintEngineEx route, which is NOT restated: the oracle and the HIP stepper refuse such configurations).
"""
import os, sys, math
os.environ["QT_QPA_PLATFORM"]="offscreen"
from PyQt5.QtGui import QImage, QPainter, QGuiApplication, QColor, QPen, QBrush
from PyQt5.QtCore import QRectF, QRect, Qt
import numpy as np
app = QGuiApplication(sys.argv[:1])
def canvas():
    img = QImage(64,64,QImage.Format_RGB32); img.fill(QColor(0,0,0)); return img
def arr(img):
    ptr=img.constBits(); ptr.setsize(64*64*4); return np.frombuffer(bytes(ptr),np.uint32).reshape(64,64)&0xffffff
def line(x1,y1,x2,y2):
    img=canvas(); p=QPainter(img)
    p.setBrush(QBrush(QColor(252,186,3))); p.setPen(QPen(QColor(252,186,3),0))
    p.drawLine(int(x1),int(y1),int(x2),int(y2)); p.end()
    return arr(img)!=0
def ellipse(x,y,w,h):
    img=canvas(); p=QPainter(img)
    p.setBrush(QBrush(QColor(168,166,158))); p.setPen(QPen(QColor(168,166,158),1))
    p.drawEllipse(QRectF(x,y,w,h)); p.end()
    return arr(img)!=0
print("=== cosmetic line model")
def tdiv(a,b):
    q=abs(a)//abs(b); return q if (a>=0)==(b>0) else -q
def cosmetic(X1,Y1,X2,Y2):
    m=np.zeros((64,64),bool)
    x1=X1*64; y1=Y1*64; x2=X2*64; y2=Y2*64
    dx=abs(x2-x1); dy=abs(y2-y1)
    def put(x,y):
        if 0<=x<64 and 0<=y<64: m[y,x]=True
    if dx<dy:
        if y1>y2: x1,x2=x2,x1; y1,y2=y2,y1
        xinc=tdiv((x2-x1)<<16, y2-y1)
        x=x1<<10
        y1-=32; x-=xinc>>1; y2+=32
        y=(y1+32)>>6; ys=(y2+32)>>6; rnd=32 if xinc>0 else 0
        if y!=ys:
            x+=(((y<<6)+rnd-y1)*xinc)>>6
            while True:
                put(x>>16,y); x+=xinc; y+=1
                if not y<ys: break
    else:
        if dx==0: return m
        if x1>x2: x1,x2=x2,x1; y1,y2=y2,y1
        yinc=tdiv((y2-y1)<<16, x2-x1)
        y=y1<<10
        x1-=32; y-=yinc>>1; x2+=32
        x=(x1+32)>>6; xs=(x2+32)>>6; rnd=32 if yinc>0 else 0
        if x!=xs:
            y+=(((x<<6)+rnd-x1)*yinc)>>6
            while True:
                put(x,y>>16); y+=yinc; x+=1
                if not x<xs: break
    return m
bad=0; shown=0
for dx in range(-9,10):
    for dy in range(-9,10):
        q=line(30,30,30+dx,30+dy); mm=cosmetic(30,30,30+dx,30+dy)
        if not np.array_equal(q,mm):
            bad+=1
            if shown<4:
                shown+=1; print("dx,dy",dx,dy)
                for y in range(20,41): print(''.join('#' if q[y,x] else '.' for x in range(20,41)),'  ',''.join('#' if mm[y,x] else '.' for x in range(20,41)))
print("cosmetic model mismatches:",bad,"of",19*19)

# ---- ellipse ----
def canvas():
    img = QImage(64,64,QImage.Format_RGB32); img.fill(QColor(10,20,30)); return img
def arr(img):
    ptr=img.constBits(); ptr.setsize(64*64*4); return np.frombuffer(bytes(ptr),np.uint32).reshape(64,64).copy()
def q_ellipse(x,y,w,h,mode):
    img=canvas(); p=QPainter(img)
    if mode=='penbrush':
        p.setBrush(QBrush(QColor(168,166,158))); p.setPen(QPen(QColor(168,166,158),1)); p.drawEllipse(QRectF(x,y,w,h))
    else:
        p.setBrush(QColor(255,255,255,120)); p.setPen(Qt.NoPen); p.drawEllipse(QRect(int(x),int(y),int(w),int(h)))
    p.end(); return arr(img)
def byte_mul(x,a):
    t=(x&0xff00ff)*a; t=(t+((t>>8)&0xff00ff)+0x800080)>>8; t&=0xff00ff
    x=((x>>8)&0xff00ff)*a; x=(x+((x>>8)&0xff00ff)+0x800080); x&=0xff00ff00
    return (x|t)&0xffffffff
def model(x,y,w,h,mode):
    out=np.full((64,64),0xff0a141e,np.uint32)
    if mode=='penbrush':
        rx=int(x); ry=int(y); rw=int(x+w)-int(x) if False else math.ceil(x+w)-math.floor(x); rh=math.ceil(y+h)-math.floor(y)
        rw=int(math.ceil(x+w))-int(x); rh=int(math.ceil(y+h))-int(y)
    else:
        rx=int(x); ry=int(y); rw=int(w); rh=int(h)
    if rw<=0 or rh<=0: return out
    pen = mode=='penbrush'
    def span(sx,sy,ln,kind):
        for X in range(sx,sx+ln):
            if 0<=X<64 and 0<=sy<64:
                if mode=='penbrush': out[sy,X]=0xffa8a69e
                else:
                    s=0x78787878  # premultiplied (255,255,255,120)
                    d=int(out[sy,X]); out[sy,X]=(s+byte_mul(d,255-120))&0xffffffff
    def points(px,py,length):
        if length==0: return
        midx=rx+(rw+1)//2; midy=ry+(rh+1)//2
        X=px+midx; Y=midy-py
        o0x=midx+midx-X-(length-1)-(rw&1); o0len=min(length,X-o0x); o0y=Y
        o1x=X; o1len=length; o1y=Y
        o2x=o0x; o2len=o0len; o2y=midy+midy-Y-(rh&1)
        o3x=X; o3len=length; o3y=o2y
        if o0x+o0len<o1x:
            f0x=o0x+o0len-1; f0len=max(0,o1x-f0x); f0y=o1y
            f1x=o2x+o2len-1; f1len=max(0,o3x-f1x); f1y=o3y
            n=1 if f0y>=f1y else 2
            span(f0x,f0y,f0len,'f')
            if n==2: span(f1x,f1y,f1len,'f')
        if pen:
            n=2 if o1y>=o2y else 4
            span(o0x,o0y,o0len,'o'); span(o1x,o1y,o1len,'o')
            if n==4: span(o2x,o2y,o2len,'o'); span(o3x,o3y,o3len,'o')
    a=rw/2.0; b=rh/2.0
    d=b*b-(a*a*b)+0.25*a*a
    X=0; Y=(rh+1)//2; startx=X
    while a*a*(2*Y-1) > 2*b*b*(X+1):
        if d<0:
            d+=b*b*(2*X+3); X+=1
        else:
            d+=b*b*(2*X+3)+a*a*(-2*Y+2)
            points(startx,Y,X-startx+1)
            X+=1; startx=X; Y-=1
    points(startx,Y,X-startx+1)
    d=b*b*(X+0.5)*(X+0.5)+a*a*((Y-1)*(Y-1)-b*b)
    miny=rh&1
    while Y>miny:
        if d<0:
            d+=b*b*(2*X+2)+a*a*(-2*Y+3); X+=1
        else:
            d+=a*a*(-2*Y+3)
        Y-=1
        points(X,Y,1)
    return out
rng=np.random.RandomState(1)
for mode in ('penbrush','nopen'):
    bad=0; n=0; ex=[]
    for t in range(1500):
        if False:
            x=float(np.float32(rng.uniform(0,50))); y=float(np.float32(rng.uniform(0,50))); w=float(np.float32(rng.uniform(0.5,20))); h=float(np.float32(rng.uniform(0.5,20)))
        else:
            x=float(rng.randint(0,50)); y=float(rng.randint(0,50)); w=float(rng.randint(1,20)); h=float(rng.randint(1,20))
        q=q_ellipse(x,y,w,h,mode); mo=model(x,y,w,h,mode); n+=1
        if not np.array_equal(q,mo):
            bad+=1
            if len(ex)<3: ex.append((x,y,w,h,int((q!=mo).sum())))
    print(mode,n,bad,ex)

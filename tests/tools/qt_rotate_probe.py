"""
TEST TOOL (run with /opt/conda/bin/python3.9 = PyQt5 5.9.7, the Qt the reference oracle links): pins the rule of
the reference's rotated sprite draw (reference src/basic-abstract-game.cpp:902-906: translate, rotate, drawImage)
for the non-antialiased raster engine:
  * rotation with a non-null sine -> qt_transform_image (inverse 16.16 mapping, u0/v0 = qCeil(..)-1, three
    trapezoids with 16.16 edge walkers, source coordinates clamped);
  * 180 degrees (sine fuzzy-null) -> the scale path with a negative scale (qFloor(..)+1 from the far edge).
Usage: qt_rotate_probe.py <seed> <trials>.  Result on this image: 0 mismatches in several thousand trials.
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
def draw(img, x,y,w,h, deg):
    dst = QImage(64, 64, QImage.Format_RGB32)
    dst.fill(QColor(0,0,255))
    p = QPainter(dst)
    p.translate(x + w/2, y + h/2); p.rotate(deg)
    p.drawImage(QRectF(-w/2, -h/2, w, h), img); p.end()
    ptr = dst.constBits(); ptr.setsize(64*64*4)
    return np.frombuffer(bytes(ptr), np.uint32).reshape(64,64).copy()
def qround(d): return int(d+0.5) if d>=0 else int(d-float(int(d-1))+0.5)+int(d-1)
def cint(d): return int(d)  # trunc toward zero
def fuzzy_null(d): return abs(d) <= 0.000000000001
def model_scale(src,x,y,w,h,c):
    sh,sw=src.shape
    out=np.full((64,64),0xff0000ff,np.uint32)
    cx=x+w/2; cy=y+h/2
    x1=c*(-w/2)+cx; y1=c*(-h/2)+cy; x2=c*(-w/2+w)+cx; y2=c*(-h/2+h)+cy
    L,T,W,H=x1,y1,x2-x1,y2-y1
    R=L+W; B=T+H
    sx=W/sw; sy=H/sh
    ix=int(65536/sx); iy=int(65536/sy)
    tx1=qround(L); tx2=qround(R); ty1=qround(T); ty2=qround(B)
    if tx2<tx1: tx1,tx2=tx2,tx1
    if ty2<ty1: ty1,ty2=ty2,ty1
    tx1=max(tx1,0); ty1=max(ty1,0); tx2=min(tx2,64); ty2=min(ty2,64)
    ww=tx2-tx1; hh=ty2-ty1
    if ww<=0 or hh<=0: return out
    M=1<<32
    basex=(sw*65536 + math.floor((tx1+0.5-R)*ix)+1)%M if sx<0 else (math.ceil((tx1+0.5-L)*ix)-1)%M
    srcy=(sh*65536 + math.floor((ty1+0.5-B)*iy)+1)%M if sy<0 else (math.ceil((ty1+0.5-T)*iy)-1)%M
    if (srcy>>16)>=sh and iy<0: srcy=(srcy+iy)%M; hh-=1
    if (basex>>16)>=sw and ix<0: basex=(basex+ix)%M; ww-=1
    if (((srcy+iy*(hh-1))%M)>>16)>=sh: hh-=1
    if (((basex+ix*(ww-1))%M)>>16)>=sw: ww-=1
    for yy in range(hh):
        for xx in range(ww):
            out[ty1+yy,tx1+xx]=src[((srcy+iy*yy)%M)>>16,((basex+ix*xx)%M)>>16]
    return out
def model(src, x,y,w,h,deg):
    sh,sw=src.shape
    out=np.full((64,64),0xff0000ff,np.uint32)
    cx=x+w/2; cy=y+h/2
    if deg==0: return None
    if deg==90 or deg==-270: s,c=1.0,0.0
    elif deg==270 or deg==-90: s,c=-1.0,0.0
    elif deg==180: s,c=0.0,-1.0
    else:
        b=0.017453292519943295769*deg; s=math.sin(b); c=math.cos(b)
    if fuzzy_null(s):
        return model_scale(src,x,y,w,h,c)
    def mp(px,py): return (c*px + (-s)*py + cx, s*px + c*py + cy)
    L,T=-w/2,-h/2; R=L+w; B=T+h
    V=[[*mp(L,T),0.0,0.0],[*mp(R,T),float(sw),0.0],[*mp(R,B),float(sw),float(sh)],[*mp(L,B),0.0,float(sh)]]
    top=0
    for i in range(1,4):
        if V[i][1]<V[top][1]: top=i
    V=V[top:]+V[:top]
    dx1=V[1][0]-V[0][0]; dy1=V[1][1]-V[0][1]; dx2=V[3][0]-V[0][0]; dy2=V[3][1]-V[0][1]
    if dx1*dy2-dx2*dy1>0: V[1],V[3]=V[3],V[1]
    u=[V[1][k]-V[0][k] for k in range(4)]; ww=[V[2][k]-V[0][k] for k in range(4)]
    det=u[0]*ww[1]-u[1]*ww[0]
    if det==0: return out
    inv=1.0/det
    m11=(u[2]*ww[1]-u[1]*ww[2])*inv; m12=(u[0]*ww[2]-u[2]*ww[0])*inv
    m21=(u[3]*ww[1]-u[1]*ww[3])*inv; m22=(u[0]*ww[3]-u[3]*ww[0])*inv
    mdx=V[0][2]-m11*V[0][0]-m12*V[0][1]; mdy=V[0][3]-m21*V[0][0]-m22*V[0][1]
    dudx=cint(m11*65536); dvdx=cint(m21*65536); dudy=cint(m12*65536); dvdy=cint(m22*65536)
    u0=math.ceil((0.5*m11+0.5*m12+mdx)*65536)-1; v0=math.ceil((0.5*m21+0.5*m22+mdy)*65536)-1
    def rast(tl,bl,tr,br,topY,botY):
        fromY=max(qround(topY),0); toY=min(qround(botY),64)
        if fromY>=toY: return
        ls=(bl[0]-tl[0])/(bl[1]-tl[1]); rs=(br[0]-tr[0])/(br[1]-tr[1])
        dxl=cint(ls*65536); dxr=cint(rs*65536)
        xl=cint((tl[0]+(0.5+fromY-tl[1])*ls+0.5)*65536); xr=cint((tr[0]+(0.5+fromY-tr[1])*rs+0.5)*65536)
        for Y in range(fromY,toY):
            fx=max(xl>>16,0); tx=min(xr>>16,64)
            for X in range(fx,tx):
                uu=(u0+X*dudx+Y*dudy)>>16; vv=(v0+X*dvdx+Y*dvdy)>>16
                uu=min(max(uu,0),sw-1); vv=min(max(vv,0),sh-1)
                out[Y,X]=src[vv,uu]
            xl+=dxl; xr+=dxr
    if V[1][1]<V[3][1]:
        rast(V[0],V[1],V[0],V[3],V[0][1],V[1][1])
        rast(V[1],V[2],V[0],V[3],V[1][1],V[3][1])
        rast(V[1],V[2],V[3],V[2],V[3][1],V[2][1])
    else:
        rast(V[0],V[1],V[0],V[3],V[0][1],V[3][1])
        rast(V[0],V[1],V[3],V[2],V[3][1],V[1][1])
        rast(V[1],V[2],V[3],V[2],V[1][1],V[2][1])
    return out
if __name__=="__main__":
    rng=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
    n=0; bad=0; ex=[]
    for trial in range(int(sys.argv[2]) if len(sys.argv)>2 else 500):
        sw=int(rng.choice([17,64,100,128])); sh=int(rng.choice([17,64,90,128]))
        src,buf,img=mk(sw,sh)
        if trial%3==0:
            unit=np.float32(64)/np.float32(rng.choice([13,16,10,20,8,12,15]))
            x=float(np.float32(np.float32(rng.randint(0,12))*unit*np.float32(rng.choice([1,0.5]))+np.float32(rng.choice([0,0.5,0.25]))))
            y=float(np.float32(np.float32(rng.randint(0,12))*unit+np.float32(rng.choice([0,0.5]))))
            w=float(unit*np.float32(rng.choice([1,2,0.8,1.5]))); h=float(unit*np.float32(rng.choice([1,2,0.8,3])))
        else:
            x=float(np.float32(rng.uniform(-10,50))); y=float(np.float32(rng.uniform(-10,50))); w=float(np.float32(rng.uniform(2,30))); h=float(np.float32(rng.uniform(2,30)))
        k=trial%4
        if k==0: deg=float(rng.choice([45,-45,135,-135,90,-90,180,-180,360,30,60]))
        elif k==1: deg=float(np.float32(np.float32(rng.uniform(-7,7))*np.float32(180)/np.float32(np.pi)))
        elif k==2: deg=float(np.float32(np.float32(rng.uniform(-1500,1500))*np.float32(180)/np.float32(np.pi)))
        else: deg=float(np.float32(np.float32(rng.choice([np.pi/2,-np.pi/2,np.pi,1e-7,np.pi+1e-6]))*np.float32(180)/np.float32(np.pi)))
        q=draw(img,x,y,w,h,deg)
        mo=model(src,x,y,w,h,deg)
        if mo is None: continue
        n+=1
        if not np.array_equal(q,mo):
            bad+=1
            if len(ex)<6:
                d=np.argwhere(q!=mo); ex.append((sw,sh,x,y,w,h,deg,len(d),d[:3].tolist(), [hex(int(q[a,b])) for a,b in d[:3]], [hex(int(mo[a,b])) for a,b in d[:3]]))
    print(n,bad)
    for e in ex: print(e)

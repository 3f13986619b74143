import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
x=np.linspace(0,7,20001)
Phi=0.5*(1+erf(x/np.sqrt(2)))
def sig(z): return 1/(1+np.exp(-z))
def model(c,x):
    x2=x*x
    p=c[-1]
    for a in c[-2::-1]: p=p*x2+a
    return sig(x*p)
for deg in (2,3,4):
    c0=[1.5976,0.07056]+[0.0]*(deg-1)
    c0=c0[:deg+1] if len(c0)>deg+1 else c0
    # minimax-ish via high-power norm on gelu error weight
    f=lambda c: ((model(c,x)-Phi))*1e4
    r=least_squares(f,c0[:deg+1],method='lm')
    c=r.x
    for it in range(30):   # iteratively reweighted to approximate minimax
        e=np.abs(model(c,x)-Phi); w=(e/e.max())**2+0.05
        r=least_squares(lambda c: (model(c,x)-Phi)*w*1e4,c,method='lm'); c=r.x
    xx=np.linspace(-8,8,64001)
    ph=model(c,np.abs(xx)); ph=np.where(xx<0,1-ph,ph)
    g=xx*ph; gt=xx*0.5*(1+erf(xx/np.sqrt(2)))
    # derivative
    x2=xx*xx
    print(deg,c, "max|Phi err|",np.abs(model(c,x)-Phi).max(),"max|gelu err|",np.abs(g-gt).max())
print("---- derivative check for deg 2")
c=np.array([1.59499531e+00, 7.40885562e-02, -7.23764583e-04])
xx=np.linspace(-9,9,72001); xc=np.clip(xx,-7,7); x2=xc*xc
p=(c[2]*x2+c[1])*x2+c[0]; s=sig(xc*p)
g=xx*s; gt=xx*0.5*(1+erf(xx/np.sqrt(2)))
q=(5*c[2]*x2+3*c[1])*x2+c[0]
d=s+xc*s*(1-s)*q
dt=0.5*(1+erf(xx/np.sqrt(2)))+xx*np.exp(-0.5*xx*xx)/np.sqrt(2*np.pi)
print("gelu err",np.abs(g-gt).max(),"at",xx[np.abs(g-gt).argmax()],"grad err",np.abs(d-dt).max(),"at",xx[np.abs(d-dt).argmax()])
# float32 evaluation error
import numpy as np
x32=xx.astype(np.float32); xc32=np.clip(x32,-7,7); x232=xc32*xc32
L2E=np.float32(1.4426950408889634)
p32=((np.float32(c[2])*x232+np.float32(c[1]))*x232+np.float32(c[0]))
s32=1/(1+np.exp2(-(xc32*p32)*L2E))
print("f32 gelu err",np.abs(x32*s32-gt).max())

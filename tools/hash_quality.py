import numpy as np
M=0xffffffff
def mul24(a,b): return ((a&0xffffff).astype(np.uint64)*(np.uint64(b&0xffffff)))&np.uint64(M)
def rot(x,r): return ((x>>np.uint64(r))|(x<<np.uint64(32-r)))&np.uint64(M)
def h4(idx,s0,s1):
    x=(idx+s0)&M; x=x.astype(np.uint64); x^=x>>np.uint64(15)
    a=mul24(x,0x9E3779); b=mul24(rot(x,9),0x85EBCB)
    x=((a^((b<<np.uint64(7))&np.uint64(M)))+np.uint64(s1))&np.uint64(M); x^=x>>np.uint64(13)
    a=mul24(x,0xC2B2AF); b=mul24(rot(x,10),0x27D4EB)
    x=(a+((b<<np.uint64(9))&np.uint64(M))+(x>>np.uint64(3)))&np.uint64(M); x^=x>>np.uint64(16); return x
def h2(idx,s0,s1):
    x=(idx+s0)&M; x=x.astype(np.uint64); x^=x>>np.uint64(15)
    a=(mul24(x,0x9E3779)+np.uint64(s1))&np.uint64(M); x=a^(a>>np.uint64(13))
    b=(mul24(x,0xC2B2AF)+rot(x,7))&np.uint64(M)
    return b^(b>>np.uint64(16))
def h3(idx,s0,s1):
    x=(idx+s0)&M; x=x.astype(np.uint64); x^=x>>np.uint64(15)
    a=(mul24(x,0x9E3779)+np.uint64(s1))&np.uint64(M); x=a^(a>>np.uint64(13))
    b=(mul24(x,0xC2B2AF)+rot(x,7))&np.uint64(M); x=b^(b>>np.uint64(11))
    c=(mul24(x,0x85EBCB)+(x>>np.uint64(5)))&np.uint64(M)
    return c^(c>>np.uint64(16))
def stats(h,name):
    n=1<<22
    for s0,s1 in [(0x12345678,0x9abcdef0),(1,2),(0xdeadbeef,0)]:
        idx=np.arange(n,dtype=np.uint64)+np.uint64(98304*128)
        v=h(idx,s0,s1); lo=(v&np.uint64(0xffff)).astype(np.float64); hi=(v>>np.uint64(16)).astype(np.float64)
        thr=6554
        klo=(lo>=thr).astype(np.float64); khi=(hi>=thr).astype(np.float64)
        def corr(a,b): return np.corrcoef(a,b)[0,1]
        v2=h(idx,s0+1,s1); k2=((v2&np.uint64(0xffff))>=thr).astype(np.float64)
        v3=h(idx,s0,s1^1); k3=((v3&np.uint64(0xffff))>=thr).astype(np.float64)
        # chi-square over 256 buckets of top 8 bits
        cnt=np.bincount((lo.astype(np.int64)>>8),minlength=256); chi=((cnt-n/256)**2/(n/256)).sum()
        cnt2=np.bincount((hi.astype(np.int64)>>8),minlength=256); chi2=((cnt2-n/256)**2/(n/256)).sum()
        print(name,"keep",klo.mean().round(5),khi.mean().round(5),"c(lo,hi)",corr(klo,khi).round(5),"adj",corr(klo[1:],klo[:-1]).round(5),"row256",corr(klo[256:],klo[:-256]).round(5),"row768",corr(khi[768:],khi[:-768]).round(5),"s0+1",corr(klo,k2).round(5),"s1^1",corr(klo,k3).round(5),"chi",chi.round(0),chi2.round(0))
stats(h4,"h4");stats(h2,"h2");stats(h3,"h3")

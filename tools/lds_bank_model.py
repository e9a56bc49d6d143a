import itertools
GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
        [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59],[36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def cycles(addr):  # addr(lane)->byte address of 16B read; banks 64x4B
    tot=0
    for g in GROUPS:
        bank={}
        for l in g:
            a=addr(l)
            for k in range(4):
                b=((a//4)+k)%64
                bank.setdefault(b,set()).add(a+4*k)
        tot+=max(len(v) for v in bank.values())
    return tot
def act(stride, LW=10, xt=0, swz=None):
    def f(l):
        lr=l&15; lq=l>>4
        lh=2*xt+(lr>>3); lw=lr&7
        v=lh*LW+lw
        c=lq if swz is None else swz(v,lq)
        return v*stride+c*16
    return f
def wgt(stride, swz=None):
    def f(l):
        lr=l&15; lq=l>>4
        c=lq if swz is None else swz(lr,lq)
        return lr*stride+c*16
    return f
print("ideal = 4 cycles per ds_read_b128")
for st in (64,80,96,112,144):
    print("act stride",st,[cycles(act(st,LW,xt)) for LW in (10,9,8) for xt in (0,1)], "wgt",cycles(wgt(st)))
# xor swizzles on 64B stride
for name,sw in (("v>>2&3",lambda v,c:c^((v>>2)&3)),("v&3",lambda v,c:c^(v&3)),("(v>>1)&3",lambda v,c:c^((v>>1)&3)),("(v^(v>>2))&3",lambda v,c:c^((v^(v>>2))&3))):
    print("swz64",name,[cycles(act(64,LW,xt,sw)) for LW in (10,9,8) for xt in (0,1,2,3)],"wgt",cycles(wgt(64,sw)))
print("---- search")
def act2(stride, LWp, xt, swz, w0=1):
    def f(l):
        lr=l&15; lq=l>>4
        lh=2*xt+(lr>>3)+1; lw=(lr&7)+w0
        v=lh*LWp+lw
        c=swz(v,lq)
        return v*stride+c*16
    return f
swzs={"none":lambda v,c:c,"v>>1":lambda v,c:c^((v>>1)&3),"v>>2":lambda v,c:c^((v>>2)&3),"v":lambda v,c:c^(v&3),"v>>3":lambda v,c:c^((v>>3)&3),
      "v+v>>3":lambda v,c:(c+v+(v>>3))&3,"(v>>1)+(v>>3)":lambda v,c:c^(((v>>1)+(v>>3))&3), "v>>1^v>>4":lambda v,c:c^(((v>>1)^(v>>4))&3)}
best=[]
for st in (64,80,96,112,128):
  for LWp in (8,9,10,11,12,13,14,16):
    for name,sw in swzs.items():
        worst=max(cycles(act2(st,LWp,xt,sw,w0)) for xt in range(4) for w0 in (0,1,2))
        best.append((worst,st,LWp,name))
best.sort()
for b in best[:15]: print(b)
print("---- search2: pitch P, swizzle s(lh,lw); cs = op(lq, s)")
def act3(P, xt, s, op, w0, h0):
    def f(l):
        lr=l&15; lq=l>>4
        lh=2*xt+(lr>>3)+h0; lw=(lr&7)+w0
        v=lh*P+lw
        return v*64+op(lq,s(lh,lw))*16
    return f
f1s={"0":lambda w:0,"w":lambda w:w,"w>>1":lambda w:w>>1,"w>>2":lambda w:w>>2}
f2s={"0":lambda h:0,"h":lambda h:h,"2h":lambda h:2*h,"3h":lambda h:3*h,"h>>1":lambda h:h>>1}
ops={"xor":lambda c,s:c^(s&3),"add":lambda c,s:(c+s)&3}
res=[]
for P in (10,9,11,12):
  for n1,f1 in f1s.items():
    for n2,f2 in f2s.items():
      for comb in ("+","^"):
        s=(lambda f1,f2:(lambda h,w:(f1(w)+f2(h)) if comb=="+" else (f1(w)^f2(h))))(f1,f2)
        for on,op in ops.items():
            worst=max(cycles(act3(P,xt,s,op,w0,h0)) for xt in range(4) for w0 in (0,1,2) for h0 in (0,1,2))
            res.append((worst,P,n1,comb,n2,on))
res.sort()
for r in res[:12]: print(r)
print("---- gemm rows (BK=64 -> 128 B per row), fragment chunk = ks*4 + lq")
def gm(RS, sw):
    def mk(ks):
        def f(l):
            lr=l&15; lq=l>>4
            c=ks*4+lq
            return lr*RS+sw(lr,c)*16
        return f
    return max(cycles(mk(0)),cycles(mk(1)))
for name,RS,sw in (("pad144",144,lambda r,c:c),("pad160",160,lambda r,c:c),("128 c^(r&7)",128,lambda r,c:c^(r&7)),("128 c^((r>>1)&7)",128,lambda r,c:c^((r>>1)&7)),
                   ("128 c^(r>>1&3)",128,lambda r,c:c^((r>>1)&3)),("128 c^((r&1)<<2|(r>>1&3))",128,lambda r,c:c^(((r&1)<<2)|((r>>1)&3)))):
    print(name, gm(RS,sw))
print("---- convT fused: 16 lanes = 4(h) x 4(w) voxels of a 5x5 plane (pitch PW), tap offsets h0,w0 in {0,1}")
def act4(PW, s, op, h0, w0, d0=0, PHW=None):
    def f(l):
        lr=l&15; lq=l>>4
        h=(lr>>2)+h0; w=(lr&3)+w0
        v=(d0*(PHW or 5*PW))+h*PW+w
        return v*64+op(lq,s(h,w,v))*16
    return f
cands={"0":lambda h,w,v:0,"h":lambda h,w,v:h,"2h":lambda h,w,v:2*h,"w":lambda h,w,v:w,"v":lambda h,w,v:v,"v>>1":lambda h,w,v:v>>1,"v>>2":lambda h,w,v:v>>2,
       "h+w":lambda h,w,v:h+w,"h^(w>>1)":lambda h,w,v:h^(w>>1),"(h>>1)":lambda h,w,v:h>>1,"h+(w>>1)":lambda h,w,v:h+(w>>1),"2h+(w>>1)":lambda h,w,v:2*h+(w>>1)}
res=[]
for PW in (5,6,7,8):
    for n,sf in cands.items():
        for on,op in ops.items():
            worst=max(cycles(act4(PW,sf,op,h0,w0,d0,PHW)) for h0 in (0,1) for w0 in (0,1) for d0 in (0,1,2) for PHW in ((5*PW),))
            res.append((worst,PW,n,on))
res.sort()
for r in res[:10]: print(r)

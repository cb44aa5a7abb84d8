"""CPU model of image_build.hip: ChainRun -- the linking of a heavy overflow run's records into chains (device_common.hpp) -- with an exhaustive check
that every record is found through exactly one chain and that near misses are not found.  python tools/chain_sim.py"""
import random, bisect
random.seed(5)
def sim(nrel, div, extra_random=0):
    L=48
    root=[random.randrange(4) for _ in range(L)]
    recs=set()
    for g in range(nrel):
        s=list(root)
        for i in list(range(16))+list(range(32,48)):
            if random.random()<div: s[i]=random.randrange(4)
        A=s[:16]; B=s[32:]
        a=0
        for x in A: a=a*4+x
        b=0
        for x in B: b=b*4+x
        for p in range(17):
            if random.random()<0.9:
                rest=(((a<<32)|b)>>(2*p))&0xFFFFFFFF
                recs.add((p,rest))
    for _ in range(extra_random):
        recs.add((random.randrange(17), random.getrandbits(32)))
    recs=sorted(recs)
    n=len(recs)
    g=[0]*18
    p=0
    for q,(pq,_) in enumerate(recs):
        while p<=pq and p<18: g[p]=q; p+=1
    while p<18: g[p]=n; p+=1
    rest=lambda i: recs[i][1]
    pos=lambda i: recs[i][0]
    def lb(p,v):
        lo,hi=g[p],g[p+1]
        while lo<hi:
            mid=(lo+hi)>>1
            if rest(mid)<v: lo=mid+1
            else: hi=mid
        return lo
    def pred(i,p):
        if p==0: return None
        ri=rest(i); low30=ri&0x3FFFFFFF; x=ri>>30; t=0
        for y in range(x):
            v=(y<<30)|low30; a=lb(p,v)
            if a<g[p+1] and rest(a)==v: t+=1
        a=lb(p-1,(low30<<2)&0xFFFFFFFF)
        return a+t if (a+t<g[p] and (rest(a+t)>>2)==low30) else None
    def succ(j,p):
        if p>=16: return None
        low30=rest(j)>>2
        t=j-lb(p,(low30<<2))
        for y in range(4):
            v=(y<<30)|low30; a=lb(p+1,v)
            if a<g[p+2] and rest(a)==v:
                if t==0: return a
                t-=1
        return None
    chains=[]; ranks=[]; covered=[0]*n
    for i in range(n):
        p0=pos(i)
        if pred(i,p0) is not None: continue
        cur=i; p=p0; first=len(ranks)
        while True:
            ranks.append(cur); covered[cur]+=1
            nx=succ(cur,p)
            if nx is None: break
            assert pred(nx,p+1)==cur, "asymmetric"
            cur=nx; p+=1
        rl=rest(cur); rf=rest(i)
        A=(rl>>(32-2*p)) if p else 0
        B=((rf<<(2*p0))&0xFFFFFFFF) if p0<16 else 0
        chains.append(((A<<32)|B,p0,p,first))
    assert all(c==1 for c in covered), "coverage"
    def find(pp,rr):
        out=[]
        for (w0,p0,p1,first) in chains:
            if p0<=pp<=p1 and ((w0>>(2*pp))&0xFFFFFFFF)==rr: out.append(first+pp-p0)
        return out
    for i,(pp,rr) in enumerate(recs):
        f=find(pp,rr); assert len(f)==1 and ranks[f[0]]==i, (i,f)
    S=set(recs)
    for _ in range(3000):
        i=random.randrange(n); pp,rr=recs[i]
        rr2=rr^(1<<random.randrange(32))
        if (pp,rr2) not in S: assert find(pp,rr2)==[]
        pp2=random.randrange(17)
        if (pp2,rr) not in S: assert find(pp2,rr)==[]
    return n,len(chains)
for args in [(25,0.015,0),(25,0.03,0),(40,0.1,0),(25,0.03,50),(5,0.3,200),(60,0.02,10)]:
    print(args, [sim(*args) for _ in range(4)])

"""Pure-Python model of the GPU prefix-beam kernel's per-entry formulation
(wenet_amd/csrc/ctc.hip): every key of the reference's `next_hyps` dict gets at
most a blank, a repeat and an extend contribution, so one step is a parallel
evaluation of <= beam + beam^2 entries followed by a stable rank.  Used by
tests/test_prefix_beam_formulation.py to check that decomposition against the
oracle on the CPU (`canonical=False` reproduces the node-identity bug the
sequence hash fixes)."""
import math

NEG=-float('inf')
def log_add2(a,b):
    if a==NEG and b==NEG: return NEG
    m=max(a,b); return m+math.log(math.exp(a-m)+math.exp(b-m))
class H: pass
def emul(logp, T, beam, blank=0, canonical=True, graph=None):
    root=H(); root.key=(); root.node=0; root.last=-1; root.par=None; root.s=0.0; root.ns=NEG; root.vs=0.0; root.vns=0.0
    root.ts=[]; root.tns=[]; root.score=0.0; root.vit=0.0; root.tim=[]
    root.parkey=None
    root.cs=0; root.cx=0.0   # context state / bonus (graph is not None)
    beamh=[root]; nodec=[1]
    for t in range(T):
        lp=logp[t]; tv,ti=lp.topk(beam); tok=ti.tolist(); l=[float(x) for x in tv.tolist()]
        nb=len(beamh); ents=[]
        ident=(lambda h: h.key) if canonical else (lambda h: h.node)
        for r,K in enumerate(beamh):
            qb=-1; ql=-1
            for q in range(beam):
                if tok[q]==blank: qb=q
                if K.last>=0 and tok[q]==K.last: ql=q
            if qb<0 and ql<0: continue
            E=H(); E.s=NEG;E.ns=NEG;E.vs=NEG;E.vns=NEG;E.ts=[];E.tns=[];E.key=K.key;E.node=K.node;E.par=K.par;E.last=K.last;E.parkey=K.parkey; seq=1<<30
            if qb>=0:
                p=l[qb]; E.s=K.score+p; E.vs=K.vit+p; E.ts=K.tim; seq=min(seq,(qb*nb+r)*2)
            if ql>=0:
                p=l[ql]; u=K.last
                rp=-1
                for j,Hj in enumerate(beamh):
                    if (canonical and Hj.key==K.parkey) or ((not canonical) and Hj.node==K.par): rp=j
                xa=K.ns+p; va=K.vns+p; seq=min(seq,(ql*nb+r)*2)
                v=NEG; ctp=NEG; tl=[]
                def rep(lst): 
                    x=list(lst); x[-1]=t; return x
                if rp<0:
                    E.ns=xa
                    if v<va: v=va; tl=rep(K.tns)
                else:
                    P=beamh[rp]
                    if P.last==u: xb=P.s+p; vb=P.vs+p; tb=P.ts; sub=1
                    else: xb=P.score+p; vb=P.vit+p; tb=P.tim; sub=0
                    seq=min(seq,(ql*nb+rp)*2+sub)
                    if r<rp:
                        E.ns=log_add2(xa,xb)
                        if v<va: v=va; ctp=p; tl=rep(K.tns)
                        if v<vb: v=vb; ctp=p; tl=list(tb)+[t]
                    else:
                        E.ns=log_add2(xb,xa)
                        if v<vb: v=vb; ctp=p; tl=list(tb)+[t]
                        if v<va:
                            v=va
                            if ctp<p: ctp=p; tl=rep(K.tns)
                E.vns=v; E.tns=tl
            if graph is not None:
                # first visitor of key K in the reference's loop order decides
                first=1<<30; from_parent=False
                if qb>=0: first=qb*nb+r
                if ql>=0:
                    first=min(first,ql*nb+r)
                    if rp>=0 and ql*nb+rp<first: from_parent=True
                if from_parent:
                    sc,st=graph.forward_one_step(beamh[rp].cs,K.last); E.cx=beamh[rp].cx+sc; E.cs=st
                else:
                    E.cx=K.cx; E.cs=K.cs
            E.seq=seq; ents.append(E)
        for r,P in enumerate(beamh):
            for q in range(beam):
                u=tok[q]
                if u==blank: continue
                ck=P.key+(u,)
                merged=any(((Hj.key==ck) if canonical else (Hj.par==P.node and Hj.last==u)) for Hj in beamh)
                if merged: continue
                p=l[q]
                if u==P.last: x=P.s+p; v=P.vs+p; tb=P.ts; sub=1
                else: x=P.score+p; v=P.vit+p; tb=P.tim; sub=0
                E=H(); E.s=NEG;E.ns=x;E.vs=NEG;E.vns=NEG;E.ts=[];E.tns=[]
                if v>NEG: E.vns=v; E.tns=list(tb)+[t]
                E.key=ck; E.node=None; E.par=P.node; E.last=u; E.parkey=P.key; E.seq=(q*nb+r)*2+sub
                if graph is not None:
                    sc,st=graph.forward_one_step(P.cs,u); E.cx=P.cx+sc; E.cs=st
                ents.append(E)
        for E in ents:
            E.score=log_add2(E.s,E.ns)
            E.total=E.score+E.cx if graph is not None else E.score
        ents.sort(key=lambda E:(-E.total if E.total>NEG else float('inf'),E.seq))
        newb=[]
        for E in ents[:beam]:
            if E.node is None: E.node=nodec[0]; nodec[0]+=1
            E.vit=E.vs if E.vs>E.vns else E.vns
            E.tim=E.ts if E.vs>E.vns else E.tns
            newb.append(E)
        beamh=newb
    if graph is not None:
        return [(h.key,h.score+graph.finalize(h.cs)[0],h.tim) for h in beamh]
    return [(h.key,h.score,h.tim) for h in beamh]

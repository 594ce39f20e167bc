"""Inclusion checks per query of the folded cloth (CPU oracle, depth-first walk per query): the distribution that decides whether
a STREAMING first stage of the narrow phase (64 queries per wave, <= K checks each, survivors handed to the persistent kernel --
VERDICT r03, task 1) can pay.  python tools/checks_histogram.py [cloth side, default 708] -> profiles/r04_ab/checks_per_query_histogram.txt"""
import sys, time, numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'oracle')); sys.path.insert(0,os.path.join(ROOT,'scalable-ccd_amd'))
import orc
from sccd import scenes
n=int(sys.argv[1]) if len(sys.argv)>1 else 708
V0,V1,E,F=scenes.folded_cloth(n)
t0=time.time()
vb,eb,fb=orc.build_boxes(V0,V1,E,F)
vf=orc.sort_and_sweep(vb,fb,0,nthreads=8,sort=False)[0]
ee=orc.sort_and_sweep(eb,None,0,nthreads=8,sort=False)[0]
print("pairs",len(vf),len(ee),time.time()-t0)
toi=1.0
for name,pairs,is_vf in (("VF",vf,True),("EE",ee,False)):
    # seeded with final toi to emulate late pruning: run twice
    t1,chk=orc.narrow_phase_mt(V0,V1,E,F,pairs,is_vf,toi=toi,nthreads=8,want_checks=True)
    print(name,"toi",t1,"checks",chk.sum(),"mean",chk.mean())
    t2,chk2=orc.narrow_phase_mt(V0,V1,E,F,pairs,is_vf,toi=t1,nthreads=8,want_checks=True)
    print(name,"seeded with final: checks",chk2.sum(),"mean",chk2.mean())
    for c in (chk,chk2):
        h=np.bincount(np.minimum(c,40))
        tot=c.sum()
        print(" hist(0..40+):",h.tolist())
        for K in (4,6,8,10,12,16,24,32):
            surv=(c>K)
            print("  K=%d survivors %.4f of queries, checks within K: %.3f of all, mean min(c,K)=%.2f, survivors' remaining checks mean %.1f"%(K,surv.mean(),np.minimum(c,K).sum()/tot,np.minimum(c,K).mean(), (c[surv]-K).mean() if surv.any() else 0))
    toi=t1

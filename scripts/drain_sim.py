"""Drain model of C3's phase B (CPU): after time slicing has advanced all chains together to x0 evaluations (the point at
which no chain waits any more), every survivor owns a lane group and a wavefront steps in T(k) microseconds with k
wavefronts on its SIMD (measured: 1.87 / 2.6 / 3.0).  Compares the launch as it is with an IDEAL compaction inside each
four-wavefront workgroup (survivors packed into the fewest wavefronts, the two workgroups of a CU towards opposite SIMDs).
usage: drain_sim.py traces.npz   (evals[nb]: tests/fuzz/gen_branch_traces.py)"""
import numpy as np, sys
rng=np.random.default_rng(1)
ev=np.load(sys.argv[1])['evals'].astype(int)
T={0:0,1:1.87,2:2.6,3:3.0}
NW=2048; SLOTS=NW*4
# PS phase: find x0 such that number alive at x0 <= SLOTS
xs=np.sort(ev)
x0=next(x for x in range(0,900) if (ev>x).sum()<=SLOTS)
work=np.minimum(ev,x0).sum()          # group-steps done in the fill phase
t_fill=work/SLOTS*T[2]                # 2 waves per SIMD, all groups busy
alive=ev[ev>x0]-x0
print('x0',x0,'alive',len(alive),'fill time us',round(t_fill))
def drain(compact):
    rem=np.zeros(SLOTS,int); idx=rng.permutation(SLOTS)[:len(alive)]; rem[idx]=alive
    rem=rem.reshape(256,2,4,4)      # CU, WG, wave(SIMD), group
    # time-stepped in units of chain steps per wave; simulate per CU independently (SIMD s hosts wave s of both WGs)
    tmax=0
    for cu in range(256):
        r=rem[cu].copy()            # [wg][wave][group]
        t=0.0
        prog=np.zeros((2,4))        # fractional progress of each wave
        while r.sum()>0:
            if compact:             # per WG: pack alive chains into fewest waves; WG0 fills waves 0.., WG1 fills waves 3..
                for wg in range(2):
                    ch=np.sort(r[wg][r[wg]>0])[::-1]
                    newr=np.zeros((4,4),int)
                    order=[0,1,2,3] if wg==0 else [3,2,1,0]
                    for i,c in enumerate(ch): newr[order[i//4], i%4]=c
                    r[wg]=newr
            live=(r.sum(axis=2)>0)                     # [wg][wave]
            k=live.sum(axis=0)                         # per SIMD
            # advance until next chain finishes anywhere: each live wave steps at rate 1/T(k_simd)
            rate=np.where(live, 1.0/np.vectorize(T.get)(np.maximum(k,1))[None,:].repeat(2,0),0.0)
            # steps to next event per wave = min positive rem in that wave
            nxt=np.where(live, np.where(r>0,r,10**9).min(axis=2),10**9)
            dt=(nxt/np.where(rate>0,rate,1e-9)).min()
            steps=np.floor(rate*dt+1e-9).astype(int)
            steps=np.maximum(steps, (nxt/np.where(rate>0,rate,1e-9)==dt).astype(int)*nxt.clip(max=10**8)*0+steps)
            # simple: advance each live wave by floor(rate*dt) steps, at least 1 for the event wave
            ew=np.unravel_index(np.argmin(nxt/np.where(rate>0,rate,1e-9)),nxt.shape)
            steps[ew]=nxt[ew]
            r=np.maximum(r-steps[:,:,None],0)*(r>0)
            t+=dt
        tmax=max(tmax,t)
    return tmax
for c in (False,True):
    d=drain(c); print('compact' if c else 'plain  ','drain us',round(d),'total ms',round((t_fill+d)/1e3,3))

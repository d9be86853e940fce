import sys, time, numpy as np
sys.path.insert(0,'.')
import alphabeta_rs_amd as A
rows=[[float(t) for t in ln.replace("\t"," ").split()] for ln in open('./tests/golden/pedigree_generated.txt').read().splitlines()[1:] if ln.strip()]
ped=np.asarray(rows); p0=0.6554051647850447
W,S,B=300,100,100
rng=np.random.default_rng(1)
D=np.abs(ped[:,3][None,:]*rng.uniform(0.8,1.2,(W,1)))
with A.Context(0) as ctx:
    for order in (0,-1,0,-1):
        plan=A.Plan(ctx,ped[:,:3],W,S,B,options=A.default_options(seed=5,strict_order=order))
        plan.set_windows(D,np.full(W,p0)); plan.run(); plan.sync()
        t0=time.perf_counter()
        for _ in range(5): plan.run()
        plan.sync(); dt=(time.perf_counter()-t0)/5
        print('order',order,'ms/step %.2f'%(dt*1e3), plan.kernel_ms(), plan.last_kernels()); plan.close()

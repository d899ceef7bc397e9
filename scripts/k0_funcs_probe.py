#!/usr/bin/env python3
"""K0 (mlx_eos_map) per function -- density, alpha, beta, drho_dtemp -- on 16 steps of the bench grid,
float64 and float32 fields: the pointwise maps behind derived.calc_rho / calc_alpha / calc_beta.

    python scripts/k0_funcs_probe.py
"""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from momlevel_amd import core, synthetic
nz,ny,nx=75,1080,1440; nt=16
g=synthetic.make_grid(ny,nx,nz); vol0=torch.from_numpy(g["volcello"]).cuda()
pz=torch.from_numpy(g["z_l"]*1e4+101325.0).cuda()
out={}
for dt,name in ((torch.float64,"f64"),(torch.float32,"f32")):
    kw=dict(seed=synthetic.SEED,mask3d=vol0)
    T=core.synth_field((nt,nz,ny,nx),dt,field_id=1,lo=-2.0,scale=34.0,**kw); S=core.synth_field((nt,nz,ny,nx),dt,field_id=2,lo=30.0,scale=10.0,**kw)
    cells=T.numel(); bpc=(16 if dt==torch.float64 else 8)+8
    for func in ("density","alpha","beta","drho_dtemp"):
        core.eos_map(T,S,pz,func=func); torch.cuda.synchronize(); best=1e9
        for _ in range(3):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record(); r=core.eos_map(T,S,pz,func=func); e1.record(); torch.cuda.synchronize(); best=min(best,e0.elapsed_time(e1)); del r
        out[f"{name}_{func}"]={"ms":round(best,3),"frac_of_8TBs":round(bpc*cells/best/1e6/8000,3)}
    del T,S; torch.cuda.empty_cache()
print(json.dumps(out))

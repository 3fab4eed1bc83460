#!/usr/bin/env python3
"""Whole C2 frame, product vs CPU oracle: how many pixels moved by more than 2e-6, which Gaussians carry a gradient error above
1e-4 of the tensor's largest entry, and how far those Gaussians are from a moved pixel (a blend whose alpha sits on the 1/255
threshold is taken by one expf and dropped by the other: that pixel moves by <= 1/255 and the Gaussians blended at it inherit
the weight of one pixel in their gradients).  Round-2 result: 2 pixels of 2,073,600; 1 Gaussian of 1,000,000, 4.7 px away."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "stopthepop-rasterization_amd"), ROOT): sys.path.insert(0, p)
import numpy as np
from helpers import GpuRun, oracle_run, settings_dict, FULL_STP
from diff_gaussian_rasterization import scenes
sc=scenes.config("C2"); sd=settings_dict(**FULL_STP)
g=GpuRun(sc, sd, backward=True)
f,og=oracle_run(sc, sd, backward=True)
d=np.abs(g.color.astype(np.float64)-f.color)
mv=np.argwhere((d>2e-6).any(axis=0))
print("moved pixels", len(mv), "max", d.max())
for k in ("dL_dmeans2D","dL_dopacity","dL_dmeans3D","dL_dscales","dL_drotations","dL_dsh"):
    a,b=g.grads[k].astype(np.float64),og[k].astype(np.float64)
    if k=="dL_dmeans2D": a,b=a[:,:2],b[:,:2]
    e=np.abs(a-b).reshape(a.shape[0],-1).max(axis=1)
    i=int(np.argmax(e)); print(k,"max abs",e[i],"rel",e[i]/np.abs(b).max(),"gaussian",i,"n>1e-4rel",int((e>1e-4*np.abs(b).max()).sum()))
m2=f.array("means2D").reshape(-1,2)
e=np.abs(g.grads["dL_dmeans2D"][:,:2].astype(np.float64)-og["dL_dmeans2D"][:,:2]).max(axis=1)
bad=np.argsort(-e)[:8]
for i in bad:
    dist=np.min(np.hypot(mv[:,1]-m2[i,0], mv[:,0]-m2[i,1])) if len(mv) else -1
    print("gaussian",i,"err",e[i],"mean2D",m2[i],"radius",f.radii[i],"nearest moved pixel dist",dist, "opacity", sc.opacities[i,0])
for y,x in mv[:10]:
    print("pixel",x,y,d[:,y,x])

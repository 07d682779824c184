"""CPU simulation of the per-quartet screening of the tiled J/K kernels on a large molecule: survivors per (bra tile pair,
ket tile pair) and the fill of the 256-lane batches of the lane-per-quartet kernels (NKS ket pairs staged per iteration).
Uses the oracle's Schwarz matrix (TEST INFRASTRUCTURE) and the predicate of jk_tile.hip (reference screen_jk_tasks.cu:202-261).
usage: python tools/survivor_stats.py       -> profiles/r02_survivor_statistics.txt was produced with it"""
import os, sys, time, math; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from joltqc_amd.pyscf.basis import BasisLayout
from joltqc_amd.constants import tile_width
from joltqc_amd.pyscf import jk as jkmod
mol,_=bench.load_workload("0112-elongated-nitrogenous")
lay=BasisLayout.from_mol(mol, alignment=tile_width)
from oracle import jk as O
Q=O.schwarz(lay.packed)
lq=np.log(Q+1e-300).astype(np.float32); lq[lay.pad_id,:]=-100; lq[:,lay.pad_id]=-100
tt=jkmod._TileTables(lay, 0.0, q_host=lq)
log_max_dm=8.931080446508306; log_cut=math.log(1e-13)
goff,gkey=lay.group_offset,lay.group_key
rng=np.random.default_rng(0)
NKS={(3,2,1,0):4,(2,1,1,0):2,(1,0,1,0):1,(2,1,1,1):4,(1,0,0,0):1,(3,1,2,0):4,(2,0,1,0):1}
for ang,nks in NKS.items():
    recs=[]
    for gi in range(lay.ngroups):
      for gj in range(gi+1):
        for gk in range(gi+1):
          for gl in range(gk+1):
            a=(int(gkey[gi,0]),int(gkey[gj,0]),int(gkey[gk,0]),int(gkey[gl,0]))
            if a!=ang or (gi,gj) not in tt.q_host or (gk,gl) not in tt.q_host: continue
            qij=tt.q_host[gi,gj]; qkl=tt.q_host[gk,gl]
            oi=tt.offset[gi,gj]; ok=tt.offset[gk,gl]
            nb=len(qij); sel=rng.choice(nb, min(nb,30), replace=False)
            wi,wj,wk,wl=[tile_width(x) for x in ang]
            for b in sel:
                thr=log_cut-log_max_dm-qij[b]
                nk=int(np.searchsorted(-qkl,-thr,side='left'))
                if nk==0: continue
                sh=tt.sh_host[oi+b]; i0,j0=int(sh>>16),int(sh&0xffff)
                qb=lq[i0:i0+wi,j0:j0+wj]
                # sample consecutive groups of nks ket pairs (what one iteration stages)
                starts=rng.choice(max(nk//nks,1), min(max(nk//nks,1),20), replace=False)*nks
                w=(nk/nks)/len(starts)*nb/len(sel)
                for s0 in starts:
                    n=0
                    for k in range(s0,min(s0+nks,nk)):
                        shk=tt.sh_host[ok+k]; k0,l0=int(shk>>16),int(shk&0xffff)
                        qk=lq[k0:k0+wk,l0:l0+wl]
                        est=qb[:,:,None,None]+qk[None,None,:,:]+log_max_dm
                        I,J,K,L=np.meshgrid(np.arange(i0,i0+wi),np.arange(j0,j0+wj),np.arange(k0,k0+wk),np.arange(l0,l0+wl),indexing='ij')
                        n+=int(((I>=J)&(K>=L)&(I*lay.nbasis+J>=K*lay.nbasis+L)&(est>log_cut)).sum())
                    recs.append((n,w))
    r=np.array(recs); n,w=r[:,0],r[:,1]
    batches=np.ceil(n/256); batches[n==0]=0
    tot_q=(n*w).sum(); tot_b=(batches*w).sum(); tot_it=w.sum()
    out=[f"{ang} NKS={nks}: iterations {tot_it:.3g}, quartets {tot_q:.3g}, batches {tot_b:.3g}, lane utilisation {tot_q/(256*tot_b):.2f}, empty iterations {100*w[n==0].sum()/tot_it:.0f}%"]
    for T in (16,32,64,128):
        s=(n<T)&(n>0)
        nb2=(batches*w)[~s].sum()
        out.append(f"n<{T}: {100*w[s].sum()/tot_it:.0f}% of iterations ({100*(batches*w)[s].sum()/tot_b:.0f}% of batches), {100*(n*w)[s].sum()/tot_q:.1f}% of quartets; rest util {((n*w)[~s].sum())/(256*nb2):.2f}")
    print("; ".join(out),flush=True)

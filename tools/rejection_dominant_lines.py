import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import rejection_study as rs
from slslam_amd import synth
from oracle import pyoracle
import scipy.sparse as sp, scipy.sparse.linalg as spla
# find a 300-line window with several rejected steps at the end
for seed in range(5000,5040):
    w=synth.make_window(seed,num_lines=300)
    x,s,tr=pyoracle.lba_solve(w,linear_solver=1)
    tail=[q['step_is_successful'] for q in tr[-4:]]
    if sum(tail)==0: break
print("seed",seed,s['num_successful_steps'],s['num_unsuccessful_steps'])
# re-run to the accepted point before the final rejections: take oracle's final x (accepted point), linearise there with numpy
C,L=w['num_cameras'],w['num_lines']
ww=dict(w); ww['parameters']=x
# build J at x via rejection_study internals: reuse independent_lm's code by running 1 iteration with huge radius? simpler: copy blocks
cam_idx,line_idx,obs=np.asarray(w['camera_index']),np.asarray(w['line_index']),np.asarray(w['observations']).reshape(-1,8)
fi=np.asarray(w['fixed_index']).reshape(-1,2); cam_const=np.zeros(C,bool); cam_const[cam_idx[fi[:,0]!=0]]=True
free=[c for c in range(C) if not cam_const[c]]
ccol={c:6*k for k,c in enumerate(free)}; n=6*len(free)+4*L
rows,cols,vals=[],[],[]; r_all=np.zeros(4*len(cam_idx))
from make_golden import line_residual_np, central, A_HUBER
for i,(c,l) in enumerate(zip(cam_idx,line_idx)):
    cam,ln=x[6*c:6*c+6],x[6*C+4*l:6*C+4*l+4]
    r=line_residual_np(cam,ln,obs[i]); sq=r@r
    rp=A_HUBER/np.sqrt(sq) if sq>A_HUBER**2 else 1.0; sr=np.sqrt(rp); r_all[4*i:4*i+4]=sr*r
    if c in ccol:
        Jc=sr*central(lambda q: line_residual_np(q,ln,obs[i]),cam)
        for a in range(4):
            for b in range(6): rows.append(4*i+a); cols.append(ccol[c]+b); vals.append(Jc[a,b])
    Jl=sr*central(lambda q: line_residual_np(cam,q,obs[i]),ln)
    for a in range(4):
        for b in range(4): rows.append(4*i+a); cols.append(6*len(free)+4*l+b); vals.append(Jl[a,b])
J=sp.csr_matrix((vals,(rows,cols)),shape=(4*len(cam_idx),n))
g=J.T@r_all; H=(J.T@J).tocsc(); d=H.diagonal()
radius=tr[-1]['trust_region_radius']
scale=1/(1+np.sqrt(d))   # approx (scale at x0 in the real loop)
Js=J@sp.diags(scale); gs=Js.T@r_all; Hs=(Js.T@Js).tocsc(); d2=np.clip(Hs.diagonal(),1e-6,1e32)/radius
y=spla.spsolve(Hs+sp.diags(d2).tocsc(),gs); delta=-scale*y
order=np.argsort(-np.abs(delta))[:8]
print("radius",radius,"|delta|",np.linalg.norm(delta))
for k in order:
    if k<6*len(free): print("  cam param",k,delta[k]); continue
    l=(k-6*len(free))//4; comp=(k-6*len(free))%4
    u=x[6*C+4*l:6*C+4*l+4]
    # 4x4 block of the line and its eigenvalues
    blk=H[6*len(free)+4*l:6*len(free)+4*l+4,6*len(free)+4*l:6*len(free)+4*l+4].toarray()
    ev=np.linalg.eigvalsh(blk)
    print("  line %d comp %s delta % .3e | u=(a % .3f, b % .3f, g % .3f, t %.4f) cos(b)=%.2e cot(t)=%.1f | diag(H_ll) %s | eig(H_ll) %s | nobs %d"%(l,'abgt'[comp],delta[k],u[0],u[1],u[2],u[3],np.cos(u[1]),1/np.tan(u[3]),np.array2string(np.diag(blk),precision=2),np.array2string(ev,precision=2),(line_idx==l).sum()))

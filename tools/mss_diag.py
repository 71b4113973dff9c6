import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from deepgrp_amd import synthetic
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence, stream_ptr
for name, w in (("gain3", synthetic.synthetic_weights(128,5,False,7,3.0)), ("gain1", synthetic.synthetic_weights(128,5,False,7,1.0))):
    m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
    st, d_idx = upload_sequence(synthetic.synthetic_chromosome(1_000_000))
    pipe = ContigPipeline(m); merged = pipe.merged(d_idx)
    n = merged.shape[0]
    sc = torch.empty(n, dtype=torch.float64, device="cuda"); cl = torch.empty(n, dtype=torch.int8, device="cuda")
    check(lib().dgrp_scores(merged.data_ptr(), n, 5, sc.data_ptr(), cl.data_ptr(), stream_ptr()))
    S = sc.cpu().numpy()[:300000]
    pos = S > 0
    print(name, "frac positive %.3f" % pos.mean(), "mean score %.3f" % S.mean(), "runs", int((pos[1:] & ~pos[:-1]).sum()), "score quantiles", np.round(np.quantile(S,[0.01,0.25,0.5,0.75,0.99]),3))
    # instrumented stack algorithm
    stack=[]; L=0.0; peak=-1e30; i=0; nS=len(S); ev=dict(merge=0,push_grow=0,flush=0,reset=0,maxdepth=0,runs=0, depth_hist={})
    xdrop=2297.56
    while i<nS:
        if S[i]>0:
            R=L+S[i]; k=i+1
            while k<nS and S[k]>0: R+=S[k]; k+=1
            if R>peak: peak=R
            t=[i,k,L,R,-1]; ev["runs"]+=1
            d=len(stack); ev["depth_hist"][min(d,5)]=ev["depth_hist"].get(min(d,5),0)+1
            while True:
                j=len(stack)-1
                while j>=0:
                    if stack[j][2]<t[2]: break
                    j=stack[j][4] if stack[j][4]>=0 else j-1
                if j>=0 and stack[j][3]<t[3]:
                    t[0]=stack[j][0]; t[2]=stack[j][2]; del stack[j:]; ev["merge"]+=1
                else:
                    if j<0: stack.clear(); peak=R; ev["flush"]+=1
                    else: ev["push_grow"]+=1
                    t[4]=j; stack.append(t); break
            ev["maxdepth"]=max(ev["maxdepth"],len(stack)); L=R; i=k
        else:
            if L+S[i]+xdrop<peak: stack.clear(); L=0.0; peak=-1e30; ev["reset"]+=1
            L+=S[i]; i+=1
    print("   events", ev)
    m.close()

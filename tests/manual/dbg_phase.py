import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests/manual")
import numpy as np, torch, math
import stress_collide as sc
import fun_ofdm_amd as foa
from oracle import pyoracle as po
rx=foa.Receiver(0)
ltc=po.lts_time_domain_conj()
for seed in (100000,100002):
    s=sc.make_stream(seed)
    descs=foa.find_alignments(s); od=po.find_alignments_f32(s)
    d_iq=torch.from_numpy(s.view(np.float32).reshape(-1,2).copy()).to("cuda:0")
    cap=s.size//300+64
    d_desc=torch.zeros(cap*48,dtype=torch.uint8,device="cuda:0"); d_end=torch.zeros(cap,dtype=torch.int64,device="cuda:0")
    m=rx.sync_dev(d_iq,d_desc,d_end)
    dd=d_desc.cpu().numpy()[:m*48].view(foa.frame_desc_dtype)
    for i in range(min(m,descs.size)):
        p=int(descs[i]["lts1_pos"]); idx=p-24+159
        a=complex(s[idx]); mm=a*complex(ltc[63])
        ph=math.atan2(mm.imag,mm.real)
        print(seed,i,"sample",a,"m",mm)
        print("   python  c,s", repr(math.cos(ph)), repr(math.sin(ph)), "phase", repr(ph))
        print("   host    c,s", repr(float(descs[i]["c"])), repr(float(descs[i]["s"])))
        print("   oracle  c,s", repr(float(od[i]["c"])), repr(float(od[i]["s"])))
        print("   device  c,s", repr(float(dd[i]["c"])), repr(float(dd[i]["s"])), "phase from device c,s", repr(math.atan2(float(dd[i]["s"]),float(dd[i]["c"]))))

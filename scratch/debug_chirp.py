import numpy as np, torch, sys
sys.path.insert(0,'.')
import gpu_sdr_amd as g, oracle
from gpu_sdr_amd.source import device_chirp
dev=torch.device('cuda:0')
rate,L=200_000_000,1_000_000
args=(rate,-100_000_000,100_000_000,1_000_000,1.0)
cp=g.chirp_derive(*args); ocp=oracle.chirp_params(*args)
x=torch.empty(L,dtype=torch.complex64,device=dev)
for last in (0,L,2*L):
    device_chirp(x,last,cp,scale=0.5); torch.cuda.synchronize()
    xr=oracle.chirp_gen(ocp,last,L,0.5)
    d=np.abs(x.cpu().numpy()-xr)
    bad=np.nonzero(d>1e-5)[0]
    print('src last',last,'max',d.max(),'nbad',len(bad),bad[:10])
p=g.param(mode="RX",rate=rate,buffer_len=L,decim=1,freq=[args[1]],chirp_f=[args[2]],swipe_s=[args[3]],chirp_t=[args[4]],wave_type=[g.w_type.CHIRP])
dem=g.RX_buffer_demodulator(p,device_index=0)
ref=oracle.Chirp(*args,1,L)
out=torch.empty(dem.out_capacity,dtype=torch.complex64,device=dev)
for c in range(3):
    xr=oracle.chirp_gen(ocp,c*L,L,0.5)
    x.copy_(torch.from_numpy(xr))
    n=dem.process(x,out); torch.cuda.synchronize()
    y=out[:n].cpu().numpy(); yr=ref.process(xr)
    d=np.abs(y-yr); bad=np.nonzero(d>1e-5)[0]
    print('dem buf',c,'max',d.max(),'nbad',len(bad),bad[:10], 'oracle dev from .5:',np.abs(yr-0.5).max())
    if len(bad): print(y[bad[:5]], yr[bad[:5]])

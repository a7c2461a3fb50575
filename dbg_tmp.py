import sys, os, numpy as np, torch
sys.path.insert(0,'.')
os.environ['WSA_DBG']=sys.argv[1] if len(sys.argv)>1 else '16'
import webspeechanalyzer_amd as w
from webspeechanalyzer_amd.synth import synth_clips
fs,n,ns=16000,1024,160000
pcm=synth_clips(n,ns,fs=fs,seed=1000,device='cuda')
an=w.Analyzer(w.Config(output_level=5)); b=an.batch([ns]*n,fs); b.enable_trace()
st=torch.cuda.current_stream().cuda_stream
for _ in range(3): b.run(pcm.data_ptr(),pcm.stride(0),st); r=b.device_result(st)
tr=b.trace(st)
nsp=r.n_segments
t=tr[:nsp]
clk=2.1e9
fr=t[:,0]/clk*1e3; fi=t[:,1]/clk*1e3
print('dbg',os.environ['WSA_DBG'],'spans',nsp,'stage',b.stage_ms()[1],'frames ms: mean %.3f max %.3f  finalize ms: mean %.3f max %.3f'%(fr.mean(),fr.max(),fi.mean(),fi.max()))
import collections
pw=collections.defaultdict(float)
for k in range(nsp): pw[int(t[k,6])]+=fr[k]+fi[k]
v=np.array(list(pw.values())); print('  per-wave busy ms: mean %.3f max %.3f waves %d'%(v.mean(),v.max(),len(v)))

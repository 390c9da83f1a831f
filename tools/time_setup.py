import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ribodetector_amd.model import model as M
from ribodetector_amd.parse_config import ConfigParser
t0=time.perf_counter(); torch.cuda.init(); torch.zeros(1,device='cuda'); torch.cuda.synchronize(); print('cuda init %.3f'%(time.perf_counter()-t0))
cfg = ConfigParser.from_json("ribodetector_amd/config.json")
t0=time.perf_counter(); m = cfg.init_obj("arch", M); m.load_state_dict(cfg.load_state_dict("mcc")); print('weights %.3f'%(time.perf_counter()-t0))
m.set_prefix_table(0)
t0=time.perf_counter(); m.to("cuda:0").eval(); torch.cuda.synchronize(); print('to(cuda) k=0 %.3f'%(time.perf_counter()-t0))
for k in (8,10,11,12,13,12,12):
    torch.cuda.synchronize(); t0=time.perf_counter(); m.set_prefix_table(0); m.set_prefix_table(k); torch.cuda.synchronize(); print('set_prefix_table(%d) %.4f s'%(k,time.perf_counter()-t0))
torch.cuda.empty_cache()
for k in (12,13):
    torch.cuda.synchronize(); t0=time.perf_counter(); m.set_prefix_table(0); torch.cuda.empty_cache(); m.set_prefix_table(k); torch.cuda.synchronize(); print('cold alloc set_prefix_table(%d) %.4f s'%(k,time.perf_counter()-t0))
t0=time.perf_counter(); x=torch.empty(218<<20,dtype=torch.uint8,pin_memory=True); print('pinned 218MB %.4f'%(time.perf_counter()-t0))
t0=time.perf_counter(); y=torch.empty(218<<20,dtype=torch.uint8,pin_memory=True); print('pinned 218MB again %.4f'%(time.perf_counter()-t0))
del x; t0=time.perf_counter(); x=torch.empty(218<<20,dtype=torch.uint8,pin_memory=True); print('pinned 218MB reuse %.4f'%(time.perf_counter()-t0))

"""where do the 3.3 s of set_prefix_table(13) go: allocation or the level launches?   python tools/time_k13.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ribodetector_amd import _native as N
from ribodetector_amd.model import model as M
from ribodetector_amd.parse_config import ConfigParser
cfg = ConfigParser.from_json("ribodetector_amd/config.json")
m = cfg.init_obj("arch", M); m.load_state_dict(cfg.load_state_dict("mcc")); m.set_prefix_table(0); m.to("cuda:0").eval()
lib = N.lib()
for k in (12, 13):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tab = torch.empty(int(lib.rd_prefix_table_bytes(k)), dtype=torch.uint8, device="cuda:0")
    scr = torch.empty(int(lib.rd_prefix_scratch_bytes(k)), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    N.check(lib.rd_set_prefix_table(m._handle, k, N.ptr(tab), tab.numel(), N.ptr(scr), scr.numel(), N.stream_ptr(m.device)), "set")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    N.check(lib.rd_set_prefix_table(m._handle, k, N.ptr(tab), tab.numel(), N.ptr(scr), scr.numel(), N.stream_ptr(m.device)), "set")
    torch.cuda.synchronize(); t3 = time.perf_counter()
    tab.zero_(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("k=%d alloc %.3f s  build(first touch) %.3f s  build(again) %.3f s  memset table %.3f s" % (k, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    N.check(lib.rd_set_prefix_table(m._handle, 0, None, 0, None, 0, N.stream_ptr(m.device)), "set0")
    del tab, scr
    torch.cuda.empty_cache()

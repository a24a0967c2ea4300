"""Which earlier leg of the default bench run changes a later one?  gpurun -- python tools/check/leg_order.py gp3 l lv l gp4 l"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
import la3dm_amd
from la3dm_amd import _lib

side = argparse.Namespace(steps=10, warmup=2, no_cpu=True)
for leg in sys.argv[1:]:
    if leg == "l":
        print("bgkl", bench.l_leg(side, torch, la3dm_amd, cpu=False)["ms_per_step"], flush=True)
    elif leg == "lv":
        print("lv", bench.lv_leg(side, torch, la3dm_amd, _lib, cpu=False)["synthetic_50k"]["ms_per_insert"], flush=True)
    elif leg in ("gp3", "gp4"):
        print(leg, bench.gp_leg(side, torch, la3dm_amd, _lib, depth=int(leg[2]), cpu=False)["ms_per_step"], flush=True)
    elif leg == "big":
        print("big", bench.out_of_cache_leg(la3dm_amd, _lib, torch, torch.device("cuda:0"))["kernel_ms"], flush=True)

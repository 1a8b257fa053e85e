"""where bench.py's placement goes wrong: the process's affinity and the GPU's locality before / after torch is imported"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
print("affinity at start:", len(os.sched_getaffinity(0)))
import torch
import torch.distributed as dist
print("affinity after import torch:", len(os.sched_getaffinity(0)))
import kinematic_icp_amd as K
from kinematic_icp_amd import synthetic as syn
print("cuda available", torch.cuda.is_available())
torch.cuda.set_device(0)
print("affinity after set_device:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8])
print("locality:", K.device_locality(0)[0], len(K.device_locality(0)[1]), "near:", len(K.cpus_near_gpu(0)), "near (node):", len(K.cpus_near_gpu(0, one_l3_domain=False)))
for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "KICP_NUMA", "OMP_PROC_BIND", "GOMP_CPU_AFFINITY"):
    print(k, os.environ.get(k))

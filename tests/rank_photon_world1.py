"""One rank under torchrun with backend nccl (RCCL): build_photon_map_sharded runs its all-gathers through the process
group (world = 1) and must give grt_photon_build's map bit for bit.  Started by tests/test_gpu_nccl_world1.py."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("go-raytracer_amd")
shard = importlib.import_module("go-raytracer_amd.shard")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", device_id=dev)
scene = pkg.scenes.cornell(64, 64, "cornell8d")
a = shard.build_photon_map_sharded(scene, 20000, caustics=False, rank=dist.get_rank(), world=dist.get_world_size(), seed=1)
b = pkg.tracer.PhotonMap(20000).build(scene, caustics=False, seed=1)
pa, wa, ta = a.read()
pb, wb, tb = b.read()
same = bool((pa == pb).all() and (wa == wb).all() and (ta == tb).all())
# the reduce of SampleShardRenderer through the group
ren = shard.SampleShardRenderer(pkg.scenes.cornell(64, 64), 8, 0, 1, seed=1)
ren.render()
before = ren.rgb_sum.clone()
after = ren.reduce(0).clone()
torch.cuda.synchronize()
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "photons": int(a.count()), "maps_equal": same,
                  "emitted_equal": a.emitted() == b.emitted(), "reduce_is_identity": bool((before == after).all())}))
dist.barrier()
dist.destroy_process_group()

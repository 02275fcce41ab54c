"""One rank of the sharded front-end rehearsal (launched by tests/test_halo_gpu.py; RANK / WORLD_SIZE / MASTER_* in the env):
rank g runs ORB on frames [g*B, (g+1)*B) of one sequence and matches its B-1 inner pairs plus the pair across the shard boundary."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.distributed as dist
from boslam_amd import Context, frontend, synth

out, B = sys.argv[1], int(sys.argv[2])
backend = os.environ.get("BH_TEST_BACKEND", "gloo")
dist.init_process_group(backend=backend)
rank = dist.get_rank()
torch.cuda.set_device(0)
ctx = Context(0)
params = frontend.make_params(nfeatures=600)
bgr, depth = synth.render_batch(B, start=rank * B)
fe = frontend.DeviceFrontEnd(ctx, B, 480, 640, params, torch_device="cuda:0", dist=dist, backend=backend)
b = torch.from_numpy(bgr).cuda(); d = torch.from_numpy(depth.view(np.int16)).cuda()
for _ in range(2):                                   # twice: the second pass reuses the halo buffers
    fe.run(b, d)
ctx.synchronize(); torch.cuda.synchronize()
np.savez(os.path.join(out, f"rank{rank}.npz"), n_kp=fe.n_kp.cpu().numpy(), desc=fe.desc.cpu().numpy(), n_match=fe.n_match.cpu().numpy(),
         qidx=fe.qidx.cpu().numpy(), tidx=fe.tidx.cpu().numpy(), dist=fe.dist.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
ctx.close()

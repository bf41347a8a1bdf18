import os, sys
ROOT = "/root/repo" if not os.environ.get("GRAFT_REPO_ROOT") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "spark-s3-shuffle_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import s3shuffle
from s3shuffle import datagen
mib, ntask, algo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = s3shuffle.Codec(0)
tasks = []
keep = []
for t in range(ntask):
    data, offs = datagen.skew_block(mib << 20, "terasort", seed=5, map_id=t)
    d_src = torch.from_numpy(data).cuda()
    cap = c.max_compressed_size(1, offs)
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    keep.append((d_src, d_dst))
    tasks.append((d_src.data_ptr(), offs, d_dst.data_ptr(), cap))
torch.cuda.synchronize()
res = c.compress_map_outputs_batch_device(1, algo, tasks)
print("ok", mib, ntask, algo, [int(r[0]) for r in res])

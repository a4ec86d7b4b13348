"""Can two ranks of libidgrec's RCCL communicator share ONE GPU?  (RCCL normally refuses duplicate devices; if it does
not, this is a real 2-rank check of idg_comm_* on a 1-GPU box.)  Launch: python -m torch.distributed.run --nproc-per-node 2
--master-addr 127.0.0.1 scripts/native_comm_2ranks_1gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idgrec_amd.sharded import NativeComm  # noqa: E402

dist.init_process_group("gloo")
torch.cuda.set_device(0)
try:
    comm = NativeComm(dist, 0)
    print("rank", dist.get_rank(), "communicator up; self-test:", comm.self_test(), flush=True)
    comm.close()
except Exception as exc:  # noqa: BLE001
    print("rank", dist.get_rank(), "FAILED:", str(exc)[:300], flush=True)
dist.destroy_process_group()

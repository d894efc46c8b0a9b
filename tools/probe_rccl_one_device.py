#!/usr/bin/env python3
"""Can RCCL host two ranks on ONE device?  (Decides whether the library-owned RCCL exchange can be tested at
world size 2 on a one-GPU box.)  Prints the outcome; never raises."""
import os
import socket
import sys


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank %d: all_reduce on one device ok -> %s" % (rank, t[0].item()), flush=True)
        dist.destroy_process_group()
    except Exception as e:                                # noqa: BLE001
        print("rank %d: RCCL refused two ranks on one device: %s" % (rank, str(e).splitlines()[0][:300]), flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        mp.spawn(worker, args=(2, port), nprocs=2, join=True)
    except Exception as e:                                # noqa: BLE001
        print("probe ended with", str(e).splitlines()[0][:300])
    sys.exit(0)

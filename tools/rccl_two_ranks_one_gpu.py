#!/usr/bin/env python3
"""Probe: does RCCL accept two ranks on ONE device (the only way to run a real 2-rank RCCL collective on a
1-GPU box)?  Spawns two processes, both on cuda:0, and tries one all_gather_into_tensor."""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        x = torch.full((1024,), rank + 1, dtype=torch.uint8, device="cuda")
        out = torch.empty((2048,), dtype=torch.uint8, device="cuda")
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print(f"rank {rank}: OK {out[0].item()} {out[-1].item()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29577)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    print("exit codes", [p.exitcode for p in ps])

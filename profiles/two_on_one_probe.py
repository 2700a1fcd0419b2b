import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        t = torch.ones(4, device="cuda:0") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "ok", t.tolist(), flush=True)
    except Exception as e:
        print("rank", rank, "FAILED", repr(e)[:300], flush=True)
if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2, join=True)

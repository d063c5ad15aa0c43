"""Probe (GPU box, ONE GPU): which torch.distributed backends accept two ranks that share cuda:0, and which of the collectives the
sharded path uses work there.  Launched as  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... THIS BACKEND"""
import os, sys, torch, torch.distributed as dist
backend = sys.argv[1]
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
def attempt(name, fn):
    try:
        fn(); torch.cuda.synchronize(); print(f"[{backend} r{rank}] {name}: ok", flush=True)
    except Exception as e:
        print(f"[{backend} r{rank}] {name}: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
try:
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
except Exception as e:
    print(f"[{backend} r{rank}] init FAILED {e}", flush=True); sys.exit(0)
x = torch.full((1024,), float(rank + 1), device=dev)
attempt("all_reduce", lambda: dist.all_reduce(x))
out = torch.empty(world * 1024, device=dev, dtype=torch.bfloat16); src = torch.full((1024,), rank + 1.0, device=dev, dtype=torch.bfloat16)
attempt("all_gather_into_tensor bf16", lambda: dist.all_gather_into_tensor(out, src))
o8 = torch.empty(world * 1024, device=dev, dtype=torch.uint8); s8 = torch.full((1024,), rank + 1, device=dev, dtype=torch.uint8)
attempt("all_gather_into_tensor u8 async", lambda: dist.all_gather_into_tensor(o8, s8, async_op=True).wait())
def p2p():
    r = torch.empty(1024, device=dev); ops = [dist.P2POp(dist.isend, x, (rank + 1) % world), dist.P2POp(dist.irecv, r, (rank - 1) % world)]
    for w in dist.batch_isend_irecv(ops): w.wait()
attempt("batch_isend_irecv", p2p)
def a2a():
    o = torch.empty(world * 8, device=dev); i = torch.arange(world * 8, device=dev, dtype=torch.float32)
    dist.all_to_all_single(o, i, [8] * world, [8] * world)
attempt("all_to_all_single", a2a)
attempt("barrier", lambda: dist.barrier())
dist.destroy_process_group()

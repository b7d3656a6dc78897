"""hospital / rents / flights end to end with every class sweep sharded over the GPUs of one node:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port 29511 scripts/run_distributed.py flights [particles] [mh|pg] [iters]

One process per GPU; observations and the trace are replicated, the rows of each class sweep are
block-partitioned, the exchange runs over RCCL (backend "nccl").  Also runs as a plain single process.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pclean_amd import experiments as ex
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import initialize_trace, run_inference
from pclean_amd.model import LoweredModel
from pclean_amd.parallel import Comm
from pclean_amd.trace import Trace


def main(program, particles=2, mh=True, iters=2, seed=0):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = None
    if world > 1 or os.environ.get("PCLEAN_FORCE_DIST"):  # PCLEAN_FORCE_DIST: exercise RCCL with one rank
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device = f"cuda:{local_rank}"
    comm = Comm(device)
    dirty, clean = getattr(ex, f"{program}_data")()
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)  # same permutation on every rank
    m = ex.hospital_model(ex.possibilities_of(dirty)) if program == "hospital" else getattr(ex, f"{program}_model")(dirty)
    lw = LoweredModel(m, getattr(ex, f"{program}_query")(m), dirty)
    obs = lw.encode_observations(dirty)
    eng = Engine(lw, obs, device=local_rank)
    tr = Trace(lw, obs.shape[1], seed)
    cfg = InferenceConfig(iters, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500)
    t0 = time.time()
    initialize_trace(eng, tr, cfg, seed, max_batch=512, comm=comm)
    run_inference(eng, tr, cfg, seed, comm=comm)
    dt = time.time() - t0
    tr.check_consistency()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    if comm.rank == 0:
        print(f"{program}: {world} rank(s), {obs.shape[1]} rows, init + {iters} iterations in {dt:.2f}s", acc, flush=True)
    eng.close()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "flights", particles=int(a[1]) if len(a) > 1 else 2, mh=(a[2] == "mh") if len(a) > 2 else True,
         iters=int(a[3]) if len(a) > 3 else 2)

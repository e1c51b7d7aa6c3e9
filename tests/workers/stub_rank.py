"""Stand-in rank process for the launcher test: gloo instead of RCCL, no GPU.  Every rank joins the process group the
launcher's environment describes, contributes a 1 to an all_reduce, and rank 0 prints the JSON line the launcher relays."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1)
    dist.all_reduce(ones)
    print("rank %d stdout (must not reach the parent's stdout unless rank 0)" % rank, file=sys.stderr if rank == 0 else sys.stdout)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "rccl_ranks": int(ones.item()), "local_rank": os.environ["LOCAL_RANK"],
                          "master": os.environ["MASTER_ADDR"], "args": sys.argv[1:]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

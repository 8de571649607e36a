"""One rank of a CPU-only world for tests/test_launch.py (gloo): `python tests/launch_worker.py <mode>`.

  ok     every rank joins a barrier and exits 0; rank 0 prints one line
  hang   rank 1 never enters the barrier (it sleeps): the others wait in it for ever -- what a rank whose GPU stopped
         answering looks like to its peers
  fail   rank 1 writes to stderr and exits 7 while rank 0 waits in the barrier
"""
import os
import sys
import time

import torch.distributed as dist


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.stderr.write("rank %d of %d is up (pid %d)\n" % (rank, world, os.getpid()))
    sys.stderr.flush()
    if mode == "hang" and rank == 1:
        time.sleep(3600)
    if mode == "fail" and rank == 1:
        sys.stderr.write("rank 1: simulated failure\n")
        sys.stderr.flush()
        os._exit(7)
    dist.barrier()
    if rank == 0:
        print('{"ok": true, "world": %d}' % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

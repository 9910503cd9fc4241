"""Child-process probe: can this machine capture an RCCL all-reduce into a HIP graph with the given world size?

Run as ``python -m dctn_amd._probe_allreduce_capture`` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in the environment (a port of its own: the children of all ranks form their own process group).
Exit code 0 = the captured collective replays and gives the right sum; anything else = do not capture.

Why a child: a capture that fails leaves the HIP runtime of the process in a state in which later collectives
return "invalid argument" (NOTEBOOK.md, round-2 section 7) - so the attempt is made where a failure costs nothing.  The
parent starts this before (or regardless of) its own GPU work and never replaces itself with it.
"""
import os
import sys


def main() -> int:
    import torch
    import torch.distributed as dist

    rank = int(os.environ["RANK"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ["WORLD_SIZE"])
    numel = int(os.environ.get("DCTN_PROBE_NUMEL", "29098"))
    dtype = getattr(torch, os.environ.get("DCTN_PROBE_DTYPE", "bfloat16"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        buf = torch.full((numel,), float(rank + 1), dtype=dtype, device=dev)
        dist.all_reduce(buf)                       # communicator comes up outside the capture
        torch.cuda.synchronize(dev)
        want = float(sum(range(1, world + 1)))
        static = torch.empty_like(buf)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=torch.cuda.Stream(dev), capture_error_mode="thread_local"):
            dist.all_reduce(static, op=dist.ReduceOp.SUM)
        for _ in range(3):
            static.fill_(float(rank + 1))
            graph.replay()
            torch.cuda.synchronize(dev)
            if not bool((static.float() == want).all()):
                return 4
        return 0
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    try:
        code = main()
    except BaseException as e:   # noqa: BLE001 - the exit code is the whole interface
        print(f"[probe] all-reduce capture failed: {type(e).__name__}: {e}", file=sys.stderr)
        code = 3
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)

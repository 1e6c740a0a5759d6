"""Batch-axis sharding of independent iLQR problems over the GPUs of one node (one process per GPU).

The reference has no multi-GPU path (no NCCL/MPI anywhere); its natural shard axis is the batch of
independent problems (MPC rollouts, BASELINE configs[3]).  Every rank owns problems {r : r % world == rank} with all
A alphas and all M segments local, so a DDP sweep needs NO collective.  The only exchanges are per POLL, not per sweep:
  * allgather_costs   -- the per-problem cost trace column(s), so every rank can select the globally best rollout;
  * all_done          -- "has every problem on every rank exited?" (a max-reduce of one integer).
torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests.
"""
import os
from dataclasses import dataclass

import numpy as np


@dataclass
class ShardCtx:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: str = ""
    device: int = 0          # HIP device ordinal of this rank: LOCAL_RANK, unless PDDP_FORCE_DEVICE overrides it (single-GPU dry runs of the N > 1 path)


def owned_problems(total, rank, world):
    """Round-robin ownership (mode R): global problem ids of `rank`."""
    return list(range(rank, total, world))


def init_from_env(n_gpus_flag=1, backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run; a bare `python bench.py` is world 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = int(os.environ.get("PDDP_FORCE_DEVICE", local_rank))
    if world == 1:
        return ShardCtx(0, 1, local_rank, "", device)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("PDDP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # PDDP_DIST_BACKEND=gloo: dry run of the N > 1 path on one GPU
    if backend == "nccl":
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", device)         # binds the communicator to this rank's GPU (no guessing in barrier())
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except TypeError:                                          # older torch without device_id
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return ShardCtx(rank, world, local_rank, backend, device)


def _dev(ctx):
    import torch
    return torch.device("cuda", ctx.device) if ctx.backend == "nccl" else torch.device("cpu")


def barrier(ctx):
    if ctx.world > 1:
        import torch.distributed as dist
        if ctx.backend == "nccl":
            dist.barrier(device_ids=[ctx.device])
        else:
            dist.barrier()


def max_over_ranks(ctx, value):
    if ctx.world == 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=_dev(ctx))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_done(ctx, done_local):
    """True when every problem on every rank has met an exit condition."""
    flag = 0 if bool(np.all(np.asarray(done_local) != 0)) else 1
    if ctx.world == 1:
        return flag == 0
    import torch
    import torch.distributed as dist
    t = torch.tensor([flag], dtype=torch.int32, device=_dev(ctx))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item()) == 0


def allgather_costs(ctx, jout, batch, stride, last_col):
    """Gather columns 0 and `last_col` of every rank's cost trace Jout[batch][stride] -> array [world*batch][2] in GLOBAL
    problem order (problem g lives on rank g % world at local index g // world).
    `jout` is a pyddp DeviceArray (HBM, zero-copy through __cuda_array_interface__) or a host numpy array (gloo tests)."""
    import torch
    if isinstance(jout, np.ndarray):
        local = torch.from_numpy(np.ascontiguousarray(jout)).reshape(batch, stride)
    else:
        local = torch.as_tensor(jout, device=torch.device("cuda", ctx.device)).reshape(batch, stride)       # zero-copy view of the solver's array in HBM
    cols = local[:, [0, last_col]].contiguous()
    if ctx.world == 1:
        return cols.cpu().numpy()
    import torch.distributed as dist
    cols = cols.to(_dev(ctx))
    gathered = torch.empty((ctx.world * batch, 2), dtype=cols.dtype, device=cols.device)     # rank-major concatenation
    dist.all_gather_into_tensor(gathered, cols)
    gathered = gathered.reshape(ctx.world, batch, 2)
    # [rank][local][2] -> global order g = local * world + rank
    return gathered.permute(1, 0, 2).reshape(ctx.world * batch, 2).cpu().numpy()


def allgather_cost_table(ctx, J_local, batch, A):
    """The per-iteration cost table (SURVEY.md 8(e) mode R; include/pddp.h pddp_comm_cost_table_begin / _end is the native twin): J[batch][A] of every rank's last line
    search -> [world * batch][A] in GLOBAL problem order (problem g lives on rank g % world at local index g // world).  J_local: DeviceArray or numpy."""
    import torch
    if isinstance(J_local, np.ndarray):
        local = torch.from_numpy(np.ascontiguousarray(J_local, dtype=np.float64)).reshape(batch, A)
    else:
        local = torch.as_tensor(J_local, device=torch.device("cuda", ctx.device)).reshape(batch, A).to(torch.float64)
    if ctx.world == 1:
        return local.cpu().numpy()
    import torch.distributed as dist
    local = local.contiguous().to(_dev(ctx))
    gathered = torch.empty((ctx.world * batch, A), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local)
    return gathered.reshape(ctx.world, batch, A).permute(1, 0, 2).reshape(ctx.world * batch, A).cpu().numpy()


def best_rollout(costs_global):
    """Index (global problem id) and cost of the best rollout -- what an MPC caller picks its control from."""
    final = np.asarray(costs_global)[:, -1]
    g = int(np.argmin(final))
    return g, float(final[g])


def finalize(ctx):
    if ctx.world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def solve_sharded(ctx, make_solver, x0_all, u0_all, xg_all, poll_every=8, max_sweeps=100000, **load_kw):
    """MPC-rollout mode (BASELINE configs[3]): `total` independent problems, rank g solves {r : r % world == g} with a solver
    created by make_solver(batch); sweeps run with no collective; every `poll_every` sweeps the ranks agree on "all done?"
    (one max-reduce) and at the end exchange the cost table (one all-gather).
    Returns dict(costs [total][2] = (J_initial, J_final) in global order, best = (problem id, cost), local = the rank's store())."""
    import numpy as np
    total = len(x0_all)
    mine = owned_problems(total, ctx.rank, ctx.world)
    per_rank = (total + ctx.world - 1) // ctx.world
    if total % ctx.world:
        raise ValueError("the number of problems must be a multiple of the number of ranks (pad the batch)")
    s = make_solver(per_rank)
    s.load(np.stack([x0_all[i] for i in mine]), np.stack([u0_all[i] for i in mine]), np.stack([xg_all[i] for i in mine]), **load_kw)
    sweeps = 0
    while sweeps < max_sweeps:
        s.iterate(poll_every)
        sweeps += poll_every
        done, iters = s.status()
        if all_done(ctx, done):
            break
    out = s.store()
    done, iters = s.status()
    stride = s.cfg.max_iter + 2
    J = out["Jout"].reshape(per_rank, stride)
    first_last = np.stack([J[:, 0], J[np.arange(per_rank), iters]], axis=1).astype(np.float64)
    padded = np.zeros((per_rank, stride)); padded[:, 0] = first_last[:, 0]; padded[:, 1] = first_last[:, 1]
    costs = allgather_costs(ctx, padded, per_rank, stride, 1)
    return dict(costs=costs, best=best_rollout(costs), local=out, sweeps=sweeps, iters=iters)

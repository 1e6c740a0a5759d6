"""Multi-GPU path on CPU: world_size 2, gloo.  The per-rank solver is the host emulation of the kernels (test tool); what is
under test is the host logic of pyddp.shard -- round-robin ownership, no collective per sweep, the exit-flag reduce, the
cost all-gather in global problem order and the best-rollout pick -- against a single-process solve of the same batch."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_hostsim_once():
    """the ranks load the host emulation through backends.hostsim_path(), which runs make: bring the library up to date HERE, before several ranks race to rebuild it"""
    from backends import hostsim_path
    hostsim_path()

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "parallel-ddp_amd"))
import pyddp
from pyddp import shard
from backends import hostsim_path
from oracle_binding import example_inputs
ctx = shard.init_from_env(2, backend="gloo")
assert ctx.world == 2 and ctx.backend == "gloo"
kw = dict(N=32, M=4, A=4, wafr_urdf=1, tol_cost=1e-3, total_time=0.5, max_iter=12)
rng = np.random.default_rng(3)
total = 6
xs, us, gs = [], [], []
for b in range(total):
    x0, u0, xg = example_inputs(4, 32, np.float32, noise=rng.normal(0, 0.01 * (b + 1), (32, 14)))
    xs.append(x0); us.append(u0); gs.append(xg)
path = hostsim_path()
mk = lambda batch: pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=batch, **kw), _lib_path=path)
res = shard.solve_sharded(ctx, mk, xs, us, gs, poll_every=3)
tmax = shard.max_over_ranks(ctx, 1.0 + ctx.rank)
assert tmax == 2.0
# the per-iteration cost table (SURVEY 8(e) mode R): every rank's J[batch][A] of the last line search, on every rank, in global problem order
sl = mk(total // 2)
mine = shard.owned_problems(total, ctx.rank, 2)
sl.load(np.stack([xs[i] for i in mine]), np.stack([us[i] for i in mine]), np.stack([gs[i] for i in mine]))
sl.iterate(2); sl.sync()
table = shard.allgather_cost_table(ctx, sl.get("J"), total // 2, kw["A"])
assert table.shape == (total, kw["A"]) and np.array_equal(table[ctx.rank::2], sl.get("J").reshape(-1, kw["A"]).astype(np.float64))
assert shard.owned_problems(total, ctx.rank, 2) == list(range(ctx.rank, total, 2))
if ctx.rank == 0:
    json.dump(dict(costs=res["costs"].tolist(), best=res["best"], sweeps=res["sweeps"], table=table.tolist()), open(sys.argv[1], "w"))
shard.finalize(ctx)
'''


def test_world_size_2_gloo_matches_single_process(tmp_path):
    out = tmp_path / "res.json"
    script = tmp_path / "worker.py"
    _build_hostsim_once()
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", "29613", str(script), str(out)], env=env, timeout=600)
    import json
    res = json.load(open(out))
    # single-process reference: all 6 problems in one handle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyddp
    from backends import hostsim_path
    from oracle_binding import example_inputs
    kw = dict(N=32, M=4, A=4, wafr_urdf=1, tol_cost=1e-3, total_time=0.5, max_iter=12)
    rng = np.random.default_rng(3)
    xs, us, gs = [], [], []
    for b in range(6):
        x0, u0, xg = example_inputs(4, 32, np.float32, noise=rng.normal(0, 0.01 * (b + 1), (32, 14)))
        xs.append(x0); us.append(u0); gs.append(xg)
    path = hostsim_path()
    s = pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=6, **kw), _lib_path=path)
    one = s.solve(np.stack(xs), np.stack(us), np.stack(gs))
    J0 = one["Jout"][:, 0]; Jf = one["Jout"][np.arange(6), one["iters"]]
    costs = np.asarray(res["costs"])
    assert np.array_equal(costs[:, 0].astype(np.float32), J0) and np.array_equal(costs[:, 1].astype(np.float32), Jf)
    assert res["best"][0] == int(np.argmin(Jf))
    assert res["sweeps"] % 3 == 0
    # the gathered cost table of two sweeps == the single-process handle's J[6][A] after two sweeps, row for row
    s2 = pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=6, **kw), _lib_path=path)
    s2.load(np.stack(xs), np.stack(us), np.stack(gs)); s2.iterate(2); s2.sync()
    assert np.array_equal(np.asarray(res["table"]), s2.get("J").reshape(6, kw["A"]).astype(np.float64))


WORKER4 = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "parallel-ddp_amd"))
import pyddp
from pyddp import shard
from backends import hostsim_path
from oracle_binding import example_inputs
ctx = shard.init_from_env(4, backend="gloo")
assert ctx.world == 4
kw = dict(N=16, M=2, A=2, wafr_urdf=1, tol_cost=1e-3, total_time=0.25, max_iter=6)
rng = np.random.default_rng(4)
path = hostsim_path()
mk = lambda batch: pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=batch, **kw), _lib_path=path)
def problems(total):
    out = [example_inputs(4, 16, np.float32, noise=rng.normal(0, 0.01 * (b + 1), (16, 14))) for b in range(total)]
    return [p[0] for p in out], [p[1] for p in out], [p[2] for p in out]
rejected = False
try:
    shard.solve_sharded(ctx, mk, *problems(6), poll_every=2)          # 6 problems do not split over 4 ranks
except ValueError as e:
    rejected = "multiple of the number of ranks" in str(e)
assert rejected
res = shard.solve_sharded(ctx, mk, *problems(8), poll_every=2)       # 2 per rank
assert shard.owned_problems(8, ctx.rank, 4) == [ctx.rank, ctx.rank + 4]
assert res["costs"].shape == (8, 2) and (res["costs"][:, 1] <= res["costs"][:, 0] + 3e-3).all()   # (a solve that rejects everything records prevJ = J + 2 TOL_COST, nisInitHelpers.cuh:393)
if ctx.rank == 0:
    json.dump(dict(costs=res["costs"].tolist(), best=res["best"]), open(sys.argv[1], "w"))
shard.finalize(ctx)
'''


def test_world_size_4_gloo_uneven_batch_is_rejected_and_an_even_one_gathers_in_global_order(tmp_path):
    out = tmp_path / "res4.json"
    script = tmp_path / "worker4.py"
    _build_hostsim_once()
    script.write_text(WORKER4 % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                           "--master-port", "29614", str(script), str(out)], env=env, timeout=900)
    import json
    res = json.load(open(out))
    costs = np.asarray(res["costs"])
    assert costs.shape == (8, 2) and res["best"][0] == int(np.argmin(costs[:, 1]))
    assert len(set(np.round(costs[:, 0], 3))) == 8              # eight different problems, each reported once


WORKER8 = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "parallel-ddp_amd")); sys.path.insert(0, %(root)r)
import pyddp, bench
from pyddp import shard
from backends import hostsim_path
ctx = shard.init_from_env(8, backend="gloo")
assert ctx.world == 8
kw = %(kw)r
total, N = 64, kw["N"]
x_all, u_all, g_all = bench.ee_inputs(N, np.random.default_rng(77), total)          # bench.py --workload config3: the 64 rollouts of BASELINE configs[3]
path = hostsim_path()
mk = lambda batch: pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=batch, **kw), _lib_path=path)
mine = shard.owned_problems(total, ctx.rank, 8)
assert mine == list(range(ctx.rank, total, 8)) and len(mine) == 8
res = shard.solve_sharded(ctx, mk, list(x_all), list(u_all), list(g_all), poll_every=2)
sl = mk(8)
sl.load(x_all[mine], u_all[mine], g_all[mine])
sl.iterate(1); sl.sync()
table = shard.allgather_cost_table(ctx, sl.get("J"), 8, kw["A"])
assert table.shape == (total, kw["A"]) and np.array_equal(table[ctx.rank::8], sl.get("J").reshape(-1, kw["A"]).astype(np.float64))
if ctx.rank == 0:
    json.dump(dict(costs=res["costs"].tolist(), best=res["best"], sweeps=res["sweeps"], table=table.tolist()), open(sys.argv[1], "w"))
shard.finalize(ctx)
'''


def test_world_size_8_gloo_the_split_of_baseline_config3(tmp_path):
    """BASELINE configs[3] as the first 8-GPU run will shard it (bench.py --workload config3 --gpus 8; VERDICT r5 task 9): 64 Kuka MPC rollouts with the end-effector cost,
    rank g owns {r : r % 8 == g}, 8 per rank, no data-path collective; the exit poll (all-reduce), the cost all-gather and the per-iteration [64 x A] cost table in GLOBAL problem
    order -- eight gloo processes on the CPU (the per-rank solver is the host emulation of the kernels) against ONE process solving all 64."""
    kw = dict(N=32, M=4, A=4, ee_cost=1, wafr_urdf=1, mpc_mode=1, tol_cost=1e-5, total_time=0.5, ignore_max_rho_exit=0, max_iter=4)      # (a shorter horizon and fewer step sizes than the bench: eight emulated ranks share this machine's cores)
    out = tmp_path / "res8.json"
    script = tmp_path / "worker8.py"
    _build_hostsim_once()
    script.write_text(WORKER8 % dict(root=ROOT, kw=kw))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                           "--master-port", "29618", str(script), str(out)], env=env, timeout=1500)
    import json
    res = json.load(open(out))
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
    import bench
    import pyddp
    from backends import hostsim_path
    x_all, u_all, g_all = bench.ee_inputs(kw["N"], np.random.default_rng(77), 64)
    path = hostsim_path()
    s = pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=64, **kw), _lib_path=path)
    one = s.solve(x_all, u_all, g_all)
    J0 = one["Jout"][:, 0]; Jf = one["Jout"][np.arange(64), one["iters"]]
    costs = np.asarray(res["costs"])
    assert costs.shape == (64, 2)
    assert np.array_equal(costs[:, 0].astype(np.float32), J0) and np.array_equal(costs[:, 1].astype(np.float32), Jf)       # global problem order, bit for bit the single-process batch
    assert res["best"][0] == int(np.argmin(Jf)) and res["sweeps"] % 2 == 0
    s2 = pyddp.Solver(pyddp.default_config(4, _lib_path=path, batch=64, **kw), _lib_path=path)
    s2.load(x_all, u_all, g_all); s2.iterate(1); s2.sync()
    assert np.array_equal(np.asarray(res["table"]), s2.get("J").reshape(64, kw["A"]).astype(np.float64))


@pytest.mark.gpu
def test_native_cost_table_exchange_beside_the_solver_stream_world_of_one():
    """pddp_comm_cost_table_begin / _end: the [B x A] table of the last line search leaves on the communicator's own stream behind an event; the solver iterates on meanwhile
    and the table is the one of the sweep it was taken behind (not of a later one)."""
    import pyddp
    from oracle_binding import example_inputs
    kw = dict(N=32, M=4, A=8, wafr_urdf=1, tol_cost=0.0, total_time=0.5, max_iter=20)
    B = 5
    s = pyddp.Solver(pyddp.default_config(4, batch=B, **kw))
    rng = np.random.default_rng(6)
    xs, us, gs = zip(*[example_inputs(4, 32, np.float32, noise=rng.normal(0, 0.01 * (b + 1), (32, 14))) for b in range(B)])
    s.load(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    comm = pyddp.Comm(0, 1, 0)
    s.iterate(3)
    comm.cost_table_begin(s)                 # behind sweep 3 ...
    s.iterate(4)                             # ... while four more sweeps are enqueued
    table = comm.cost_table_end()
    s.sync()
    later = s.get("J").reshape(B, 8).astype(np.float64)
    s2 = pyddp.Solver(pyddp.default_config(4, batch=B, **kw))
    s2.load(np.concatenate(xs), np.concatenate(us), np.concatenate(gs)); s2.iterate(3); s2.sync()
    assert table.shape == (B, 8) and np.array_equal(table, s2.get("J").reshape(B, 8).astype(np.float64))
    assert not np.array_equal(table, later)
    with pytest.raises(pyddp.PddpError):
        comm.cost_table_end()                # nothing in flight
    comm.close(); s.close(); s2.close()


@pytest.mark.gpu
def test_native_rccl_exchanges_of_the_c_abi_world_of_one():
    """include/pddp.h "multi-GPU": pddp_comm_init / pddp_comm_all_done / pddp_comm_allgather_costs / pddp_comm_allreduce_max with RCCL on the solver's
    stream.  The GPU box has one device, so the communicator has one rank: the collectives run for real (ncclAllReduce, ncclAllGather) and must
    return exactly what the handle holds."""
    import pyddp
    from oracle_binding import example_inputs
    kw = dict(N=32, M=4, A=4, wafr_urdf=1, tol_cost=1e-3, total_time=0.5, max_iter=12)
    B = 6
    s = pyddp.Solver(pyddp.default_config(4, batch=B, **kw))
    rng = np.random.default_rng(5)
    xs, us, gs = zip(*[example_inputs(4, 32, np.float32, noise=rng.normal(0, 0.01 * (b + 1), (32, 14))) for b in range(B)])
    s.load(np.concatenate(xs), np.concatenate(us), np.concatenate(gs))
    c = pyddp.Comm(0, 1, 0)
    assert not c.all_done(s)
    polls = 0
    while not c.all_done(s) and polls < 100:
        s.iterate(4); polls += 1
    done, iters = s.status()
    assert done.all() and c.all_done(s)
    costs = c.allgather_costs(s)
    out = s.store()
    assert costs.shape == (B, 2)
    np.testing.assert_array_equal(costs[:, 0], out["Jout"][:, 0].astype(np.float64))
    np.testing.assert_array_equal(costs[:, 1], out["Jout"][np.arange(B), iters].astype(np.float64))
    assert c.max_over_ranks(3.5) == 3.5
    c.close(); s.close()

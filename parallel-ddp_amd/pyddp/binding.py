import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PLANT_DIMS = {1: (1, 2, 1), 2: (2, 4, 1), 3: (6, 12, 4), 4: (7, 14, 7)}  # npos, n, m (config.cuh:24-46)
PHASE_BP, PHASE_FP, PHASE_LS, PHASE_NIS, PHASE_INIT_NIS, PHASE_INIT_COST, PHASE_BP_COOP, PHASE_BP_FUSED, PHASE_SWEEP_FUSED, PHASE_ROLLOUT = range(10)


class PddpError(RuntimeError):
    pass


# pddp_kernel_selection (include/pddp.h): value names per field, in the order of the header (index + 1 = the C value; 0 / None = the library's choice)
KERNEL_NAMES = {"bp": ("mx", "lg", "coop", "wide"), "fp": ("tl", "lg", "coop", "tl2", "tl4"), "sweep": ("alpha", "st", "wg", "maps"), "ls": ("many", "wg"), "ab": ("full",),
                "cf": ("ts", "coop"), "cf_bp": ("ts", "coop", "gl", "gl32", "cl", "mq"), "cf_fp": ("ts", "coop", "cf"), "cf_nis": ("ts", "coop", "gl", "gl8", "kb16", "kb32", "kb64", "kb20")}


class PddpKernelSelection(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("bp", "fp", "sweep", "ls", "ab", "cf", "cf_bp", "cf_fp", "cf_nis")]


def set_kernels(cfg, **names):
    """cfg.kernels from names: set_kernels(cfg, bp="mx", fp="tl").  None / "" / "auto" leave the library's choice."""
    for field, name in names.items():
        if field not in KERNEL_NAMES:
            raise ValueError(f"unknown kernel-selection field {field!r} (one of {sorted(KERNEL_NAMES)})")
        if name in (None, "", "auto"):
            setattr(cfg.kernels, field, 0)
        elif name in KERNEL_NAMES[field]:
            setattr(cfg.kernels, field, KERNEL_NAMES[field].index(name) + 1)
        else:
            raise ValueError(f"kernel selection {field}={name!r}: one of {KERNEL_NAMES[field]}")
    return cfg


class PddpConfig(C.Structure):
    """pddp_config (include/pddp.h): the reference's config.cuh macros as a run-time record."""
    _fields_ = [
        ("plant", C.c_int), ("dtype", C.c_int), ("N", C.c_int), ("M", C.c_int), ("A", C.c_int), ("integrator", C.c_int),
        ("batch", C.c_int), ("max_iter", C.c_int), ("wafr_urdf", C.c_int), ("mpc_mode", C.c_int),
        ("ignore_max_rho_exit", C.c_int), ("device", C.c_int), ("use_graph", C.c_int),
        ("total_time", C.c_double), ("alpha_base", C.c_double), ("rho_init", C.c_double), ("max_defect", C.c_double),
        ("tol_cost", C.c_double), ("exp_red_min", C.c_double), ("exp_red_max", C.c_double),
        ("Q1", C.c_double), ("Q2", C.c_double), ("R", C.c_double), ("QF1", C.c_double), ("QF2", C.c_double),
        ("ee_cost", C.c_int), ("ee_cost_shift", C.c_int),
        ("Q_EE1", C.c_double), ("Q_EE2", C.c_double), ("QF_EE1", C.c_double), ("QF_EE2", C.c_double), ("R_EE", C.c_double),
        ("Q_xEE", C.c_double), ("QF_xEE", C.c_double), ("Q_xdEE", C.c_double), ("QF_xdEE", C.c_double), ("ee_on_link_z", C.c_double),
        ("ee_initial_cost_fix", C.c_int),
        ("use_finite_diff", C.c_int), ("finite_diff_epsilon", C.c_double),
        ("boundary_cost_to_go_only", C.c_int),
        ("use_smooth_abs", C.c_int),
        ("smooth_abs_alpha", C.c_double),
        ("use_limits", C.c_int),
        ("ee_type", C.c_int),
        ("kernels", PddpKernelSelection),
    ]


class PddpState(C.Structure):
    _fields_ = [("rho", C.c_double), ("drho", C.c_double), ("prevJ", C.c_double), ("dJ", C.c_double), ("z", C.c_double),
                ("iter", C.c_int), ("alphaIndex", C.c_int), ("ignore_defect", C.c_int), ("accepted", C.c_int),
                ("done", C.c_int), ("cur", C.c_int), ("cur2", C.c_int), ("bp_retries", C.c_int), ("pw", C.c_int)]


class DeviceArray:
    """A 1-D device buffer owned by a Solver (valid until Solver.close())."""

    def __init__(self, ptr, count, dtype):
        self.ptr, self.count, self.dtype = ptr, count, dtype
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": dtype.str, "data": (ptr, False), "version": 2}


def algorithmic_bytes(n, m, N, A, M, s, ee_cost=False):
    """Algorithmic HBM bytes of ONE problem per launch of each sweep kernel: every array read/written once per phase
    that consumes/produces it, in the reference's phase decomposition (DESIGN.md "roofline").
    k_fp = sweep + rollout + cost of all A candidates (the three reference kernels it fuses).  ee_cost: the reference accumulates the
    cost inside the rollout (fpHelpers.cuh:259-265), so there is no separate pass over x, u."""
    nm = n + m
    bp = (n * nm + nm * nm + nm) * (N - 1) + n * n + n + (M - 1) * (n * n + 4 * n) + (2 * n * n + n * m + 2 * n + m) * (N - 1)
    sweep = ((n * n + n) * (N - 1) + 3 * n * N + n * (M - 1)) if M > 1 else 0
    sim = (n * m + m) * (N - 1) + 3 * n * N + 2 * m * (N - 1) + n * (M - 1)
    cost = 0 if ee_cost else n * N + m * (N - 1)
    nis = 2 * (n * N + m * (N - 1)) + n * nm * (N - 1) + (nm * nm + nm) * N + 2 * (n * n + n) * N + A * (2 * n + m) * N + (2 * n + m) * N + n * N
    return {"k_bp": bp * s, "k_fp": A * (sweep + sim + cost) * s, "k_ls": (3 * A + 2 * M + 16) * s, "k_nis": nis * s}


def algorithmic_bytes_per_kernel(n, m, N, A, M, s):
    """The same accounting split by KERNEL of the large-batch arm selection (joint-space cost): backward pass; linear sweep of the A candidates;
    rollouts + cost + defect of the A candidates; line search; the copies of nextIterationSetupGPU (winner -> all slots, xp/up/dp, Pp/pp: what the
    winner kernel and the index flips stand for); the derivative kernels of nextIterationSetupGPU (AB, H, g written, x, u read)."""
    nm = n + m
    tot = algorithmic_bytes(n, m, N, A, M, s)
    sweep = ((n * n + n) * (N - 1) + 3 * n * N + n * (M - 1)) if M > 1 else 0
    derivs = 2 * (n * N + m * (N - 1)) + n * nm * (N - 1) + (nm * nm + nm) * N
    return {"k_bp": tot["k_bp"], "k_sweep": A * sweep * s, "k_sim": tot["k_fp"] - A * sweep * s, "k_ls": tot["k_ls"],
            "k_nis_copies": tot["k_nis"] - derivs * s, "k_nis_derivs": derivs * s}


def library_path():
    return os.path.join(os.path.dirname(_HERE), "lib", "libpddp.so")


def build_id(lib_path=None):
    """What a set of profile counters has to be bound to (VERDICT r5 task 7): `sources` = sha256 over the library's device / host sources (csrc/*.hip, csrc/*.hpp,
    include/pddp.h, the Makefile's flags) in the tree this module runs from, `library` = sha256 of the shared object itself.  tools/make_traffic_json.py and
    tools/make_rows_traffic.py store both next to the counters; bench.py attaches counter traffic to a run only when `sources` matches the tree it is running from."""
    import glob
    import hashlib
    pkg = os.path.dirname(_HERE)
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.hpp")) + glob.glob(os.path.join(pkg, "csrc", "*.h")))
    files += [os.path.join(os.path.dirname(pkg), "include", "pddp.h"), os.path.join(pkg, "Makefile")]
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(b"\0")
        h.update(open(f, "rb").read())
    path = lib_path or library_path()
    lib = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None
    return {"sources": h.hexdigest()[:16], "library": lib}


_LIBS = {}


def _load(path):
    if path not in _LIBS:
        if not os.path.exists(path):
            raise PddpError(f"{path} not found: build it with `make -C parallel-ddp_amd` (there is no CPU fallback)")
        lib = C.CDLL(path)
        lib.pddp_last_error.restype = C.c_char_p
        _LIBS[path] = lib
    return _LIBS[path]


def default_config(plant, _lib_path=None, kernels=None, **kw):
    """pddp_default_config + overrides by field name.  kernels: {"bp": "mx", "fp": "tl", ...} pins kernel families (pddp_kernel_selection); default: the library chooses."""
    lib = _load(_lib_path or library_path())
    c = PddpConfig()
    rc = lib.pddp_default_config(C.byref(c), int(plant))
    if rc:
        raise PddpError(lib.pddp_last_error().decode())
    if kernels:
        set_kernels(c, **kernels)
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Comm:
    """The C ABI's multi-GPU communicator (include/pddp.h "multi-GPU"): RCCL collectives on the solvers' own streams.  One per process / GPU.
    Rendezvous through a file: rank 0 writes the unique id, the others wait for it (any other transport of the 128 bytes works as well)."""

    def __init__(self, rank, world, device, id_path=None, timeout_s=120.0, _lib_path=None):
        import time
        self.lib = _load(_lib_path or library_path())
        self.rank, self.world = rank, world
        blob = C.create_string_buffer(128)
        if world > 1 and id_path is None:
            raise PddpError("Comm: a world > 1 needs a rendezvous file path shared by the ranks")
        if rank == 0:
            if self.lib.pddp_comm_unique_id(blob):
                raise PddpError(self.lib.pddp_last_error().decode())
            if id_path:
                tmp = id_path + ".tmp"
                with open(tmp, "wb") as f:
                    f.write(blob.raw)
                os.replace(tmp, id_path)
        else:
            t0 = time.time()
            while not os.path.exists(id_path):
                if time.time() - t0 > timeout_s:
                    raise PddpError("Comm: timed out waiting for the rendezvous file " + id_path)
                time.sleep(0.01)
            blob = C.create_string_buffer(open(id_path, "rb").read(), 128)
        self.h = C.c_void_p()
        if self.lib.pddp_comm_init(C.byref(self.h), int(rank), int(world), blob, int(device)):
            raise PddpError(self.lib.pddp_last_error().decode())

    def _chk(self, rc):
        if rc:
            raise PddpError(f"pddp error {rc}: {self.lib.pddp_last_error().decode()}")

    def all_done(self, solver):
        flag = C.c_int(0)
        self._chk(self.lib.pddp_comm_all_done(self.h, solver.h, C.byref(flag)))
        return bool(flag.value)

    def allgather_costs(self, solver):
        out = np.zeros((self.world * solver.cfg.batch, 2), np.float64)
        self._chk(self.lib.pddp_comm_allgather_costs(self.h, solver.h, _p(out)))
        return out

    def cost_table_begin(self, solver):
        """pddp_comm_cost_table_begin: enqueue the all-gather of the last line search's J[batch][A] beside the solver's stream (returns at once)."""
        self._table_shape = (self.world * solver.cfg.batch, solver.cfg.A)
        self._chk(self.lib.pddp_comm_cost_table_begin(self.h, solver.h))

    def cost_table_end(self):
        out = np.zeros(self._table_shape, np.float64)
        self._chk(self.lib.pddp_comm_cost_table_end(self.h, _p(out)))
        return out

    def max_over_ranks(self, value):
        v = C.c_double(float(value))
        self._chk(self.lib.pddp_comm_allreduce_max(self.h, C.byref(v)))
        return v.value

    def barrier(self):
        self.max_over_ranks(0.0)

    def close(self):
        if self.h:
            self.lib.pddp_comm_destroy(self.h)
            self.h = C.c_void_p()


class Solver:
    """One pddp handle = the buffers of allocateMemory_GPU for `batch` problems + the solver kernels."""

    def __init__(self, cfg, _lib_path=None):
        self.path = _lib_path or library_path()
        if _lib_path is None and "hostsim" in self.path:
            raise PddpError("the product binding never loads the host emulation")
        self.lib = _load(self.path)
        self.cfg = cfg
        self.dtype = np.dtype(np.float32 if cfg.dtype == 0 else np.float64)
        if cfg.plant in PLANT_DIMS:
            self.npos, self.n, self.m = PLANT_DIMS[cfg.plant]
        else:                                      # a user plant (make user PLANT_POLICY=...): the library knows its sizes
            self.n, self.m = self.lib.pddp_state_size(cfg.plant), self.lib.pddp_control_size(cfg.plant)
            self.npos = self.n // 2
        self.h = C.c_void_p()
        self._chk(self.lib.pddp_create(C.byref(cfg), C.byref(self.h)))

    def _chk(self, rc):
        if rc:
            raise PddpError(f"pddp error {rc}: {self.lib.pddp_last_error().decode()}")

    def close(self):
        if self.h:
            self.lib.pddp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def arr(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    # ---- runiLQR_GPU pieces
    def load(self, x0, u0, xGoal, clear_vars=1, ignore_first_defect=1, forward_rollout=0, KT0=None, P0=None, p0=None, d0=None):
        """loadVarsGPU + initAlgGPU (pddp_load_ex): warm-start arrays are used when clear_vars == 0."""
        B, N = self.cfg.batch, self.cfg.N
        x0, u0, xGoal = self.arr(x0), self.arr(u0), self.arr(xGoal)
        assert x0.size == B * N * self.n and u0.size == B * N * self.m and xGoal.size == B * self.n
        ws = [None if a is None else self.arr(a) for a in (KT0, P0, p0, d0)]
        self._chk(self.lib.pddp_load_ex(self.h, _p(x0), _p(u0), _p(xGoal), _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]),
                                        int(forward_rollout), int(clear_vars), int(ignore_first_defect)))

    def iterate(self, sweeps=1):
        self._chk(self.lib.pddp_iterate(self.h, int(sweeps)))

    def sync(self):
        self._chk(self.lib.pddp_sync(self.h))

    def status(self):
        B = self.cfg.batch
        done, iters = np.zeros(B, np.int32), np.zeros(B, np.int32)
        self._chk(self.lib.pddp_status(self.h, _p(done), _p(iters)))
        return done, iters

    def store(self):
        B, N, n, m, mi = self.cfg.batch, self.cfg.N, self.n, self.m, self.cfg.max_iter
        out = dict(x=np.zeros((B, N, n), self.dtype), u=np.zeros((B, N, m), self.dtype), KT=np.zeros((B, N, m, n), self.dtype),
                   Jout=np.zeros((B, mi + 2), self.dtype), alphaOut=np.zeros((B, mi + 2), np.int32), dmax=np.zeros(B, self.dtype))
        self._chk(self.lib.pddp_store(self.h, _p(out["x"]), _p(out["u"]), _p(out["KT"]), _p(out["Jout"]), _p(out["alphaOut"]), _p(out["dmax"])))
        return out

    def solve(self, x0, u0, xGoal, clear_vars=1, ignore_first_defect=1, max_sweeps=None, chunk=8, **load_kw):
        """load -> iterate until every problem has exited -> store (the shape of runiLQR_GPU)."""
        self.load(x0, u0, xGoal, clear_vars, ignore_first_defect, **load_kw)
        limit = max_sweeps if max_sweeps is not None else 4 * (self.cfg.max_iter + 2) + 250
        sweeps = 0
        while sweeps < limit:
            self.iterate(chunk)
            sweeps += chunk
            done, iters = self.status()
            if done.all():
                break
        out = self.store()
        out["done"], out["iters"], out["sweeps"] = done, iters, sweeps
        return out

    def mpc_solve(self, xActual, xGoal, shift, clear_vars=0, full_rollout=1, ignore_first_defect=1, max_iter=None, time_budget_ms=0.0, poll_every=4):
        """runiLQR_MPC_GPU (pddp_mpc_solve): warm start from the handle's previous solution shifted by `shift` knots and rolled out
        from the measured state, iterate, fall back to the shifted previous solution when no step was taken."""
        B, N, n, m, mi = self.cfg.batch, self.cfg.N, self.n, self.m, self.cfg.max_iter
        xActual, xGoal = self.arr(xActual), self.arr(xGoal)
        shift = np.ascontiguousarray(np.broadcast_to(np.asarray(shift, np.int32), (B,)))
        out = dict(x=np.zeros((B, N, n), self.dtype), u=np.zeros((B, N, m), self.dtype), KT=np.zeros((B, N, m, n), self.dtype),
                   Jout=np.zeros((B, mi + 2), self.dtype), alphaOut=np.zeros((B, mi + 2), np.int32), success=np.zeros(B, np.int32), iters=np.zeros(B, np.int32))
        self.lib.pddp_mpc_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.pddp_mpc_solve(self.h, _p(xActual), _p(xGoal), _p(shift), int(clear_vars), int(full_rollout), int(ignore_first_defect),
                                          int(max_iter if max_iter is not None else mi), float(time_budget_ms), int(poll_every),
                                          _p(out["x"]), _p(out["u"]), _p(out["KT"]), _p(out["Jout"]), _p(out["alphaOut"]), _p(out["success"]), _p(out["iters"])))
        return out

    def solve_timed(self, x0, u0, xGoal, clear_vars=1, ignore_first_defect=1):
        """pddp_solve: the whole runiLQR_GPU call in C (load, init, sweeps until every problem exits, store), with the
        reference's two timers: ms_total (*tTime) and ms_init (*initTime); ms_loop = their difference."""
        B, mi = self.cfg.batch, self.cfg.max_iter
        x0, u0, xGoal = self.arr(x0).copy(), self.arr(u0).copy(), self.arr(xGoal)
        Jout, aout = np.zeros((B, mi + 2), self.dtype), np.zeros((B, mi + 2), np.int32)
        times = (C.c_double * 2)()
        self._chk(self.lib.pddp_solve(self.h, _p(x0), _p(u0), _p(xGoal), _p(Jout), _p(aout), int(clear_vars), int(ignore_first_defect), times))
        done, iters = self.status()
        return dict(x=x0.reshape(B, self.cfg.N, self.n), u=u0.reshape(B, self.cfg.N, self.m), Jout=Jout, alphaOut=aout, done=done,
                    iters=int(iters[0]) if B == 1 else iters, ms_total=times[0], ms_init=times[1], ms_loop=times[0] - times[1],
                    J_final=float(Jout[0][iters[0]]))

    def solve_phase_timed(self, x0, u0, xGoal, clear_vars=1, ignore_first_defect=1, poll_every=8):
        """pddp_solve_ex with per-iteration phase timers (HIP events per kernel, like the reference's bpTime[] / simTime[] / nisTime[], DDPWrappers.cuh:54-105):
        phase_ms[5][max_iter + 2] = backward pass, sweep + rollouts, line search, next-iteration setup of every iteration, and the linear sweep's own kernel (a part of row 1) (batch 1)."""
        B, mi = self.cfg.batch, self.cfg.max_iter
        x0, u0, xGoal = self.arr(x0).copy(), self.arr(u0).copy(), self.arr(xGoal)
        Jout, aout = np.zeros((B, mi + 2), self.dtype), np.zeros((B, mi + 2), np.int32)
        times = (C.c_double * 2)()
        phase = np.zeros((5, mi + 2), np.float64)
        sweeps = C.c_int(0)
        self.lib.pddp_solve_ex.argtypes = [C.c_void_p] * 10 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.pddp_solve_ex(self.h, _p(x0), _p(u0), _p(xGoal), None, None, None, None, _p(Jout), _p(aout), 0, int(clear_vars), int(ignore_first_defect),
                                         int(poll_every), times, _p(phase), C.byref(sweeps)))
        done, iters = self.status()
        return dict(Jout=Jout, alphaOut=aout, iters=iters, phase_ms=phase, ms_total=times[0], ms_init=times[1], sweeps=sweeps.value)

    def simulate(self, x, u, KT, t0_us, elapsed_us, substeps=150, goal_xyz=None, xActual=None):
        """pddp_simulate: the lock-step simulated robot (simulateForward).  Returns (xActual after elapsed_us, average tracking error, failed)."""
        x, u, KT = self.arr(x), self.arr(u), self.arr(KT)
        xa = self.arr(xActual).copy()
        g = None if goal_xyz is None else self.arr(goal_xyz)
        err, failed = C.c_double(0), C.c_int(0)
        self.lib.pddp_simulate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.pddp_simulate(self.h, _p(x), _p(u), _p(KT), float(t0_us), float(elapsed_us), int(substeps), _p(g), _p(xa), C.byref(err), C.byref(failed)))
        return xa, err.value, failed.value

    def ee_pos(self, x):
        """pddp_ee_pos: tool point (x, y, z, roll, pitch, yaw) of states [count][n]."""
        x = self.arr(x).reshape(-1, self.n)
        out = np.zeros((x.shape[0], 6), self.dtype)
        self._chk(self.lib.pddp_ee_pos(self.h, x.shape[0], _p(x), _p(out)))
        return out

    def set_cost(self, Q1, Q2, R, QF1, QF2):
        """pddp_set_cost: joint-space cost weights for the following loads / solves."""
        self.lib.pddp_set_cost.argtypes = [C.c_void_p] + [C.c_double] * 5
        self._chk(self.lib.pddp_set_cost(self.h, Q1, Q2, R, QF1, QF2))

    def set_cost_ee(self, Q_EE1, Q_EE2, QF_EE1, QF_EE2, R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE):
        """pddp_set_cost_ee: end-effector cost weights for the following loads / solves."""
        self.lib.pddp_set_cost_ee.argtypes = [C.c_void_p] + [C.c_double] * 9
        self._chk(self.lib.pddp_set_cost_ee(self.h, Q_EE1, Q_EE2, QF_EE1, QF_EE2, R_EE, Q_xEE, QF_xEE, Q_xdEE, QF_xdEE))

    # ---- measurement
    def set_benchmark_mode(self, on):
        self._chk(self.lib.pddp_set_benchmark_mode(self.h, int(on)))

    def hbm_calibration(self, nbytes, reps):
        self.lib.pddp_hbm_calibration.argtypes = [C.c_int, C.c_size_t, C.c_int]
        self._chk(self.lib.pddp_hbm_calibration(self.cfg.device, nbytes, reps))

    def time_sweeps(self, sweeps, phases=False):
        tot = C.c_float(0)
        ph = (C.c_float * 4)()
        self._chk(self.lib.pddp_time_sweeps(self.h, int(sweeps), C.byref(tot), ph if phases else None))
        return tot.value, [v for v in ph]

    def time_kernels(self, sweeps):
        """pddp_time_kernels: [(kernel name, average ms per launch)] of one sweep in launch order (HIP events on the solver's stream)."""
        ms = (C.c_float * 6)()
        names = C.create_string_buffer(6 * 32)
        self.lib.pddp_time_kernels.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        self._chk(self.lib.pddp_time_kernels(self.h, int(sweeps), ms, names, 32))
        out = []
        for k in range(6):
            nm = names.raw[32 * k: 32 * k + 32].split(b"\0")[0].decode()
            if nm:
                out.append((nm, float(ms[k])))
        return out

    # ---- teacher-forced hooks
    def _adtype(self, name):
        return np.int32 if name in ("err", "alphaOut", "tshift", "shift") else self.dtype

    def get(self, name):
        nb = C.c_size_t(0)
        self._chk(self.lib.pddp_array_bytes(self.h, name.encode(), C.byref(nb)))
        a = np.zeros(nb.value // np.dtype(self._adtype(name)).itemsize, self._adtype(name))
        self._chk(self.lib.pddp_get_array(self.h, name.encode(), _p(a), C.c_size_t(nb.value)))
        return a

    def refresh_reference_views(self):
        """pddp_refresh_reference_views: d_ApBK / d_Bdu and the accepted trajectory in every step-size slot, as the reference leaves them after a solve."""
        self._chk(self.lib.pddp_refresh_reference_views(self.h))

    def set(self, name, a):
        a = np.ascontiguousarray(a, dtype=self._adtype(name)).ravel()
        self._chk(self.lib.pddp_set_array(self.h, name.encode(), _p(a), C.c_size_t(a.nbytes)))

    def device_array(self, name):
        """Zero-copy view of a solver array in HBM for torch (`torch.as_tensor(view, device="cuda")`), via
        __cuda_array_interface__; used for the RCCL exchange of the cost table."""
        ptr, nb = C.c_void_p(), C.c_size_t(0)
        self._chk(self.lib.pddp_array_ptr(self.h, name.encode(), C.byref(ptr), C.byref(nb)))
        return DeviceArray(ptr.value, nb.value // np.dtype(self._adtype(name)).itemsize, np.dtype(self._adtype(name)))

    def get_cost_to_go(self):
        """(P, p) the last backward pass wrote and (Pp, pp) the one before: the double buffer's halves in their current roles."""
        pw = [st.pw for st in self.get_state()]
        B = self.cfg.batch
        P, Pp, p, pp = (self.get(k).reshape(B, -1) for k in ("P", "Pp", "p", "pp"))
        w = np.asarray(pw)[:, None].astype(bool)
        return np.where(w, Pp, P), np.where(w, pp, p), np.where(w, P, Pp), np.where(w, p, pp)

    def get_state(self):
        st = (PddpState * self.cfg.batch)()
        self._chk(self.lib.pddp_get_state(self.h, st))
        return st

    def set_state(self, st):
        self._chk(self.lib.pddp_set_state(self.h, st))

    def run_phase(self, phase):
        self._chk(self.lib.pddp_run_phase(self.h, int(phase)))

    def plant_eval_paths(self):
        """`what` codes of pddp_plant_eval per plant function: every implementation of the plant the library carries."""
        if self.cfg.plant != 4:
            return {"dynamics": [0], "gradient": [1]}
        return {"dynamics": [0, 4, 6, 7], "gradient": [1, 5, 8]}    # cooperative wave; lane group; lane group, packed rows; one thread per evaluation

    def plant_eval(self, what, x, u):
        x, u = self.arr(x).reshape(-1, self.n), self.arr(u).reshape(-1, self.m)
        count = x.shape[0]
        nm = self.n + self.m
        osz = [self.npos, self.npos * nm, self.n, self.n * nm, self.npos, self.npos * nm, self.npos, self.npos, self.npos * nm, 48][what]    # 9: tool point[6] + Jacobian[7][6] (thread lanes)
        out = np.zeros((count, osz), self.dtype)
        self._chk(self.lib.pddp_plant_eval(self.h, int(what), count, _p(x), _p(u), _p(out)))
        return out
